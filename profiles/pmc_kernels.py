#!/usr/bin/env python3
"""Median per-kernel PMC counter values from rocprofv3 --pmc output directories (csv):
    python profiles/pmc_kernels.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import re
import sys


def main(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rpde::", "").replace("rpde::", "")[:60]
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in sorted(acc.items()):
        if not any(k in name for k in ("k_dft", "k_spec", "gemm", "k_ff", "k_mix", "k_wgrad", "k_conv", "k_cf")):
            continue
        print(name)
        for c, v in sorted(cs.items()):
            v.sort()
            print(f"    {c:28s} median {v[len(v) // 2]:16.0f}  (n={len(v)})")


if __name__ == "__main__":
    main(sys.argv[1:])
