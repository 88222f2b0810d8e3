#!/usr/bin/env python3
"""Fold the two PMC passes into profiles/traffic.json.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and on gfx950
FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import json
import os
import sys


def per_kernel(path, counter):
    acc = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return acc


def main(fetch_dir, write_dir, B, out):
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for name in fe:
        if name not in wr:
            continue
        f = sorted(fe[name])[len(fe[name]) // 2]
        w = sorted(wr[name])[len(wr[name]) // 2]
        res[name] = {"launches": len(fe[name]), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                     "hbm_bytes_per_launch": (2 * f + w) * 1024}
    # dominant kernel = split-bf16 NT 128x128 (its epilogue writes h and d); pmc_target.py launches that
    # instantiation for the forward and for the backward-data GEMM: take the larger (forward, 3 P*256 floats)
    dom = [k for k in res if "gemm_bf16x3_kernel<2, 2, 2, 2, true, true" in k]
    blob = json.load(open(out)) if os.path.exists(out) else {}
    if dom:
        blob.setdefault("ff_gemm_256x256", {})[f"B{B}"] = max(res[k]["hbm_bytes_per_launch"] for k in dom)
    blob.setdefault("per_kernel", {})[f"B{B}"] = res
    json.dump(blob, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k[:90]:90s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  (x{v['launches']})")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
