#!/usr/bin/env python3
"""Ordered kernel list of the LAST training step in a rocprofv3 --kernel-trace CSV (steps end with k_adamw):
    python profiles/step_sequence.py <kernel_trace.csv>"""
import csv
import re
import sys


def short(name):
    m = re.search(r"gemm_\w+_kernel<([^>]*)>", name)
    if m:
        return "gemm<" + m.group(1).replace(" ", "") + ">"
    name = re.sub(r"\(.*", "", name)
    return name[-70:]


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_adamw" in r["Kernel_Name"]]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
print(f"{hi - lo} dispatches in the last step")
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  {wg:>6d}x{r['Grid_Size_Y']:>3s}  {short(r['Kernel_Name'])}")
