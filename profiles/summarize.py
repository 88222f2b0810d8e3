#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, launch geometry) time."""
import collections
import csv
import re
import sys


def main(path, top=30):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        name = r["Kernel_Name"]
        m = re.search(r"gemm_f32_kernel<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+)>", name)
        if m:
            name = "gemm<%s%s%s%s,%s,%s>" % (m.group(1), m.group(2), m.group(3), m.group(4),
                                             "Ak" if m.group(5) == "true" else "Ax", "Bk" if m.group(6) == "true" else "Bx")
        else:
            name = name.split("(")[0][-44:]
        key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"],
               r["LDS_Block_Size"], r["VGPR_Count"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(v[1] for v in agg.values())
    print(f"total kernel time {tot / 1e6:.2f} ms over {len(rows)} dispatches")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k[0]:44s} blocks {k[1]:>7d} x{k[2]:>5s} x{k[3]:>2s} lds {k[4]:>6s} vgpr {k[5]:>4s} "
              f"calls {v[0]:4d} avg {v[1] / v[0] / 1e3:9.1f} us total {v[1] / 1e6:8.2f} ms {100 * v[1] / tot:5.1f}%")
    # The top kernel instantiation serves several problem shapes inside a training step (K = 64 and 256, with
    # and without the activation epilogue), so its overall average is a mix.  bench.py's roofline line times
    # ONE shape back to back (2 warm-up + 20 timed launches): report the longest uninterrupted run of the
    # top kernel -- that run is the microbenchmark, and its average is the number to compare with
    # roofline.ms_per_launch.
    top_name = max(collections.Counter(r["Kernel_Name"] for r in rows).items(),
                   key=lambda kv: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == kv[0]))[0]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    best, cur = [], []
    for r in rows:
        if r["Kernel_Name"] == top_name:
            cur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        else:
            if len(cur) > len(best):
                best = cur
            cur = []
    if len(cur) > len(best):
        best = cur
    if len(best) >= 4:
        timed = best[2:]
        print(f"longest back-to-back run of the top kernel: {len(best)} launches, avg of the last {len(timed)}: "
              f"{sum(timed) / len(timed) / 1e3:.1f} us  (= bench.py roofline microbenchmark of one shape)")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
