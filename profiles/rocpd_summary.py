#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (--kernel-trace, default output format of ROCm 7.2):
    python profiles/rocpd_summary.py <results.db> [top]"""
import collections
import re
import sqlite3
import sys


def main(path, top=25):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select * from kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    agg = collections.OrderedDict()
    for r in rows:
        name = r[ix["name"]]
        name = re.sub(r"\(.*", "", name)
        name = name.replace("void rpde::", "").replace("rpde::", "")[:78]
        key = (name, r[ix["grid_x"]] // max(1, r[ix["workgroup_x"]]) if "grid_x" in ix else 0)
        d = r[ix["end"]] - r[ix["start"]]
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(v[1] for v in agg.values())
    print(f"total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k[0]:78s} blocks {k[1]:>7d} calls {v[0]:5d} avg {v[1] / v[0] / 1e3:9.1f} us total {v[1] / 1e6:9.3f} ms "
              f"{100 * v[1] / tot:5.1f}%")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
