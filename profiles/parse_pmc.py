#!/usr/bin/env python3
"""Fold the PMC passes into profiles/traffic.json.
    python profiles/parse_pmc.py <kernels FETCH dir> <kernels WRITE dir> <step FETCH dir> <step WRITE dir> <B> <steps in the step run> <out.json> <source tag>
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM); Infinity-Cache hits are counted
as fetches."""
import csv
import glob
import json
import os
import re
import sys


def per_kernel(path, counter):
    acc = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rpde::", "").replace("rpde::", "")
            acc.setdefault(name, []).append(float(r["Counter_Value"]))
    return acc


def main(kf, kw, sf, sw, B, steps, out, source):
    fe, wr = per_kernel(kf, "FETCH_SIZE"), per_kernel(kw, "WRITE_SIZE")
    res = {}
    for name in fe:
        if name not in wr:
            continue
        f = sorted(fe[name])[len(fe[name]) // 2]
        w = sorted(wr[name])[len(wr[name]) // 2]
        res[name] = {"launches": len(fe[name]), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                     "hbm_bytes_per_launch": (2 * f + w) * 1024}
    blob = {"source": source,
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = (2 FETCH + WRITE) KiB "
                   "(gfx950: FETCH_SIZE counts half of wide streaming reads); median over the launches of a kernel"}
    dom = [k for k in res if k.startswith("k_ff3_fwd_h2<1>")]
    if dom:
        blob["dominant_kernel"] = {f"B{B}": res[dom[0]]["hbm_bytes_per_launch"], "kernel": dom[0]}
    blob["per_kernel"] = {f"B{B}": res}
    # FSpectralConv2d.forward_fourier = weight preparation + analysis + mode mix + synthesis (no skip tensor)
    parts = [k for k in res if k.startswith(("k_mix_prep", "k_dft_analysis_rr_h2", "k_dft_analysis_sq_h2", "k_dft_analysis_h2",
                                             "k_mix_h2", "k_dft_synthesis4_h2")) or
             (k.startswith("k_dft_synthesis3_h2") and "false" in k)]
    if parts:
        blob["spectral_forward"] = {f"B{B}": sum(res[k]["hbm_bytes_per_launch"] for k in parts), "kernels": sorted(parts)}
    sfe, swr = per_kernel(sf, "FETCH_SIZE"), per_kernel(sw, "WRITE_SIZE")
    rd = sum(sum(v) for v in sfe.values()) * 2 * 1024 / steps
    wb = sum(sum(v) for v in swr.values()) * 1024 / steps
    blob["train_step"] = {f"B{B}": {"read_bytes": rd, "write_bytes": wb, "hbm_bytes_per_step": rd + wb,
                                   "how": f"all kernels of `bench.py --steps-only` summed / {steps} steps"}}
    json.dump(blob, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:14]:
        print(f"{k[:84]:84s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  (x{v['launches']})")
    print(f"train step: {(rd + wb) / 1e9:.1f} GB  (read {rd / 1e9:.1f}, written {wb / 1e9:.1f})")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7], sys.argv[8])
