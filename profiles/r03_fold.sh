#!/bin/bash
# Fold the outputs of profiles/r03_collect.sh (gpurun_out/r03/) into the tracked profiles/r03_* files.
O=gpurun_out/r03
H=$(git rev-parse --short HEAD)
python3 profiles/parse_pmc.py $O/pmc_kf $O/pmc_kw $O/pmc_sf $O/pmc_sw 32 4 profiles/traffic.json "rocprofv3 --pmc passes of profiles/r03_collect.sh at commit $H (round 3)" | tail -1
python3 profiles/summarize.py $O/prof_bench/bench_kernel_trace.csv 44 > profiles/r03_bench_kernel_summary.txt
cp $O/prof_bench/bench_kernel_stats.csv profiles/r03_bench_kernel_stats.csv
python3 profiles/summarize.py $O/prof_aten/aten_kernel_trace.csv 30 > profiles/r03_aten_baseline_kernel_summary.txt
python3 profiles/pmc_kernels.py $O/pmc_sq > profiles/r03_sq_counters.txt
cp $O/bench.json profiles/r03_bench.json
cp $O/bench_prof.json profiles/r03_bench_under_rocprof.json
cp $O/other_configs.txt profiles/r03_other_configs.txt
cp $O/mres_timing.txt profiles/r03_mres_timing.txt
