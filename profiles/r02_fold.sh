#!/bin/bash
# Fold the outputs of profiles/r02_collect.sh (gpurun_out/r02/) into the tracked profiles/r02_* files.
O=gpurun_out/r02
H=$(git rev-parse --short HEAD)
python3 profiles/parse_pmc.py $O/pmc_kf $O/pmc_kw $O/pmc_sf $O/pmc_sw 32 4 profiles/traffic.json "rocprofv3 --pmc passes of profiles/r02_collect.sh at commit $H (round 2)" | tail -1
python3 profiles/summarize.py $O/prof_bench/bench_kernel_trace.csv 40 > profiles/r02_bench_kernel_summary.txt
cp $O/prof_bench/bench_kernel_stats.csv profiles/r02_bench_kernel_stats.csv
python3 profiles/pmc_kernels.py $O/pmc_sq > profiles/r02_sq_counters.txt
cp $O/bench.json profiles/r02_bench.json
cp $O/bench_prof.json profiles/r02_bench_under_rocprof.json
cp $O/bench_2rank.json profiles/r02_bench_2rank.json
cp $O/other_configs.txt profiles/r02_other_configs.txt
cp $O/gpu_tests.log profiles/r02_gpu_tests.txt
