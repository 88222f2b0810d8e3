#!/bin/bash
# Fold the outputs of profiles/r04_collect.sh (gpurun_out/r04/) into the tracked profiles/r04_* files.
O=gpurun_out/r04
H=$(git rev-parse --short HEAD)
python3 profiles/parse_pmc.py $O/pmc_kf $O/pmc_kw $O/pmc_sf $O/pmc_sw 32 4 profiles/traffic.json "rocprofv3 --pmc passes of profiles/r04_collect.sh at commit $H (round 4)" | tail -1
python3 profiles/summarize.py $O/prof_bench/bench_kernel_trace.csv 44 > profiles/r04_bench_kernel_summary.txt
cp $O/prof_bench/bench_kernel_stats.csv profiles/r04_bench_kernel_stats.csv
python3 profiles/pmc_kernels.py $O/pmc_sq > profiles/r04_sq_counters.txt
cp $O/bench.json profiles/r04_bench.json
cp $O/bench_prof.json profiles/r04_bench_under_rocprof.json
cp $O/other_configs.txt profiles/r04_other_configs.txt
cp $O/mres_timing.txt profiles/r04_mres_timing.txt
cp gpurun_out/r04final_cfg5_kernels.txt profiles/r04_cfg5_kernel_summary.txt
tail -n 4 $O/gpu_tests.txt > profiles/r04_gpu_tests.txt
