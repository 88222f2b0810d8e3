#!/bin/bash
# round-4 kernel experiments: per-kernel times of the spectral microbenchmark under env-selected variants
#   bash profiles/r04_exp.sh TAG "VAR=val VAR=val ..."    -> gpurun_out/TAG_spectral_kernels.txt
TAG=${1:-cur}
for kv in $2; do export "$kv"; done
O=gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o sb -- python3 profiles/spectral_bench.py 32 10 > $O/out.txt 2>&1
f=$(find $O -name "sb_kernel_trace.csv" | head -1)
python3 profiles/summarize.py "$f" 24 > gpurun_out/${TAG}_spectral_kernels.txt 2>&1
grep -v amdgpu.ids $O/out.txt >> gpurun_out/${TAG}_spectral_kernels.txt
rm -rf $O
