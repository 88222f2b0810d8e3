#!/bin/bash
# per-kernel times of the spectral microbenchmark (rocprofv3 --kernel-trace), summary -> gpurun_out/<tag>_spectral_kernels.txt
TAG=${1:-cur}
O=gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o sb -- python3 profiles/spectral_bench.py 32 10 > $O/out.txt 2>&1
f=$(find $O -name "sb_kernel_trace.csv" | head -1)
python3 profiles/summarize.py "$f" 24 > gpurun_out/${TAG}_spectral_kernels.txt 2>&1
cat $O/out.txt | grep -v amdgpu.ids >> gpurun_out/${TAG}_spectral_kernels.txt
