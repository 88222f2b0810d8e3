#!/bin/bash
# per-kernel times of the BASELINE config-5 evaluation (rocprofv3 --kernel-trace) -> gpurun_out/<tag>_cfg5_kernels.txt
TAG=${1:-cur}
B=${2:-16}
O=gpurun_out/prof_cfg5_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c5 -- python3 profiles/cfg5_bench.py $B 5 > $O/out.txt 2>&1
f=$(find $O -name "c5_kernel_trace.csv" | head -1)
python3 profiles/summarize.py "$f" 24 > gpurun_out/${TAG}_cfg5_kernels.txt 2>&1
grep -E "forward|rollout|rel-L2" $O/out.txt >> gpurun_out/${TAG}_cfg5_kernels.txt
