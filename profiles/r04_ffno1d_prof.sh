#!/bin/bash
# per-kernel dispatch list of the FFNO1D (BASELINE config 2 shape) training step: gpurun_out/<tag>_ffno1d_kernels.txt
TAG=${1:-cur}
O=gpurun_out/prof_ff1d_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o f1 -- python3 profiles/prof_ffno1d.py 12 > $O/out.txt 2>&1
f=$(find $O -name "f1_kernel_trace.csv" | head -1)
python3 profiles/summarize.py "$f" 60 > gpurun_out/${TAG}_ffno1d_kernels.txt 2>&1
python3 profiles/step_sequence.py "$f" > gpurun_out/${TAG}_ffno1d_sequence.txt 2>&1
rm -rf $O
