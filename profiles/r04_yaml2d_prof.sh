#!/bin/bash
# per-kernel dispatch list of the shipped-yaml FFNO2D (n_modes 64) training step: gpurun_out/<tag>_yaml2d_{kernels,sequence}.txt
TAG=${1:-cur}
O=gpurun_out/prof_yaml2d_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o y2 -- python3 profiles/prof_yaml2d.py 8 6 > $O/out.txt 2>&1
f=$(find $O -name "y2_kernel_trace.csv" | head -1)
python3 profiles/summarize.py "$f" 40 > gpurun_out/${TAG}_yaml2d_kernels.txt 2>&1
python3 profiles/step_sequence.py "$f" > gpurun_out/${TAG}_yaml2d_sequence.txt 2>&1
rm -rf $O
