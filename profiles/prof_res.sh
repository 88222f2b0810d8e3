#!/bin/bash
# per-kernel view of the headline model's training step at a smaller grid: gpurun_out/res<R>_{kernels,sequence}.txt
R=${1:-64}
O=gpurun_out/prof_res_$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r -- python3 profiles/prof_res.py $R 32 12 > $O/out.txt 2>&1
f=$(find $O -name "r_kernel_trace.csv" | head -1)
python3 profiles/summarize.py "$f" 30 > gpurun_out/res${R}_kernels.txt 2>&1
python3 profiles/step_sequence.py "$f" > gpurun_out/res${R}_sequence.txt 2>&1
rm -rf $O
