#!/bin/bash
# PMC passes (separate runs, --pmc only) over the spectral microbenchmark -> gpurun_out/<tag>_spectral_pmc.txt
TAG=${1:-cur}
O=gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o k -- python3 profiles/spectral_bench.py 32 3 > $O/f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o k -- python3 profiles/spectral_bench.py 32 3 > $O/w.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/t -o k -- python3 profiles/spectral_bench.py 32 3 > $O/t.log 2>&1 &&
rocprofv3 --pmc $SQ --output-format csv -d $O/q -o k -- python3 profiles/spectral_bench.py 32 3 > $O/q.log 2>&1
python3 profiles/pmc_kernels.py $O/f $O/w $O/t $O/q > gpurun_out/${TAG}_spectral_pmc.txt 2>&1
