#!/bin/bash
# Round-4 evidence run (one gpurun call on a 1-GPU MI355X box); outputs under gpurun_out/r04/, the judged summaries are
# copied into profiles/r04_* afterwards (profiles/r04_fold.sh).  Counter passes are separate runs with --pmc only (+ the
# implicit kernel records), as the pool requires.
O=gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
python3 -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
python3 bench.py > $O/bench.json 2> $O/bench.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py --no-cpu-baseline --no-aten-baseline --no-graph --no-mres > $O/bench_prof.json 2> $O/bench_prof.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_kf -o k -- python3 profiles/pmc_target.py 32 > $O/pmc_kf.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_kw -o k -- python3 profiles/pmc_target.py 32 > $O/pmc_kw.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_sf -o s -- python3 bench.py --steps-only --steps 3 --warmup 1 > $O/pmc_sf.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_sw -o s -- python3 bench.py --steps-only --steps 3 --warmup 1 > $O/pmc_sw.log 2>&1 &&
rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq -o q -- python3 profiles/pmc_target.py 32 > $O/pmc_sq.log 2>&1 &&
python3 profiles/other_configs.py > $O/other_configs.txt 2>&1 &&
python3 profiles/mres_timing.py > $O/mres_timing.txt 2>&1 &&
bash profiles/prof_cfg5.sh r04final 16
echo "rc=$?" > $O/done.txt
cat $O/done.txt; tail -n 3 $O/bench.err; tail -n 2 $O/gpu_tests.txt
