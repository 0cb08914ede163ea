#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block cites.  Run ON the GPU box:
#   gpurun -- 'bash tools/profile_bench.sh <tag> [extra bench args]'
# Kernel trace and each PMC group are separate passes (gpurun refuses --pmc mixed with traces;
# FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 -- MI355X_MICROARCH.md §rocprofv3 PMC slots).
set -e
TAG=${1:-run}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-side-records "$@" > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-records "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-records "$@" > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-records "$@" > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_F64 SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-records "$@" > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/misc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-records "$@" > $OUT/misc.log 2>&1 || true
python3 tools/pmc_summary.py $OUT | tee $OUT/summary.txt
