#!/bin/bash
# Counter passes of bench.py with the exchange variant of the path kernel switched on (MI355PT_XCHG=1 is
# exported by the caller) or off.  usage: tools/profile_xchg.sh <tag>  -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-xchg}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rmse"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT/pmc_sq3 -- $BENCH > $OUT/pmc_sq3.log 2>&1 || exit 1
echo profile done
