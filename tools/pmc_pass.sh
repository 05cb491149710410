#!/bin/bash
# PMC passes of one path-B variant on a GPU box:  bash tools/pmc_pass.sh TAG [RT_BENCH_TUNE value] [workload]
#   -> gpurun_out/prof/pmc_summary_TAG.txt (per-kernel counters, lanes per vector instruction, L2 hit rate)
# Counters in their own runs with --kernel-trace only, the program directly after "--" (MI355X_MICROARCH.md).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
TAG=$1
W=${3:-tri1m_1080p_4spp}
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
if [ -n "$2" ]; then export RT_BENCH_TUNE=$2; else unset RT_BENCH_TUNE; fi
P="python3 $R/bench.py --traffic-child --workload $W"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/pmc_${TAG}_a -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $O/pmc_${TAG}_b -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_${TAG}_c -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/pmc_${TAG}_d -- $P > /dev/null 2>&1
if [ "$4" = traffic ]; then
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG}_e -- $P > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG}_f -- $P > /dev/null 2>&1
fi
(cd $R && python3 tools/pmc_summary.py $O/pmc_${TAG}_?) > $O/pmc_summary_$TAG.txt
echo "done pmc $TAG"
