#!/bin/bash
# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun):
#   bash tools/refresh_profiles.sh        -> writes gpurun_out/prof/*, copy what should be judged into profiles/
# rocprofv3 passes follow /opt/skills/guides/MI355X_MICROARCH.md: kernel-trace/stats in one run, PMC
# counters in their own runs with --kernel-trace only, the program directly after "--".
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
rm -rf $O
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
for W in tri1m_1080p_4spp terrain1m_1080p_4spp spheres8_1080p_4spp; do
    python3 $R/bench.py --workload $W > $O/bench_$W.json 2> $O/bench_$W.err
    # kernel durations: one frame at a time (what roofline.avg_kernel_ms measures, with HIP events, on lane 0 alone) ...
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 $R/bench.py --workload $W --frames-in-flight 1 --no-traffic --no-cpu-baseline > $O/bench_1lane_$W.json 2> /dev/null
    # ... and the default command (three frame lanes: kernels of different frames overlap and stretch each other)
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3_$W -- python3 $R/bench.py --workload $W --no-traffic --no-cpu-baseline > /dev/null 2>&1
    echo "done $W"
done
P="python3 $R/bench.py --traffic-child --workload tri1m_1080p_4spp"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/pmc_a -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_b -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_c -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/pmc_d -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_e -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_f -- $P > /dev/null 2>&1
echo "done pmc"
cd $R
python3 tools/pmc_summary.py $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d $O/pmc_e $O/pmc_f > $O/pmc_summary.txt
python3 tools/run_configs.py --out $O/configs.json > $O/configs.md
[ -x tools/_bin/valu ] && tools/_bin/valu > $O/valu_issue_rates.txt
echo "done all"
