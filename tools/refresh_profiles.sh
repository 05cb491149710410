#!/bin/bash
# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun):
#   bash tools/refresh_profiles.sh [A|B|all]   -> writes gpurun_out/prof/*, copy what should be judged into profiles/ (named per round)
#   (A = bench lines + kernel-trace statistics + the default PMC passes, B = the variants' PMC passes + the tools; each fits a 20-minute gpurun call)
# rocprofv3 passes follow /opt/skills/guides/MI355X_MICROARCH.md: kernel-trace/stats in one run, PMC
# counters in their own runs with --kernel-trace only, the program directly after "--".
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
STAGE=${1:-all}
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
if [ $STAGE != B ]; then
for W in tri1m_1080p_4spp terrain1m_1080p_4spp spheres8_1080p_4spp; do
    python3 $R/bench.py --workload $W > $O/bench_$W.json 2> $O/bench_$W.err
    # kernel durations: one frame at a time (what roofline.avg_kernel_ms measures, with HIP events, on lane 0 alone) ...
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 $R/bench.py --workload $W --frames-in-flight 1 --no-traffic --no-cpu-baseline --no-parity > $O/bench_1lane_$W.json 2> /dev/null
    # ... and the default command (three frame lanes: kernels of different frames overlap and stretch each other)
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3_$W -- python3 $R/bench.py --workload $W --no-traffic --no-cpu-baseline --no-parity > /dev/null 2>&1
    echo "done $W"
done
fi
pmc() {  # $1 = output tag; RT_BENCH_TUNE (exported by the caller) selects the variant
    P="python3 $R/bench.py --traffic-child --workload tri1m_1080p_4spp"
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/pmc_$1_a -- $P > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $O/pmc_$1_b -- $P > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_$1_c -- $P > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/pmc_$1_d -- $P > /dev/null 2>&1
    if [ "$1" = default ]; then
        rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_$1_e -- $P > /dev/null 2>&1
        rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_$1_f -- $P > /dev/null 2>&1
    fi
    (cd $R && python3 tools/pmc_summary.py $O/pmc_$1_?) > $O/pmc_summary_$1.txt
    echo "done pmc $1"
}
unset RT_BENCH_TUNE
if [ $STAGE != B ]; then pmc default; fi
if [ $STAGE = A ]; then echo "done stage A"; exit 0; fi
# the round's two experiments on pt_trace, each against the default: camera rays through the per-lane kernel (no packet kernel);
# bounce / shadow rays sorted in LDS by (direction octant, origin cell) inside each 1024-ray workgroup of the shade stage
export RT_BENCH_TUNE=tune_no_packet=1
pmc no_packet
export RT_BENCH_TUNE=tune_sort_rays=1
pmc sort_rays
unset RT_BENCH_TUNE
cd $R
# round 3: the packet kernel's node tests (per-ray slab tests of all eight children / interval test + per-ray tests of the children that pass;
# the default - interval test only - is in pmc_summary_default.txt) and their A/B on both scenes
bash tools/pmc_pass.sh pk_exact tune_no_packet=2
bash tools/pmc_pass.sh pk_interval tune_no_packet=3
python3 tools/ab_tri_mode.py --variants "tune_no_packet=2;tune_no_packet=3;tune_no_packet=4;tune_no_packet=5;tune_no_packet=1;tune_no_packet=2;tune_no_packet=4" > $O/ab_packet.txt 2>&1
python3 tools/ab_tri_mode.py --scene terrain --variants "tune_no_packet=2;tune_no_packet=3;tune_no_packet=4;tune_no_packet=5;tune_no_packet=1;tune_no_packet=2;tune_no_packet=4" > $O/ab_packet_terrain.txt 2>&1
python3 tools/ab_pt_variants.py > $O/ab_pt_variants.txt
python3 tools/partition_scaling.py --lanes 6 > $O/partition_scaling.txt
python3 tools/two_level_bvh.py > $O/two_level_bvh.txt
python3 tools/run_configs.py --out $O/configs.json > $O/configs.md
[ -x tools/_bin/valu ] && tools/_bin/valu > $O/valu_issue_rates.txt
[ -x tools/_bin/l1bench ] && tools/_bin/l1bench > $O/l1_gather_bench.txt
echo "done all"
