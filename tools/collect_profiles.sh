#!/bin/bash
# Copies the newest outputs of tools/refresh_profiles.sh (gpurun_out/prof/) into profiles/ under the round's names:
#   bash tools/collect_profiles.sh r03
set -e
R=${1:?round tag, e.g. r03}
O=gpurun_out/prof
P=profiles
newest() { ls -t $1 2>/dev/null | head -1; }
cpn() { local f; f=$(newest "$1"); if [ -n "$f" ]; then cp "$f" "$2"; echo "$2 <- $f"; fi; }
cpn "$O/bench_tri1m_1080p_4spp.json" $P/${R}_path_b_bench.json
cpn "$O/bench_1lane_tri1m_1080p_4spp.json" $P/${R}_path_b_bench_1lane.json
cpn "$O/stats_tri1m_1080p_4spp/*/*_kernel_stats.csv" $P/${R}_path_b_tri1m_1080p_4spp_kernel_stats.csv
cpn "$O/stats3_tri1m_1080p_4spp/*/*_kernel_stats.csv" $P/${R}_path_b_tri1m_1080p_4spp_kernel_stats_3lanes.csv
cpn "$O/pmc_summary_default.txt" $P/${R}_path_b_pmc_summary.txt
cpn "$O/bench_terrain1m_1080p_4spp.json" $P/${R}_path_b_terrain_bench.json
cpn "$O/stats_terrain1m_1080p_4spp/*/*_kernel_stats.csv" $P/${R}_path_b_terrain1m_1080p_4spp_kernel_stats.csv
cpn "$O/bench_spheres8_1080p_4spp.json" $P/${R}_path_a_bench.json
cpn "$O/bench_1lane_spheres8_1080p_4spp.json" $P/${R}_path_a_bench_1lane.json
cpn "$O/stats_spheres8_1080p_4spp/*/*_kernel_stats.csv" $P/${R}_path_a_spheres8_1080p_4spp_kernel_stats.csv
cpn "$O/stats3_spheres8_1080p_4spp/*/*_kernel_stats.csv" $P/${R}_path_a_spheres8_1080p_4spp_kernel_stats_3lanes.csv
for t in pk_exact pk_interval; do cpn "$O/pmc_summary_$t.txt" $P/${R}_path_b_pmc_summary_$t.txt; done
