#!/bin/bash
# The 16 M-triangle context workload (scene = 6x the Infinity Cache) under rocprofv3: kernel-trace statistics of bench.py with
# one frame lane, and the PMC passes incl. FETCH_SIZE / WRITE_SIZE.   bash tools/profile_tri16m.sh   (from the repo root, on a GPU box)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tri16m -- python3 $R/bench.py --workload tri16m_1080p_4spp --steps 10 --frames-in-flight 1 --no-traffic --no-cpu-baseline --no-parity > $O/bench_1lane_tri16m.json 2> /dev/null
echo "done stats"
bash $R/tools/pmc_pass.sh tri16m "" tri16m_1080p_4spp traffic
