#!/bin/bash
# Condense the rocprofv3 outputs of `tools/profile_round.sh <tag> ...` (merged back under gpurun_out/) into profiles/:
#   bash tools/profile_collect.sh <tag> "<description of the bench command>" [workload users variant [kernel substring]]
# writes profiles/<tag>_summary.txt and profiles/<tag>_kernel_stats.csv; with a workload also updates profiles/traffic.json.
tag=$1; what=$2
S=$(find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1)
P=$(find gpurun_out/${tag}_pmc_* -name '*counter_collection.csv' | sort | tr '\n' ' ')
python tools/profile_summary.py $tag $S "$what" $P > profiles/${tag}_summary.txt
grep -E "^\"?Name|dmx::" $S > profiles/${tag}_kernel_stats.csv
if [ -n "$3" ]; then
  W=$(find gpurun_out/${tag}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  F=$(find gpurun_out/${tag}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
  python tools/make_traffic_json.py $3 $4 $5 $W $F ${tag}_summary.txt ${6:-dmx::k2_fd}
fi
head -12 profiles/${tag}_summary.txt
