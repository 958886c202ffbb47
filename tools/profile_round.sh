#!/bin/bash
# rocprofv3 evidence for a bench line, run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh <tag> [bench.py args]'; default
# = the headline workload, e.g. `bash tools/profile_round.sh r2_d8 --workload d8_default_arrays`).
# One --kernel-trace --stats run, then one --pmc pass per counter set (never combined with trace domains other than
# the kernel trace; the program itself follows `--`).  Outputs land under gpurun_out/<tag>_*/ ; condense them at home:
#   python tools/profile_summary.py <tag> <stats csv> <pmc csvs...> > profiles/<tag>_summary.txt
#   python tools/make_traffic_json.py c3_headline 100000 0 <write csv> <fetch csv> profiles/<tag>_summary.txt
set -o pipefail
tag=${1:-r1}
shift || true
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
out=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 20 --warmup 5 --cpu-users 0 --skip-adaptive "$@" > $out/${tag}_stats.log 2>&1 || exit 1
for set in WRITE_SIZE FETCH_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --output-format csv -d $out/${tag}_pmc_$name -- python3 bench.py --steps 1 --warmup 0 --cpu-users 0 --skip-adaptive "$@" > $out/${tag}_pmc_$name.log 2>&1 || exit 1
done
find $out -name '*kernel_stats.csv' -o -name '*counter_collection.csv' | sort
