#!/bin/bash
# Every bench workload of one library build on one box -> gpurun_out/<tag>_bench_<workload>.json (copy into profiles/),
# the small-panel sweep of the automatic and the folded kernel, and the loader's kernel / memory-copy trace.
#   gpurun --timeout 1100 -- 'bash tools/bench_all.sh r3'
tag=${1:-r3}
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
python bench.py > $out/${tag}_bench_c3_headline.json 2>$out/${tag}_bench_c3_headline.err || exit 1
python bench.py --random-valid --cpu-users 0 > $out/${tag}_bench_c3_random_valid.json 2>/dev/null || exit 1
for w in c2_asu_shape c4_shard c5_massive d8_default_arrays d16_k256 d64_k256 c3_beam_power c3_time_domain c3_rx_filter load_asu_shape; do
    python bench.py --workload $w > $out/${tag}_bench_$w.json 2>$out/${tag}_bench_$w.err || exit 1
    echo "$w done"
done
FOLD_SWEEP_VARIANTS="0 12" bash tools/fold_sweep.sh > $out/${tag}_fold_sweep.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/${tag}_loader_stats -- python3 bench.py --workload load_asu_shape --steps 10 --warmup 3 > $out/${tag}_loader_stats.log 2>&1 || exit 1
echo all done
