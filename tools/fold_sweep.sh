#!/bin/bash
# Stage-2 time of the small-panel shapes (DeepMIMO's default arrays: channel.py:36-46) - run on the GPU box.
#   variants: 0 automatic, 1 fp32 vector, 2 matrix cores (one half-empty tile per strip), 9 small-output, 12 folded matrix cores
# 200k users x 25 paths all valid; shape = BS0 BS1 UE0 UE1 L N(=K).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
VARS="${FOLD_SWEEP_VARIANTS:-0 1 2 12}"
for shape in "8 1 1 1 25 64" "8 1 1 1 25 256" "8 1 1 1 25 512" "4 4 1 1 25 64" "4 4 1 1 25 256" "4 4 1 1 25 512" \
             "8 4 1 1 25 64" "8 4 1 1 25 256" "8 4 1 1 25 512" "8 1 1 1 25 32" "8 1 1 1 25 1024" "8 8 1 1 25 64" "8 8 1 1 25 512" "4 1 1 1 25 512" "8 1 1 1 10 512"; do
    echo "== shape (bs bs ue ue L N=K): $shape"
    python tools/ab_bench.py --variants $VARS --rounds 7 --users 200000 --shape $shape 2>&1 | grep -v amdgpu.ids | grep -v "^workload"
done
