#!/bin/bash
# Stage-2 time over shapes with few selected subcarriers (the DeepMIMO default is ONE subcarrier): run on the GPU box.
#   variants: 0 automatic, 1 fp32 vector (lane = subcarrier), 2 matrix cores, 9 small-output (wave per user)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for shape in "8 8 1 1 25 1" "8 8 1 1 25 8" "16 16 1 1 25 8" "16 16 2 2 25 2" "4 4 1 1 25 4" "8 1 1 1 25 1" "8 8 2 2 25 16" "8 8 2 2 25 64" "32 1 1 1 25 64" "8 1 1 1 25 64" "8 1 1 1 25 16"; do
    echo "== shape (bs bs ue ue L N=K): $shape"
    python tools/ab_bench.py --variants 0 1 2 9 --rounds 7 --users 200000 --shape $shape 2>&1 | grep -v amdgpu.ids | grep -v "^workload"
done
