#!/bin/bash
# needs the tuning build: make -C deepmimo_amd/csrc alt ALTFLAGS=-DDMX_TUNING_HOOKS, then DMX_LIB_PATH=deepmimo_amd/lib/alt/libdeepmimo_amd.so
# (the shipped library reads no environment variable, csrc/dmx_tuning.h)
# second sweep of the folded kernel: the edges of its dispatch region (K 16..32, 48-64 pairs) and the chunk size
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for shape in "8 1 1 1 25 4" "8 1 1 1 25 8" "8 1 1 1 25 12" "8 4 1 1 25 8" "8 4 1 1 25 12" "8 8 1 1 25 8" "8 8 1 1 25 16" "8 8 1 1 25 32" "8 1 1 1 25 16" "8 1 1 1 25 24" "8 2 1 1 25 16" "8 4 1 1 25 16" "8 4 1 1 25 24" "8 6 1 1 25 64" "8 6 1 1 25 128" "8 6 1 1 25 256" "8 6 1 1 25 512" "8 8 1 1 25 128" "8 8 1 1 25 256" "4 1 1 1 25 64" "2 1 1 1 25 512" "1 1 1 1 25 512"; do
    echo "== shape (bs bs ue ue L N=K): $shape"
    python tools/ab_bench.py --variants 0 1 2 9 12 --rounds 5 --users 200000 --shape $shape 2>&1 | grep -v amdgpu.ids | grep -v "^workload"
done
for ch in 8 16 32; do
  for shape in "8 1 1 1 25 512" "4 4 1 1 25 512" "8 4 1 1 25 512" "8 8 1 1 25 128"; do
    echo "== DMX_FOLD_CHUNK=$ch shape: $shape"
    DMX_FOLD_CHUNK=$ch python tools/ab_bench.py --variants 12 --rounds 5 --users 200000 --shape $shape 2>&1 | grep "^variant"
  done
done
