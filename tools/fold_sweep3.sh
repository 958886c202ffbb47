#!/bin/bash
# needs the tuning build: make -C deepmimo_amd/csrc alt ALTFLAGS=-DDMX_TUNING_HOOKS, then DMX_LIB_PATH=deepmimo_amd/lib/alt/libdeepmimo_amd.so
# (the shipped library reads no environment variable, csrc/dmx_tuning.h)
# folded kernel with one table set per workgroup (33-128 antenna pairs) against the plain matrix-core kernel, and the two
# table modes against each other at 32 pairs
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for shape in "8 6 1 1 25 64" "8 6 1 1 25 256" "8 6 1 1 25 512" "8 8 1 1 25 64" "8 8 1 1 25 128" "8 8 1 1 25 256" "8 8 1 1 25 512" "8 8 1 1 25 1024" \
             "8 6 2 1 25 64" "8 6 2 1 25 256" "8 6 2 1 25 512" "8 8 2 1 25 64" "8 8 2 1 25 256" "8 8 2 1 25 512"; do
    echo "== shape (bs bs ue ue L N=K): $shape"
    python tools/ab_bench.py --variants 2 12 --rounds 5 --users 100000 --shape $shape 2>&1 | grep "^variant"
done
for sh in 0 1; do for shape in "8 4 1 1 25 64" "8 4 1 1 25 256" "8 4 1 1 25 512" "4 4 1 1 25 512" "8 8 1 1 25 256"; do
    echo "== DMX_FOLD_SHARED=$sh shape: $shape"
    DMX_FOLD_SHARED=$sh python tools/ab_bench.py --variants 12 --rounds 5 --users 100000 --shape $shape 2>&1 | grep "^variant"
done; done
