#!/bin/bash
# A/B of two BUILDS of the library on one box: the in-tree build against deepmimo_amd/lib/alt/libdeepmimo_amd.so
# (built here with extra flags: `make -C deepmimo_amd/csrc alt ALTFLAGS=-DDMX_SPLIT_FMAMIX`), alternating processes.
#   bash tools/ab_two_libs.sh [ab_bench.py args...]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
ALT=$PWD/deepmimo_amd/lib/alt/libdeepmimo_amd.so
for i in 1 2 3; do
  echo "-- base"; python tools/ab_bench.py "$@" 2>&1 | grep "^variant"
  echo "-- alt";  DMX_LIB_PATH=$ALT python tools/ab_bench.py "$@" 2>&1 | grep "^variant"
done
