#!/bin/bash
# Is this box one of those on which the single-accumulator build of the folded kernel (deepmimo_amd/lib/alt_prev) is not
# bit-reproducible?  If so, run the current build on it at length; if not, stop after a minute.
# alt_prev is not kept in the tree: `git archive b313e09 deepmimo_amd/csrc include | tar -x -C /tmp/prev && make -C
# /tmp/prev/deepmimo_amd/csrc` and copy /tmp/prev/deepmimo_amd/lib/libdeepmimo_amd.so to deepmimo_amd/lib/alt_prev/ first.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
L=$PWD/deepmimo_amd/lib
prev=$(DMX_LIB_PATH=$L/alt_prev/libdeepmimo_amd.so python tools/repro_stress.py --launches 1000 2>&1 | grep "differing user-launches" | tail -1)
echo "single-accumulator build: $prev"
n=$(echo "$prev" | sed 's/.*: \([0-9]*\) differing.*/\1/')
if [ "${n:-0}" -lt 3 ]; then echo "not a susceptible box"; exit 0; fi
echo "== current (two accumulators): $(python tools/repro_stress.py --launches 3000 2>&1 | grep 'differing user-launches' | tail -1)"
echo "== single-accumulator again:   $(DMX_LIB_PATH=$L/alt_prev/libdeepmimo_amd.so python tools/repro_stress.py --launches 2000 2>&1 | grep 'differing user-launches' | tail -1)"
echo "== current, 64 pairs:          $(python tools/repro_stress.py --bs 8x8 --K 256 --users 100000 --launches 2000 2>&1 | grep 'differing user-launches' | tail -1)"
echo "== current again:              $(python tools/repro_stress.py --launches 3000 2>&1 | grep 'differing user-launches' | tail -1)"
