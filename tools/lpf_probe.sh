#!/bin/bash
# what the gains table costs the rx_filter contraction (tuning build): all users reading user 0's rows (L2 hits), and the
# plain kernel with the register-lean tile loop the table-fed form uses (the run in profiles/r3_lpf_experiments.txt also had the
# 8-wave x 128-row-block DMA form, DMX_LPF_DMA=2, which was not kept)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export DMX_LIB_PATH=$PWD/deepmimo_amd/lib/alt/libdeepmimo_amd.so
users=${1:-100000}
for rep in 1 2; do
  for cfg in "DMX_LPF_DMA=0" "DMX_LPF_DMA=0 DMX_LPF_ALIAS_TABLE=1" "DMX_LPF_DMA=1" "DMX_LPF_DMA=1 DMX_LPF_ALIAS_TABLE=1"; do
    echo -n "rep $rep $cfg: "
    env $cfg python tools/lpf_bench.py --quick --users $users --rounds 7 2>&1 | grep "rx_filter = 1" || exit 1
  done
  echo -n "rep $rep plain, pipelined tile loop: "; python tools/lpf_bench.py --plain-only --users $users --rounds 7 2>&1 | grep "plain"
  echo -n "rep $rep plain, register-lean tile loop: "; DMX_PLAIN_TILE_MODE=0 python tools/lpf_bench.py --plain-only --users $users --rounds 7 2>&1 | grep "plain"
done
