#!/bin/bash
# rx_filter path, same box: register-loaded gains (DMX_LPF_DMA=0) against the LDS-DMA form (1 = 16 waves x 256 rows).
# (The round-3 measurement in profiles/r3_lpf_experiments.txt also had 2 = 8 waves x 128-row blocks and the pipelined tile
# loop on top of the DMA forms; only 0 and 1 exist in the library now.)  Needs the tuning build:
#   make -C deepmimo_amd/csrc alt ALTFLAGS=-DDMX_TUNING_HOOKS ; gpurun -- 'bash tools/lpf_dma_ab.sh [users]'
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export DMX_LIB_PATH=$PWD/deepmimo_amd/lib/alt/libdeepmimo_amd.so
users=${1:-100000}
for rep in 1 2; do
  for dma in 0 1; do
    for tm in 0; do
      echo -n "rep $rep DMX_LPF_DMA=$dma DMX_LPF_TILE_MODE=$tm: "
      DMX_LPF_DMA=$dma DMX_LPF_TILE_MODE=$tm python tools/lpf_bench.py --quick --users $users --rounds 7 2>&1 | grep "rx_filter = 1" || exit 1
    done
  done
done
