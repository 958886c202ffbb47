#!/bin/bash
# rx_filter path, same box: register-loaded gains (DMX_LPF_DMA=0) against the LDS-DMA forms (1 = 16 waves x 256 rows,
# 2 = 8 waves x 128 rows), each with the register-lean (0) and the pipelined (2) tile loop.  Needs the tuning build:
#   make -C deepmimo_amd/csrc alt ALTFLAGS=-DDMX_TUNING_HOOKS ; gpurun -- 'bash tools/lpf_dma_ab.sh [users]'
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export DMX_LIB_PATH=$PWD/deepmimo_amd/lib/alt/libdeepmimo_amd.so
users=${1:-100000}
for rep in 1 2; do
  for dma in 0 1 2; do
    for tm in 0 2; do
      echo -n "rep $rep DMX_LPF_DMA=$dma DMX_LPF_TILE_MODE=$tm: "
      DMX_LPF_DMA=$dma DMX_LPF_TILE_MODE=$tm python tools/lpf_bench.py --quick --users $users --rounds 7 2>&1 | grep "rx_filter = 1" || exit 1
    done
  done
done
