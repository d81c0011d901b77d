#!/bin/bash
# gram build (MFMA kernel): tiles per workgroup (GPX_GRAM_TPW), same box
set -o pipefail
mkdir -p gpurun_out
OUT=${1:-gpurun_out/gram_tpw.txt}
{
  for rep in 1 2; do
    for cfg in "2048 8 0" "4096 8 0" "4096 16 1" "8192 8 0" "16384 8 0" "16384 16 1"; do
      for t in 1 2 3 4 6 8; do
        GPX_GRAM_TPW=$t timeout -k 10 200 python3 tools/gram_time.py $cfg | sed "s/^/tpw=$t  /" || exit 1
      done
    done
  done
} 2>&1 | tee "$OUT"
