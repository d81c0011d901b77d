#!/bin/bash
# gram build: the MFMA distance kernel (default) against the direct-difference kernel (GPX_GRAM_DIRECT=1), same box, alternated
set -o pipefail
mkdir -p gpurun_out
OUT=${1:-gpurun_out/gram_ab.txt}
{
  for rep in 1 2 3; do
    for cfg in "4096 8 0" "4096 16 1" "8192 8 0" "16384 8 0" "16384 16 1" "32768 8 0"; do
      GPX_GRAM_DIRECT=1 timeout -k 10 200 python3 tools/gram_time.py $cfg | sed 's/^/direct  /' || exit 1
      timeout -k 10 200 python3 tools/gram_time.py $cfg | sed 's/^/mfma    /' || exit 1
    done
  done
} 2>&1 | tee "$OUT"
