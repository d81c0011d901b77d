#!/bin/bash
# A/B of the strided lazy bulk update ("bulk_stride") against the round-2 schedule, same box, alternated (tools/opt_ab3.py)
set -o pipefail
mkdir -p gpurun_out
OUT=${1:-gpurun_out/stride_ab.txt}
SIZES=${2:-"2048 4096 8192 16384"}
shift; shift
{
  for n in $SIZES; do
    timeout -k 10 400 python3 tools/opt_ab3.py $n "$@" || exit 1
  done
} 2>&1 | tee "$OUT"
