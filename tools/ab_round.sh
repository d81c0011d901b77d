#!/bin/bash
# Same-box A/B of the shipped library against a baseline build: bit comparison of every output (tools/bits_r1.py), then
# ALT alternations of tools/ab_lib.py per size (the same library differs by 5-7 % from box to box: only same-box,
# alternated numbers count).  Usage: bash tools/ab_round.sh <baseline.so> "<sizes>" <alternations> <out.txt>
set -o pipefail
BASE=$1; SIZES=${2:-"2048 4096"}; ALT=${3:-6}; OUT=${4:-gpurun_out/ab.txt}
mkdir -p gpurun_out
{
  echo "# A/B: $BASE vs gpax_amd/libgpx_hip.so (HEAD ${GPX_HEAD:-?}), $ALT alternations"
  python3 tools/bits_r1.py dump "$BASE" /tmp/bits_a.npz > /tmp/bits_a.log 2>&1 && python3 tools/bits_r1.py dump default /tmp/bits_b.npz > /tmp/bits_b.log 2>&1 \
    && python3 tools/bits_r1.py cmp /tmp/bits_a.npz /tmp/bits_b.npz
  for n in $SIZES; do
    for i in $(seq $ALT); do
      python3 tools/ab_lib.py "$BASE" $n
      python3 tools/ab_lib.py default $n
    done
  done
} 2>&1 | tee "$OUT"
