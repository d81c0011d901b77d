#!/bin/bash
# same-box A/B of library builds: default vs tools/libgpx_hip_<tag>.so ... (REPS alternations per size)
set -e
TAGS="${TAGS:-old}"
for n in ${SIZES:-1024 4096 8192 16384}; do
  for rep in $(seq 1 ${REPS:-2}); do
    python tools/ab_lib.py default $n
    for t in $TAGS; do python tools/ab_lib.py tools/libgpx_hip_$t.so $n; done
  done
done
