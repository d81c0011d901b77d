#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c8
./tools/potrf_stamps 2>&1 | tee gpurun_out/c8/potrf_stamps.txt | grep -E "NB=|first leaf|totals|max"
for n in 1024 4096 8192; do python tools/ab_lib.py tools/libgpx_hip_r1.so $n 2>&1 | tail -1; python tools/ab_lib.py default $n 2>&1 | tail -1; done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/c8/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c8/pytest.log
