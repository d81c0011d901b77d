#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c5
for n in 1024 4096 8192; do python tools/ab_lib.py tools/libgpx_hip_r1.so $n 2>&1 | tail -1; python tools/ab_lib.py default $n 2>&1 | tail -1; done
python tools/diag_modes.py 4096 1000 2>&1 | tail -6
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/c5/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c5/pytest.log
GPX_LIB_COMPAT=1 python tools/timeline2.py 4096 20 23 > gpurun_out/c5/timeline_4096.txt 2>&1; tail -45 gpurun_out/c5/timeline_4096.txt
