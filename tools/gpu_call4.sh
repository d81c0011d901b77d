#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/c4/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c4/pytest.log



for n in 4096 8192; do python tools/ab_lib.py default $n 2>&1 | tail -1; done
