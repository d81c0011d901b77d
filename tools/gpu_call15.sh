#!/bin/bash
set -e
mkdir -p gpurun_out
for n in 1024 4096; do python tools/ab_lib.py tools/libgpx_hip_old.so $n; python tools/ab_lib.py default $n; python tools/ab_lib.py tools/libgpx_hip_old.so $n; python tools/ab_lib.py default $n; done
GPX_LIB_COMPAT=1 python3 tools/timeline2.py 4096 12 13 > gpurun_out/tl4096.txt 2>&1
tail -3 gpurun_out/tl4096.txt
python -m pytest tests -m gpu -x -q > gpurun_out/call15_pytest.log 2>&1 || { tail -30 gpurun_out/call15_pytest.log; exit 1; }
tail -3 gpurun_out/call15_pytest.log
