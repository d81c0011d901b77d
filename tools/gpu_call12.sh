#!/bin/bash
set -e
mkdir -p gpurun_out
GPX_LIB_COMPAT=1 python3 tools/timeline2.py 4096 6 8 > gpurun_out/tl4096_mid.txt 2>&1
cat gpurun_out/tl4096_mid.txt | head -80
