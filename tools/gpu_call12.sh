#!/bin/bash
set -e
mkdir -p gpurun_out
GPX_LIB_COMPAT=1 python3 tools/timeline2.py 4096 12 13 > gpurun_out/tl4096.txt 2>&1
tail -4 gpurun_out/tl4096.txt
GPX_LIB_COMPAT=1 python3 tools/timeline2.py 1024 2 3 > gpurun_out/tl1024.txt 2>&1
tail -4 gpurun_out/tl1024.txt
