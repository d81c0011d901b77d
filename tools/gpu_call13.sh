#!/bin/bash
set -e
for n in 256 1024 2048 4096; do python tools/ab_lib.py tools/libgpx_hip_old.so $n; python tools/ab_lib.py default $n; python tools/ab_lib.py tools/libgpx_hip_old.so $n; python tools/ab_lib.py default $n; done
