#!/bin/bash
# same-box A/B of the library against tools/libgpx_hip_old.so (a copy of the previous build) + bit comparison
set -e
mkdir -p gpurun_out
for n in 1024 4096 8192 16384; do python tools/ab_lib.py tools/libgpx_hip_old.so $n; python tools/ab_lib.py default $n; python tools/ab_lib.py tools/libgpx_hip_old.so $n; python tools/ab_lib.py default $n; done
python tools/bits_r1.py dump tools/libgpx_hip_old.so /tmp/bits_old.npz > /dev/null
python tools/bits_r1.py dump default /tmp/bits_new.npz > /dev/null
python tools/bits_r1.py cmp /tmp/bits_old.npz /tmp/bits_new.npz | tail -3
