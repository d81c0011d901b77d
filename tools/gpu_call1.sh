#!/bin/bash
# round 2, first GPU call: bits vs the round-1 library, the GPU test suite, the bench line
set -o pipefail
mkdir -p gpurun_out/c1
python tools/bits_r1.py dump tools/libgpx_hip_r1.so gpurun_out/c1/r1.npz > gpurun_out/c1/bits_r1.log 2>&1 || { echo "r1 dump failed"; tail -5 gpurun_out/c1/bits_r1.log; }
python tools/bits_r1.py dump default gpurun_out/c1/r2.npz > gpurun_out/c1/bits_r2.log 2>&1 || { echo "r2 dump failed"; tail -20 gpurun_out/c1/bits_r2.log; }
python tools/bits_r1.py cmp gpurun_out/c1/r1.npz gpurun_out/c1/r2.npz 2>&1 | tee gpurun_out/c1/bits_cmp.log
for n in 1024 4096 16384; do
  python tools/ab_lib.py tools/libgpx_hip_r1.so $n 2>&1 | tail -1 | tee -a gpurun_out/c1/ab.log
  python tools/ab_lib.py default $n 2>&1 | tail -1 | tee -a gpurun_out/c1/ab.log
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/c1/pytest.log
grep -E "ill-cond|inv-branch" gpurun_out/c1/pytest.log | head -40
timeout -k 10 600 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/c1/bench.json; tail -5 gpurun_out/c1/bench.err
