#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c6
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/c6/pytest_dist.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c6/pytest_dist.log
for cu in 0 8 16 32; do
  GPX_DIST_RESERVED_CUS=$cu timeout -k 10 300 python bench.py --workload dist --dist-rehearsal --n 16384 --steps 3 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dist one-rank N=16384 reserved_cus=$cu', d['ms_per_step'], 'ms')"
done
GPX_DIST_RESERVED_CUS=8 timeout -k 10 300 python bench.py --workload dist --dist-rehearsal --n 32768 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dist one-rank N=32768 reserved 8', d['ms_per_step'], 'ms')"
python tools/ab_lib.py default 16384 | tail -1
