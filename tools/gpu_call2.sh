#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c2
python tools/batch_sweep.py 4096 1,2,8,16 28:40,8:40,4:8,2:4,1:2 2>&1 | tee gpurun_out/c2/sweep4096.txt
python tools/batch_sweep.py 4096 8 28:40,4:8,1:2 1 2>&1 | tee -a gpurun_out/c2/sweep4096.txt
python tools/batch_sweep.py 1024 1,16 28:40,4:8,2:4 2>&1 | tee gpurun_out/c2/sweep1024.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c2/stats -o k -- python3 $GRAFT_REPO_ROOT/tools/batch_sweep.py 4096 8 28:40 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/c2/stats.err
cd $GRAFT_REPO_ROOT
find gpurun_out/c2/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/c2/kernel_stats_b8.csv \;
rm -rf gpurun_out/c2/stats
cat gpurun_out/c2/kernel_stats_b8.csv | cut -c1-200
