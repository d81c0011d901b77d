#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c7
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c7/stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --workload dist --dist-rehearsal --n 16384 --steps 3 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/c7/stats.err
cd $GRAFT_REPO_ROOT
find gpurun_out/c7/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/c7/kernel_stats_dist.csv \;
rm -rf gpurun_out/c7/stats
cut -c1-170 gpurun_out/c7/kernel_stats_dist.csv | head -20
