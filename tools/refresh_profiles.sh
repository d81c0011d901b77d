#!/bin/bash
# Regenerates the judged evidence under gpurun_out/refresh/ on the GPU box (copy what is kept into profiles/):
#   bench line, rocprofv3 kernel stats of the same command, PMC traffic passes (separate runs), size sweep, step timeline.
# Usage (from the repo root on the box): bash tools/refresh_profiles.sh
set -e -o pipefail
R=$PWD
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; tail -c 300 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py --no-cpu-baseline --no-sweep > $OUT/stats_bench.json 2> $OUT/stats.err
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-sweep --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-sweep --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc MfmaUtil --output-format csv -d $OUT/pmc_mfma -o p -- python3 $R/bench.py --no-cpu-baseline --no-sweep --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_mfma.err
echo "pmc done"
cd $R
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/traffic.json $OUT/pmc_mfma > /dev/null
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/stats
python3 tools/sweep.py --out $OUT/sweep.json > $OUT/sweep.md 2> $OUT/sweep.err
echo "sweep done"
python3 tools/timeline2.py 4096 12 15 > $OUT/step_timeline.txt 2>&1 || true
python3 tools/small_n.py > $OUT/small_n.txt 2>&1 || true
ls -la $OUT
