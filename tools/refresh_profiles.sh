#!/bin/bash
# Regenerates the judged evidence under gpurun_out/refresh/ on the GPU box (copy what is kept into profiles/ as rNN_*):
#   bench line, rocprofv3 kernel stats of the same command, PMC traffic passes (separate runs), size sweep, critical-path table
#   of the chain (timeline build), same-box A/B against the previous round's library, batched-draw / batched-gradient numbers,
#   gram-build rates, one-rank distributed kernel stats, peak microbenchmarks, small-N latencies.  Every file is stamped with
#   the HEAD it was built from (GPX_HEAD, set by the caller: the box has no .git).
# Usage (from the repo root on the box): GPX_HEAD=<sha> bash tools/refresh_profiles.sh
set -o pipefail
R=$PWD
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
echo "HEAD ${GPX_HEAD:-?}  $(date -u +%FT%TZ)" > $OUT/STAMP
# the diagnostic builds the tools below need (git-ignored binaries: a fresh checkout does not have them)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
[ -x tools/peaks ] || $HIPCC --offload-arch=gfx950 -O3 -o tools/peaks tools/peaks.hip > $OUT/build_tools.log 2>&1
# (these two include the library's sources: always rebuilt, a stale diagnostic build would describe another library)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o tools/potrf_stamps tools/potrf_stamps.hip >> $OUT/build_tools.log 2>&1
rm -f tools/libgpx_hip_tl.so; rm -rf gpax_amd/csrc/build_tl; make -C gpax_amd/csrc timeline -j8 >> $OUT/build_tools.log 2>&1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; tail -c 200 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py --no-cpu-baseline --no-sweep > $OUT/stats_bench.json 2> $OUT/stats.err
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-sweep --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-sweep --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc MfmaUtil --output-format csv -d $OUT/pmc_mfma -o p -- python3 $R/bench.py --no-cpu-baseline --no-sweep --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_mfma.err
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b -o k -- python3 $R/tools/batch_sweep.py 4096 16 28:40 > $OUT/batched16.txt 2> $OUT/stats_b.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_d -o k -- python3 $R/bench.py --workload dist --dist-rehearsal --n 16384 --steps 3 --warmup 2 > $OUT/dist_rehearsal.json 2> $OUT/stats_d.err
cd $R
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/traffic.json $OUT/pmc_mfma > /dev/null
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats_b -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_batched16.csv \;
find $OUT/stats_d -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_dist_one_rank.csv \;
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/stats $OUT/stats_b $OUT/stats_d
python3 tools/sweep.py --out $OUT/sweep.json > $OUT/sweep.md 2> $OUT/sweep.err
echo "sweep done"
python3 tools/critical_path.py 4096 $OUT/critical_path.txt > /dev/null 2> $OUT/critical_path.err || true
python3 tools/critical_path.py 2048 $OUT/critical_path_2048.txt > /dev/null 2>> $OUT/critical_path.err || true
if [ -f tools/ab/libgpx_hip_r2.so ]; then
  bash tools/ab_round.sh tools/ab/libgpx_hip_r2.so "1024 2048 4096" 6 $OUT/ab_vs_r2.txt > /dev/null 2>&1 || true
  bash tools/ab_round.sh tools/ab/libgpx_hip_r2.so "8192 16384" 3 $OUT/ab_vs_r2_large.txt > /dev/null 2>&1 || true
fi
python3 tools/gradbatch_bench.py > $OUT/grad_batched.txt 2>&1 || true
for cfg in "4096 8 0" "4096 16 1" "8192 8 0" "16384 8 0" "16384 16 1" "32768 8 0"; do python3 tools/gram_time.py $cfg; done > $OUT/gram.txt 2>&1 || true
python3 tools/batch_sweep.py 4096 1,2,4,8,16 28:40 > $OUT/batch_sweep_4096.txt 2>&1
python3 tools/batch_sweep.py 1024 1,4,16,64 28:40 > $OUT/batch_sweep_1024.txt 2>&1
python3 tools/small_n.py > $OUT/small_n.txt 2>&1 || true
./tools/peaks > $OUT/peaks.txt 2>&1 || true
./tools/potrf_stamps > $OUT/potrf_diag_stamps.txt 2>&1 || true
python3 tools/fit_bench.py > $OUT/fit_bench.txt 2>&1 || true
for f in $OUT/*.txt $OUT/*.md; do sed -i "1i # built from HEAD ${GPX_HEAD:-?}" $f; done
ls -la $OUT
