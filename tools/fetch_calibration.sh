#!/bin/bash
# Calibration of rocprofv3's FETCH_SIZE for the bulk kernel's two kinds of loads (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports
# exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on
# a known byte count in your own access pattern").  The kernel's operand loads are 16 B per lane, its C-tile loads 8 B per lane in
# 128-B row segments.  Full M x N rectangle on the 64x64 kernel with and without the C read (beta = 1 / 0): the difference of the
# two FETCH_SIZE readings against the known M N 8 bytes of C is the factor of the C loads.
R=$PWD; OUT=$R/gpurun_out/fcal; rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-result -o /tmp/gs tools/gemm_shape.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for M in 4096 8192; do
  RECT=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/b1_$M -o p -- /tmp/gs $M $M 128 > $OUT/b1_$M.log 2>&1
  RECT=1 BETA0=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/b0_$M -o p -- /tmp/gs $M $M 128 > $OUT/b0_$M.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob
def mean_fetch(d):
    v=[float(r["Counter_Value"]) for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if r.get("Counter_Name")=="FETCH_SIZE" and "gemm_nt_kernel" in r["Kernel_Name"]]
    return sum(v)/len(v)*1024.0, len(v)
for M in (4096, 8192):
    b1,n1=mean_fetch("$OUT/b1_%d"%M); b0,n0=mean_fetch("$OUT/b0_%d"%M)
    cbytes=M*M*8.0; ops=2*M*128*8.0
    print(f"M=N={M} K=128: FETCH_SIZE raw with C read {b1/1e6:.1f} MB ({n1} launches), without {b0/1e6:.1f} MB ({n0}); C tile = {cbytes/1e6:.1f} MB, operands read once = {ops/1e6:.1f} MB")
    print(f"   C loads (8 B/lane, 128-B row segments): counted at {(b1-b0)/cbytes:.3f} of their bytes;  operand loads (16 B/lane): raw {b0/1e6:.1f} MB = {b0/ops:.2f} x one pass -> x2 = {2*b0/ops:.2f} passes through the L2s")
PY
rm -rf $OUT/b1_* $OUT/b0_*/ 2>/dev/null; ls $OUT | head -3
