#!/bin/bash
# Does a power-of-two leading dimension cost the bulk kernel?  Same trapezoid, ld = 4096 + pad doubles.
cd /root/repo
for rep in 1 2; do
for pad in 896 904 912 928 960 1024 1040; do
  echo -n "ld=$((3200+pad)) : "; LDPAD=$pad ./tools/gemm_shape 4224 3200 128 | tr '\n' ' '; echo
done
done
for pad in 0 16 32; do echo -n "N=8192 ld=$((6144+2048+pad)): "; LDPAD=$((2048+pad)) ./tools/gemm_shape 8192 6144 128 | tr '\n' ' '; echo; done
