#!/bin/bash
# A/B of two builds of the library on the E-path projection shapes:  bash tools/ab_gemm.sh libA.so libB.so  (GPU box only)
out=gpurun_out/ab.log; : > $out
for lib in "$@"; do
  for shape in "64 512" "256 64" "64 128" "128 64"; do
    MBVAE_LIB=$lib timeout -k 10 120 python tools/gemm_probe.py $shape 2>/dev/null | grep -v amdgpu | sed "s|^|$(basename $lib) |" >> $out || exit 1
  done
done
