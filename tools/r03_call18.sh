#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03r; mkdir -p $out
export ARTSPEECH_DIAG_LIB=1
for sg in 0 10 11 0 10; do
  step 120 $out/heads_s$sg.log env AS_LIN_STAGGER=$sg python tools/bench_heads.py 20
  echo "stagger $sg: $(grep -h 'gemm1\|gemm2\|dx3\|dx2' $out/heads_s$sg.log | tr '\n' ' ')"
done
step 60 $out/stamps.log python tools/lin_stamps.py
head -12 $out/stamps.log
