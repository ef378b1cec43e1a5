#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03o; mkdir -p $out
export ARTSPEECH_DIAG_LIB=1
for cfg in "3 1" "3 0" "2 0" "2 1" "2 2"; do
  set -- $cfg
  step 120 $out/heads_n$1_s$2.log env AS_LIN_NBUF=$1 AS_LIN_STAGGER=$2 python tools/bench_heads.py 20
  echo "nbuf $1 stagger $2: $(grep -h 'gemm1\|gemm2\|dx3\|dx2\|finite' $out/heads_n$1_s$2.log | tr '\n' ' ')"
done
