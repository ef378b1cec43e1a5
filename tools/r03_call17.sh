#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03q; mkdir -p $out
export ARTSPEECH_DIAG_LIB=1
for abl in 0 1 2; do
  step 120 $out/heads_abl$abl.log env AS_LIN_ABL=$abl python tools/bench_heads.py 20
  echo "abl $abl: $(grep -h 'gemm1\|gemm2\|dx3\|dx2' $out/heads_abl$abl.log | tr '\n' ' ')"
done
