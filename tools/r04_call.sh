#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
ARTSPEECH_DIAG_LIB=artspeech_amd/libartspeech_hip_diag_trace.so AS_LIN_NARROW=1 step 200 $O/trace.log python tools/s6_trace.py
cat $O/trace.log
