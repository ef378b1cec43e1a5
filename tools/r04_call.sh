#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
step 300 $O/linear.log python tools/bench_linear.py
cat $O/linear.log
