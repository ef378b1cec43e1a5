#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
step 300 $O/dbg.log python tools/scratch/dbg_det.py
cat $O/dbg.log
