#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
step 900 $O/t1.log python -m pytest tests/test_gpu_parity.py -x -q -k "split_matrix or full_size or 1024_frames"
tail -15 $O/t1.log
