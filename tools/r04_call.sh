#!/bin/bash
. tools/gpu_steps.sh
O=gpurun_out/r04a; mkdir -p $O
step 900 $O/gpu_tests.log python -m pytest tests -x -q -m gpu
tail -5 $O/gpu_tests.log
step 300 $O/bench.log python bench.py
tail -2 $O/bench.log | cut -c1-1500
