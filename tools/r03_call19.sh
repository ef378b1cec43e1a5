#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03s; mkdir -p $out
step 500 $out/tests.log python -m pytest tests -m gpu -q -x
tail -3 $out/tests.log
