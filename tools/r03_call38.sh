#!/bin/bash
source tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03t; mkdir -p $O
step 900 $O/ttests.log python -m pytest tests/test_gpu_transformer.py tests/test_gpu_pipeline.py -q || exit 1
tail -2 $O/ttests.log
grep -q failed $O/ttests.log && exit 1
cd /tmp && export TMPDIR=/tmp
step 400 $O/tstep.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/tstep -- python3 $R/tools/profile_transformer_step.py 32 200 2 || exit 1
cd $R
python3 tools/trace_by_shape.py $(ls -t $O/tstep/*/*_kernel_trace.csv | head -1) 45 > $O/tstep_by_shape.txt
rm -f $O/tstep/*/*_kernel_trace.csv
grep "attn_\|total kernel" $O/tstep_by_shape.txt | head -8
step 300 $O/bench_transformer.log python tools/bench_transformer.py 32 200 5 || exit 1
grep "transformer f" $O/bench_transformer.log
