#!/bin/bash
source tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step 300 $O/tfwd.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/tfwd -- python3 $R/tools/profile_transformer_forward.py || exit 1
cd $R
python3 tools/trace_by_shape.py $(ls -t $O/tfwd/*/*_kernel_trace.csv | head -1) 30 > $O/tfwd_by_shape.txt
rm -f $O/tfwd/*/*_kernel_trace.csv
cat $O/tfwd_by_shape.txt
