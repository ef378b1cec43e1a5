#!/bin/bash
# round 3, call 21: kernel traces of the transformer forward and training step after the ChannelBlocks fusion
source tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step 400 $O/tstep.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/tstep -- python3 $R/tools/profile_transformer_step.py 32 200 2 || exit 1
step 300 $O/tfwd.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/tfwd -- python3 $R/tools/profile_transformer_forward.py || exit 1
cd $R
python3 tools/trace_by_shape.py $(ls -t $O/tstep/*/*_kernel_trace.csv | head -1) 45 > $O/tstep_by_shape.txt
python3 tools/trace_by_shape.py $(ls -t $O/tfwd/*/*_kernel_trace.csv | head -1) 30 > $O/tfwd_by_shape.txt
rm -f $O/tstep/*/*_kernel_trace.csv $O/tfwd/*/*_kernel_trace.csv
