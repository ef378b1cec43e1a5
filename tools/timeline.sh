#!/bin/bash
# One step's kernel timeline under rocprofv3 (GPU box): bash tools/timeline.sh <tag> [ENV=1 ...]  -> gpurun_out/<tag>_timeline.txt
set -e
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=$GRAFT_REPO_ROOT/gpurun_out/kt_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras --no-exact > $out/kt.log 2>&1
f=$(ls $out/*/*_kernel_trace.csv | head -1)
test -n "$f"
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py "$f" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_timeline.txt
