#!/bin/bash
# Collect the round's measurement evidence on the GPU box into gpurun_out/<tag>/ (copied to profiles/ afterwards by
# tools/collect_profiles.py and by hand).  usage (inside gpurun): bash tools/collect_round.sh r02
set -e
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $out/bench_final.json 2> $out/bench_final.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile > $out/fetch.json 2> $out/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile > $out/write.json 2> $out/write.log
rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras > $out/kt.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $out/kt/*/*_kernel_trace.csv > $out/step_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tstep -- python3 $GRAFT_REPO_ROOT/tools/profile_transformer_step.py 32 200 2 > $out/tstep.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/bench_wgrad.py 20 > $out/wgrad_microbench.log 2>&1
ARTSPEECH_DIAG_LIB=1 AS_NO_WGRAD=1 python3 tools/bench_wgrad.py 20 >> $out/wgrad_microbench.log 2>&1
python3 tools/bench_heads.py 20 > $out/heads_microbench.log 2>&1
ARTSPEECH_DIAG_LIB=1 AS_NO_LIN=1 python3 tools/bench_heads.py 20 >> $out/heads_microbench.log 2>&1
python3 tools/lin_stamps.py > $out/lin_stamps.log 2>&1
hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak > $out/mfma_peak.log 2>&1
python3 tools/bench_gru.py 20 > $out/recurrence_microbench.log 2>&1
echo done
