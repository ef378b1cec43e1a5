#!/bin/bash
# Collect the round's measurement evidence on the GPU box into gpurun_out/<tag>/ (copied to profiles/ afterwards by
# tools/collect_profiles.py and by hand).  usage (inside gpurun): bash tools/collect_round.sh r04 [a|b|c]
# (three parts, each inside gpurun's 20-minute limit: a = bench + counter passes of the BiGRU step, b = transformer passes,
#  c = microbenchmarks, arithmetic A/B, two-rank rehearsal)
source tools/gpu_steps.sh
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
part=${2:-abc}
cd /tmp && export TMPDIR=/tmp
if [[ $part == *a* ]]; then
step 500 $out/bench_final.json python3 $R/bench.py
step 300 $out/bench_under_rocprof.json rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-exact
step 300 $out/fetch.json rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile --no-exact
step 300 $out/write.json rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile --no-exact
step 300 $out/kt.log rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras --no-exact
python3 $R/tools/step_timeline.py $(ls -t $out/kt/*/*_kernel_trace.csv | head -1) > $out/step_timeline.txt
step 300 $out/mfma.json rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile --no-exact
fi
if [[ $part == *b* ]]; then
# the three small kernels north_star names: per-kernel time + HBM bytes of a driver-style loop
step 200 $out/mk_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d $out/mk_stats -- python3 $R/tools/bench_metrics_kernels.py 50
step 200 $out/mk_fetch.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/mk_fetch -- python3 $R/tools/bench_metrics_kernels.py 10
step 200 $out/mk_write.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/mk_write -- python3 $R/tools/bench_metrics_kernels.py 10
step 400 $out/tstep.log rocprofv3 --kernel-trace --stats --output-format csv -d $out/tstep -- python3 $R/tools/profile_transformer_step.py 32 200 2
python3 $R/tools/trace_by_shape.py $(ls -t $out/tstep/*/*_kernel_trace.csv | head -1) 45 > $out/transformer_step_by_shape.txt
rm -f $out/tstep/*/*_kernel_trace.csv
# matrix-pipe occupancy of the transformer step (north_star: "MFMA utilisation against CDNA4 peak"), counters in their own pass
step 600 $out/tmfma.log rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/tmfma -- python3 $R/tools/profile_transformer_step.py 32 200 1
fi
cd $R
if [[ $part == *c* ]]; then
step 200 $out/gemm_ext_microbench.log python3 tools/bench_gemm_ext.py 10
step 200 $out/attention_microbench.log python3 tools/bench_attention.py
step 120 $out/metrics_kernels.log python3 tools/bench_metrics_kernels.py 50 --json $out/metrics_kernels.json
step 120 $out/recurrence_in_step.log python3 tools/recurrence_stamps.py 50
step 120 $out/recurrence_microbench.log python3 tools/bench_gru.py 20
step 120 $out/heads_microbench.log python3 tools/bench_heads.py 20
step 120 $out/wgrad_microbench.log python3 tools/bench_wgrad.py 20
step 120 $out/linear_microbench.log python3 tools/bench_linear.py 20
step 120 $out/lin_stamps.log python3 tools/lin_stamps.py
for m in bf16x6 fp32 bf16x6 fp32; do ARTSPEECH_MATRIX_ARITH=$m step 200 $out/ab_$m.json python3 bench.py --no-extras --no-cpu-baseline --no-profile --no-exact; grep '^{' $out/ab_$m.json | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'])" >> $out/arith_ab.log; done
step 300 $out/epoch.log python3 tools/bench_epoch.py 4
step 120 $out/corun.log python3 tools/check_corun.py
step 300 $out/bench_gpus2_rehearsal.json python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline
fi
echo done
