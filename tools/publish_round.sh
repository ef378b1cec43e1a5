#!/bin/bash
# gpurun_out/<tag>/ (written by tools/collect_round.sh on the GPU box) -> the summaries kept under profiles/.
# usage (here, after the gpurun call): bash tools/publish_round.sh r03
set -e
tag=${1:-r04}
G=gpurun_out/$tag
P=profiles
last_json() { grep '^{' "$1" | tail -1; }
last_json $G/bench_final.json | python3 -m json.tool > $P/${tag}_bench_final.json
last_json $G/bench_under_rocprof.json | python3 -m json.tool > $P/${tag}_bench_under_rocprof.json
last_json $G/bench_gpus2_rehearsal.json | python3 -m json.tool > $P/${tag}_bench_gpus2_rehearsal.json
python3 tools/collect_profiles.py $tag $G/stats $G/fetch $G/write > /dev/null
python3 tools/collect_profiles.py $tag $G/mk_stats $G/mk_fetch $G/mk_write metrics_kernels > /dev/null
python3 tools/mfma_busy.py $(ls -t $G/mfma/*/*_counter_collection.csv | head -1) $P/${tag}_mfma_busy.json "python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile" > /dev/null
python3 tools/mfma_busy.py $(ls -t $G/tmfma/*/*_counter_collection.csv | head -1) $P/${tag}_transformer_mfma_busy.json "python3 tools/profile_transformer_step.py 32 200 1" > /dev/null
cp $G/step_timeline.txt $P/${tag}_step_timeline.txt
cp $(ls -t $G/tstep/*/*_kernel_stats.csv | head -1) $P/${tag}_transformer_step_kernel_stats.csv
cp $G/transformer_step_by_shape.txt $P/${tag}_transformer_step_by_shape.txt
cp $G/metrics_kernels.json $P/${tag}_metrics_kernels.json
for f in metrics_kernels recurrence_in_step recurrence_microbench heads_microbench wgrad_microbench linear_microbench lin_stamps arith_ab epoch corun gemm_ext_microbench attention_microbench; do
  grep -v "amdgpu.ids" $G/$f.log > $P/${tag}_$f.log
done
[ -f gpurun_out/parity_worst_errors.json ] && cp gpurun_out/parity_worst_errors.json $P/${tag}_parity_worst_errors.json
[ -f gpurun_out/parity_relu_flips.json ] && cp gpurun_out/parity_relu_flips.json $P/${tag}_parity_relu_flips.json
[ -f gpurun_out/c4_full_grad_errors.json ] && cp gpurun_out/c4_full_grad_errors.json $P/${tag}_transformer_full_width_parity.json
echo published $tag
