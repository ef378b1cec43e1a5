#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03d; mkdir -p $out
step 400 $out/tests.log python -m pytest tests -m gpu -q -x
tail -5 $out/tests.log
step 200 $out/bench.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
step 200 $out/bench_nopipe.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras --no-pipeline
cd /tmp && export TMPDIR=/tmp
step 300 $GRAFT_REPO_ROOT/$out/kt.log rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras
cd $GRAFT_REPO_ROOT
python3 tools/step_timeline.py $out/kt/*/*_kernel_trace.csv > $out/step_timeline.txt
cat $out/step_timeline.txt
python - <<'PY'
import json
for n in ["bench", "bench_nopipe"]:
    try:
        d = json.loads(open(f"gpurun_out/r03d/{n}.json").read().strip().splitlines()[-1])
        k = d["kernels_us_per_step"]
        print(n, d["ms_per_step"], d["loss"], {p: k[p]["us_per_step"] for p in ("gru.fwd_l0", "gru.fwd_l1", "headb.dw31", "headb.dw2", "headb.unfold", "grub.dx1", "grub.dw_hh", "grub.dw_ih1", "gru.bwd_l0", "gru.bwd_l1") if p in k})
    except Exception as e:
        print(n, "unreadable", e)
PY
