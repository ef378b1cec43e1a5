#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03c; mkdir -p $out
step 300 $out/tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x
tail -3 $out/tests.log
step 200 $out/bench.json python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
export ARTSPEECH_DIAG_LIB=1
step 200 $out/bench_one.json env AS_HEAD_DW_ONE=1 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras
cd /tmp && export TMPDIR=/tmp
unset ARTSPEECH_DIAG_LIB
step 300 $GRAFT_REPO_ROOT/$out/kt.log rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-extras
cd $GRAFT_REPO_ROOT
python3 tools/step_timeline.py $out/kt/*/*_kernel_trace.csv > $out/step_timeline.txt
cat $out/step_timeline.txt
python - <<'PY'
import json
for n in ["bench", "bench_one"]:
    try:
        d = json.loads(open(f"gpurun_out/r03c/{n}.json").read().strip().splitlines()[-1])
        k = d["kernels_us_per_step"]
        print(n, d["ms_per_step"], {p: k[p]["us_per_step"] for p in ("headb.dw_fused", "headb.dw31", "headb.dw2", "headb.unfold", "grub.dx1", "grub.dw_hh", "grub.dw_ih1", "trunkb.dw", "gru.bwd_l0", "gru.bwd_l1") if p in k})
    except Exception as e:
        print(n, "unreadable", e)
PY
