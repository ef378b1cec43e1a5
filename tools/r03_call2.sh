#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03b; mkdir -p $out
step 400 $out/tests.log python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_transformer.py::test_transformer_loops_match_reference_fixture
tail -5 $out/tests.log
step 100 $out/dbg_nan.log python tools/scratch/dbg_nan.py
cat $out/dbg_nan.log | grep -v amdgpu
step 200 $out/bench.json python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras
export ARTSPEECH_DIAG_LIB=1
step 120 $out/heads.log python tools/bench_heads.py 20
cat $out/heads.log | grep -v amdgpu
for c in 8 10 12 16 20 25 34; do
  step 200 $out/bench_chunk$c.json env AS_WGRAD_MULTI_CHUNK=$c python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras
done
step 200 $out/bench_nomulti.json env AS_NO_WGRAD_MULTI=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras
python - <<'PY'
import json
for n in ["bench", "bench_nomulti"] + [f"bench_chunk{c}" for c in (8, 10, 12, 16, 20, 25, 34)]:
    try:
        d = json.loads(open(f"gpurun_out/r03b/{n}.json").read().strip().splitlines()[-1])
        k = d["kernels_us_per_step"]
        print(n, d["ms_per_step"], {p: k[p]["us_per_step"] for p in ("headb.dw_fused", "headb.dw2", "headb.unfold", "grub.dx1", "grub.dw_hh", "grub.dw_ih1", "trunkb.dw", "gru.bwd_l0", "gru.bwd_l1") if p in k})
    except Exception as e:
        print(n, "unreadable", e)
PY
echo done
