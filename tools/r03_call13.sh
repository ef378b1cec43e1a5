#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03m; mkdir -p $out
export ARTSPEECH_DIAG_LIB=1
run() { step 200 $out/bench_$1.json env $2 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras; }
run base X=1
run dx32 AS_DX1_LIN=32
run dx64 AS_DX1_LIN=64
run xp64 AS_XPROJ_LIN=64
run xp32 AS_XPROJ_LIN=32
python - <<'PY'
import json
for n in ("base", "dx32", "dx64", "xp64", "xp32"):
    d = json.loads([l for l in open(f"gpurun_out/r03m/bench_{n}.json").read().strip().splitlines() if l.startswith("{")][-1])
    k = d["kernels_us_per_step"]
    print(n, d["ms_per_step"], d["loss"], {p: v["us_per_step"] for p, v in k.items() if p in ("gru.xproj1", "grub.dx1", "gru.fwd_l1", "gru.bwd_l0")})
PY
