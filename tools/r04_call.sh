#!/bin/bash
source $GRAFT_REPO_ROOT/tools/gpu_steps.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
F="--no-extras --no-cpu-baseline --no-exact --steps 50 --warmup 10"
step 200 $O/p_new.log python bench.py $F
python3 - <<'PY'
import json
l=[x for x in open("gpurun_out/p_new.log") if x.startswith('{')][-1]
d=json.loads(l)
k=d['kernels_us_per_step']
print(d['ms_per_step'], {n:k[n]['us_per_step'] for n in ['gru.fwd_l0','gru.bwd_l1','gru.bwd_l0','grub.dx1','trunkb.dx']}, d['roofline']['us_per_launch'])
PY
