#!/bin/bash
source tools/gpu_steps.sh
out=gpurun_out/r03e; mkdir -p $out
step 400 $out/tests.log python -m pytest tests -m gpu -q -x
tail -5 $out/tests.log
step 120 $out/metrics_kernels.log python tools/bench_metrics_kernels.py 50 --json $out/metrics_kernels.json
grep -v amdgpu $out/metrics_kernels.log
step 300 $out/epoch.log python tools/bench_epoch.py 4
grep -v amdgpu $out/epoch.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step 200 $R/$out/mk_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/mk_stats -- python3 $R/tools/bench_metrics_kernels.py 50
step 200 $R/$out/mk_fetch.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$out/mk_fetch -- python3 $R/tools/bench_metrics_kernels.py 10
step 200 $R/$out/mk_write.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$out/mk_write -- python3 $R/tools/bench_metrics_kernels.py 10
cd $R
ls $out/mk_stats/*/ | head; head -12 $out/mk_stats/*/*kernel_stats.csv | cut -c1-160
echo done
