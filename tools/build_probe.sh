#!/bin/bash
# builds tools/_build/libsplit_probe.so (+ its ISA listing) from tools/split_gemm_probe.hip
set -e
cd "$(dirname "$0")"
mkdir -p _build
hipcc --offload-arch=gfx950 -O3 -shared -fPIC -save-temps=obj split_gemm_probe.hip -o _build/libsplit_probe.so
cd _build && rm -f *.bc *.hipi *.o *.out *.txt *host*
grep -E "\.vgpr_count|vgpr_spill_count" *.s
