#!/bin/bash
# asm_stats.sh [extra hipcc flags]: compile cgconv.hip (fast bf16 instantiation only) to /tmp/prod.s and print register/spill stats
cd /root/repo/matdeeplearn_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DMDL_CG_FAST_ONLY "$@" -S --cuda-device-only -o /tmp/prod.s csrc/cgconv.hip 2>&1 | grep -E "error"
awk '/^_ZN3mdl17cgconv_(bwd|fwd)_kernelItLi64/{k=substr($1,1,28)} /; NumVgprs:|; NumAgprs:|ScratchSize|; Occupancy/{if (k!="") printf "%s %s ", k, $0; if ($0 ~ /Occupancy/) {print ""; k=""}}' /tmp/prod.s
