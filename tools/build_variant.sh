#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>  -> matdeeplearn_amd/lib/variants/<name>.so (for A/B runs via MDL_HIP_LIB)
# the extra flags go to the two CGConv translation units (cgconv.hip, cgconv_ep.hip), or to the ones named in VAR_FILES
# (e.g. VAR_FILES="linear gemm_tn")
set -e
name=$1; shift
cd /root/repo/matdeeplearn_amd
mkdir -p lib/variants lib/obj_$name
for f in csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""; { [ "$b" = "cgconv_ep" ] || [ "$b" = "cfconv" ]; } && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  for vf in ${VAR_FILES:-cgconv cgconv_ep}; do [ "$b" = "$vf" ] && extra="$extra $@"; done
  [ -n "${NO_VGPR_FORM:-}" ] && [ "$b" = "cgconv_ep" ] && extra="$@"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $extra -c $f -o lib/obj_$name/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/$name.so lib/obj_$name/*.o
rm -rf lib/obj_$name
echo built lib/variants/$name.so
