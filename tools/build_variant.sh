#!/bin/bash
# Builds a variant of the library for A/B runs on the GPU box: tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>
# -> mb-vae-dti_amd/variants/libmbvae_<name>.so = the in-tree objects with <source.hip> recompiled under the extra flags
# (select it with MBVAE_LIB=<path>).  Run `python mb-vae-dti_amd/build.py` first.  Base flags as in build.py.
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)/mb-vae-dti_amd
mkdir -p $root/variants/obj_$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=on -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c $root/csrc/$src -o $root/variants/obj_$name/${src%.hip}.o
objs=""
for o in $root/build/*.o; do
  b=$(basename $o)
  if [ "$b" = "${src%.hip}.o" ]; then objs="$objs $root/variants/obj_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/variants/libmbvae_$name.so $objs
echo $root/variants/libmbvae_$name.so
