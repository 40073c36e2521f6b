#!/bin/bash
# Development: build a variant of libhpvg.so with extra -D flags on ONE source (ablations / A-B builds), in-tree so that it
# travels to the GPU box: usage tools/build_variant.sh <name> <source.hip> <flags...>  ->  hp-vae-gan_amd/build/libhpvg_<name>.so
# (select it with HPVG_LIB=hp-vae-gan_amd/build/libhpvg_<name>.so)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
B=hp-vae-gan_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include "$@" -c hp-vae-gan_amd/csrc/$src -o $B/${src}_$name.o
objs=""
for s in conv_mfma.hip conv_wgrad.hip elementwise.hip frames.hip graph.hip variants.hip; do
  if [ "$s" == "$src" ]; then objs="$objs $B/${src}_$name.o"; else objs="$objs $B/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libhpvg_$name.so $objs
echo built $B/libhpvg_$name.so
