#!/bin/bash
# Build a VARIANT of the library for an A/B run on the MI355X: one translation unit recompiled with extra -D flags, linked with
# the product's other objects into tools/variants/<name>.so (git-ignored, travels with gpurun).  Usage:
#   tools/build_variant.sh xfetch conv_halo -DPP_HALO_XFETCH_EARLY
# The A/B script then swaps it in:  cp tools/variants/xfetch.so comfyui_propainter_nodes_amd/libpropainter_mi355.so
set -eu
name=$1; tu=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/comfyui_propainter_nodes_amd/build/hip
mkdir -p $root/tools/variants /tmp/variant_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=off -fno-slp-vectorize -I $root/comfyui_propainter_nodes_amd/csrc -I $root/include "$@" \
  -c $root/comfyui_propainter_nodes_amd/csrc/$tu.hip -o /tmp/variant_$name/$tu.o
objs=$(for f in $root/comfyui_propainter_nodes_amd/csrc/*.hip; do b=$(basename $f .hip); [ $b = $tu ] || echo $obj/$b.o; done)   # (only objects of current sources)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_$name/$tu.o -o $root/tools/variants/$name.so
ls -la $root/tools/variants/$name.so
