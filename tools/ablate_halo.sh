#!/bin/bash
# Ablation builds of the halo-tile kernels: whole library with conv_halo.hip / conv_halo_f16.hip compiled -DPP_ABLATE=<mask> (1 no MFMA, 2 no pixel
# loads, 4 no weight copies, 16 no LDS fragment reads, 32 no barriers).  --build here (CPU); without arguments on the MI355X:
# convbench against every variant.  The hooks are not in the product sources: the variants compile from the scratch copy that
# tools/ablate_src.sh patches with tools/ablate_hooks.patch (results of an ablated kernel are meaningless; only the time differences are read).
cd "$(dirname "$0")/.."
PKG=comfyui_propainter_nodes_amd
if [ "${1:-}" = "--build" ]; then
  SRC=$(tools/ablate_src.sh) || exit 1
  for m in 1 2 4 6 16 32; do
    mkdir -p tools/ablate/$m
    for f in conv_halo conv_halo_f16_h conv_halo_f16_f; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPP_ABLATE=$m -I $SRC -I include -c $SRC/$f.hip -o tools/ablate/$m/$f.o &
    done
  done; wait
  for m in 1 2 4 6 16 32; do
    objs=$(ls $PKG/build/hip/*.o | grep -v "conv_halo.o\|conv_halo_f16_h.o\|conv_halo_f16_f.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/ablate/$m/conv_halo.o tools/ablate/$m/conv_halo_f16_h.o tools/ablate/$m/conv_halo_f16_f.o -o tools/ablate/$m/libpropainter_mi355.so && rm tools/ablate/$m/*.o
  done; ls -la tools/ablate/*/; exit 0
fi
O=gpurun_out/ablate_halo; mkdir -p $O
SH="raft_convc2_f32x2 raft_gru_1x5_f32x2 enc_3x3_256_384_f16 dcn_offset_f16"
timeout 60 tools/convbench $SH > $O/m0.json
for m in 1 2 4 6 16 32; do LD_LIBRARY_PATH=tools/ablate/$m timeout 60 tools/convbench $SH > $O/m$m.json 2>&1; done
for m in 0 1 2 4 6 16 32; do echo "mask $m: $(cat $O/m$m.json | python3 -c 'import sys,json; print("  ".join("%s %.3f" % (json.loads(l)["name"][:14], json.loads(l)["ms"]) for l in sys.stdin if l.startswith("{")))')"; done
