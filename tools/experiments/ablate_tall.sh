#!/bin/bash
# ARCHIVED with conv_halo_tall.hip (not built any more; needs the hooks of tools/ablate_hooks.patch in pp_device.h).
# Ablation builds of the 16-row halo kernel (conv_halo_tall.hip -DPP_ABLATE=<mask>: 1 no MFMA, 2 no pixel loads, 4 no weight
# copies, 16 no LDS fragment reads, 32 no barriers, 64 no pixel split/store).  --build here (CPU); without arguments on the MI355X.
cd "$(dirname "$0")/.."
PKG=comfyui_propainter_nodes_amd
MASKS="1 2 4 6 16 32 64 118"
if [ "${1:-}" = "--build" ]; then
  for m in $MASKS; do
    mkdir -p tools/ablate/t$m
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPP_ABLATE=$m -I $PKG/csrc -I include -c $PKG/csrc/conv_halo_tall.hip -o tools/ablate/t$m/conv_halo_tall.o &
  done; wait
  for m in $MASKS; do
    objs=$(ls $PKG/build/hip/*.o | grep -v "conv_halo_tall.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/ablate/t$m/conv_halo_tall.o -o tools/ablate/t$m/libpropainter_mi355.so && rm tools/ablate/t$m/*.o
  done; ls -la tools/ablate/t*/; exit 0
fi
O=gpurun_out/ablate_tall; mkdir -p $O
export PP_CONV_HALO_TALL=force
SH="raft_convc2_f32x2 raft_gru_1x5_f32x2 raft_fh1_f32x2"
timeout 60 tools/convbench $SH > $O/m0.json
for m in $MASKS; do LD_LIBRARY_PATH=tools/ablate/t$m timeout 60 tools/convbench $SH > $O/m$m.json 2>&1; done
for m in 0 $MASKS; do echo "mask $m: $(cat $O/m$m.json | python3 -c 'import sys,json; print("  ".join("%s %.3f" % (json.loads(l)["name"][:14], json.loads(l)["ms"]) for l in sys.stdin if l.startswith("{")))')"; done
