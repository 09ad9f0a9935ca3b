#!/bin/bash
# Prints the path of a scratch copy of csrc/ with the PP_ABLATE profiling hooks re-inserted (tools/ablate_hooks.patch).  The product
# sources carry no ablation code; the ablation tools compile their variants from this copy with -DPP_ABLATE=<mask>.
set -e
cd "$(dirname "$0")/.."
S=tools/ablate/src; rm -rf $S; mkdir -p $S/comfyui_propainter_nodes_amd
cp -r comfyui_propainter_nodes_amd/csrc $S/comfyui_propainter_nodes_amd/; cp -r include $S/
(cd $S && patch -s -p1 < ../../ablate_hooks.patch)
grep -q "PP_ABLATE & 1" $S/comfyui_propainter_nodes_amd/csrc/pp_device.h
echo "$PWD/$S/comfyui_propainter_nodes_amd/csrc"
