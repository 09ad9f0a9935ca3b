#!/bin/bash
# A/B of library variants (tools/variants/<name>.so) against the product with the C++ client only (no Python, starts in well under a
# second): tools/ab_convbench.sh "<variant names>" <convbench shapes...>; prints time, TF/s and a hash of the output bits per shape.
set -u
cd "$GRAFT_REPO_ROOT"
L=comfyui_propainter_nodes_amd/libpropainter_mi355.so
VARIANTS=$1; shift
cp $L /tmp/product.so
export PP_CONVBENCH_SUM=1
for v in product $VARIANTS product; do
  [ $v = product ] && cp /tmp/product.so $L || cp tools/variants/$v.so $L
  echo "== $v"; timeout 40 tools/convbench "$@" 2>&1 | tail -8
done
cp /tmp/product.so $L
