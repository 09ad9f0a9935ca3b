#!/bin/bash
cd "$GRAFT_REPO_ROOT"
SH="raft_gru_1x5_f32x2 raft_convc2_f32x2 enc_3x3_256_384_f16 dcn_offset_f16"
for d in 0 4 8 16 32; do echo "dephase $d: $(PP_CONV_DEPHASE=$d timeout 60 tools/convbench $SH | python3 -c 'import sys,json; print("  ".join("%s %.3f" % (json.loads(l)["name"][:14], json.loads(l)["ms"]) for l in sys.stdin if l.startswith("{")))')"; done
