"""r06 diagnostic: upsample2x / the <= 4-channel convolution / deform_cols on fixed inputs, one build per process (solo, one stream):
   python tools/diag_compare_kernels.py dump <tag> [lib.so] ; python tools/diag_compare_kernels.py diff <a> <b>"""
import os
import shutil
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
LIB = ROOT / "comfyui_propainter_nodes_amd" / "libpropainter_mi355.so"
if sys.argv[1] == "dump":
    swap = len(sys.argv) > 3
    if swap:
        shutil.copy(LIB, "/tmp/product_keep.so")
        shutil.copy(sys.argv[3], LIB)
    try:
        from comfyui_propainter_nodes_amd import lib, ops
        lib.load()
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(9)
        x = torch.randn(3, 90, 160, 64, generator=g).half().to(dev)
        up = ops.upsample2x(x, torch.empty(3, 180, 320, 64, device=dev, dtype=torch.float16))
        ref_up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
        w = torch.randn(3, 64, 3, 3, generator=g) * 0.05
        b = torch.randn(3, generator=g)
        sp = ops.make_conv_spec(w, b, torch.float16, padding=1).to(dev)
        y = torch.empty(3, 180, 320, 4, device=dev, dtype=torch.float16)
        ops.conv2d(sp, [up], y[..., :3], act="tanh")
        ref_y = torch.tanh(F.conv2d(up.float().permute(0, 3, 1, 2), w.half().float().to(dev), b.to(dev), padding=1)).permute(0, 2, 3, 1)
        print(f"upsample2x vs torch (align_corners=True): max {float((up.float() - ref_up).abs().max()):.3e}; 64->3 conv + tanh vs torch: max {float((y[..., :3].float() - ref_y).abs().max()):.3e}")
        # one transformer block, op by op, on fixed tokens (which kernel gives different bits in the two builds?)
        import math
        os.environ["PP_ALLOW_SYNTHETIC_WEIGHTS"] = "1"
        from comfyui_propainter_nodes_amd import generator, pipeline, weights
        gen = pipeline.models_from_state_dicts(weights.synth_state_dicts(0), dev).inpaint_model
        t, h, w = 6, 90, 160
        fh, fw = generator.token_grid(h, w)
        tok = (torch.randn(t, fh, fw, 512, generator=g) * 0.5).half().to(dev)
        WIN = generator.WIN
        Hp, Wp = math.ceil(fh / WIN[0]) * WIN[0], math.ceil(fw / WIN[1]) * WIN[1]
        ph, pw = Hp // 4, Wp // 4
        E = lambda *s: torch.empty(*s, device=dev, dtype=torch.float16)
        xn = torch.zeros(t, Hp, Wp, 512, device=dev, dtype=torch.float16)
        qkv, pooled, pkv, att, yy, f1, folded, tok2, tok3 = (E(t, Hp, Wp, 1536), E(t, ph, pw, 512), E(t, ph, pw, 1024), E(t, fh, fw, 512),
                                                           E(t, fh, fw, 512), E(t, fh, fw, 1960), E(t, h, w, 40), E(t, fh, fw, 512), E(t, fh, fw, 512))
        B = gen.blocks[0]
        nwin = (Hp // WIN[0]) * (Wp // WIN[1])
        flags = (torch.arange(nwin, device=dev) % 3 == 0).to(torch.int32)
        ops.layernorm(tok, xn, B["n1w"], B["n1b"])
        ops.conv2d(B["qkv"], [xn], qkv)
        ops.pool_tokens(xn, pooled, B["pool_w"], B["pool_b"])
        ops.conv2d(B["kv"], [pooled], pkv)
        ops.window_attention(qkv, pkv.view(t, ph * pw, 1024), flags, torch.arange(0, t, 2, dtype=torch.int32, device=dev), att)
        ops.conv2d(B["proj"], [att], tok2, epi="add", aux1=tok)
        ops.layernorm(tok2, yy, B["n2w"], B["n2b"])
        ops.conv2d(B["fc1"], [yy], f1)
        ops.fold(f1.view(t, fh * fw, 1960), folded, fh, fw, True, gelu=True)
        ops.linear_of_unfold(B["fc2"], folded, tok3, 7, 3, 3, epi="add", aux1=tok2)
        torch.save({"up": up.cpu(), "y": y[..., :3].cpu(), "xn": xn.cpu(), "qkv": qkv.cpu(), "pooled": pooled.cpu(), "pkv": pkv.cpu(), "att": att.cpu(),
                    "tok2": tok2.cpu(), "ln2": yy.cpu(), "f1": f1.cpu(), "folded": folded.cpu(), "tok3": tok3.cpu()}, f"/tmp/kcmp_{sys.argv[2]}.pt")
    finally:
        if swap:
            shutil.copy("/tmp/product_keep.so", LIB)
else:
    a, b = torch.load(f"/tmp/kcmp_{sys.argv[2]}.pt"), torch.load(f"/tmp/kcmp_{sys.argv[3]}.pt")
    for k in a:
        d = (a[k].float() - b[k].float()).abs()
        print(f"{k}: equal {bool(torch.equal(a[k], b[k]))}, differing {int((d > 0).sum())} of {d.numel()}, max {float(d.max()):.3e}")
