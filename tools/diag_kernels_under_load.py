"""Which kernel of the flow-completion step changes its bits when another stream keeps the chip busy?  (MI355X; diagnostic)"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from comfyui_propainter_nodes_amd import lib, ops, pipeline, weights  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
models = pipeline.models_from_state_dicts(weights.synth_state_dicts(0), dev, "enable")
S = models.flow_model.prop["forward_"]
B, h, w = 2, 45, 80
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).half()
prop, cur, n2, t128 = rnd(B, h, w, 128), rnd(B, h, w, 128), rnd(B, h, w, 128), rnd(B, h, w, 128)
bwd = rnd(B, h, w, 128)
om = (torch.randn(B, h, w, 432, device=dev, generator=g) * 2).float()
# a load generator: a large PP_F32X2 convolution on the launch stream
big = torch.randn(64, 90, 160, 128, device=dev)
wl = torch.randn(128, 128, 3, 3) * 0.05
lspec = ops.make_conv_spec(wl, torch.zeros(128), torch.float32, padding=1, split=True).to(dev)
lout = torch.empty(64, 90, 160, 128, device=dev)
side = torch.cuda.Stream(dev)
import os
REPS = int(os.environ.get('REPS', '400'))

only = os.environ.get('ONLY', '')
cases = {
    "off0  3x3 384->128 (split-K)": lambda o: ops.conv2d(S["off0"], [prop, cur, n2], o, act="leaky", act_param=0.1),
    "off2  3x3 128->128 (split-K)": lambda o: ops.conv2d(S["off2"], [t128], o, act="leaky", act_param=0.1),
    "off6  3x3 128->432 f32 out (halo f16)": lambda o: ops.conv2d(S["off6"], [t128], o, act="tanh", out_scale=5.0, act2="sigmoid", act_split=288),
    "dcn   deformable 3x3 256->128 (one launch)": lambda o: ops.deform_conv(S["dcn"], prop, n2, om, o),
    "bb0   3x3 384->128 (split-K)": lambda o: ops.conv2d(S["bb0"], [cur, bwd, prop], o, act="leaky", act_param=0.1),
    "bb2   3x3 128->128 + residual (split-K)": lambda o: ops.conv2d(S["bb2"], [t128], o, epi="add", aux1=prop),
    "off6  without activations": lambda o: ops.conv2d(S["off6"], [t128], o),
    "off6  f16 output": lambda o: ops.conv2d(S["off6"], [t128], o),
    "off2  via the halo ct kernel (128 wide)": lambda o: ops.conv2d(S["off2"], [t128], o, act="leaky", act_param=0.1),
}
for name, fn in cases.items():
    if only and not any(k in name for k in only.split(',')):
        continue
    odt, oc = (torch.float32, 432) if "off6" in name else (torch.float16, 128)
    if "f16 output" in name:
        odt = torch.float16
    if "via the halo" in name:
        os.environ["PP_CONV_KSPLIT"] = "0"; os.environ["PP_CONV_HALO"] = "force"; lib.reload_options()
    first_bad = None
    ref = torch.empty(B, h, w, oc, device=dev, dtype=odt)
    fn(ref)
    torch.cuda.synchronize()
    bad = 0
    worst = 0.0
    for rep in range(REPS):
        out = torch.full_like(ref, float("nan"))
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            for _ in range(5):
                fn(out)
        ops.conv2d(lspec, [big], lout)          # ~1 ms of MFMA work next to it
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            if first_bad is None:
                d = (out.float() - ref.float()).abs().nan_to_num(1e9)
                idx = torch.nonzero(d > 0)
                first_bad = (int(idx.shape[0]), idx[0].tolist(), idx[-1].tolist())
            worst = max(worst, float((out.float() - ref.float()).abs().nan_to_num(1e9).max()))
    print(f"{name:45s}: {bad:3d} of {REPS} runs (x5 launches) under load differ from the quiet run (worst |diff| {worst:.3e}; first bad run: #elements, first, last index {first_bad})", flush=True)
