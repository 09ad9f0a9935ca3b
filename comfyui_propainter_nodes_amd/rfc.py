"""Recurrent flow completion on the MI355X (fp16 activations / fp32 accumulate, like the reference
runs this net `.half()`: utils/model_utils.py:55-58).

Replaces RecurrentFlowCompleteNet.forward_bidirect_flow + combine_flow
(model/recurrent_flow_completion.py:315-400).  Scheduling decisions:

  * the forward-flow and the (time-flipped) backward-flow applications share weights, so they
    run as ONE batch of 2 through every kernel, laid out [T][2][h][w][C] so that each step of
    the second-order recurrence (:96-131) is a dense batch of 2 images;
  * Conv3d(1,k,k) layers are 2-D convolutions over the T*2 images, Conv3d(3,1,1,dilation 2)
    layers are 2-D convolutions over a [T] x [2*h*w] "image" (kernel 3x1) -- same MFMA kernel;
  * every torch.cat is a K-segment list, the deformable conv is sampling (pp_deform_cols) +
    the same GEMM kernel, residual adds and activations are conv epilogues.
"""
from __future__ import annotations

import torch

from . import graphs, ops

F16 = torch.float16


class FlowCompleter:
    def __init__(self, sd: dict, device, dtype: torch.dtype = F16):
        """`dtype` = storage type of the activations: f16 (the node's fp16 "enable": the reference runs this net
        `.half()`) or f32 (fp16 "disable": fp32 like the reference; the convolutions then multiply on the f16 matrix pipe
        with the two-term operand split of PP_F32X2, or on the f32 MFMA instructions under PP_F32_GEMM=exact)."""
        self.device = torch.device(device)
        self.dt = dtype
        self.split = dtype == torch.float32 and ops.f32_split_enabled()
        self._graphs = graphs.GraphCache()
        p = {k: v.float() for k, v in sd.items() if not k.startswith("edgeDetector")}

        def c2(name, **kw):  # Conv3d (1,k,k) or Conv2d
            w = p[name + ".weight"]
            if w.dim() == 5:
                w = w[:, :, 0]
            return ops.make_conv_spec(w, p[name + ".bias"], self.dt, split=self.split, **kw).to(device)

        def ct(name):  # Conv3d (3,1,1), dilation 2, padding 2 -> kernel 3x1 over [T] x [pixels]
            w = p[name + ".weight"][:, :, :, 0, 0].unsqueeze(-1)  # [Co,Ci,3,1]
            return ops.make_conv_spec(w, p[name + ".bias"], self.dt, padding=(2, 0), dilation=(2, 1), split=self.split).to(device)

        wd = p["downsample.0.weight"][:, :, 0]  # [32,3,5,5] -> im2col GEMM (replicate padding)
        self.down_kpad = ops.pad32(75)
        self.down = ops.make_conv_spec(wd.permute(0, 2, 3, 1).reshape(32, 75, 1, 1), p["downsample.0.bias"], self.dt,
                                       seg_channels=[self.down_kpad], seg_valid=[75], split=self.split).to(device)
        self.enc = [
            (c2("encoder1.0.conv1.0", padding=1, many_images=True), ct("encoder1.0.conv2.0")),
            (c2("encoder1.2.conv1.0", stride=2, padding=1, many_images=True), ct("encoder1.2.conv2.0")),
            (c2("encoder2.0.conv1.0", padding=1, many_images=True), ct("encoder2.0.conv2.0")),
            (c2("encoder2.2.conv1.0", stride=2, padding=1, many_images=True), ct("encoder2.2.conv2.0")),
        ]
        # r06 (ABI v12 `many_images`): the layers that see every image of the sub-video at once.  At 640x360 their 45 x 80 images and
        # 36-chunk reductions look like the recurrences' one-time-step launches to a selection rule that never looks at the batch, and
        # they ran on the split-K kernel at 224 TF/s (0.75 ms each, 3 ms per clip); with the hint: the halo tiles, ~800 TF/s
        self.mid = [c2(f"mid_dilation.{i}", padding=d, dilation=d, many_images=True) for i, d in ((0, 3), (2, 2), (4, 1))]
        fp = "feat_prop_module."
        self.prop = {}
        for name, nseg in (("backward_", 2), ("forward_", 3)):
            da = f"{fp}deform_align.{name}."
            wm = p[da + "weight"]  # [128,256,3,3] -> GEMM over the sampled columns (tap-major)
            self.prop[name] = {
                "off0": c2(da + "conv_offset.0", padding=1, seg_channels=[128, 128, 128]),
                "off2": c2(da + "conv_offset.2", padding=1),
                "off4": c2(da + "conv_offset.4", padding=1),
                "off6": c2(da + "conv_offset.6", padding=1),
                "dcn": ops.make_conv_spec(wm.permute(0, 2, 3, 1).reshape(128, 9 * 256, 1, 1), p[da + "bias"], self.dt, split=self.split).to(device),
                "bb0": c2(f"{fp}backbone.{name}.0", padding=1, seg_channels=[128] * nseg),
                "bb2": c2(f"{fp}backbone.{name}.2", padding=1),
            }
        self.fusion = c2(fp + "fusion", seg_channels=[128, 128], many_images=True)
        self.dec2_0 = c2("decoder2.0", padding=1, many_images=True)
        self.dec2_2 = c2("decoder2.2.conv", padding=1, many_images=True)
        self.dec1_0 = c2("decoder1.0", padding=1, many_images=True)
        self.dec1_2 = c2("decoder1.2.conv", padding=1, many_images=True)
        self.up_0 = c2("upsample.0", padding=1, many_images=True)
        self.up_2 = c2("upsample.2.conv", padding=1, many_images=True)

    # ------------------------------------------------------------------------------------
    def _encode(self, x: torch.Tensor):
        """x f16 [T,2,H,W,4] -> (e1 [T*2,H/4,W/4,64], mid [T,2,H/8,W/8,128])."""
        dev = x.device
        T, B, H, W, _ = x.shape
        n = T * B
        h2, w2 = (H + 4 - 5) // 2 + 1, (W + 4 - 5) // 2 + 1
        cols = torch.empty(n, h2, w2, self.down_kpad, device=dev, dtype=self.dt)
        ops.im2col(x.view(n, H, W, 4)[..., 0:3], cols, 5, 5, stride=2, padding=2, pad_mode="replicate")
        cur = torch.empty(n, h2, w2, 32, device=dev, dtype=self.dt)
        ops.conv2d(self.down, [cols], cur, act="leaky", act_param=0.2)
        del cols
        e1 = None
        for li, (sp, tp) in enumerate(self.enc):
            _, h, w, _ = cur.shape
            ho, wo = sp.out_hw(h, w)
            a = torch.empty(n, ho, wo, sp.cout, device=dev, dtype=self.dt)
            ops.conv2d(sp, [cur], a, act="leaky", act_param=0.2)
            b = torch.empty(n, ho, wo, sp.cout, device=dev, dtype=self.dt)
            # temporal conv: rows = T, columns = B*ho*wo pixels
            ops.conv2d(tp, [a.view(1, T, B * ho * wo, sp.cout)], b.view(1, T, B * ho * wo, sp.cout), act="leaky",
                       act_param=0.2)
            cur = b
            if li == 1:
                e1 = b
        for sp in self.mid:
            nxt = torch.empty_like(cur)
            ops.conv2d(sp, [cur], nxt, act="leaky", act_param=0.2)
            cur = nxt
        _, h8, w8, _ = cur.shape
        return e1, cur.view(T, B, h8, w8, 128)

    def _propagate(self, mid: torch.Tensor, capture: bool = True) -> torch.Tensor:
        """The two sweeps are ~16 small dependent launches per frame: captured once per clip shape into a hipGraph and
        replayed (graphs.py); the result lives in the graph's static buffer until the next call with this shape."""
        if not capture:
            return self._propagate_eager(mid)
        return self._graphs.run(("rfc_propagate", ops.deform_fused(mid.shape[2], mid.shape[3])), self._propagate_eager, mid)

    def _propagate_eager(self, mid: torch.Tensor) -> torch.Tensor:
        """BidirectionalPropagation.forward (:77-143) on mid [T,2,h,w,128] -> [T,2,h,w,128]."""
        dev = mid.device
        T, B, h, w, C = mid.shape
        outs = {}
        zeros = torch.zeros(B, h, w, C, device=dev, dtype=self.dt)
        t128 = torch.empty(B, h, w, 128, device=dev, dtype=self.dt)
        u128 = torch.empty(B, h, w, 128, device=dev, dtype=self.dt)
        om = torch.empty(B, h, w, 432, device=dev, dtype=torch.float32)
        fused = self.dt == torch.float16 and ops.deform_fused(h, w)
        cols = None if fused else torch.empty(B, h, w, 9 * 256, device=dev, dtype=self.dt)
        aligned = torch.empty(B, h, w, 128, device=dev, dtype=self.dt)
        for name in ("backward_", "forward_"):
            S = self.prop[name]
            order = list(range(T - 1, -1, -1)) if name == "backward_" else list(range(T))
            out = torch.empty(T, B, h, w, C, device=dev, dtype=self.dt)
            prop = zeros
            for i, idx in enumerate(order):
                cur = mid[idx]
                if i > 0:
                    n2 = out[order[i - 2]] if i > 1 else zeros
                    ops.conv2d(S["off0"], [prop, cur, n2], t128, act="leaky", act_param=0.1)
                    ops.conv2d(S["off2"], [t128], u128, act="leaky", act_param=0.1)
                    ops.conv2d(S["off4"], [u128], t128, act="leaky", act_param=0.1)
                    # offset = 5*tanh(first 288), mask = sigmoid(last 144)  (:36-42)
                    ops.conv2d(S["off6"], [t128], om, act="tanh", out_scale=5.0, act2="sigmoid", act_split=288)
                    if cols is None:   # one launch, no column tensor (pp_deform_conv)
                        ops.deform_conv(S["dcn"], prop, n2, om, aligned)
                    else:
                        ops.deform_cols(prop, n2, om, cols)
                        ops.conv2d(S["dcn"], [cols], aligned)
                    prop = aligned
                segs = [cur] + ([outs["backward_"][idx]] if name == "forward_" else []) + [prop]
                ops.conv2d(S["bb0"], segs, t128, act="leaky", act_param=0.1)
                ops.conv2d(S["bb2"], [t128], out[idx], epi="add", aux1=prop)
                prop = out[idx]
            outs[name] = out
        fused = torch.empty(T * B, h, w, C, device=dev, dtype=self.dt)
        ops.conv2d(self.fusion, [outs["backward_"].view(T * B, h, w, C), outs["forward_"].view(T * B, h, w, C)], fused,
                   epi="add", aux1=mid.view(T * B, h, w, C))
        return fused

    def _decode(self, prop: torch.Tensor, e1: torch.Tensor) -> torch.Tensor:
        dev = prop.device
        n, h, w, _ = prop.shape

        def new(hh, ww, c):
            return torch.empty(n, hh, ww, c, device=dev, dtype=self.dt)

        a = ops.conv2d(self.dec2_0, [prop], new(h, w, 128), act="leaky", act_param=0.2)
        up = ops.upsample2x(a, new(2 * h, 2 * w, 128))
        b = ops.conv2d(self.dec2_2, [up], new(2 * h, 2 * w, 64), act="leaky", act_param=0.2, epi="add", aux1=e1)
        c = ops.conv2d(self.dec1_0, [b], new(2 * h, 2 * w, 64), act="leaky", act_param=0.2)
        up = ops.upsample2x(c, new(4 * h, 4 * w, 64))
        d = ops.conv2d(self.dec1_2, [up], new(4 * h, 4 * w, 32), act="leaky", act_param=0.2)
        e = ops.conv2d(self.up_0, [d], new(4 * h, 4 * w, 32), act="leaky", act_param=0.2)
        up = ops.upsample2x(e, new(8 * h, 8 * w, 32))
        return ops.conv2d(self.up_2, [up], new(8 * h, 8 * w, 2))

    def __call__(self, flows: torch.Tensor, masks_u8: torch.Tensor, trace: dict | None = None) -> torch.Tensor:
        """flows fp32 [2,T,H,W,2] (forward, backward), masks u8 [T+1,H,W] (flow masks of the T+1 frames)
        -> completed + combined flows fp32 [2,T,H,W,2]."""
        _, T, H, W, _ = flows.shape
        x = torch.empty(T, 2, H, W, 4, device=flows.device, dtype=self.dt)
        ops.rfc_prep(flows, masks_u8, x)
        e1, mid = self._encode(x)
        prop = self._propagate(mid, capture=trace is None)
        pred = self._decode(prop, e1).view(T, 2, H, W, 2)
        if trace is not None:
            trace.update(mid=mid, prop=prop, pred=pred)
        out = torch.empty_like(flows)
        ops.flow_combine(pred, flows, masks_u8, out)
        return out
