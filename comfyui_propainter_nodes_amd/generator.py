"""ProPainter InpaintGenerator on the MI355X (f16 activations, fp32 accumulate / statistics /
coordinates), replacing InpaintGenerator.forward (model/propainter.py:358-453) and the per-window
work of feature_propagation (propainter_inference.py:254-281).

Scheduling decisions (all numerically equivalent re-orderings of the reference):
  * the encoder is per-frame work: it runs ONCE per clip frame (`prepare_clip`), the reference
    re-encodes every frame for each of the 2-3 windows it appears in;
  * 1/4-resolution flows, fb-consistency planes and mask planes are per-clip as well;
  * SoftComp and the decoder only touch the l_t local frames whose output is used (:450-451);
  * torch.cat -> K segments, residual adds / activations -> conv epilogues, Linear -> 1x1
    pp_conv2d, F.fold/unfold -> gather kernels on TAP-MAJOR token vectors (weights permuted here).
"""
from __future__ import annotations

import math
import os

import torch

from . import graphs, ops

F16 = torch.float16
WIN = (5, 9)


def token_grid(h: int, w: int) -> tuple[int, int]:
    return (h + 6 - 7) // 3 + 1, (w + 6 - 7) // 3 + 1


class ClipState:
    """Per-clip tensors shared by all windows (all on the device)."""

    def __init__(self):
        self.enc = None        # f16 [T,h,w,128] encoder features
        self.flow_f = None     # fp32 [T-1,h,w,2] 1/4-res completed forward flows (/4)
        self.flow_b = None
        self.aux_b = None      # f16 [T-1,h,w,8] (flow_f, fb-valid, masks of frame t)    -> backward pass at frame t
        self.aux_f = None      # f16 [T-1,h,w,8] (flow_b, fb-valid, masks of frame t+1)  -> forward pass at frame t+1
        self.maskpair = None   # f16 [T,h,w,8] (m_in, m_updated, 0...)
        self.tokmask = None    # u8 [T,fh,fw]: MaxPool2d(7,3,3) of the 1/4-res dilated mask
        self.H = self.W = 0
        # static masks (one MASK frame replicated over the clip, or the two border planes of an outpaint canvas): the token
        # mask is the same in every frame, so the masked-window set is ONE constant of the clip -- or, with a hashable key
        # (outpaint geometry), of every clip with that geometry
        self.static_masks = False
        self.flags = None
        # r06: soft-split tokens of the REFERENCE frames, computed once per clip frame (a reference frame enters the transformer
        # of ~8 windows with the same encoder features, so its tokens are the same every time): frame -> row of ref_tok
        self.ref_tok = None      # f16 [n, fh, fw, 512]
        self.ref_row: dict[int, int] = {}


class InpaintGeneratorMI355:
    def __init__(self, sd: dict, device, dtype: torch.dtype = F16):
        """`dtype` = storage type of the activations: f16 (the node's fp16 "enable": the reference runs the generator
        `.half()`) or f32 (fp16 "disable").  In f32 the convolutions / linears multiply with the two-term operand split of
        PP_F32X2 and the attention core rounds q, k, v and the probabilities to f16 for its MFMAs (scores, statistics,
        accumulators and outputs in fp32); under PP_F32_GEMM=exact both run on the f32 MFMA instructions with fp32 operands
        (r06: ops.attention_exact_enabled(), ABI v11) -- the reference's fp32 generator at fp32 rounding level."""
        self.device = torch.device(device)
        self.dt = dtype
        self._geometry_flags: dict = {}   # outpaint geometry -> masked-window flags (window_mask_flags)
        self.split = dtype == torch.float32 and ops.f32_split_enabled()
        self._graphs = graphs.GraphCache()
        self._side = None   # second stream of propagate_windows()
        p = {k: v.float() for k, v in sd.items() if v.is_floating_point()}
        dev = device

        def conv(name, **kw):
            return ops.make_conv_spec(p[name + ".weight"], p[name + ".bias"], self.dt, split=self.split, **kw).to(dev)

        def linear(w, b, **kw):
            return ops.make_conv_spec(w.reshape(w.shape[0], w.shape[1], 1, 1), b, self.dt, split=self.split, **kw).to(dev)

        # ---- encoder (propainter.py:234-275) ----------------------------------------------------
        w0 = p["encoder.layers.0.weight"]  # [64,5,3,3] -> im2col over the 5 packed channels
        self.enc0_kpad = ops.pad32(45)
        self.enc0 = ops.make_conv_spec(w0.permute(0, 2, 3, 1).reshape(64, 45, 1, 1), p["encoder.layers.0.bias"], self.dt,
                                       seg_channels=[self.enc0_kpad], seg_valid=[45], split=self.split).to(dev)
        self.enc2 = conv("encoder.layers.2", padding=1)
        self.enc4 = conv("encoder.layers.4", stride=2, padding=1)
        self.enc6 = conv("encoder.layers.6", padding=1)
        self.enc8 = conv("encoder.layers.8", padding=1)
        # grouped layers read [x0 group | running group] per group (:269-273)
        self.enc10 = conv("encoder.layers.10", padding=1, groups=2, seg_channels=[128, 192])
        self.enc12 = conv("encoder.layers.12", padding=1, groups=4, seg_channels=[64, 128])
        self.enc14 = conv("encoder.layers.14", padding=1, groups=8, seg_channels=[32, 48])
        self.enc16 = conv("encoder.layers.16", padding=1, seg_channels=[256, 256])
        # ---- feature propagation (propainter.py:85-231) -------------------------------------------
        fp = "feat_prop_module."
        self.prop = {}
        for name in ("backward_1", "forward_1"):
            da = f"{fp}deform_align.{name}."
            self.prop[name] = {
                "off0": conv(da + "conv_offset.0", padding=1, seg_channels=[128, 128, 8], seg_valid=[128, 128, 5]),
                "off2": conv(da + "conv_offset.2", padding=1),
                "off4": conv(da + "conv_offset.4", padding=1),
                "off6": conv(da + "conv_offset.6", padding=1),
                "dcn": ops.make_conv_spec(p[da + "weight"].permute(0, 2, 3, 1).reshape(128, 9 * 128, 1, 1), p[da + "bias"],
                                          self.dt, split=self.split).to(dev),
                "bb0": conv(f"{fp}backbone.{name}.0", padding=1, seg_channels=[128, 128, 8], seg_valid=[128, 128, 2]),
                "bb2": conv(f"{fp}backbone.{name}.2", padding=1),
            }
        self.fuse0 = conv(fp + "fuse.0", padding=1, seg_channels=[128, 128, 8], seg_valid=[128, 128, 2])
        self.fuse2 = conv(fp + "fuse.2", padding=1)
        # ---- soft split / composition (sparse_transformer.py:8-64) --------------------------------
        self.ss = ops.make_conv_spec(p["ss.embedding.weight"].view(512, 128, 7, 7), p["ss.embedding.bias"], self.dt, stride=3,
                                     padding=3, split=self.split).to(dev)
        wsc = p["sc.embedding.weight"].view(128, 49, 512).permute(1, 0, 2).reshape(6272, 512)  # rows -> tap-major
        bsc = p["sc.embedding.bias"].view(128, 49).t().reshape(6272)
        self.sc = linear(wsc, bsc)
        self.sc_bias_conv = conv("sc.bias_conv", padding=1)
        # ---- transformer blocks ---------------------------------------------------------------------
        self.blocks = []
        for i in range(8):
            t = f"transformers.transformer.{i}."
            a = t + "attention."
            wqkv = torch.cat([p[a + "query.weight"], p[a + "key.weight"], p[a + "value.weight"]], 0)
            bqkv = torch.cat([p[a + "query.bias"], p[a + "key.bias"], p[a + "value.bias"]], 0)
            wkv = torch.cat([p[a + "key.weight"], p[a + "value.weight"]], 0)
            bkv = torch.cat([p[a + "key.bias"], p[a + "value.bias"]], 0)
            w1 = p[t + "mlp.fc1.0.weight"].view(40, 49, 512).permute(1, 0, 2).reshape(1960, 512)
            b1 = p[t + "mlp.fc1.0.bias"].view(40, 49).t().reshape(1960)
            w2 = p[t + "mlp.fc2.1.weight"].view(512, 40, 49).permute(0, 2, 1).reshape(512, 1960)
            self.blocks.append({
                "qkv": linear(wqkv, bqkv), "kv": linear(wkv, bkv),
                "proj": linear(p[a + "proj.weight"], p[a + "proj.bias"]),
                "pool_w": p[a + "pool_layer.weight"].view(512, 16).t().contiguous().to(dev),
                "pool_b": p[a + "pool_layer.bias"].contiguous().to(dev),
                "n1w": p[t + "norm1.weight"].to(dev), "n1b": p[t + "norm1.bias"].to(dev),
                "n2w": p[t + "norm2.weight"].to(dev), "n2b": p[t + "norm2.bias"].to(dev),
                "fc1": linear(w1, b1), "fc2": linear(w2, p[t + "mlp.fc2.1.bias"]),
            })
        # ---- decoder (propainter.py:304-312) ----------------------------------------------------------
        self.dec0 = conv("decoder.0.conv", padding=1)
        self.dec2 = conv("decoder.2", padding=1)
        self.dec4 = conv("decoder.4.conv", padding=1)
        self.dec6 = conv("decoder.6", padding=1)

    # ------------------------------------------------------------------------------------------------
    def encode(self, packed: torch.Tensor, chunk: int = 16) -> torch.Tensor:
        """packed f16 [T,H,W,8] (rgb, m_in, m_updated, 0..) -> f16 [T,H/4,W/4,128]."""
        dev = packed.device
        T, H, W, _ = packed.shape
        h2, w2 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        h4, w4 = (h2 + 2 - 3) // 2 + 1, (w2 + 2 - 3) // 2 + 1
        out = torch.empty(T, h4, w4, 128, device=dev, dtype=self.dt)
        for s in range(0, T, chunk):
            e = min(T, s + chunk)
            n = e - s

            def new(hh, ww, c):
                return torch.empty(n, hh, ww, c, device=dev, dtype=self.dt)

            cols = ops.im2col(packed[s:e][..., 0:5], new(h2, w2, self.enc0_kpad), 3, 3, stride=2, padding=1)
            a = ops.conv2d(self.enc0, [cols], new(h2, w2, 64), act="leaky", act_param=0.2)
            del cols
            b = ops.conv2d(self.enc2, [a], new(h2, w2, 64), act="leaky", act_param=0.2)
            c = ops.conv2d(self.enc4, [b], new(h4, w4, 128), act="leaky", act_param=0.2)
            x0 = ops.conv2d(self.enc6, [c], new(h4, w4, 256), act="leaky", act_param=0.2)
            d = ops.conv2d(self.enc8, [x0], new(h4, w4, 384), act="leaky", act_param=0.2)
            f = ops.conv2d(self.enc10, [x0, d], new(h4, w4, 512), act="leaky", act_param=0.2)
            g = ops.conv2d(self.enc12, [x0, f], new(h4, w4, 384), act="leaky", act_param=0.2)
            hh = ops.conv2d(self.enc14, [x0, g], new(h4, w4, 256), act="leaky", act_param=0.2)
            ops.conv2d(self.enc16, [x0, hh], out[s:e], act="leaky", act_param=0.2)
        return out

    def prepare_clip(self, packed: torch.Tensor | None, flows: torch.Tensor, masks_in_u8: torch.Tensor,
                     masks_upd_u8: torch.Tensor, enc: torch.Tensor | None = None, static_masks=False) -> ClipState:
        """packed f16 [T,H,W,8] (or precomputed encoder features `enc` f16 [T,h,w,128], e.g. gathered from
        other ranks); flows fp32 [2,T-1,H,W,2] (completed); masks u8 [T,H,W] on the device."""
        st = ClipState()
        st.static_masks = static_masks
        T, H, W = masks_in_u8.shape
        dev = masks_in_u8.device
        st.H, st.W = H, W
        st.enc = self.encode(packed) if enc is None else enc
        h, w = st.enc.shape[1:3]
        ds = torch.empty(2 * (T - 1), h, w, 2, device=dev)
        ops.flow_down4(flows.view(2 * (T - 1), H, W, 2), ds)
        st.flow_f, st.flow_b = ds[:T - 1], ds[T - 1:]
        # nearest x1/4 mask planes + MaxPool2d(7,3,3) token masks (propainter.py:409-428): binary plumbing on the device
        st.maskpair, st.tokmask = ops.clip_masks(masks_in_u8.contiguous(), masks_upd_u8.contiguous(), *token_grid(h, w),
                                                 dtype=self.dt)
        st.aux_b = torch.empty(T - 1, h, w, 8, device=dev, dtype=self.dt)
        st.aux_f = torch.empty(T - 1, h, w, 8, device=dev, dtype=self.dt)
        ops.featprop_aux(st.flow_f, st.flow_b, st.maskpair[:T - 1], st.aux_b)
        ops.featprop_aux(st.flow_b, st.flow_f, st.maskpair[1:], st.aux_f)
        return st

    # ------------------------------------------------------------------------------------------------
    def propagate_windows(self, st: ClipState, windows: list[list[int]], ready: list | None = None) -> list[torch.Tensor]:
        """BidirectionalPropagation(learnable=True) (propainter.py:118-231) for the local frames of EVERY
        window of the clip.  Windows are independent, so all windows with the same number of local
        frames advance through the recurrence together as one batch: step i of the pass is a batch of
        `nw` images (one per window) instead of `nw` launches of 1 image.  Returns, per window, the
        propagated local features f16 [l_t,h,w,128].
        `ready` (r06, PP_FEATPROP_PIPE): a list the caller passes to take part in the schedule -- entry wi is None when
        window wi's tensor is ordered by the launch stream, or an event its consumer must wait for: the second half of a
        large group then runs on the side stream BEHIND the first half (not next to it), i.e. next to the transformer of the
        first windows, and nobody joins the side stream here."""
        result: list[torch.Tensor | None] = [None] * len(windows)
        groups: dict[int, list[int]] = {}
        for wi, nb in enumerate(windows):
            groups.setdefault(len(nb), []).append(wi)
        # r06: a large group runs as two halves next to each other on two streams (PP_FEATPROP_LANES=1: one batch): the sweep
        # alternates gather-bound kernels (flow warp, deformable sampling) with matrix-bound convolutions, and two independent
        # halves in flight overlap one's gathers with the other's MFMAs (~2 ms per clip).  A layer's result per window does not
        # depend on its batch (kernel selection never looks at it): the same bits as one batch.  (This lane is how the packed-fp32
        # `op_sel` erratum was found -- pp_deform_cols next to another stream's convolutions computed wrong sample positions until
        # the library was rebuilt without SLP vectorisation: profiles/r06_pk_f32_op_sel_erratum.md, tests/test_isa_audit.py;
        # tests/test_e2e.py::test_stream_lanes_are_bit_identical_and_reproducible runs every lane against the serial schedule.)
        two = (os.environ.get("PP_FEATPROP_LANES", "2") != "1" and st.enc.is_cuda and not torch.cuda.is_current_stream_capturing()
               and ops.CONV_PROFILE is None)
        if ready is not None:
            ready[:] = [None] * len(windows)
        for lt, wis in groups.items():
            if two and len(wis) >= 8:
                dev = st.enc.device
                main = torch.cuda.current_stream(dev)
                if self._side is None:
                    self._side = torch.cuda.Stream(dev)
                side, half = self._side, (len(wis) + 1) // 2
                if ready is not None:
                    out_a = self._feature_propagation_batch(st, [windows[wi][0] for wi in wis[:half]], lt)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        out_b = self._feature_propagation_batch(st, [windows[wi][0] for wi in wis[half:]], lt)
                        done = torch.cuda.Event()
                        done.record(side)
                    for j, wi in enumerate(wis):
                        result[wi] = out_a[:, j] if j < half else out_b[:, j - half]
                        if j >= half:
                            ready[wi] = done
                    continue
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    out_b = self._feature_propagation_batch(st, [windows[wi][0] for wi in wis[half:]], lt)
                out_a = self._feature_propagation_batch(st, [windows[wi][0] for wi in wis[:half]], lt)
                main.wait_stream(side)
                for j, wi in enumerate(wis):
                    result[wi] = out_a[:, j] if j < half else out_b[:, j - half]
                continue
            out = self._feature_propagation_batch(st, [windows[wi][0] for wi in wis], lt)
            for j, wi in enumerate(wis):
                result[wi] = out[:, j]
        return result  # type: ignore[return-value]

    def _feature_propagation_batch(self, st: ClipState, g0s: list[int], lt: int) -> torch.Tensor:
        """nw windows starting at clip frames g0s, each with lt local frames -> f16 [lt, nw, h, w, 128].  The sweep is
        2 x lt steps of ~10 small dependent launches: captured once per (window set, clip shape) into a hipGraph and
        replayed on static copies of the per-clip tensors (graphs.py)."""
        # (the key carries what the captured launches depend on besides the input shapes: the window set and the form of
        # the deformable convolution, an environment knob)
        # r06: the rows the sweep reads are gathered (window-minor: row i * nw + j = local frame i of window j) straight into
        # the captured graph's static inputs, ONE index_select per source tensor and replay; r02-r05 copied the whole clip's
        # tensors into static twins (enc alone: 295 MB at cfg 2, per window group) and gathered per step inside the graph.
        fused = ops.deform_fused(st.enc.shape[1], st.enc.shape[2])
        dev = st.enc.device
        nw = len(g0s)
        rows = ops.device_ints([g + i for i in range(lt) for g in g0s], dev)
        rows1 = ops.device_ints([g + i for i in range(lt - 1) for g in g0s], dev)   # frame g + i: flows_forward[g+i], flows_backward[g+i]
        srcs = ((st.enc, rows), (st.maskpair, rows), (st.flow_f, rows1), (st.flow_b, rows1), (st.aux_b, rows1), (st.aux_f, rows1))

        def fill(bufs):
            if bufs is None:
                return [t.index_select(0, ix) for t, ix in srcs]
            for b, (t, ix) in zip(bufs, srcs):
                torch.index_select(t, 0, ix, out=b)
            return bufs

        key = ("featprop", tuple(g0s), lt, fused, tuple(st.enc.shape[1:]), st.enc.dtype)
        return self._graphs.run_filled(key, lambda *t: self._featprop_eager(nw, lt, *t), fill, dev)

    def _featprop_eager(self, nw: int, lt: int, x, mp, flow_f, flow_b, aux_b, aux_f) -> torch.Tensor:
        """x [lt*nw,h,w,128], mp [lt*nw,h,w,8]: row i * nw + j = local frame i of window j; flows / aux [(lt-1)*nw, ...]: row
        i * nw + j = the pair (local frame i, i + 1) of window j."""
        dev = x.device
        _, h, w, _ = x.shape
        x, mp = x.view(lt, nw, h, w, 128), mp.view(lt, nw, h, w, 8)
        flow_f, flow_b = flow_f.view(max(lt - 1, 0), nw, h, w, 2), flow_b.view(max(lt - 1, 0), nw, h, w, 2)
        aux_b, aux_f = aux_b.view(max(lt - 1, 0), nw, h, w, 8), aux_f.view(max(lt - 1, 0), nw, h, w, 8)

        def buf(c, dt=None):
            return torch.empty(nw, h, w, c, device=dev, dtype=dt or self.dt)

        t128, u128, warped, aligned = buf(128), buf(128), buf(128), buf(128)
        om = buf(432, torch.float32)
        fused = self.dt == torch.float16 and ops.deform_fused(h, w)
        cols = None if fused else buf(9 * 128)
        outs = {}
        src = x
        for name in ("backward_1", "forward_1"):
            S = self.prop[name]
            out = torch.empty(lt, nw, h, w, 128, device=dev, dtype=self.dt)
            order = list(range(lt - 1, -1, -1)) if name == "backward_1" else list(range(lt))
            prop = None
            for i, idx in enumerate(order):
                cur = src[idx]
                if i == 0:
                    prop = cur
                else:
                    if name == "backward_1":   # frame g uses flows_forward[g] / aux_b[g]
                        flow, aux = flow_f[idx], aux_b[idx]
                    else:                      # frame g uses flows_backward[g-1] / aux_f[g-1]
                        flow, aux = flow_b[idx - 1], aux_f[idx - 1]
                    ops.flow_warp(prop, flow, warped)
                    ops.conv2d(S["off0"], [cur, warped, aux], t128, act="leaky", act_param=0.1)
                    ops.conv2d(S["off2"], [t128], u128, act="leaky", act_param=0.1)
                    ops.conv2d(S["off4"], [u128], t128, act="leaky", act_param=0.1)
                    ops.conv2d(S["off6"], [t128], om, act="tanh", out_scale=3.0, act2="sigmoid", act_split=288)
                    if cols is None:   # one launch, no column tensor (pp_deform_conv)
                        ops.deform_conv(S["dcn"], prop, None, om, aligned, flow=flow)
                    else:
                        ops.deform_cols(prop, None, om, cols, flow=flow)
                        ops.conv2d(S["dcn"], [cols], aligned)
                    prop = aligned
                ops.conv2d(S["bb0"], [cur, prop, mp[idx]], t128, act="leaky", act_param=0.2)
                ops.conv2d(S["bb2"], [t128], out[idx], epi="add", aux1=prop)
                prop = out[idx]
            outs[name] = out
            src = out
        n = lt * nw
        tmp = torch.empty(n, h, w, 128, device=dev, dtype=self.dt)
        res = torch.empty(lt, nw, h, w, 128, device=dev, dtype=self.dt)
        ops.conv2d(self.fuse0, [outs["backward_1"].view(n, h, w, 128), outs["forward_1"].view(n, h, w, 128), mp.view(n, h, w, 8)],
                   tmp, act="leaky", act_param=0.2)
        ops.conv2d(self.fuse2, [tmp], res.view(n, h, w, 128), epi="add", aux1=x.view(n, h, w, 128))
        return res

    def _transformer(self, tok: torch.Tensor, hw: tuple[int, int], win_masked: torch.Tensor) -> torch.Tensor:
        dev = tok.device
        t, fh, fw, _ = tok.shape
        h, w = hw
        Hp, Wp = math.ceil(fh / WIN[0]) * WIN[0], math.ceil(fw / WIN[1]) * WIN[1]
        ph, pw = Hp // 4, Wp // 4
        xn = torch.zeros(t, Hp, Wp, 512, device=dev, dtype=self.dt)  # pad tokens stay zero (:212-216)
        qkv = torch.empty(t, Hp, Wp, 1536, device=dev, dtype=self.dt)
        pooled = torch.empty(t, ph, pw, 512, device=dev, dtype=self.dt)
        pkv = torch.empty(t, ph, pw, 1024, device=dev, dtype=self.dt)
        att = torch.empty(t, fh, fw, 512, device=dev, dtype=self.dt)
        y = torch.empty(t, fh, fw, 512, device=dev, dtype=self.dt)
        f1 = torch.empty(t, fh, fw, 1960, device=dev, dtype=self.dt)
        fused_fc2 = self.dt == torch.float16 and os.environ.get("PP_FC2_UNFOLD", "fused") != "copy"
        f2 = None if fused_fc2 else torch.empty(t, fh, fw, 1960, device=dev, dtype=self.dt)
        folded = torch.empty(t, h, w, 40, device=dev, dtype=self.dt)
        tok2 = torch.empty_like(tok)
        t_inds = [torch.arange(i, t, 2, dtype=torch.int32, device=dev) for i in (0, 1)]
        for i, B in enumerate(self.blocks):
            ops.layernorm(tok, xn, B["n1w"], B["n1b"])
            ops.conv2d(B["qkv"], [xn], qkv)
            ops.pool_tokens(xn, pooled, B["pool_w"], B["pool_b"])
            ops.conv2d(B["kv"], [pooled], pkv)
            ops.window_attention(qkv, pkv.view(t, ph * pw, 1024), win_masked, t_inds[i % 2], att)
            ops.conv2d(B["proj"], [att], tok2, epi="add", aux1=tok)
            ops.layernorm(tok2, y, B["n2w"], B["n2b"])
            ops.conv2d(B["fc1"], [y], f1)
            # (r04: the GELU once per folded value in pp_fold instead of on each of its <= 9 unfolded copies: bit-identical)
            ops.fold(f1.view(t, fh * fw, 1960), folded, fh, fw, True, gelu=True)
            if fused_fc2:   # fc2 gathers its 7x7 / stride-3 patches from the 40-channel map itself: no 49x unfolded copy
                ops.linear_of_unfold(B["fc2"], folded, tok, 7, 3, 3, epi="add", aux1=tok2)
            else:
                ops.unfold_gelu(folded, f2.view(t, fh * fw, 1960), fh, fw, pre_activated=True)
                ops.conv2d(B["fc2"], [f2], tok, epi="add", aux1=tok2)
        return tok

    def window_mask_flags(self, st: ClipState, nb: list[int]) -> torch.Tensor:
        """Integer logic of sparse_transformer.py:321-326: a 5x9 token window is 'masked' iff any
        local-frame token mask inside it is set (computed on the device, no host sync)."""
        if not st.static_masks:
            return ops.window_flags(st.tokmask, nb[0], len(nb), WIN)
        if st.flags is None:
            key = (st.static_masks, st.H, st.W, str(st.tokmask.device)) if st.static_masks is not True else None
            st.flags = self._geometry_flags.get(key) if key is not None else None
            if st.flags is None:
                st.flags = ops.window_flags(st.tokmask, 0, 1, WIN)   # any frame: they are all the same
                if key is not None:
                    if len(self._geometry_flags) > 64:
                        self._geometry_flags.clear()
                    self._geometry_flags[key] = st.flags
        return st.flags

    def reference_tokens(self, st: ClipState, frames: list[int]) -> torch.Tensor:
        """The soft-split tokens of clip-state frames `frames` (encoder features -> 7x7 / stride-3 embedding), kept per clip:
        rows are appended for frames not seen before (pipeline.run_inpainting asks for the whole schedule's reference frames up
        front: one batched convolution per clip).  -> st.ref_tok, indexed through st.ref_row."""
        new = [f for f in dict.fromkeys(frames) if f not in st.ref_row]
        if new:
            dev = st.enc.device
            fh, fw = token_grid(*st.enc.shape[1:3])
            n0 = 0 if st.ref_tok is None else st.ref_tok.shape[0]
            grown = torch.empty(n0 + len(new), fh, fw, 512, device=dev, dtype=self.dt)
            if n0:
                grown[:n0] = st.ref_tok
            src = st.enc.index_select(0, ops.device_ints(new, dev))
            ops.conv2d(self.ss, [src], grown[n0:])
            for i, f in enumerate(new):
                st.ref_row[f] = n0 + i
            st.ref_tok = grown
        return st.ref_tok

    def forward_window(self, st: ClipState, nb: list[int], refs: list[int], trace: dict | None = None,
                       local_prop: torch.Tensor | None = None, lane: int = 0) -> torch.Tensor:
        """One neighbour+reference window -> tanh image of the local frames, f16 [l_t,H,W,4] (3 used).
        `local_prop` = this window's entry of propagate_windows() (computed here when not given)."""
        dev = st.enc.device
        lt, t = len(nb), len(nb) + len(refs)
        _, h, w, _ = st.enc.shape
        if local_prop is None:
            local_prop = self.propagate_windows(st, [nb])[0]
        fh, fw = token_grid(h, w)
        tok = torch.empty(t, fh, fw, 512, device=dev, dtype=self.dt)
        # r06: soft split (sparse_transformer.py:8-37) of the propagated local frames straight from propagate_windows()'s tensor,
        # the reference frames' tokens from the per-clip table -- r01-r05 assembled [local | enc[refs]] into a `feat` copy per
        # window and re-ran the 7x7 / stride-3 embedding on every reference frame of every window (8 x 16 frame-convolutions at
        # cfg 2 for 8 distinct frames).  A convolution's result per image does not depend on its batch: same tokens.
        local_prop = local_prop.contiguous()
        ops.conv2d(self.ss, [local_prop], tok[:lt])
        if refs:
            torch.index_select(self.reference_tokens(st, refs), 0,
                               ops.device_ints([st.ref_row[r] for r in refs], dev), out=tok[lt:])
        flags = self.window_mask_flags(st, nb)
        if trace is not None:
            trace.update(local_prop=local_prop.clone(), tok=tok.clone())
        # r06: the 8 blocks (~80 launches on fixed shapes) replay as ONE hipGraph per token-tensor shape -- a clip has 2-3 distinct
        # window lengths; inputs: the tokens (28 MB copy) and the 36 window flags; the result lives in the graph's
        # static buffer and is consumed by the soft composition right below
        # (`lane`: two windows in flight on two streams -- pipeline.run_inpainting -- replay two INSTANCES of a shape's graph: a
        #  captured sweep owns its static buffers)
        tok = self._graphs.run(("transformer", h, w, os.environ.get("PP_FC2_UNFOLD", "fused"), lane,
                                self.dt == torch.float32 and ops.attention_exact_enabled()),
                               lambda tk, fl: self._transformer(tk, (h, w), fl), tok, flags)
        # soft composition + residual, only for the local frames that are decoded (:443-451)
        emb = torch.empty(lt, fh, fw, 6272, device=dev, dtype=self.dt)
        ops.conv2d(self.sc, [tok[:lt]], emb)
        comp = torch.empty(lt, h, w, 128, device=dev, dtype=self.dt)
        ops.fold(emb.view(lt, fh * fw, 6272), comp, fh, fw, False)
        enc3 = torch.empty(lt, h, w, 128, device=dev, dtype=self.dt)
        ops.conv2d(self.sc_bias_conv, [comp], enc3, epi="add", aux1=local_prop)
        if trace is not None:
            trace.update(tok_out=tok, enc3=enc3)
        H, W = st.H, st.W

        def new(hh, ww, c):
            return torch.empty(lt, hh, ww, c, device=dev, dtype=self.dt)

        up = ops.upsample2x(enc3, new(2 * h, 2 * w, 128))
        a = ops.conv2d(self.dec0, [up], new(2 * h, 2 * w, 128), act="leaky", act_param=0.2)
        b = ops.conv2d(self.dec2, [a], new(2 * h, 2 * w, 64), act="leaky", act_param=0.2)
        up = ops.upsample2x(b, new(H, W, 64))
        c = ops.conv2d(self.dec4, [up], new(H, W, 64), act="leaky", act_param=0.2)
        out = new(H, W, 4)
        ops.conv2d(self.dec6, [c], out[..., 0:3], act="tanh")
        return out
