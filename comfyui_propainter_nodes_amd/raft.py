"""RAFT bidirectional optical flow on the MI355X (fp32, like the reference keeps RAFT in fp32:
utils/model_utils.py:55-56, propainter_inference.py:74).

Host-side scheduling only: every FLOP and gather runs in libpropainter_mi355 kernels.
Replaces RAFT_bi.forward / RAFT.forward (model/modules/flow_comp_raft.py:39-58,
RAFT/raft.py:94-152) and compute_flow's clip chunking (propainter_inference.py:61-99) with a
scheduler designed for 288 GB of HBM:

  * fnet / cnet run ONCE per frame (the reference recomputes both for every pair and direction),
  * all pair-directions of the clip are batched through every update-block kernel,
  * the mask head and the convex upsampling run only after the last iteration
    (the reference evaluates and discards them every iteration, raft.py:141-150),
  * eval-mode BatchNorm of cnet is folded into the convolution weights.

Results per pair are independent of the reference's clip chunking (instance norm is per
sample), so only the pair batching is memory-capped (`max_volume_bytes`).
"""
from __future__ import annotations

import os

import torch

from . import graphs, ops


def _strip(sd: dict, prefix: str) -> dict:
    pre = ("module." + prefix) if any(k.startswith("module.") for k in sd) else prefix
    return {k[len(pre):]: v.float() for k, v in sd.items() if k.startswith(pre) and v.is_floating_point()}


def _conv_spec(*args, **kw):
    """RAFT runs on f32 tensors (the reference keeps RAFT out of its fp16 mode, propainter_inference.py).  The
    constant-weight convolutions multiply on the f16 matrix pipe with two-term operand splits (PP_F32X2: fp32-GEMM
    accuracy at several times the f32 MFMA rate); PP_F32_GEMM=exact selects the plain f32 MFMA kernels."""
    return ops.make_conv_spec(*args, split=ops.f32_split_enabled(), **kw)


def _fold_bn(w, b, p, name):
    s = p[name + ".weight"].double() / torch.sqrt(p[name + ".running_var"].double() + 1e-5)
    w2 = (w.double() * s.view(-1, 1, 1, 1)).float()
    b2 = ((b.double() - p[name + ".running_mean"].double()) * s + p[name + ".bias"].double()).float()
    return w2, b2


class _Encoder:
    """BasicEncoder (extractor.py:121-193) as a list of conv specs; `kind` = instance | batch."""

    def __init__(self, p: dict, kind: str, device):
        self.kind = kind
        dt = torch.float32

        def conv(name, norm=None, **kw):
            w, b = p[name + ".weight"], p[name + ".bias"]
            if kind == "batch" and norm is not None:
                w, b = _fold_bn(w, b, p, norm)
            return w, b, kw

        w, b, _ = conv("conv1", "norm1")
        # 7x7 s2 on 3 channels -> im2col (k = (ky,kx,c)) + GEMM
        self.c1_kpad = ops.pad32(147)
        self.conv1 = _conv_spec(w.permute(0, 2, 3, 1).reshape(64, 147, 1, 1), b, dt, seg_channels=[self.c1_kpad],
                                        seg_valid=[147]).to(device)
        self.blocks = []
        cin = 64
        for layer, dim, stride in (("layer1", 64, 1), ("layer2", 96, 2), ("layer3", 128, 2)):
            for bi, st in ((0, stride), (1, 1)):
                pre = f"{layer}.{bi}."
                w1, b1, _ = conv(pre + "conv1", pre + "norm1")
                w2, b2, _ = conv(pre + "conv2", pre + "norm2")
                blk = {
                    "c1": _conv_spec(w1, b1, dt, stride=st, padding=1).to(device),
                    "c2": _conv_spec(w2, b2, dt, padding=1).to(device),
                    "down": None, "dim": dim, "stride": st,
                }
                if st != 1:
                    wd, bd, _ = conv(pre + "downsample.0", pre + "norm3")
                    blk["down"] = _conv_spec(wd, bd, dt, stride=st).to(device)
                self.blocks.append(blk)
                cin = dim
        self.conv2 = _conv_spec(p["conv2.weight"], p["conv2.bias"], dt).to(device)

    def __call__(self, frames: torch.Tensor, out: torch.Tensor, *, split_tanh_relu: bool) -> torch.Tensor:
        """frames [n,H,W,3] fp32 -> out [n,H/8,W/8,256]."""
        dev = frames.device
        n, H, W, _ = frames.shape
        inst = self.kind == "instance"
        h2, w2 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        x = torch.empty(n, h2, w2, 64, device=dev)
        if self.conv1.split and ops.patch_conv_enabled():
            # r06: the 7x7 / stride-2 stem gathers its patches from the 3-channel frames itself (conv_patch.hip): no im2col tensor
            ops.conv2d_patch(self.conv1, frames, x, 7, 7, stride=2, padding=3, act=None if inst else "relu")
        else:
            cols = torch.empty(n, h2, w2, self.c1_kpad, device=dev)
            ops.im2col(frames, cols, 7, 7, stride=2, padding=3)
            ops.conv2d(self.conv1, [cols], x, act=None if inst else "relu")
            del cols
        if inst:
            ops.instnorm(x, x, relu_pre=True)
        for blk in self.blocks:
            _, h, w, _ = x.shape
            ho, wo = blk["c1"].out_hw(h, w)
            y1 = torch.empty(n, ho, wo, blk["dim"], device=dev)
            y2 = torch.empty(n, ho, wo, blk["dim"], device=dev)
            if inst:
                ops.conv2d(blk["c1"], [x], y1)
                ops.instnorm(y1, y1, relu_pre=True)
                ops.conv2d(blk["c2"], [y1], y2)
                skip = x
                if blk["down"] is not None:
                    skip = torch.empty(n, ho, wo, blk["dim"], device=dev)
                    ops.conv2d(blk["down"], [x], skip)
                    ops.instnorm(skip, skip)
                ops.instnorm(y2, y2, relu_pre=True, skip=skip, relu_post=True)
            else:
                ops.conv2d(blk["c1"], [x], y1, act="relu")
                skip = x
                if blk["down"] is not None:
                    skip = torch.empty(n, ho, wo, blk["dim"], device=dev)
                    ops.conv2d(blk["down"], [x], skip)
                ops.conv2d(blk["c2"], [y1], y2, act="relu", epi="add_relu", aux1=skip)
            x = y2
        if split_tanh_relu:  # raft.py:119-122: net = tanh(c[:128]), inp = relu(c[128:])
            ops.conv2d(self.conv2, [x], out, act="tanh", act2="relu", act_split=128)
        else:
            ops.conv2d(self.conv2, [x], out)
        return out


class RaftFlow:
    def __init__(self, sd: dict, device, *, max_volume_bytes: int = 48 << 30, enc_chunk: int = 16):
        self.device = torch.device(device)
        self.max_volume_bytes = max_volume_bytes
        self.enc_chunk = enc_chunk
        self._graphs = graphs.GraphCache(max_entries=6)   # (each holds the all-pairs volume of half a clip shape)
        self._ws: dict = {}       # per clip shape: the update graphs' static inputs (feature maps, per-pair context, tiled operand)
        self._side = None         # the second direction's stream
        dt = torch.float32
        self.fnet = _Encoder(_strip(sd, "fnet."), "instance", device)
        self.cnet = _Encoder(_strip(sd, "cnet."), "batch", device)
        u = _strip(sd, "update_block.")

        def spec(name, **kw):
            return _conv_spec(u[name + ".weight"], u[name + ".bias"], dt, **kw).to(device)

        self.convc1 = spec("encoder.convc1")
        self.convc2 = spec("encoder.convc2", padding=1)
        wf1 = u["encoder.convf1.weight"]  # [128,2,7,7] -> im2col GEMM
        self.f1_kpad = ops.pad32(98)
        self.convf1 = _conv_spec(wf1.permute(0, 2, 3, 1).reshape(128, 98, 1, 1), u["encoder.convf1.bias"], dt,
                                         seg_channels=[self.f1_kpad], seg_valid=[98]).to(device)
        self.convf2 = spec("encoder.convf2", padding=1)
        self.conv = spec("encoder.conv", padding=1)
        # GRU input = cat(h, inp, motion) (update.py:60-71).  `inp` (the context features) does not change over
        # the iterations, so its third of every GRU convolution is computed ONCE per clip and enters the
        # per-iteration convolution over [h | motion] as a pre-activation addend (exact, 1/3 fewer GRU FLOPs).
        # The z and r gates read the same input (update.py:41-43 / :49-51): ONE 256-channel convolution computes both
        # (r03; sigmoid on all channels, the r * h product as an epilogue on channels 128..255 only: `epi_from`), its two
        # 128-channel tiles of a pixel tile run next to each other and share the pixel stage in L2.
        self.gru, self.gru_ctx = {}, {}
        for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
            ws = {g: u[f"gru.conv{g}{sfx}.weight"] for g in "zrq"}
            bs = {g: u[f"gru.conv{g}{sfx}.bias"] for g in "zrq"}
            for key, w, b in (("zr", torch.cat([ws["z"], ws["r"]], 0), torch.cat([bs["z"], bs["r"]], 0)), ("q", ws["q"], bs["q"])):
                w_dyn = torch.cat([w[:, 0:128], w[:, 256:384]], 1)
                self.gru[key + sfx] = _conv_spec(w_dyn, b, dt, padding=pad, seg_channels=[128, 128]).to(device)
                self.gru_ctx[key + sfx] = _conv_spec(w[:, 128:256].contiguous(), None, dt, padding=pad).to(device)
        self.fh1 = spec("flow_head.conv1", padding=1)
        self.fh2 = spec("flow_head.conv2", padding=1)
        self.mask0 = spec("mask.0", padding=1)
        self.mask2 = spec("mask.2")

    # ------------------------------------------------------------------------------------
    def encode(self, frames: torch.Tensor, fmap: torch.Tensor | None = None,
               ctx: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """frames [T,H,W,3] fp32 in [-1,1] -> (fmap [T,h8,w8,256], ctx [T,h8,w8,256] = tanh|relu); written into `fmap` / `ctx`
        when given."""
        T, H, W, _ = frames.shape
        h8, w8 = H // 8, W // 8
        if fmap is None:
            fmap = torch.empty(T, h8, w8, 256, device=frames.device)
        if ctx is None:
            ctx = torch.empty(T, h8, w8, 256, device=frames.device)
        # r06: the two encoders share nothing but their input: cnet runs on a second stream next to fnet (PP_ENC_LANES=1: one
        # after the other); fnet's instance-norm passes are bandwidth-bound, cnet's folded-BatchNorm convolutions matrix-bound
        two = (frames.is_cuda and os.environ.get("PP_ENC_LANES", "2") != "1" and not torch.cuda.is_current_stream_capturing()
               and ops.CONV_PROFILE is None)
        if two:
            main = torch.cuda.current_stream(frames.device)
            if self._side is None:
                self._side = torch.cuda.Stream(frames.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                for s in range(0, T, self.enc_chunk):
                    e = min(T, s + self.enc_chunk)
                    self.cnet(frames[s:e], ctx[s:e], split_tanh_relu=True)
        for s in range(0, T, self.enc_chunk):
            e = min(T, s + self.enc_chunk)
            self.fnet(frames[s:e], fmap[s:e], split_tanh_relu=False)
            if not two:
                self.cnet(frames[s:e], ctx[s:e], split_tanh_relu=True)
        if two:
            main.wait_stream(self._side)
        return fmap, ctx

    def _tiled_operand(self, fm: torch.Tensor, h: int, w: int, out: tuple | None = None) -> torch.Tensor:
        """The second operand of the all-pairs GEMM for every frame of fm [n+1,hw,256]: its pixels (the GEMM's output-channel index)
        in the 4 x 8 tile order of the level-0 planes, zero rows for the tile padding (45 -> 48 rows at 640x360), split-packed for
        PP_F32X2 -- ONCE per frame (r06; r01-r05 built it per pair-direction from three 582 MB concatenations)."""
        n1 = fm.shape[0]
        fz = torch.cat([fm, torch.zeros(n1, 1, 256, device=fm.device)], 1)
        idx = ops.tiled_order_index(h, w, fm.device)
        if out is None:
            ft = fz.index_select(1, idx)                                    # [n+1, pitch0, 256]
            return ops.split_pack(ft) if ops.f32_split_enabled() else ft
        gathered, packed = out                                              # (`out`: a graph's static buffers, filled in place)
        if not ops.f32_split_enabled():
            return torch.index_select(fz, 1, idx, out=packed)
        torch.index_select(fz, 1, idx, out=gathered)
        return ops.split_pack(gathered, out=packed)

    def _update_pairs(self, fm, ft, ctx, iters: int, flow_up: tuple, trace: dict | None = None, direction: int | None = None) -> None:
        """The update block over the n + 1 frames of `fm` [n+1,hw,256] (`ft` = _tiled_operand(fm)).  direction None: ONE batch of the
        n forward pairs (i -> i+1) then the n backward pairs (i+1 -> i), ctx [2n,h,w,256] (context of each pair's first frame),
        flow_up = (forward [n,8h,8w,2], backward [n,8h,8w,2]) views that receive the upsampled flows; direction 0 / 1: that half
        alone (ctx [n,...], flow_up[direction] written) -- bidirectional() runs the two halves on two streams."""
        dev = fm.device
        P, h, w, _ = ctx.shape
        n = fm.shape[0] - 1
        hw = h * w
        # corr.py:52-60 (/sqrt(256)); both operands are activations: the second one is split-packed on the device (PP_F32X2).
        # Levels 0 and 1 of the pyramid are stored in 4 x 8 tiles of 128 bytes (r03): a 12 x 12 lookup window then touches
        # ~9 cache lines instead of ~17 on 320-byte rows (r02: 2.0x the algorithmic HBM traffic).  The all-pairs GEMM
        # writes that layout for free (its second operand is in tile order); the small levels 2 and 3 stay row-major.
        pitch0 = ops.tiled_pitch(h, w)
        split = ops.f32_split_enabled()
        vol = torch.empty(P, 1, hw, pitch0, device=dev)
        if direction in (None, 0):
            ops.batched_gemm_nt(fm[:n].view(n, 1, hw, 256), ft[1:], vol[:n], scale=1.0 / 16.0, split=split, b_packed=split)
        if direction in (None, 1):
            ops.batched_gemm_nt(fm[1:].view(n, 1, hw, 256), ft[:n], vol[P - n:], scale=1.0 / 16.0, split=split, b_packed=split)
        pyr = [(vol.view(P, hw, pitch0), h, w, True)]
        for lvl in range(1, 4):
            _, hi, wi, ti = pyr[-1]
            ho, wo, to = hi // 2, wi // 2, lvl == 1
            nxt = torch.empty(P, hw, ops.tiled_pitch(ho, wo) if to else ho * wo, device=dev)
            ops.avgpool2x2(pyr[-1][0].view(P * hw, -1), nxt.view(P * hw, -1), hw=(hi, wi), in_tiled=ti, out_tiled=to)
            pyr.append((nxt, ho, wo, to))
        # 324 lookup channels at a pitch of 352 floats: every 32-channel chunk (128 bytes) the motion encoder's first
        # convolution gathers then starts on a cache-line boundary (at pitch 324 each chunk straddled two lines: that 1x1
        # convolution ran at 120 TF/s where its neighbours reach 250-300)
        fused_lookup = self.convc1.split and ops.lookup_fused_enabled() and trace is None
        corr = None if fused_lookup else torch.empty(P, h, w, 352, device=dev)[..., :324]
        cor1 = torch.empty(P, h, w, 256, device=dev)
        cf = torch.empty(P, h, w, 256, device=dev)       # cor (192) | flo (64)
        patch = self.convf1.split and ops.patch_conv_enabled()
        fcols = None if patch else torch.empty(P, h, w, self.f1_kpad, device=dev)
        flo1 = torch.empty(P, h, w, 128, device=dev)
        mf = torch.zeros(P, h, w, 128, device=dev)       # motion (126) | flow (2)
        flow = mf[..., 126:128]
        inp = ctx[..., 128:256]
        hcur = ctx[..., 0:128]
        hA = torch.empty(P, h, w, 128, device=dev)
        hB = torch.empty(P, h, w, 128, device=dev)
        zr = torch.empty(P, h, w, 256, device=dev)      # z (128) | r * h (128)
        z, rh = zr[..., 0:128], zr[..., 128:256]
        t256 = torch.empty(P, h, w, 256, device=dev)
        ctx_term = {}
        for key, sp in self.gru_ctx.items():
            ctx_term[key] = ops.conv2d(sp, [inp], torch.empty(P, h, w, sp.cout, device=dev))
        for it in range(iters):
            if fused_lookup:   # r06: lookup + convc1 in one launch, the 324-channel tensor never written (corr_lookup_conv.hip)
                ops.corr_lookup_conv(pyr, flow, self.convc1, cor1, act="relu")
            else:
                ops.corr_lookup(pyr, flow, corr)
                if trace is not None and it == 0:
                    trace["corr0"] = corr.clone()
                ops.conv2d(self.convc1, [corr], cor1, act="relu")
            ops.conv2d(self.convc2, [cor1], cf[..., 0:192], act="relu")
            if patch:   # r06: 7x7 on the 2-channel flow without the 291 MB patch tensor per iteration (conv_patch.hip)
                ops.conv2d_patch(self.convf1, flow, flo1, 7, 7, padding=3, act="relu")
            else:
                ops.im2col(flow, fcols, 7, 7, padding=3)
                ops.conv2d(self.convf1, [fcols], flo1, act="relu")
            ops.conv2d(self.convf2, [flo1], cf[..., 192:256], act="relu")
            ops.conv2d(self.conv, [cf], mf[..., 0:126], act="relu")
            for sfx, hout in (("1", hA), ("2", hB)):
                segs = [hcur, mf]
                ops.conv2d(self.gru["zr" + sfx], segs, zr, act="sigmoid", epi="mul", aux1=hcur, epi_from=128,
                           pre_add=ctx_term["zr" + sfx])
                ops.conv2d(self.gru["q" + sfx], [rh, mf], hout, act="tanh", epi="gru", aux1=z, aux2=hcur,
                           pre_add=ctx_term["q" + sfx])
                hcur = hout
            ops.conv2d(self.fh1, [hcur], t256, act="relu")
            ops.conv2d(self.fh2, [t256], flow, epi="add", aux1=flow)  # coords1 += delta (raft.py:139)
        ops.conv2d(self.mask0, [hcur], t256, act="relu")
        mask = torch.empty(P, h, w, 576, device=dev)
        ops.conv2d(self.mask2, [t256], mask, out_scale=0.25)           # update.py:153
        if direction in (None, 0):
            ops.convex_upsample(mask[:n], flow[:n], flow_up[0])        # straight into the caller's [2, T-1, H, W, 2] tensor
        if direction in (None, 1):
            ops.convex_upsample(mask[P - n:], flow[P - n:], flow_up[1])
        if trace is not None:
            trace.update(flow_lr=flow.clone(), net=hcur.clone(), mask=mask)

    def _bidirectional_graph(self, frames: torch.Tensor, iters: int, out: torch.Tensor | None) -> torch.Tensor:
        """r06: a clip whose pairs fit ONE batch runs its update block -- the all-pairs volume, the pyramid, `iters`
        iterations of 19 launches, the mask head and the upsampling: ~400 launches on fixed shapes -- as hipGraphs per (clip shape,
        iters), ONE PER DIRECTION, replayed next to each other on two streams (PP_RAFT_LANES=1: one graph, one stream): like the
        transformer windows (pipeline.run_inpainting), two independent halves in flight interleave one's store bursts with the
        other's matrix work; a layer's result per pair does not depend on its batch, so the flows are the same bits.  The encoders
        stay eager and write the graphs' static inputs (per-shape workspace: feature maps, tile-ordered second operand, per-pair
        context) in place, so no input is copied; the flows are copied OUT of the graphs' static results (2 x 147 MB at cfg 2):
        the caller's tensor must survive the next replay (sharded runs keep several clips' flows alive)."""
        T, H, W, _ = frames.shape
        h, w, n, dev = H // 8, W // 8, T - 1, frames.device
        key = (T, H, W, ops.f32_split_enabled())
        ws = self._ws.get(key)
        if ws is None:
            while len(self._ws) >= 3:                       # (bounded like the graph cache: a workspace is ~1.2 GB at cfg 2)
                self._ws.pop(next(iter(self._ws)))
            pitch0 = ops.tiled_pitch(h, w)
            ws = self._ws[key] = (torch.empty(T, h * w, 256, device=dev), torch.empty(2 * n, h, w, 256, device=dev),
                                  torch.empty(T, pitch0, 256, device=dev), torch.empty(T, pitch0, 256, device=dev))
        fm, cx, ft_gathered, ft = ws
        _, ctx = self.encode(frames, fm.view(T, h, w, 256))
        torch.cat([ctx[:n], ctx[1:]], 0, out=cx)
        self._tiled_operand(fm, h, w, out=(ft_gathered, ft))
        gkey = ("raft_update", iters, T, H, W, ops.f32_split_enabled(), ops.patch_conv_enabled(), ops.lookup_fused_enabled())
        if out is None:
            out = torch.empty(2, n, H, W, 2, device=dev)
        if os.environ.get("PP_RAFT_LANES", "2") == "1":
            def update(fm_, ft_, cx_):
                up = torch.empty(2, n, H, W, 2, device=dev)
                self._update_pairs(fm_, ft_, cx_, iters, (up[0], up[1]))
                return up

            out.copy_(self._graphs.run_filled(gkey + ("both",), update, lambda bufs: [fm, ft, cx], dev))
            return out
        main = torch.cuda.current_stream(dev)
        side = self._side
        if side is None:
            side = self._side = torch.cuda.Stream(dev)

        def update_dir(d):
            def fn(fm_, ft_, cx_):
                up = torch.empty(n, H, W, 2, device=dev)
                self._update_pairs(fm_, ft_, cx_, iters, (up, up), direction=d)
                return up
            return fn

        side.wait_stream(main)                              # the workspace is written
        with torch.cuda.stream(side):
            up1 = self._graphs.run_filled(gkey + (1,), update_dir(1), lambda bufs: [fm, ft, cx[n:]], dev)
            out[1].copy_(up1)
        up0 = self._graphs.run_filled(gkey + (0,), update_dir(0), lambda bufs: [fm, ft, cx[:n]], dev)
        out[0].copy_(up0)
        main.wait_stream(side)
        return out

    def __call__(self, frames: torch.Tensor, iters: int, trace: dict | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """frames [T,H,W,3] fp32 in [-1,1] (channels-last) -> (flows_fwd, flows_bwd), each [T-1,H,W,2]."""
        out = self.bidirectional(frames, iters, trace)
        return out[0], out[1]

    def bidirectional(self, frames: torch.Tensor, iters: int, trace: dict | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
        """frames [T,H,W,3] fp32 in [-1,1] -> flows fp32 [2,T-1,H,W,2] (forward, backward): the layout compute_flow hands on
        (r06: written in place by the upsampling kernel; `out`: an existing [2,T-1,H,W,2] view to fill)."""
        T, H, W, _ = frames.shape
        if H % 8 or W % 8 or H < 128 or W < 128:
            raise ValueError("RAFT needs H, W multiples of 8 and >= 128 (reference limit, SURVEY.md 9.15)")
        h, w = H // 8, W // 8
        per_pair = int(h * w * h * w * 4 * 1.34) + h * w * 4 * 3000
        # (not while bench.py times individual launches with HIP events -- ops.CONV_PROFILE: launches that share the chip with a
        #  second stream cannot be priced one by one)
        if (trace is None and T >= 2 and 2 * per_pair * (T - 1) <= self.max_volume_bytes and frames.is_cuda
                and os.environ.get("PP_GRAPHS_RAFT", "1") != "0" and ops.CONV_PROFILE is None):
            return self._bidirectional_graph(frames, iters, out)
        fmap, ctx = self.encode(frames)
        if trace is not None:
            trace.update(fmap=fmap, ctx=ctx)
        h, w = H // 8, W // 8
        hw = h * w
        fm = fmap.view(T, hw, 256)
        npair = T - 1
        if out is None:
            out = torch.empty(2, npair, H, W, 2, device=frames.device)
        per_pair = int(hw * hw * 4 * 1.34) + hw * 4 * 3000
        chunk = max(1, min(npair, self.max_volume_bytes // (2 * per_pair)))
        for s in range(0, npair, chunk):
            e = min(npair, s + chunk)
            # forward pairs (i -> i+1) then backward pairs (i+1 -> i) in ONE batch of 2n through the update block
            cx = torch.cat([ctx[s:e], ctx[s + 1:e + 1]], 0)
            fmc = fm[s:e + 1]
            ft = self._tiled_operand(fmc, h, w)
            if (trace is None and frames.is_cuda and ops.CONV_PROFILE is None and os.environ.get("PP_RAFT_LANES", "2") != "1"
                    and not torch.cuda.is_current_stream_capturing()):
                # the two directions of the chunk next to each other on two streams (as the graph path does for a whole clip)
                main = torch.cuda.current_stream(frames.device)
                if self._side is None:
                    self._side = torch.cuda.Stream(frames.device)
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    self._update_pairs(fmc, ft, cx[e - s:], iters, (None, out[1, s:e]), direction=1)
                self._update_pairs(fmc, ft, cx[:e - s], iters, (out[0, s:e], None), direction=0)
                main.wait_stream(self._side)
            else:
                self._update_pairs(fmc, ft, cx, iters, (out[0, s:e], out[1, s:e]), trace)
        return out
