"""Multi-GPU execution of ONE long clip: sub-video sharding with seam exchange.

The reference already cuts a long clip into sub-videos of `subvideo_length` frames with halos
(propainter_inference.py:115-144, :172-212) and, for T > subvideo_length, only looks
+-ref_stride*(ref_num//2) frames around a window for reference frames (:36-58).  Those chunks are the
shards: rank r owns a contiguous run of sub-video chunks, i.e. frames [F0, F1).  Per rank:

  A  RAFT on the frame pairs it owns
  x0 raw RAFT flows at the seams (5 per side: the flow-completion halos)            -- peer to peer
  A' flow completion of its chunks
  x1 completed flows at the seams (10 per side: image-propagation halos)            -- peer to peer
  B  image propagation of its chunks, blend, encoder on its frames
  x2 encoder features + updated masks of the frames its windows read (+-45 frames)  -- peer to peer
  C  feature propagation + transformer + decoder for the windows centred in [F0, F1), on a clip state over
     [F0-45, F1+45) only
  x3 the window outputs that land on frames owned by a neighbour (seam windows)     -- peer to peer
  D  uint8 compose of its own frames in GLOBAL window order (the blend is order dependent)
  x4 all_gather of the composed uint8 frames (every rank returns the clip)
A rank holds and uploads only the frames it needs (`frames_needed`: its own +-10); masks are clip-long host plumbing.

Because chunk boundaries, halos and window schedule are exactly the single-GPU ones, the sharded
result is identical to the single-process result.  The driver is written against a small backend
protocol so that the orchestration can be tested on CPU (gloo, world_size 2) with a toy backend and
on one GPU with N in-process virtual ranks; `GpuBackend` is the real thing.

xGMI is point to point, so the seams are exchanged as grouped send / receive pairs between the ranks that share them
(`dist.batch_isend_irecv`; at the defaults 5 + 10 flows and 45 frames of encoder features per side: ~0.4 GB per rank and
clip instead of the 2 x 2.4 GB an all_gather of whole per-rank slabs delivers to every rank of an 8-GPU job).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch

from . import imgprop, ops
from .pipeline import Models, ProPainterConfig, window_schedule


# ------------------------------------------------------------------------------------------------
# plan (pure integer logic)
# ------------------------------------------------------------------------------------------------
@dataclass
class ShardPlan:
    T: int
    sv: int
    world: int
    rank: int

    def __post_init__(self):
        if self.sv > 100:
            raise ValueError("sharded mode needs subvideo_length <= 100 (flow and frame chunks then coincide)")
        self.nchunks = (self.T + self.sv - 1) // self.sv
        per, rem = divmod(self.nchunks, self.world)
        counts = [per + (1 if r < rem else 0) for r in range(self.world)]
        starts = [sum(counts[:r]) for r in range(self.world)]
        self.chunk_ranges = [(starts[r], starts[r] + counts[r]) for r in range(self.world)]
        self.frame_ranges = [(min(self.T, a * self.sv), min(self.T, b * self.sv)) for a, b in self.chunk_ranges]
        nflow = self.T - 1
        self.flow_ranges = [(min(nflow, a * self.sv), min(nflow, b * self.sv)) for a, b in self.chunk_ranges]

    @property
    def frames(self) -> tuple[int, int]:
        return self.frame_ranges[self.rank]

    @property
    def flows(self) -> tuple[int, int]:
        return self.flow_ranges[self.rank]

    def owner_of_frame(self, idx: int) -> int:
        for r, (a, b) in enumerate(self.frame_ranges):
            if a <= idx < b:
                return r
        raise IndexError(idx)

    def flow_chunks(self, rank: int | None = None) -> list[tuple[int, int, int, int]]:
        """(f, e_own, s_halo, e_halo) per owned flow chunk: completes flows [s_halo, e_halo), keeps [f, e_own)."""
        rank = self.rank if rank is None else rank
        a, b = self.flow_ranges[rank]
        nflow = self.T - 1
        out = []
        for f in range(a, b, self.sv):
            out.append((f, min(nflow, f + self.sv), max(0, f - 5), min(nflow, f + self.sv + 5)))
        return out

    def frame_chunks(self, rank: int | None = None) -> list[tuple[int, int, int, int]]:
        rank = self.rank if rank is None else rank
        a, b = self.frame_ranges[rank]
        out = []
        for f in range(a, b, self.sv):
            out.append((f, min(self.T, f + self.sv), max(0, f - 10), min(self.T, f + self.sv + 10)))
        return out

    def raft_frames(self) -> tuple[int, int]:
        """Frames whose adjacent pairs this rank's flow completion reads (its flow chunks + the 5-flow halos; the
        halo flows arrive through exchange x0, RAFT itself runs on the owned pairs only)."""
        ch = self.flow_chunks()
        if not ch:
            return (0, 0)
        return (min(c[2] for c in ch), max(c[3] for c in ch) + 1)


# ------------------------------------------------------------------------------------------------
# backend protocol + the real backend
# ------------------------------------------------------------------------------------------------
class GpuBackend:
    """Stage functions on the MI355X (thin wrappers over the single-GPU pipeline pieces)."""

    def __init__(self, models: Models, config: ProPainterConfig):
        self.m, self.cfg = models, config
        self.act_dtype = models.inpaint_model.dt   # storage type of encoder features / window outputs on the wire

    def raft(self, frames):                     # fp32 [n,H,W,3] -> [2,n-1,H,W,2]
        return self.m.raft_model.bidirectional(frames, self.cfg.raft_iter)

    def complete(self, flows, masks):           # one chunk incl. halos
        return self.m.flow_model(flows.contiguous(), masks.contiguous())

    def img_prop(self, frames, masks, flows):   # one chunk incl. halos
        return imgprop.image_propagation(frames, masks, flows.contiguous())

    def encode(self, frames, prop, md, upd):
        n, H, W, _ = frames.shape
        packed = torch.empty(n, H, W, 8, device=frames.device, dtype=self.m.inpaint_model.dt)
        ops.pack_encoder_input(frames.contiguous(), prop.contiguous(), md.contiguous(), upd.contiguous(), packed)
        return self.m.inpaint_model.encode(packed)

    def make_state(self, enc, flows, md, upd):
        return self.m.inpaint_model.prepare_clip(None, flows, md, upd, enc=enc)

    def propagate_windows(self, st, windows):
        return self.m.inpaint_model.propagate_windows(st, windows)

    def forward_window(self, st, nb, refs, local_prop):
        return self.m.inpaint_model.forward_window(st, nb, refs, local_prop=local_prop)

    def compose(self, comp, pred, frame_ids, first, md, frames_u8):
        dev = comp.device
        ops.compose_u8(pred.contiguous(), ops.device_ints(frame_ids, dev, torch.int32),
                       ops.device_ints(first, dev, torch.int32), md, frames_u8, comp)

    def to_frames(self, frames_u8):             # uint8 -> fp32 in [-1,1]  (image_utils.py:191)
        return ops.frames_from_u8(frames_u8.contiguous())

    # geometry every rank can compute locally (r02 agreed on it with two 3-integer all_gathers on the critical path)
    def enc_tail(self, hw):                     # encoder features of one frame: [H/4, W/4, 128]  (propainter.py:234-275)
        return (hw[0] // 4, hw[1] // 4, 128)

    def pred_tail(self, hw):                    # generator output of one local frame: [H, W, 4] (3 used)
        return (hw[0], hw[1], 4)


# ------------------------------------------------------------------------------------------------
# the sharded driver (a generator: it yields communication requests and receives their results)
#   yield tensor                      -> all_gather: the list of every rank's tensor (same shape everywhere)
#   yield ("p2p", sends, recvs)       -> neighbour exchange: sends {peer: tensor}, recvs {peer: (shape, dtype)};
#                                        the value sent back is {peer: tensor}
#   yield ("p2p_start", sends, recvs) -> the same exchange posted asynchronously: the value sent back is a handle;
#   yield ("p2p_wait", handle)        -> ... and completed: {peer: tensor}.  Work yielded in between runs under the transfer.
# ------------------------------------------------------------------------------------------------
def _pad_first(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.shape[0] == n:
        return t.contiguous()
    pad = torch.zeros((n - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, pad], 0)


class Slab:
    """Rows [lo, lo + len) of a clip-long tensor (a rank only holds the frames it needs: its own plus halos)."""

    def __init__(self, lo: int, t: torch.Tensor):
        self.lo, self.t = lo, t

    @property
    def hi(self) -> int:
        return self.lo + self.t.shape[0]

    def rows(self, a: int, b: int) -> torch.Tensor:
        if a < self.lo or b > self.hi:
            raise IndexError(f"rows [{a},{b}) outside the slab [{self.lo},{self.hi})")
        return self.t[a - self.lo:b - self.lo]


def _interval(rng: tuple[int, int], halo: int, n: int) -> tuple[int, int]:
    a, b = rng
    return (a, a) if b <= a else (max(0, a - halo), min(n, b + halo))


def _halo_plan(own: torch.Tensor, ranges: list[tuple[int, int]], rank: int, need: list[tuple[int, int]]):
    a, b = ranges[rank]
    lo, hi = need[rank]
    sends, recvs = {}, {}
    for r, (ra, rb) in enumerate(ranges):
        if r == rank:
            continue
        x, y = max(a, need[r][0]), min(b, need[r][1])         # my rows that peer r needs
        if y > x:
            sends[r] = own[x - a:y - a].contiguous()
        x, y = max(ra, lo), min(rb, hi)                       # peer r's rows that I need
        if y > x:
            recvs[r] = ((y - x,) + tuple(own.shape[1:]), own.dtype)
    return sends, recvs


def _halo_buffer(own: torch.Tensor, ranges, rank: int, need, poison: bool = False) -> Slab:
    a, b = ranges[rank]
    lo, hi = need[rank]
    buf = torch.empty((hi - lo,) + tuple(own.shape[1:]), dtype=own.dtype, device=own.device)
    if poison and buf.is_floating_point():
        buf.fill_(float("nan"))          # (tests: a window that reads a halo row before it landed cannot go unnoticed)
    if b > a:
        buf[a - lo:b - lo] = own
    return Slab(lo, buf)


def _halo_fill(slab: Slab, got: dict, ranges, recvs) -> None:
    lo, hi = slab.lo, slab.hi
    for r, (ra, rb) in enumerate(ranges):
        if r in recvs:
            x, y = max(ra, lo), min(rb, hi)
            slab.t[x - lo:y - lo] = got[r]


def _halo_exchange(own: torch.Tensor, ranges: list[tuple[int, int]], rank: int, need: list[tuple[int, int]]):
    """Sub-generator: `own` holds this rank's rows `ranges[rank]` of a clip-long tensor (row = dim 0); every rank r needs
    the rows `need[r]` (an interval containing its own range).  Each rank sends a peer exactly the part of its rows the
    peer needs -- for halos of a few frames that is the two neighbours only -- and returns a Slab over `need[rank]`."""
    sends, recvs = _halo_plan(own, ranges, rank, need)
    got = yield ("p2p", sends, recvs)
    slab = _halo_buffer(own, ranges, rank, need)
    _halo_fill(slab, got, ranges, recvs)
    return slab


def run_rank(backend, plan: ShardPlan, config: ProPainterConfig, frames_u8, flow_masks_u8, masks_dilated_u8,
             gather_root: int | None = None):
    """Generator implementing phases A..D for one rank.  `frames_u8` is the whole clip [T,H,W,3] or a `Slab` holding at
    least `frames_needed(plan)`; the masks are clip-long (they are replicated host plumbing, 0.23 MB per frame).
    Returns the full composed clip (uint8 [T,H,W,3]) on every rank; with `gather_root` = r only rank r receives the clip
    (every other rank sends its own frames to r and returns None): the node's return value lives on one rank (r04)."""
    T = plan.T
    F0, F1 = plan.frames
    fa, fb = plan.flows
    nflow = T - 1
    fr = frames_u8 if isinstance(frames_u8, Slab) else Slab(0, frames_u8)
    f_lo, f_hi = frames_needed(plan)
    dev = fr.t.device
    frames_loc = Slab(f_lo, backend.to_frames(fr.rows(f_lo, f_hi))) if f_hi > f_lo else Slab(f_lo, fr.t[0:0].float())
    hw = tuple(fr.t.shape[1:3])
    # ---- A: RAFT on the owned pairs; x0: the 5-flow completion halos come from the neighbours (RAFT is per-pair
    # independent -- tests/test_raft.py -- so a neighbour's flows are the ones this rank would have computed itself)
    raw = backend.raft(frames_loc.rows(fa, fb + 1)) if fb > fa else torch.zeros((2, 0) + hw + (2,), device=dev)
    need0 = [_interval(r, 5, nflow) for r in plan.flow_ranges]
    raw_s = yield from _halo_exchange(raw.transpose(0, 1), plan.flow_ranges, plan.rank, need0)   # rows = flows, [n,2,H,W,2]
    own = []
    for f, e_own, s, e in plan.flow_chunks():
        gt = raw_s.rows(s, e).transpose(0, 1).contiguous()
        sub = backend.complete(gt, flow_masks_u8[s:e + 1])
        own.append(sub[:, f - s:e_own - s])
    own_flows = torch.cat(own, 1) if own else torch.zeros((2, 0) + hw + (2,), device=dev)
    schedule = window_schedule(config)
    ns = config.neighbor_length // 2
    centers = [wi * ns for wi in range(len(schedule))]

    # x1: completed flows read by rank r: its image-propagation chunks (10 flows either side) AND the flows between the
    # local (neighbour) frames of the windows centred in its frames -- feature propagation warps along them
    # (propainter.py:149-205); for neighbor_length // 2 > 10 those reach beyond the image-propagation halo (r02 left them
    # zero there: silently wrong seam windows for neighbor_length >= 22)
    def flows_for_rank(r):
        ch = plan.frame_chunks(r)     # (a rank may own frames but no flow: the last frame of the clip)
        a = plan.flow_ranges[r][0]
        if not ch:
            return (a, a)
        lo, hi = min(c[2] for c in ch), max(c[3] for c in ch) - 1
        F0r, F1r = plan.frame_ranges[r]
        for wi, c in enumerate(centers):
            if F0r <= c < F1r:
                nb = schedule[wi][0]
                lo, hi = min(lo, nb[0]), max(hi, nb[-1])     # flows nb[0] .. nb[-1]-1
        return (lo, hi)

    need1 = [flows_for_rank(r) for r in range(plan.world)]
    pred_s = yield from _halo_exchange(own_flows.transpose(0, 1), plan.flow_ranges, plan.rank, need1)
    # ---- B: image propagation of the owned chunks, blend + encoder on the owned frames ---------------
    props, upds = [], []
    for f, e_own, s, e in plan.frame_chunks():
        pr = pred_s.rows(s, e - 1).transpose(0, 1)
        p, m = backend.img_prop(frames_loc.rows(s, e), masks_dilated_u8[s:e], pr)
        props.append(p[f - s:e_own - s])
        upds.append(m[f - s:e_own - s])
    def frames_of_windows(rng):       # frames read by the windows centred in `rng` (neighbours + references)
        ids = [i for wi, c in enumerate(centers) if rng[0] <= c < rng[1] for i in schedule[wi][0] + schedule[wi][1]]
        return (min(ids + [rng[0]]), max(ids + [rng[1] - 1]) + 1) if rng[1] > rng[0] else (rng[0], rng[0])

    need2 = [frames_of_windows(r) for r in plan.frame_ranges]
    if props:
        prop, upd = torch.cat(props, 0), torch.cat(upds, 0)
        enc_own = backend.encode(frames_loc.rows(F0, F1), prop, masks_dilated_u8[F0:F1], upd)
    else:
        upd = masks_dilated_u8[0:0]
        enc_own = None
    if enc_own is None:   # an idle rank still takes part in the exchange: the geometry is a function of the frame size
        enc_own = torch.zeros((0,) + tuple(backend.enc_tail(hw)), device=dev, dtype=getattr(backend, "act_dtype", torch.float16))
    # x2: encoder features + updated masks of the frames this rank's windows read (+-45 frames at the defaults).  The masks
    # (0.23 MB per frame) travel first, blocking; the features (0.35 GB per seam at 640x360) are POSTED and travel under the
    # feature propagation of the windows whose local frames are all this rank's own (r04: it was a blocking exchange between
    # phases B and C) -- feature propagation reads the local frames of a window only (propainter.py:118-231); the reference
    # frames, up to 40 frames away, are first read by the transformer
    upd_s = yield from _halo_exchange(upd, plan.frame_ranges, plan.rank, need2)
    enc_sends, enc_recvs = _halo_plan(enc_own, plan.frame_ranges, plan.rank, need2)
    enc_handle = yield ("p2p_start", enc_sends, enc_recvs)
    enc_s = _halo_buffer(enc_own, plan.frame_ranges, plan.rank, need2, poison=os.environ.get("PP_POISON_HALOS") == "1")
    S0, S1 = need2[plan.rank]
    # ---- C: the windows centred in the owned frames, on a clip state over [S0, S1) only --------------------------
    # x3 (window outputs that land on frames of another rank) is posted as soon as the SEAM windows are done and
    # travels under the transformer of the interior windows; it is waited for where compose needs it
    mine = [wi for wi, f in enumerate(centers) if F0 <= f < F1]
    exports: list[dict[int, list[tuple[int, int]]]] = [dict() for _ in range(plan.world)]   # [src][dst] -> [(wi, idx)]
    for wi, (nb, _) in enumerate(schedule):
        src = plan.owner_of_frame(centers[wi])
        for idx in nb:
            dst = plan.owner_of_frame(idx)
            if dst != src:
                exports[src].setdefault(dst, []).append((wi, idx))
    seam_set = {wi for lst in exports[plan.rank].values() for wi, _ in lst}
    seam = [wi for wi in mine if wi in seam_set]
    interior = [wi for wi in mine if wi not in seam_set]     # (their local frames are all owned: they export nothing)
    H, W = hw
    pred_tail = tuple(backend.pred_tail(hw))
    recvs = {src: ((len(exports[src][plan.rank]),) + pred_tail, getattr(backend, "act_dtype", torch.float16))
             for src in range(plan.world) if src != plan.rank and plan.rank in exports[src]}
    preds = {}
    st = loc = None
    if mine:
        flows_loc = torch.zeros((2, max(S1 - S0 - 1, 0)) + hw + (2,), device=dev)
        x, y = max(S0, pred_s.lo), min(S1 - 1, pred_s.hi)        # flows beyond [x, y) lie between reference frames only
        if y > x:                                                #  and are never read (need1 covers the local frames)
            flows_loc[:, x - S0:y - S0] = pred_s.rows(x, y).transpose(0, 1)
        st = backend.make_state(enc_s.t, flows_loc, masks_dilated_u8[S0:S1].contiguous(), upd_s.t)
        loc = {wi: ([i - S0 for i in schedule[wi][0]], [i - S0 for i in schedule[wi][1]]) for wi in mine}
    local_props = {}
    if interior:        # under x2
        lp = backend.propagate_windows(st, [loc[wi][0] for wi in interior])
        local_props.update({wi: lp[j] for j, wi in enumerate(interior)})
    got = yield ("p2p_wait", enc_handle)
    _halo_fill(enc_s, got, plan.frame_ranges, enc_recvs)
    if mine and hasattr(backend, "enc_landed"):
        backend.enc_landed(st, enc_s.t)      # (a backend whose state copied the features refreshes its copy)

    def run_windows(group):
        todo = [wi for wi in group if wi not in local_props]
        if todo:
            lp = backend.propagate_windows(st, [loc[wi][0] for wi in todo])
            local_props.update({wi: lp[j] for j, wi in enumerate(todo)})
        for wi in group:
            preds[wi] = backend.forward_window(st, loc[wi][0], loc[wi][1], local_props.pop(wi))

    run_windows(seam)
    sends = {dst: torch.stack([preds[wi][schedule[wi][0].index(idx)] for wi, idx in lst], 0)
             for dst, lst in exports[plan.rank].items()}
    handle = yield ("p2p_start", sends, recvs)
    run_windows(interior)
    got = yield ("p2p_wait", handle)
    foreign = {}
    for src, t in got.items():
        for j, key in enumerate(exports[src][plan.rank]):
            foreign[key] = t[j]
    # ---- D: compose the owned frames in global window order -----------------------------------------------
    comp = torch.zeros((F1 - F0,) + hw + (3,), dtype=torch.uint8, device=dev)
    orig = fr.rows(F0, F1).contiguous()
    md_own = masks_dilated_u8[F0:F1].contiguous()
    seen = [False] * T
    for wi, (nb, _) in enumerate(schedule):
        ids = [idx for idx in nb if F0 <= idx < F1]
        if not ids:
            continue
        if wi in preds:
            p = torch.stack([preds[wi][nb.index(idx)] for idx in ids], 0)
        else:
            p = torch.stack([foreign[(wi, idx)] for idx in ids], 0)
        backend.compose(comp, p, [i - F0 for i in ids], [0 if seen[i] else 1 for i in ids], md_own, orig)
        for i in ids:
            seen[i] = True
    # x4: the composed uint8 frames (0.7 MB each): to every rank (all_gather), or to the one rank that returns the clip
    if gather_root is not None:
        if plan.rank != gather_root:
            yield ("p2p", ({gather_root: comp} if F1 > F0 else {}), {})
            return None
        recvs = {r: ((b - a,) + hw + (3,), torch.uint8) for r, (a, b) in enumerate(plan.frame_ranges) if r != gather_root and b > a}
        got = yield ("p2p", {}, recvs)
        return torch.cat([comp if r == gather_root else got[r] for r, (a, b) in enumerate(plan.frame_ranges) if b > a], 0)
    max_frames = max(b - a for a, b in plan.frame_ranges)
    g_comp = yield _pad_first(comp, max_frames)
    return torch.cat([g[:b - a] for g, (a, b) in zip(g_comp, plan.frame_ranges)], 0)


def frames_needed(plan: ShardPlan) -> tuple[int, int]:
    """Frames a rank has to hold: its RAFT pairs and its image-propagation chunks with their 10-frame halos."""
    F0, F1 = plan.frames
    fa, fb = plan.flows
    if F1 <= F0:
        return (F0, F0)
    return (max(0, min(F0 - 10, fa)), min(plan.T, max(F1 + 10, fb + 1)))


# ------------------------------------------------------------------------------------------------
# runners
# ------------------------------------------------------------------------------------------------
def run_distributed(backend, config: ProPainterConfig, frames_u8, flow_masks_u8, masks_dilated_u8, group=None,
                    gather_root: int | None = None, timeline: list | None = None):
    """One rank of a torch.distributed job (backend "nccl" = RCCL on the MI355X, "gloo" in CPU tests).  all_gather for the
    two whole-clip exchanges, grouped point-to-point sends / receives (xGMI is point to point: a seam travels over the one
    link between the two neighbours) for the halos."""
    import torch.distributed as dist

    plan = ShardPlan(config.video_length, config.subvideo_length, dist.get_world_size(group), dist.get_rank(group))
    gen = run_rank(backend, plan, config, frames_u8, flow_masks_u8, masks_dilated_u8, gather_root=gather_root)
    via_host = dist.get_backend(group) == "gloo"  # gloo moves host memory: stage through the host (tests only)
    dev = (frames_u8.t if isinstance(frames_u8, Slab) else frames_u8).device

    def wire(t):
        return t.cpu() if via_host and t.is_cuda else t.contiguous()

    def post(sends, recvs):
        bufs = {peer: torch.empty(shape, dtype=dtype, device="cpu" if via_host else dev)
                for peer, (shape, dtype) in recvs.items()}
        keep = [wire(ten) for ten in sends.values()]         # the wire tensors must outlive the requests
        p2p = [dist.P2POp(dist.irecv, buf, peer, group) for peer, buf in bufs.items()]
        p2p += [dist.P2POp(dist.isend, ten, peer, group) for peer, ten in zip(sends.keys(), keep)]
        return bufs, (dist.batch_isend_irecv(p2p) if p2p else []), keep

    def finish(pending):
        bufs, reqs, _keep = pending
        for req in reqs:
            req.wait()
        return {peer: buf.to(dev) for peer, buf in bufs.items()}

    # PP_P2P_ASYNC=0: complete every posted exchange at once (the blocking form of r02) -- a switch for operators should the
    # overlap of batch_isend_irecv with the compute stream misbehave on some RCCL build; the default keeps exchanges in flight
    blocking = os.environ.get("PP_P2P_ASYNC", "1") == "0"
    # `timeline` (bench.py --gpus N, one extra instrumented step): this rank's wall clock per segment, with a device synchronize
    # at every hand-over so that compute and exchange times separate -- [("compute" | kind, milliseconds, bytes sent)], kinds
    # in the order of the protocol: p2p (x0 raw-flow halos, x1 completed-flow halos, x2 masks), p2p_start / p2p_wait (x2 encoder
    # features, x3 seam-window outputs: `p2p_wait` is what the rank still had to wait for after the work it did meanwhile), p2p or
    # all_gather (x4 composed frames)
    import time as _time

    def _mark(kind, nbytes=0):
        if timeline is None:
            return
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        now = _time.perf_counter()
        timeline.append((kind, round((now - _mark.t0) * 1e3, 3), int(nbytes)))
        _mark.t0 = now
    _mark.t0 = _time.perf_counter()

    def _nbytes(sends):
        return sum(v.numel() * v.element_size() for v in sends.values())
    try:
        t = next(gen)
        while True:
            _mark("compute")
            if isinstance(t, tuple) and t[0] == "p2p_start":
                out = ("done", finish(post(t[1], t[2]))) if blocking else post(t[1], t[2])
            elif isinstance(t, tuple) and t[0] == "p2p_wait":
                out = t[1][1] if (isinstance(t[1], tuple) and len(t[1]) == 2 and t[1][0] == "done") else finish(t[1])
            elif isinstance(t, tuple):
                out = finish(post(t[1], t[2]))
            else:
                th = wire(t)
                outh = [torch.empty_like(th) for _ in range(plan.world)]
                dist.all_gather(outh, th, group=group)
                out = [o.to(t.device) for o in outh]
            if timeline is not None:
                _mark(t[0] if isinstance(t, tuple) else "all_gather",
                      _nbytes(t[1]) if isinstance(t, tuple) and isinstance(t[1], dict) else (0 if isinstance(t, tuple) else t.numel() * t.element_size()))
            t = gen.send(out)
    except StopIteration as stop:
        _mark("compute")
        return stop.value


def run_simulated(make_backend, world: int, config: ProPainterConfig, frames_u8, flow_masks_u8, masks_dilated_u8,
                  gather_root: int | None = None):
    """N virtual ranks advanced in lock-step inside ONE process (functional testing on a single GPU)."""
    plans = [ShardPlan(config.video_length, config.subvideo_length, world, r) for r in range(world)]
    gens = [run_rank(make_backend(r), plans[r], config, frames_u8, flow_masks_u8, masks_dilated_u8, gather_root=gather_root)
            for r in range(world)]
    vals = [next(g) for g in gens]
    results = [None] * world
    done = [False] * world

    class _Pending:     # what a rank holds between p2p_start and p2p_wait: opaque, so it cannot read a buffer before the wait
        def __init__(self, reply):
            self._reply = reply

    while not all(done):
        nxt = []
        for r, g in enumerate(gens):
            if done[r]:
                nxt.append(None)
                continue
            if isinstance(vals[r], tuple) and vals[r][0] == "p2p_start":
                # every rank posts in the same lock-step round: deliver at once, hand the result back at the wait
                _, _, recvs = vals[r]
                reply = {}
                for src, (shape, dtype) in recvs.items():
                    t = vals[src][1][r]
                    assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (src, r, tuple(t.shape), shape)
                    reply[src] = t.clone()
                assert all(r in vals[dst][2] for dst in vals[r][1]), "a send without a matching receive"
                reply = _Pending(reply)
            elif isinstance(vals[r], tuple) and vals[r][0] == "p2p_wait":
                reply = vals[r][1]._reply
            elif isinstance(vals[r], tuple):    # neighbour exchange: what every peer addressed to rank r
                _, _, recvs = vals[r]
                reply = {}
                for src, (shape, dtype) in recvs.items():
                    t = vals[src][1][r]
                    assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (src, r, tuple(t.shape), shape)
                    reply[src] = t.clone()
                assert all(r in vals[dst][2] for dst in vals[r][1]), "a send without a matching receive"
            else:
                reply = [v.clone() for v in vals]
            try:
                nxt.append(g.send(reply))
            except StopIteration as stop:
                results[r] = stop.value
                done[r] = True
                nxt.append(None)
        vals = nxt
    return results


# ------------------------------------------------------------------------------------------------
# in-process multi-device runner: ONE process drives N GPUs (the drop-in node itself shards, PP_GPUS=N)
# ------------------------------------------------------------------------------------------------
class _Mailbox:
    """Rendezvous of the rank threads: (sequence number, src, dst) -> (tensor, ready event).  A failing rank poisons the box so
    that its peers stop waiting."""

    def __init__(self):
        import threading

        self.cv = threading.Condition()
        self.box: dict = {}
        self.failed: BaseException | None = None

    def put(self, key, item) -> None:
        with self.cv:
            self.box[key] = item
            self.cv.notify_all()

    def take(self, key, timeout_s: float = 900.0):
        import time

        t_end = time.monotonic() + timeout_s
        with self.cv:
            while key not in self.box:
                if self.failed is not None:
                    raise RuntimeError("a peer rank failed") from self.failed
                if time.monotonic() > t_end:
                    raise TimeoutError(f"no message {key} within {timeout_s} s")
                self.cv.wait(0.5)
            return self.box.pop(key)

    def fail(self, exc: BaseException) -> None:
        with self.cv:
            if self.failed is None:
                self.failed = exc
            self.cv.notify_all()


class _PeerLink:
    """The receiving side of the peer copies of one rank thread.  A copy between two GPUs runs on a SIDE stream of the source
    device (torch issues a cross-device copy on the source device's current stream of the calling thread) bracketed by a side
    stream of this rank's device, so neither the sender's nor the receiver's compute stream waits for it until the receiver asks
    for the data (`finish`): a posted exchange travels over xGMI under the kernels both sides keep launching."""

    def __init__(self, device: torch.device):
        self.device = device
        self.cuda = device.type == "cuda"
        self.side = torch.cuda.Stream(device) if self.cuda else None
        self.src_streams: dict = {}

    def ready_event(self):
        """(sender side) the data produced so far on this rank's compute stream is complete."""
        if not self.cuda:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def fetch(self, t: torch.Tensor, ev, shape, dtype) -> tuple:
        """Start copying peer tensor `t` (complete at `ev`) to this device; returns (buffer, done event)."""
        assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (tuple(t.shape), tuple(shape), t.dtype, dtype)
        if not self.cuda:
            return t.clone(), None
        if not t.is_cuda:
            return t.to(self.device, non_blocking=True), None
        if t.device == self.device:      # ranks sharing a device (virtual ranks): an ordinary copy on this rank's side stream
            if ev is not None:
                self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                buf = torch.empty(shape, dtype=dtype, device=self.device)
                buf.copy_(t, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.side)
            t.record_stream(self.side)
            return buf, done
        s_src = self.src_streams.get(t.device)
        if s_src is None:
            s_src = self.src_streams[t.device] = torch.cuda.Stream(t.device)
        if ev is not None:
            s_src.wait_event(ev)
        # torch issues the copy on the SOURCE device's current stream and makes the destination device's current stream wait
        # for it: both are side streams here, `done` (recorded on this device's side stream) therefore covers the copy
        with torch.cuda.stream(self.side), torch.cuda.stream(s_src):
            buf = torch.empty(shape, dtype=dtype, device=self.device)
            buf.copy_(t, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.side)
        t.record_stream(s_src)          # the sender may drop its tensor: the allocator must not reuse it under the copy
        return buf, done

    def finish(self, buf: torch.Tensor, done) -> torch.Tensor:
        if done is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            buf.record_stream(cur)
        return buf


def run_multi_device(backends: list, config: ProPainterConfig, load_slab, flow_masks_u8, masks_dilated_u8, devices: list,
                     gather_root: int | None = 0):
    """ONE process, N devices, one thread per device: the sharded driver behind the node (PP_GPUS=N) -- a ComfyUI process that owns
    the 8 GPUs of a node partitions a long clip over them without torch.distributed (north_star: "the pipeline partitions clips
    over the 8 GPUs of one node", propainter_nodes.py:109 picks ONE device per call).  Same plan, same generator (`run_rank`), same
    seam exchanges as the one-process-per-GPU runner; the exchanges are peer copies (hipMemcpyPeer over xGMI) ordered by events
    instead of RCCL point-to-point calls, the whole-clip all_gather is replaced by a gather to `gather_root`.

    backends[r]     stage functions on devices[r] (GpuBackend over that device's models)
    load_slab(r, lo, hi, device) -> uint8 frames [hi-lo, H, W, 3] of the clip on that device (each rank uploads only what it needs)
    masks           clip-long uint8 tensors (any device / host): copied to every device
    Devices may repeat (virtual ranks on one GPU: the functional test of this path on a 1-GPU box).
    Returns the composed clip on devices[gather_root] (gather_root None: a list with every rank's copy)."""
    import threading

    world = len(devices)
    devices = [torch.device(d) for d in devices]
    T = config.video_length
    box = _Mailbox()
    results: list = [None] * world
    errors: list = [None] * world

    def rank_main(r: int) -> None:
        dev = devices[r]
        try:
            if dev.type == "cuda":
                torch.cuda.set_device(dev)          # (the HIP current device is per thread)
                # a compute stream of its own per rank: ranks that share a device (virtual ranks) must not meet on the legacy
                # default stream -- a launch there while another rank captures a hipGraph invalidates the capture
                compute = torch.cuda.Stream(dev)
                compute.wait_stream(torch.cuda.default_stream(dev))
                torch.cuda.set_stream(compute)
            link = _PeerLink(dev)
            plan = ShardPlan(T, config.subvideo_length, world, r)
            lo, hi = frames_needed(plan)
            fr = Slab(lo, load_slab(r, lo, hi, dev))
            fm = flow_masks_u8.to(dev, non_blocking=True)
            md = masks_dilated_u8.to(dev, non_blocking=True)
            gen = run_rank(backends[r], plan, config, fr, fm, md, gather_root=gather_root)
            seq = 0

            def post(sends, recvs):
                nonlocal seq
                ev = link.ready_event() if sends else None
                for peer, ten in sends.items():
                    box.put((seq, r, peer), (ten, ev))
                pend = {}
                for peer, (shape, dtype) in recvs.items():
                    ten, pev = box.take((seq, peer, r))
                    pend[peer] = link.fetch(ten, pev, shape, dtype)
                seq += 1
                return pend

            def finish(pend):
                return {peer: link.finish(buf, done) for peer, (buf, done) in pend.items()}

            try:
                t = next(gen)
                while True:
                    if isinstance(t, tuple) and t[0] == "p2p_start":
                        out = post(t[1], t[2])
                    elif isinstance(t, tuple) and t[0] == "p2p_wait":
                        out = finish(t[1])
                    elif isinstance(t, tuple):
                        out = finish(post(t[1], t[2]))
                    else:       # all_gather (gather_root None): every rank's tensor to every rank
                        sends = {p: t for p in range(world) if p != r}
                        recvs = {p: (tuple(t.shape), t.dtype) for p in range(world) if p != r}
                        got = finish(post(sends, recvs))
                        out = [t if p == r else got[p] for p in range(world)]
                    t = gen.send(out)
            except StopIteration as stop:
                results[r] = stop.value
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()     # the caller reads the result from another thread / stream
        except BaseException as e:  # noqa: BLE001 -- surfaced by the caller
            errors[r] = e
            box.fail(e)

    threads = [threading.Thread(target=rank_main, args=(r,), name=f"pp-rank{r}", daemon=True) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for e in errors:
        if e is not None:
            raise e
    return results if gather_root is None else results[gather_root]
