"""Multi-GPU execution of ONE long clip: sub-video sharding with seam exchange.

The reference already cuts a long clip into sub-videos of `subvideo_length` frames with halos
(propainter_inference.py:115-144, :172-212) and, for T > subvideo_length, only looks
+-ref_stride*(ref_num//2) frames around a window for reference frames (:36-58).  Those chunks are the
shards: rank r owns a contiguous run of sub-video chunks, i.e. frames [F0, F1).  Per rank:

  A  RAFT on the frame pairs it owns
  x0 all_gather of the raw RAFT flows at the seams (5 per side: the flow-completion halos)
  A' flow completion of its chunks
  x1 all_gather of the completed flows                       (RCCL over xGMI / gloo in tests)
  B  image propagation of its chunks (+10-frame halo from x1), blend, encoder on its frames
  x2 all_gather of encoder features + updated masks
  C  feature propagation + transformer + decoder for the windows centred in [F0, F1)
  x3 all_gather of the window outputs that land on frames owned by a neighbour (seam windows)
  D  uint8 compose of its own frames in GLOBAL window order (the blend is order dependent)
  x4 all_gather of the composed frames

Because chunk boundaries, halos and window schedule are exactly the single-GPU ones, the sharded
result is identical to the single-process result.  The driver is written against a small backend
protocol so that the orchestration can be tested on CPU (gloo, world_size 2) with a toy backend and
on one GPU with N in-process virtual ranks; `GpuBackend` is the real thing.

The collectives gather whole per-rank slabs (affordable: 640 frames x 3.7 MB = 2.4 GB per exchange,
xGMI is point-to-point so every peer pair moves its slice concurrently); trimming them to the seams
is a bandwidth optimisation that does not change results.
"""
from __future__ import annotations

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

    def raft(self, frames):                     # fp32 [n,H,W,3] -> [2,n-1,H,W,2]
        ff, fb = self.m.raft_model(frames, self.cfg.raft_iter)
        return torch.stack([ff, fb], 0)

    def complete(self, flows, masks):           # one chunk incl. halos
        return self.m.flow_model(flows.contiguous(), masks.contiguous())

    def img_prop(self, frames, masks, flows):   # one chunk incl. halos
        return imgprop.image_propagation(frames, masks, flows.contiguous())

    def encode(self, frames, prop, md, upd):
        n, H, W, _ = frames.shape
        packed = torch.empty(n, H, W, 8, device=frames.device, dtype=torch.float16)
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


# ------------------------------------------------------------------------------------------------
# the sharded driver (a generator: yields tensors to all_gather, receives the per-rank list)
# ------------------------------------------------------------------------------------------------
def _pad_first(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.shape[0] == n:
        return t.contiguous()
    pad = torch.zeros((n - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, pad], 0)


def run_rank(backend, plan: ShardPlan, config: ProPainterConfig, frames_u8, flow_masks_u8, masks_dilated_u8):
    """Generator implementing phases A..D for one rank. `yield tensor` = all_gather (same shape on every rank);
    the value sent back is the list of every rank's tensor.  Returns the full composed clip (uint8 [T,H,W,3])."""
    T = plan.T
    F0, F1 = plan.frames
    frames_all = backend.to_frames(frames_u8)
    # ---- A: RAFT on the owned pairs, seam exchange of the raw flows, flow completion of the owned chunks --------
    # (RAFT is per-pair independent -- tests/test_raft.py -- so a neighbour's flows are the ones this rank would
    #  have computed itself: the 5-flow completion halos are exchanged instead of recomputed)
    fa, fb = plan.flows
    HAL = 5
    dev = frames_all.device
    zero_flows = torch.zeros((2, 0) + tuple(frames_all.shape[1:3]) + (2,), device=dev)
    raw = backend.raft(frames_all[fa:fb + 1]) if fb > fa else zero_flows
    k = min(HAL, fb - fa)
    # x0: the first / last (up to) 5 raw flows of every rank
    g_seam = yield torch.cat([_pad_first(raw[:, :k].transpose(0, 1), HAL),
                              _pad_first(raw[:, fb - fa - k:].transpose(0, 1), HAL)], 0)

    def raw_flow(g: int) -> torch.Tensor:        # global flow index -> [2,H,W,2]
        if fa <= g < fb:
            return raw[:, g - fa]
        for r, (a, b) in enumerate(plan.flow_ranges):
            if a <= g < b:
                kr = min(HAL, b - a)
                if g - a < kr:
                    return g_seam[r][g - a]
                if b - g <= kr:
                    return g_seam[r][HAL + (g - (b - kr))]
                raise AssertionError(f"flow {g} is not inside a seam of rank {r}")
        raise IndexError(g)

    max_flows = max(b - a for a, b in plan.flow_ranges)
    own = []
    for f, e_own, s, e in plan.flow_chunks():
        left = [raw_flow(g) for g in range(s, min(e, fa))]
        right = [raw_flow(g) for g in range(max(s, fb), e)]
        mid = raw[:, max(s, fa) - fa:min(e, fb) - fa]
        gt = torch.cat(([torch.stack(left, 1)] if left else []) + [mid] + ([torch.stack(right, 1)] if right else []), 1)
        sub = backend.complete(gt, flow_masks_u8[s:e + 1])
        own.append(sub[:, f - s:e_own - s])
    flow_shape = (2, 0) + tuple(frames_all.shape[1:3]) + (2,)
    own_flows = torch.cat(own, 1) if own else torch.zeros(flow_shape, device=frames_all.device)
    # x1: completed flows of every rank
    gathered = yield _pad_first(own_flows.transpose(0, 1), max_flows)
    parts = [g[:b - a].transpose(0, 1) for g, (a, b) in zip(gathered, plan.flow_ranges)]
    pred = torch.cat(parts, 1).contiguous()                       # [2,T-1,H,W,2]
    # ---- B: image propagation of the owned chunks, blend + encoder on the owned frames ---------------
    max_frames = max(b - a for a, b in plan.frame_ranges)
    props, upds = [], []
    for f, e_own, s, e in plan.frame_chunks():
        p, m = backend.img_prop(frames_all[s:e], masks_dilated_u8[s:e], pred[:, s:e - 1])
        props.append(p[f - s:e_own - s])
        upds.append(m[f - s:e_own - s])
    if props:
        prop, upd = torch.cat(props, 0), torch.cat(upds, 0)
        enc_own = backend.encode(frames_all[F0:F1], prop, masks_dilated_u8[F0:F1], upd)
    else:
        upd = masks_dilated_u8[0:0]
        enc_own = None
    # x2: encoder features + updated masks of every rank
    enc_shape = yield torch.tensor(list(enc_own.shape[1:]) if enc_own is not None else [0, 0, 0], device=frames_all.device)
    eshape = [int(v) for v in max(enc_shape, key=lambda t: int(t.sum()))]
    if enc_own is None:
        enc_own = torch.zeros([0] + eshape, device=frames_all.device, dtype=torch.float16)
    g_enc = yield _pad_first(enc_own, max_frames)
    g_upd = yield _pad_first(upd, max_frames)
    enc = torch.cat([g[:b - a] for g, (a, b) in zip(g_enc, plan.frame_ranges)], 0)
    upd_all = torch.cat([g[:b - a] for g, (a, b) in zip(g_upd, plan.frame_ranges)], 0)
    st = backend.make_state(enc, pred, masks_dilated_u8, upd_all)
    # ---- C: the windows centred in the owned frames ------------------------------------------------------
    schedule = window_schedule(config)
    ns = config.neighbor_length // 2
    centers = [wi * ns for wi in range(len(schedule))]
    mine = [wi for wi, f in enumerate(centers) if F0 <= f < F1]
    lp = backend.propagate_windows(st, [schedule[wi][0] for wi in mine]) if mine else []
    preds = {wi: backend.forward_window(st, schedule[wi][0], schedule[wi][1], lp[j]) for j, wi in enumerate(mine)}
    # x3: window outputs that land on frames of another rank
    exports: list[list[tuple[int, int]]] = [[] for _ in range(plan.world)]
    for wi, (nb, _) in enumerate(schedule):
        r = plan.owner_of_frame(centers[wi])
        exports[r] += [(wi, idx) for idx in nb if plan.owner_of_frame(idx) != r]
    max_exp = max(1, max(len(e) for e in exports))
    H, W = frames_all.shape[1:3]
    pred_tail = tuple(next(iter(preds.values())).shape[1:]) if preds else (H, W, 4)
    tail_g = yield torch.tensor(list(pred_tail), device=frames_all.device)
    pred_tail = tuple(int(v) for v in max(tail_g, key=lambda t: int(t.sum())))
    exp = torch.zeros((max_exp,) + pred_tail, device=frames_all.device, dtype=torch.float16)
    for j, (wi, idx) in enumerate(exports[plan.rank]):
        exp[j] = preds[wi][schedule[wi][0].index(idx)]
    g_exp = yield exp
    foreign = {}
    for r, lst in enumerate(exports):
        for j, (wi, idx) in enumerate(lst):
            if F0 <= idx < F1:
                foreign[(wi, idx)] = g_exp[r][j]
    # ---- D: compose the owned frames in global window order -----------------------------------------------
    comp = torch.zeros((T,) + tuple(frames_u8.shape[1:]), dtype=torch.uint8, device=frames_all.device)
    seen = [False] * T
    for wi, (nb, _) in enumerate(schedule):
        ids = [idx for idx in nb if F0 <= idx < F1]
        if not ids:
            continue
        if wi in preds:
            p = torch.stack([preds[wi][nb.index(idx)] for idx in ids], 0)
        else:
            p = torch.stack([foreign[(wi, idx)] for idx in ids], 0)
        backend.compose(comp, p, ids, [0 if seen[i] else 1 for i in ids], masks_dilated_u8, frames_u8)
        for i in ids:
            seen[i] = True
    g_comp = yield _pad_first(comp[F0:F1], max_frames)
    return torch.cat([g[:b - a] for g, (a, b) in zip(g_comp, plan.frame_ranges)], 0)


# ------------------------------------------------------------------------------------------------
# runners
# ------------------------------------------------------------------------------------------------
def run_distributed(backend, config: ProPainterConfig, frames_u8, flow_masks_u8, masks_dilated_u8, group=None):
    """One rank of a torch.distributed job (backend "nccl" = RCCL on the MI355X, "gloo" in CPU tests)."""
    import torch.distributed as dist

    plan = ShardPlan(config.video_length, config.subvideo_length, dist.get_world_size(group), dist.get_rank(group))
    gen = run_rank(backend, plan, config, frames_u8, flow_masks_u8, masks_dilated_u8)
    via_host = dist.get_backend(group) == "gloo"  # gloo has no device all_gather: stage through the host (tests only)
    try:
        t = next(gen)
        while True:
            if via_host and t.is_cuda:
                th = t.cpu()
                outh = [torch.empty_like(th) for _ in range(plan.world)]
                dist.all_gather(outh, th, group=group)
                out = [o.to(t.device) for o in outh]
            else:
                out = [torch.empty_like(t) for _ in range(plan.world)]
                dist.all_gather(out, t.contiguous(), group=group)
            t = gen.send(out)
    except StopIteration as stop:
        return stop.value


def run_simulated(make_backend, world: int, config: ProPainterConfig, frames_u8, flow_masks_u8, masks_dilated_u8):
    """N virtual ranks advanced in lock-step inside ONE process (functional testing on a single GPU)."""
    plans = [ShardPlan(config.video_length, config.subvideo_length, world, r) for r in range(world)]
    gens = [run_rank(make_backend(r), plans[r], config, frames_u8, flow_masks_u8, masks_dilated_u8) for r in range(world)]
    vals = [next(g) for g in gens]
    results = [None] * world
    while any(r is None for r in results):
        nxt = []
        for r, g in enumerate(gens):
            try:
                nxt.append(g.send([v.clone() for v in vals]))
            except StopIteration as stop:
                results[r] = stop.value
                nxt.append(None)
        vals = nxt
    return results
