"""Inference driver: the MI355X re-design of propainter_inference.py (reference :17-341).

Same temporal policy as the reference (so results are comparable chunk for chunk):
  compute_flow      :61-99    all adjacent pairs (per-pair results are chunk-invariant)
  complete_flow     :102-156  sub-videos of `subvideo_length` flows with 5-flow halos
  image_propagation :159-225  sub-videos of min(100, subvideo_length) frames with 10-frame halos
  feature_propagation :228-311 sliding neighbour windows + strided reference frames, uint8 compose
but frames stay resident in HBM for the whole clip: one H2D of uint8 frames + masks, one D2H of
the composed uint8 frames; no per-window host round trips, no empty_cache() calls.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import imgprop, ops, weights
from .generator import InpaintGeneratorMI355
from .raft import RaftFlow
from .rfc import FlowCompleter


@dataclass
class ProPainterConfig:
    ref_stride: int
    neighbor_length: int
    subvideo_length: int
    raft_iter: int
    fp16: str
    video_length: int
    device: torch.device
    process_size: tuple[int, int]
    use_half: bool = field(init=False)

    def __post_init__(self) -> None:
        self.use_half = self.fp16 == "enable"
        if self.device == torch.device("cpu"):
            self.use_half = False


def get_ref_index(mid_neighbor_id: int, neighbor_ids: list[int], config: ProPainterConfig, ref_num: int = -1) -> list[int]:
    """Reference-frame ids for one window (propainter_inference.py:36-58), incl. its `len > ref_num` quirk."""
    ref_index: list[int] = []
    if ref_num == -1:
        return [i for i in range(0, config.video_length, config.ref_stride) if i not in neighbor_ids]
    start = max(0, mid_neighbor_id - config.ref_stride * (ref_num // 2))
    end = min(config.video_length, mid_neighbor_id + config.ref_stride * (ref_num // 2))
    for i in range(start, end, config.ref_stride):
        if i not in neighbor_ids:
            if len(ref_index) > ref_num:
                break
            ref_index.append(i)
    return ref_index


def window_schedule(config: ProPainterConfig) -> list[tuple[list[int], list[int]]]:
    """(neighbor_ids, ref_ids) per window, in the reference's order (:247-262)."""
    ns = config.neighbor_length // 2
    ref_num = config.subvideo_length // config.ref_stride if config.video_length > config.subvideo_length else -1
    out = []
    for f in range(0, config.video_length, ns):
        nb = list(range(max(0, f - ns), min(config.video_length, f + ns + 1)))
        out.append((nb, get_ref_index(f, nb, config, ref_num)))
    return out


def final_ranges(schedule, video_length: int) -> list[tuple[int, int]]:
    """Per window (lo, hi): the frames that have received their LAST blend once that window is composed -- no later
    window has them among its local frames -- as consecutive half-open ranges that tile [0, video_length) (hi == lo for a
    window that finishes nothing).  The local-frame ranges of window_schedule() slide forward monotonically, so the
    frames before the next window's first local frame are final."""
    out, done = [], 0
    for wi in range(len(schedule)):
        nxt = min(schedule[wi + 1][0]) if wi + 1 < len(schedule) else video_length
        nxt = max(nxt, done)
        out.append((done, nxt))
        done = nxt
    return out


@dataclass
class Models:
    raft_model: RaftFlow
    flow_model: FlowCompleter
    inpaint_model: InpaintGeneratorMI355
    provenance: str = ""


_MODEL_CACHE: dict[tuple, Models] = {}
_WARNED: set[str] = set()


def _warn_once(msg: str) -> None:
    if msg not in _WARNED:
        _WARNED.add(msg)
        print(f"[ProPainter-MI355X] WARNING: {msg}", flush=True)


def drop_model_cache() -> None:
    """Release the cached, repacked networks (e.g. from a ComfyUI 'unload models' hook)."""
    _MODEL_CACHE.clear()


def initialize_models(device: torch.device, use_half: str = "enable", seed: int = 0) -> Models:
    """Load + repack the three networks once per process (the reference reloads all checkpoints on every node
    execution, utils/model_utils.py:49-59).  The cache is keyed on the device, the precision mode and the identity of
    the checkpoint files (path, size, mtime) so that swapping files in `weights/` is picked up.

    Checkpoints are read from `weights/` (the reference downloads them there, utils/download_utils.py; there is no
    network access here).  When they are missing the node FAILS, as a user would otherwise get plausible-looking
    garbage; seeded synthetic weights are an explicit opt-in for benchmarks and tests
    (PP_ALLOW_SYNTHETIC_WEIGHTS=1)."""
    if weights.weights_available():
        ident = tuple((f, (weights.WEIGHT_DIR / f).stat().st_size, (weights.WEIGHT_DIR / f).stat().st_mtime_ns)
                      for f in weights.FILES.values())
    elif os.environ.get("PP_ALLOW_SYNTHETIC_WEIGHTS") == "1":
        ident = ("synthetic", seed, os.environ.get("PP_SYNTHETIC_VARIANT", ""))
        _warn_once(f"no checkpoints in {weights.WEIGHT_DIR}: running on SYNTHETIC weights (seed {seed}); the output is "
                   "meaningless as an inpainting result (PP_ALLOW_SYNTHETIC_WEIGHTS=1 is set)")
    else:
        raise FileNotFoundError(
            f"ProPainter checkpoints not found in {weights.WEIGHT_DIR}: place {', '.join(weights.FILES.values())} "
            f"(release {weights.RELEASE_URL}) there. Set PP_ALLOW_SYNTHETIC_WEIGHTS=1 only for benchmarks / tests.")
    key = (str(device), ops.f32_split_enabled(), use_half, ident)
    if key not in _MODEL_CACHE:
        same_dev = [k for k in _MODEL_CACHE if k[0] == key[0]]
        if len(same_dev) >= 2:  # a stale entry pins three networks in HBM: keep at most the previous one PER DEVICE
            _MODEL_CACHE.pop(same_dev[0])
        sds, prov = weights.get_state_dicts(seed)
        _MODEL_CACHE[key] = models_from_state_dicts(sds, device, use_half, prov)
    return _MODEL_CACHE[key]


def models_from_state_dicts(sds: dict, device, fp16: str = "enable", provenance: str = "explicit") -> Models:
    """RAFT is fp32 in both modes (utils/model_utils.py:55-56); fp16 "disable" keeps the other two networks on fp32
    tensors as well (utils/model_utils.py:57-58 only calls .half() for "enable")."""
    dt = torch.float16 if fp16 == "enable" else torch.float32
    return Models(RaftFlow(sds["raft"], device), FlowCompleter(sds["rfc"], device, dt),
                  InpaintGeneratorMI355(sds["gen"], device, dt), provenance)


def compute_flow(raft_model: RaftFlow, frames: torch.Tensor, config: ProPainterConfig) -> torch.Tensor:
    """frames fp32 [T,H,W,3] -> gt flows fp32 [2,T-1,H,W,2] (forward, backward)."""
    return raft_model.bidirectional(frames, config.raft_iter)     # (r06: no torch.stack of the two directions)


def complete_flow(flow_model: FlowCompleter, flows: torch.Tensor, flow_masks_u8: torch.Tensor, subvideo_length: int) -> torch.Tensor:
    """flows fp32 [2,T-1,H,W,2], flow masks u8 [T,H,W] -> completed flows fp32 [2,T-1,H,W,2]."""
    n = flows.shape[1]
    if n <= subvideo_length:
        return flow_model(flows, flow_masks_u8)
    pad = 5
    out = torch.empty_like(flows)
    for f in range(0, n, subvideo_length):
        s, e = max(0, f - pad), min(n, f + subvideo_length + pad)
        ps, pe = f - s, e - min(n, f + subvideo_length)
        sub = flow_model(flows[:, s:e].contiguous(), flow_masks_u8[s:e + 1].contiguous())
        out[:, f:f + (e - s - pe - ps)] = sub[:, ps:e - s - pe]
    return out


def flows_overlapped(models: Models, frames: torch.Tensor, flow_masks_u8: torch.Tensor, config: ProPainterConfig):
    """compute_flow + complete_flow for a clip of SEVERAL sub-videos (T - 1 > subvideo_length) with RAFT of sub-video k + 1
    running under the flow completion of sub-video k (r04; propainter_inference.py:314-341 runs the stages one
    after the other, :115-144 the sub-videos one after the other).  RAFT is per-pair independent (tests/test_raft.py), so it is
    issued range by range on the launch stream -- exactly the pairs the next sub-video's completion reads, 5-flow halo included
    -- and each completion goes to a side stream behind an event: the recurrence's ~1 600 small, latency-bound launches per
    sub-video fill the gaps of RAFT's large kernels instead of having the chip to themselves.  Same sub-video plan, same
    kernels, same arithmetic as compute_flow() + complete_flow(): bit-identical (the long-clip fixtures run through here).
    -> (raw flows, completed flows), each fp32 [2,T-1,H,W,2]."""
    T, H, W, _ = frames.shape
    n, sv, pad = T - 1, config.subvideo_length, 5
    dev = frames.device
    gt = torch.empty(2, n, H, W, 2, device=dev)
    pred = torch.empty_like(gt)
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    gt.record_stream(side)
    pred.record_stream(side)
    done = 0
    for f in range(0, n, sv):
        s, e = max(0, f - pad), min(n, f + sv + pad)
        if e > done:          # the pairs this sub-video still misses
            models.raft_model.bidirectional(frames[done:e + 1], config.raft_iter, out=gt[:, done:e])
            done = e
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            sub = models.flow_model(gt[:, s:e].contiguous(), flow_masks_u8[s:e + 1].contiguous())
            own = min(n, f + sv) - f
            pred[:, f:f + own] = sub[:, f - s:f - s + own]
    main.wait_stream(side)
    return gt, pred


_SIDE_STREAMS: dict = {}


def _side_stream(dev, k: int = 1) -> "torch.cuda.Stream":
    key = (str(dev), k)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
    return _SIDE_STREAMS[key]


def image_propagation(frames: torch.Tensor, masks_u8: torch.Tensor, flows: torch.Tensor, config: ProPainterConfig):
    """-> (prop_frames fp32 [T,H,W,3], updated_masks u8 [T,H,W]); the blend with the input frames is fused
    into the encoder-input packing kernel by the caller."""
    T = frames.shape[0]
    sub = min(100, config.subvideo_length)
    if T <= sub:
        return imgprop.image_propagation(frames, masks_u8, flows)
    pad = 10
    prop = torch.empty_like(frames)
    upd = torch.empty_like(masks_u8)
    for f in range(0, T, sub):
        s, e = max(0, f - pad), min(T, f + sub + pad)
        ps, pe = f - s, e - min(T, f + sub)
        p, m = imgprop.image_propagation(frames[s:e], masks_u8[s:e], flows[:, s:e - 1].contiguous())
        n = e - s - pe - ps
        prop[f:f + n] = p[ps:ps + n]
        upd[f:f + n] = m[ps:ps + n]
    return prop, upd


_SCHED_CACHE: dict = {}


def device_schedule(config: ProPainterConfig):
    """The window schedule as host lists plus ONE device tensor holding, for every window, the global frame ids of
    its local frames and their first-visit flags (the uint8 compose is order dependent, :283-307); cached per
    (clip length, window parameters, device) so the steady state does no H2D copies for scheduling."""
    key = (config.video_length, config.neighbor_length, config.ref_stride, config.subvideo_length, str(config.device))
    hit = _SCHED_CACHE.get(key)
    if hit is None:
        schedule = window_schedule(config)
        seen = [False] * config.video_length
        rows, spans = [], []
        for nb, _ in schedule:
            spans.append((len(rows), len(rows) + len(nb)))
            for i in nb:
                rows.append((i, 0 if seen[i] else 1))
                seen[i] = True
        table = torch.tensor(rows, dtype=torch.int32).t().contiguous().to(config.device)  # [2, sum(l_t)]
        if len(_SCHED_CACHE) > 16:
            _SCHED_CACHE.clear()
        hit = _SCHED_CACHE[key] = (schedule, spans, table)
    return hit


def run_inpainting(models: Models, frames_u8, flow_masks_u8, masks_dilated_u8, config: ProPainterConfig,
                   trace: dict | None = None, to_host: bool = True, frames_f32: torch.Tensor | None = None,
                   sink=None, static_masks=False) -> torch.Tensor:
    """uint8 arrays / tensors in ([T,H,W,3], [T,H,W], [T,H,W]) -> composed uint8 frames [T,H,W,3]
    (CPU tensor, or left in HBM when `to_host` is False).  `frames_f32` = the fp32 [-1,1] frames when the caller
    already produced them on the device (ops.frames_from_image).

    `sink` (optional): an object with `frames_final(comp, lo, hi)`, called as soon as frames [lo, hi) of the composed clip
    can no longer change (no later window has them as local frames) -- the node streams them to the host under the
    remaining windows (nodes._HostImageSink).
    `static_masks`: True when every frame has the same masks (one MASK frame, outpaint borders), or a hashable key of the
    outpaint geometry: the masked-window set of the transformer is then computed once per clip / once per geometry.

    = process_inpainting (:314-341) + feature_propagation (:228-311) of the reference."""
    dev = config.device
    timing = os.environ.get("PP_TIMING") == "1"
    marks = []

    def mark(name):
        if timing:
            torch.cuda.synchronize()
            marks.append((name, time.perf_counter()))
            print(f"[pp] {name} done", flush=True)

    mark("start")
    fr_u8 = torch.as_tensor(frames_u8).to(dev).contiguous()
    fm = torch.as_tensor(flow_masks_u8).to(dev).contiguous()
    md = torch.as_tensor(masks_dilated_u8).to(dev).contiguous()
    T, H, W, _ = fr_u8.shape
    frames = frames_f32 if frames_f32 is not None else ops.frames_from_u8(fr_u8)  # to_tensors(): x/255*2-1 (image_utils.py:191)
    # (from three sub-videos on: with two, one completion of ~a tenth of RAFT's time is all there is to hide -- cfg 5 measured
    #  2 170.8 vs 2 185.7 ms once and 2 521 ms on another box -- and the serial form keeps its lower variance)
    n_sub = -(-(T - 1) // max(1, config.subvideo_length))
    want_overlap = os.environ.get("PP_SUBVIDEO_OVERLAP", "auto")
    if (fr_u8.is_cuda and not torch.cuda.is_current_stream_capturing()
            and ((want_overlap == "auto" and n_sub >= 3) or (want_overlap == "1" and n_sub >= 2))):
        gt, pred = flows_overlapped(models, frames, fm, config)     # several sub-videos: RAFT of k+1 under completion of k
        mark("raft+flow_completion(overlapped)")
    else:
        gt = compute_flow(models.raft_model, frames, config)
        mark("raft")
        pred = complete_flow(models.flow_model, gt, fm, config.subvideo_length)
        mark("flow_completion")
    prop, upd = image_propagation(frames, md, pred, config)
    mark("image_propagation")
    gen = models.inpaint_model
    packed = torch.empty(T, H, W, 8, device=dev, dtype=gen.dt)
    updated = torch.empty(T, H, W, 3, device=dev) if trace is not None else None
    ops.pack_encoder_input(frames, prop, md, upd, packed, updated)
    st = gen.prepare_clip(packed, pred, md, upd, static_masks=static_masks)
    mark("encoder+clip_prep")
    comp = torch.zeros(T, H, W, 3, dtype=torch.uint8, device=dev)
    if trace is not None:
        trace.update(gt_flows=gt, pred_flows=pred, updated_frames=updated, updated_masks=upd, pred_imgs=[])
    schedule, spans, table = device_schedule(config)
    gen.reference_tokens(st, sorted({r for _, refs in schedule for r in refs}))   # one embedding per reference frame of the clip
    # r06 (PP_FEATPROP_PIPE=1): the second half of the large feature-propagation group runs on the generator's side stream next to
    # the transformer of the first windows instead of next to the first half; a window then waits for its own tensor's event
    pipe = (os.environ.get("PP_FEATPROP_PIPE", "0") == "1" and fr_u8.is_cuda and trace is None and ops.CONV_PROFILE is None
            and not torch.cuda.is_current_stream_capturing())
    prop_ready: list = [None] * len(schedule)
    props = gen.propagate_windows(st, [nb for nb, _ in schedule], ready=prop_ready if pipe else None)
    mark("feature_propagation(all windows batched)")
    finals = final_ranges(schedule, T) if sink is not None else None
    # r06: windows are independent until the order-dependent uint8 compose, so `lanes` consecutive windows run next to each other
    # -- the first on the launch stream, the others on side streams (PP_WINDOW_LANES, default 2; 1: one after the other).  The
    # transformer's GEMMs are identical work-groups that compute together and then write together (profiles/r06_gemm_timeline.md:
    # the store drain is as long as the K loop); windows in flight interleave one's write bursts with another's matrix work and
    # fill each other's last rounds.  Same kernels on the same values, the compose stays on the launch stream in window order:
    # bit-identical (tests/test_e2e.py).
    lanes = max(1, int(os.environ.get("PP_WINDOW_LANES", "2")))
    if not (fr_u8.is_cuda and trace is None and not torch.cuda.is_current_stream_capturing() and ops.CONV_PROFILE is None):
        lanes = 1
    lanes = min(lanes, len(schedule))
    main = torch.cuda.current_stream(dev) if lanes > 1 else None
    sides = [_side_stream(dev, k) for k in range(1, lanes)]

    def finish(wi, out):
        a, b = spans[wi]
        ops.compose_u8(out, table[0, a:b], table[1, a:b], md, fr_u8, comp)
        if trace is not None:
            trace["pred_imgs"].append(out[..., :3].float().cpu())
        if finals is not None and finals[wi][1] > finals[wi][0]:
            sink.frames_final(comp, *finals[wi])     # these frames can no longer change: stream them out

    for w0 in range(0, len(schedule), lanes):
        group = list(range(w0, min(len(schedule), w0 + lanes)))
        outs = {}
        for k, wi in enumerate(group[1:], start=1):
            side = sides[k - 1]
            side.wait_stream(main)                   # props / clip state / the previous group's buffers are ready
            with torch.cuda.stream(side):
                nb, refs = schedule[wi]
                if prop_ready[wi] is not None:
                    side.wait_event(prop_ready[wi])
                outs[wi] = gen.forward_window(st, nb, refs, local_prop=props[wi], lane=k)
        nb, refs = schedule[w0]
        if prop_ready[w0] is not None:
            torch.cuda.current_stream(dev).wait_event(prop_ready[w0])
        outs[w0] = gen.forward_window(st, nb, refs, local_prop=props[w0])
        for k, wi in enumerate(group):
            if k > 0:
                main.wait_stream(sides[k - 1])
                outs[wi].record_stream(main)
            finish(wi, outs[wi])
    mark("windows(transformer+decoder+compose)")
    if timing:
        print("[pp] stage ms: " + ", ".join(f"{b[0]} {(b[1] - a[1]) * 1e3:.1f}" for a, b in zip(marks, marks[1:])), flush=True)
    return comp.cpu() if to_host else comp
