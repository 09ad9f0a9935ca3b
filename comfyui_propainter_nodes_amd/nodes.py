"""ComfyUI node API -- the drop-in boundary.  INPUT_TYPES / RETURN_TYPES / RETURN_NAMES / FUNCTION /
CATEGORY, method names, keyword names, error behaviour and the node mappings are those of the
reference's propainter_nodes.py (:21-35, :38-154, :157-310, :313-321); the work behind them runs
on the MI355X through libpropainter_mi355.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import ops
from .image_utils import (
    ImageConfig,
    ImageOutpaintConfig,
    extrapolation,
    handle_output,
    image_to_uint8_frames,
    outpaint_geometry,
    prepare_frames_and_masks,
)
from .pipeline import ProPainterConfig, initialize_models, run_inpainting


def get_torch_device() -> torch.device:
    """comfy.model_management.get_torch_device() inside ComfyUI (propainter_nodes.py:109,247);
    outside ComfyUI (bench, tests) the first visible MI355X.  There is no CPU execution path."""
    try:
        from comfy import model_management  # type: ignore

        return model_management.get_torch_device()
    except ImportError:
        if not torch.cuda.is_available():
            raise RuntimeError("ProPainter (MI355X build) needs a ROCm GPU; none is visible") from None
        return torch.device("cuda", torch.cuda.current_device())


def check_inputs(frames: torch.Tensor, masks: torch.Tensor) -> Exception | None:
    if frames.size(dim=0) <= 1:
        raise Exception(f"""Image length must be greater than 1, but got:
                        Image length: ({frames.size(dim=0)})""")
    if frames.size(dim=0) != masks.size(dim=0) and masks.size(dim=0) != 1:
        raise Exception(f"""Image and Mask must have the same length or Mask have length 1, but got:
                        Image length: {frames.size(dim=0)}
                        Mask length: {masks.size(dim=0)}""")
    if frames.size(dim=1) != masks.size(dim=1) or frames.size(dim=2) != masks.size(dim=2):
        raise Exception(f"""Image and Mask must have the same dimensions, but got:
                        Image: ({frames.size(dim=1)}, {frames.size(dim=2)})
                        Mask: ({masks.size(dim=1)}, {masks.size(dim=2)})""")


_COMMON_INPUTS = {
    "mask_dilates": ("INT", {"default": 5, "min": 0, "max": 100}),
    "flow_mask_dilates": ("INT", {"default": 8, "min": 0, "max": 100}),
    "ref_stride": ("INT", {"default": 10, "min": 1, "max": 100}),
    "neighbor_length": ("INT", {"default": 10, "min": 2, "max": 300}),
    "subvideo_length": ("INT", {"default": 80, "min": 1, "max": 300}),
    "raft_iter": ("INT", {"default": 20, "min": 1, "max": 100}),
    "fp16": (["enable", "disable"],),
}


class _Timer:
    """Stage wall-clock of one node call (PP_TIMING=1, or `LAST_TIMING` read by bench.py)."""

    def __init__(self, device):
        self.device = device
        self.on = os.environ.get("PP_TIMING") == "1" or _Timer.collect
        self.marks = [("start", time.perf_counter())]

    collect = False
    sync = True          # False: host time stamps only (where does the launching thread spend its time?)
    last: dict = {}

    def mark(self, name: str) -> None:
        if self.on:
            if _Timer.sync:
                torch.cuda.synchronize(self.device)
            self.marks.append((name, time.perf_counter()))

    def done(self) -> None:
        if self.on:
            _Timer.last = {b[0]: round((b[1] - a[1]) * 1e3, 2) for a, b in zip(self.marks, self.marks[1:])}
            if os.environ.get("PP_TIMING") == "1":
                print("[pp] node ms: " + ", ".join(f"{k} {v}" for k, v in _Timer.last.items()), flush=True)


def _device_io_ok(image: torch.Tensor, process_size, input_size, mask: torch.Tensor | None = None) -> bool:
    """The device-side byte plumbing covers fp32 IMAGE / MASK inputs that need no resize; everything else (PIL bicubic
    resize, exotic dtypes) takes the host path of image_utils.py, which is the reference's own."""
    if os.environ.get("PP_HOST_IO") == "1":
        return False
    if tuple(process_size) != tuple(input_size) or image.dtype != torch.float32 or image.dim() != 4 or image.shape[-1] != 3:
        return False
    return mask is None or mask.dtype == torch.float32


_STAGE: dict = {}
_STAGE_LOCK = __import__("threading").Lock()


def _pinned_stage(shape: tuple):
    """A page-locked fp32 staging buffer for one node call (page-locking costs more than the copy it speeds up, so one
    buffer is kept between calls and OWNED by one call at a time; a concurrent call gets None and copies the slow way)."""
    with _STAGE_LOCK:
        buf = _STAGE.pop(shape, None)
        _STAGE.clear()
    if buf is None:
        try:
            buf = torch.empty(shape, dtype=torch.float32).pin_memory()
        except RuntimeError:
            return None
    return buf


def _pinned_release(buf: torch.Tensor) -> None:
    with _STAGE_LOCK:
        _STAGE.clear()
        _STAGE[tuple(buf.shape)] = buf


def _expand_masks(m: torch.Tensor, T: int) -> torch.Tensor:
    return m.expand(T, -1, -1).contiguous() if m.shape[0] == 1 and T != 1 else m


def _output(comp_u8: torch.Tensor, fm_u8: torch.Tensor, md_u8: torch.Tensor):
    """handle_output (image_utils.py:276-290): IMAGE fp32 k/255 on the host, the two masks fp32 on the device.
    The uint8 frames cross PCIe (a quarter of the fp32 bytes) and become float32(k) / 255 on the host cores (the same
    IEEE division as the reference's numpy expression); PP_OUTPUT=device (r03) converts on the GPU and copies fp32 instead; the r04 default
    PP_OUTPUT=overlap does that range by range under the remaining windows (_OverlapImageSink) and only falls back to this function
    when no page-locked buffer is available.
    (PP_OUTPUT=host selects the uint8 D2H + host conversion of this function; PP_OUTPUT=stream the same host arithmetic
    streamed under the window loop by _HostImageSink -- slower end to end on the r03 box, see _run.)"""
    if os.environ.get("PP_OUTPUT", "overlap") in ("device", "overlap") and comp_u8.is_cuda:
        # float32(k) / 255 on the GPU (the same IEEE division, bit-identical), then D2H through a page-locked staging buffer
        # kept between calls (a D2H into fresh pageable memory is paced by the driver's bounce buffers and the first touch
        # of 221 MB: 24 ms for the 80-frame clip) and one multi-threaded host copy into the fresh tensor ComfyUI will own
        dev_img = ops.image_from_u8(comp_u8)
        stage = _pinned_stage(tuple(dev_img.shape))
        if stage is not None:
            stage.copy_(dev_img, non_blocking=True)
            torch.cuda.current_stream(comp_u8.device).synchronize()
            images = torch.empty(dev_img.shape, dtype=torch.float32)
            images.copy_(stage)
            _pinned_release(stage)
        else:
            images = dev_img.cpu()
    else:
        images = comp_u8.cpu().to(torch.float32).div_(255.0)
    return images, fm_u8.float().squeeze(), md_u8.float().squeeze()


class _OverlapImageSink:
    """The node's IMAGE leaves the GPU while the remaining windows run, WITHOUT a worker thread (r04): every
    time pipeline.run_inpainting reports a frame range that has received its last blend, the launching thread itself enqueues --
    on a side stream, behind an event of the compute stream -- the uint8 -> fp32 k/255 conversion of that range (the same IEEE
    division as _output) and its asynchronous copy into the page-locked staging buffer kept between calls.  No host
    synchronisation until the single wait in finish(); what is left after the last kernel is the copy of the last range and the
    multi-threaded host copy into the fresh tensor ComfyUI will own.  (The r02 streaming sink converted on a worker thread that
    fought the launching thread for the interpreter: +58 ms on the pipeline, _HostImageSink.)"""

    def __init__(self, T: int, H: int, W: int, device):
        self.device = device
        self.shape = (T, H, W, 3)
        self.stage = _pinned_stage(self.shape)
        self.stream = torch.cuda.Stream(device) if self.stage is not None else None
        self.tmp = []

    @property
    def ok(self) -> bool:
        return self.stage is not None

    def frames_final(self, comp: torch.Tensor, lo: int, hi: int) -> None:
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            img = ops.image_from_u8(comp[lo:hi])
            self.stage[lo:hi].copy_(img, non_blocking=True)
            self.tmp.append(img)           # kept until finish(): freed tensors of a side stream must not be reused under the copy

    def abandon(self) -> None:
        if self.stream is not None:
            self.stream.synchronize()
            _pinned_release(self.stage)

    def finish(self) -> torch.Tensor:
        self.stream.synchronize()
        self.tmp.clear()
        images = torch.empty(self.shape, dtype=torch.float32)
        images.copy_(self.stage)
        _pinned_release(self.stage)
        return images


class _HostImageSink:
    """Streams the composed uint8 frames to the host while the remaining windows run: pipeline.run_inpainting reports every
    frame range that has received its last blend; the range is copied on a side stream into pinned memory and a worker
    thread turns it into the fp32 IMAGE rows (float32(k) / 255, the reference's expression) -- the D2H copy and the host
    conversion of an 80-frame 640x360 clip (16 ms after the last kernel) hide behind the window loop."""

    def __init__(self, T: int, H: int, W: int, device):
        import queue
        import threading

        self.device = device
        key = (T, H, W)
        # page-locking 55 MB costs more than the copy: one buffer is kept between calls and OWNED by one sink at a time
        # (a second node execution overlapping this one, or a worker that outlived an abandoned call, gets its own)
        with _HostImageSink._lock:
            buf = _HostImageSink._pinned.pop(key, None)
            _HostImageSink._pinned.clear()
        self.key = key
        self.pinned = buf if buf is not None else torch.empty(T, H, W, 3, dtype=torch.uint8).pin_memory()
        self.image = torch.empty(T, H, W, 3, dtype=torch.float32)
        self.stream = torch.cuda.Stream(device)
        self.q: "queue.Queue" = queue.Queue()
        self.error: BaseException | None = None
        self.worker = threading.Thread(target=self._convert, daemon=True)
        self.worker.start()

    _pinned: dict = {}
    _lock = __import__("threading").Lock()
    wait_mode = os.environ.get("PP_SINK_WAIT", "poll")

    def _release(self) -> None:
        with _HostImageSink._lock:
            _HostImageSink._pinned.clear()
            _HostImageSink._pinned[self.key] = self.pinned

    def frames_final(self, comp: torch.Tensor, lo: int, hi: int) -> None:
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            self.pinned[lo:hi].copy_(comp[lo:hi], non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.stream)
        self.q.put((lo, hi, copied))

    def _convert(self) -> None:
        try:
            # first touch of the IMAGE's pages (221 MB for 80 frames of 640x360) while the GPU is still in RAFT: page
            # faults, not arithmetic, are most of the host-side output cost
            # (one plain memset on this thread -- ctypes drops the GIL -- not a multi-threaded torch fill that would compete with
            # the thread issuing the kernel launches)
            import ctypes

            ptr, nbytes = self.image.data_ptr(), self.image.numel() * 4
            try:  # ask for transparent huge pages first (512x fewer faults where the kernel honours MADV_HUGEPAGE)
                lo = (ptr + (1 << 21) - 1) & ~((1 << 21) - 1)
                hi = (ptr + nbytes) & ~((1 << 21) - 1)
                if hi > lo:
                    ctypes.CDLL(None, use_errno=True).madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), 14)
            except Exception:  # advisory only
                pass
            ctypes.memset(ptr, 0, nbytes)
            while True:
                item = self.q.get()
                if item is None:
                    return
                lo, hi, copied = item
                # wait for the copy WITHOUT blocking inside the HIP runtime: hipEventSynchronize on this thread contends
                # with the thread that issues the kernel launches (measured r03: the pipeline ran 54 ms longer with it);
                # a non-blocking query + a short sleep (which also releases the GIL) does not
                if _HostImageSink.wait_mode == "sync":
                    copied.synchronize()
                else:
                    while not copied.query():
                        time.sleep(0.0004)
                dst = self.image[lo:hi]
                dst.copy_(self.pinned[lo:hi])              # uint8 -> float32(k), exact
                dst.div_(255.0)                            # / 255: the reference's IEEE division
        except BaseException as e:  # surfaced by finish()
            self.error = e

    def abandon(self) -> None:
        self.q.put(None)
        self.stream.synchronize()          # no copy into the pinned buffer may still be in flight
        self.worker.join(timeout=5.0)
        if not self.worker.is_alive():     # (a worker that is still converting keeps the buffer: it is not handed on)
            self._release()

    def finish(self) -> torch.Tensor:
        self.q.put(None)
        self.worker.join()
        self._release()
        if self.error is not None:
            raise self.error
        return self.image


def _shard_devices(config: ProPainterConfig, device: torch.device) -> list:
    """PP_GPUS=N (or "all"): the devices ONE node call spreads a long clip over (r04: the drop-in itself shards; the reference
    picks one device per call, propainter_nodes.py:109).  Sharding follows the reference's own sub-video chunks, so it applies
    when the clip has at least two of them (video_length > subvideo_length <= 100: the reference's local-reference mode);
    shorter clips stay on one GPU.  PP_GPUS_VIRTUAL=1 lets ranks share a device (functional tests on a 1-GPU box)."""
    want = os.environ.get("PP_GPUS", "1").strip().lower()
    if want in ("", "0", "1") or TRACE is not None or device.type != "cuda":
        return [device]
    have = torch.cuda.device_count()
    if want == "all":
        n = have
    else:
        try:
            n = int(want)
        except ValueError:
            n = 0
        if n < 1:      # (junk / negative values must not raise inside the node call or slice from the end)
            import warnings
            warnings.warn(f"PP_GPUS={want!r} is not a positive integer or 'all': running on one device")
            return [device]
    sv = config.subvideo_length
    nchunks = (config.video_length + sv - 1) // sv
    if sv > 100 or nchunks < 2:
        return [device]
    n = min(n, nchunks)
    if os.environ.get("PP_GPUS_VIRTUAL") == "1":
        return [device] * n
    first = device.index if device.index is not None else torch.cuda.current_device()
    order = [first] + [i for i in range(have) if i != first]
    return [torch.device("cuda", i) for i in order[:min(n, have)]]


def _run_sharded(devices: list, config: ProPainterConfig, load_slab, fm, md, tm: _Timer):
    """One clip over several GPUs from inside the node call: distributed.run_multi_device (one thread per device, sub-video
    shards, seam-only peer copies, the composed frames gathered on the first device)."""
    from . import distributed as D
    from .pipeline import models_from_state_dicts
    from . import weights as W

    virtual = len({str(d) for d in devices}) < len(devices)
    backends = []
    for d in devices:
        with torch.cuda.device(d):      # weight repacking launches kernels: the HIP current device must be the model's
            if virtual:     # ranks sharing a device must not share a model object (its captured hipGraphs own static buffers)
                sds, prov = W.get_state_dicts(0)
                m = models_from_state_dicts(sds, d, config.fp16, prov)
            else:
                m = initialize_models(d, config.fp16)
        backends.append(D.GpuBackend(m, config))
    comp = D.run_multi_device(backends, config, load_slab, fm, md, devices, gather_root=0)
    tm.mark(f"pipeline({len(devices)} ranks)")
    out = _output(comp, fm, md)
    tm.mark("output(u8->float, D2H)")
    tm.done()
    return out



TRACE: dict | None = None  # debugging / test aid: when a dict, the next node call leaves its stage tensors in it


def _run(models, config, fr_u8, fr_f32, fm, md, tm: _Timer, static_masks=False):
    if TRACE is not None:
        TRACE.update(frames_u8=fr_u8, flow_masks=fm, masks_dilated=md)
    # (r03: measured on the MI355X box, node call of the 80-frame clip: PP_OUTPUT=device 452 ms, host 466 ms, stream 496 ms --
    # with the streaming sink the pipeline itself ran 58 ms longer, whichever way its worker waits for the copies;
    # tools/node_gap.py.  The default is therefore the GPU conversion + one D2H; the sink stays selectable.)
    mode = os.environ.get("PP_OUTPUT", "overlap") if (fr_u8.is_cuda and TRACE is None) else "device"
    sink = None
    if mode == "stream":
        sink = _HostImageSink(*fr_u8.shape[:3], fr_u8.device)
    elif mode == "overlap":
        sink = _OverlapImageSink(*fr_u8.shape[:3], fr_u8.device)
        if not sink.ok:                # no page-locked buffer to be had: the plain path
            sink = None
    try:
        comp = run_inpainting(models, fr_u8, fm, md, config, trace=TRACE, to_host=False, frames_f32=fr_f32, sink=sink,
                              static_masks=static_masks)
    except BaseException:
        if sink is not None:
            sink.abandon()             # release the worker thread: a failed call must not leave it parked on the queue
        raise
    tm.mark("pipeline")
    if sink is not None:
        out = (sink.finish(), fm.float().squeeze(), md.float().squeeze())
    else:
        out = _output(comp, fm, md)
    tm.mark("output(u8->float, D2H)")
    tm.done()
    return out


class ProPainterInpaint:
    """ComfyUI Node for performing inpainting on video frames using ProPainter.

    `fp16`: "enable" = f16 storage of flow completion and the generator (the reference's `.half()`), RAFT on f32 tensors;
    "disable" = fp32 STORAGE of all three networks.  By default the matrix products run on the f16 matrix pipe in both modes: fp32
    tensors multiply as two-term f16 splits (PP_F32X2: 22 significant bits per operand, fp32 accumulation) and the attention
    core rounds q, k, v and the probabilities to f16 -- "disable" is then 61-78 dB from the reference's fp32 CPU run.  The
    environment variable PP_F32_GEMM=exact (read when the models are built) puts every product of "disable" -- convolutions,
    Linears and, since r06, the attention core -- on the f32 MFMA instructions with fp32 operands: the reference's fp32
    arithmetic up to summation order, at several times the run time."""

    def __init__(self):
        pass

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "image": ("IMAGE",),
                "mask": ("MASK",),
                "width": ("INT", {"default": 640, "min": 0, "max": 2560}),
                "height": ("INT", {"default": 360, "min": 0, "max": 2560}),
                **_COMMON_INPUTS,
            },
        }

    RETURN_TYPES = ("IMAGE", "MASK", "MASK")
    RETURN_NAMES = ("IMAGE", "FLOW_MASK", "MASK_DILATE")
    FUNCTION = "propainter_inpainting"
    CATEGORY = "ProPainter"

    def propainter_inpainting(self, image: torch.Tensor, mask: torch.Tensor, width: int, height: int, mask_dilates: int,
                              flow_mask_dilates: int, ref_stride: int, neighbor_length: int, subvideo_length: int,
                              raft_iter: int, fp16: str) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        check_inputs(image, mask)
        device = get_torch_device()
        tm = _Timer(device)
        video_length = image.size(dim=0)
        input_size = (image.size(dim=2), image.size(dim=1))
        image_config = ImageConfig(width, height, mask_dilates, flow_mask_dilates, input_size, video_length)
        config = ProPainterConfig(ref_stride, neighbor_length, subvideo_length, raft_iter, fp16, video_length, device,
                                  image_config.process_size)
        devices = _shard_devices(config, device)
        if len(devices) > 1:
            if _device_io_ok(image, image_config.process_size, input_size, mask):
                m = mask.detach().to(device).contiguous()
                fm = _expand_masks(ops.mask_dilate(m, flow_mask_dilates), video_length)
                md = _expand_masks(ops.mask_dilate(m, mask_dilates), video_length)

                def load_slab(rank, lo, hi, dev):   # each rank uploads and converts only the frames it needs
                    return ops.frames_from_image(image[lo:hi].detach().to(dev).contiguous(), want_f32=False)[0]
            else:
                frames_u8, flow_masks, masks_dilated = prepare_frames_and_masks(image_to_uint8_frames(image), mask, image_config)
                fm, md = torch.from_numpy(flow_masks).to(device), torch.from_numpy(masks_dilated).to(device)

                def load_slab(rank, lo, hi, dev):
                    return torch.from_numpy(frames_u8[lo:hi]).to(dev)
            tm.mark("input(masks)")
            print(f"\nProcessing  {config.video_length} frames on {len(devices)} GPUs...")
            return _run_sharded(devices, config, load_slab, fm, md, tm)
        models = initialize_models(device, config.fp16)
        if _device_io_ok(image, image_config.process_size, input_size, mask):
            # one H2D of the fp32 IMAGE / MASK; uint8 conversion, [-1,1] scaling and mask dilation on the device
            fr_u8, fr_f32 = ops.frames_from_image(image.detach().to(device).contiguous())
            m = mask.detach().to(device).contiguous()
            fm = _expand_masks(ops.mask_dilate(m, flow_mask_dilates), video_length)
            md = _expand_masks(ops.mask_dilate(m, mask_dilates), video_length)
        else:
            frames_u8, flow_masks, masks_dilated = prepare_frames_and_masks(image_to_uint8_frames(image), mask, image_config)
            fr_u8, fr_f32 = torch.from_numpy(frames_u8).to(device), None
            fm, md = torch.from_numpy(flow_masks).to(device), torch.from_numpy(masks_dilated).to(device)
        tm.mark("input(H2D, u8, masks)")
        print(f"\nProcessing  {config.video_length} frames...")
        return _run(models, config, fr_u8, fr_f32, fm, md, tm, static_masks=mask.shape[0] == 1)


class ProPainterOutpaint:
    """ComfyUI Node for performing outpainting on video frames using ProPainter."""

    def __init__(self):
        pass

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "image": ("IMAGE",),
                "width": ("INT", {"default": 640, "min": 0, "max": 2560}),
                "height": ("INT", {"default": 360, "min": 0, "max": 2560}),
                "width_scale": ("FLOAT", {"default": 1.2, "min": 0.0, "max": 10.0, "step": 0.01}),
                "height_scale": ("FLOAT", {"default": 1.0, "min": 0.0, "max": 10.0, "step": 0.01}),
                **_COMMON_INPUTS,
            },
        }

    RETURN_TYPES = ("IMAGE", "MASK", "INT", "INT")
    RETURN_NAMES = ("IMAGE", "OUTPAINT_MASK", "output_width", "output_height")
    FUNCTION = "propainter_outpainting"
    CATEGORY = "ProPainter"

    def propainter_outpainting(self, image: torch.Tensor, width: int, height: int, width_scale: float, height_scale: float,
                               mask_dilates: int, flow_mask_dilates: int, ref_stride: int, neighbor_length: int,
                               subvideo_length: int, raft_iter: int, fp16: str) -> tuple[torch.Tensor, torch.Tensor, int, int]:
        device = get_torch_device()
        tm = _Timer(device)
        video_length = image.size(dim=0)
        input_size = (image.size(dim=2), image.size(dim=1))
        image_config = ImageOutpaintConfig(width, height, mask_dilates, flow_mask_dilates, input_size, video_length,
                                           width_scale, height_scale)
        config = ProPainterConfig(ref_stride, neighbor_length, subvideo_length, raft_iter, fp16, video_length, device,
                                  image_config.outpaint_size)
        devices = _shard_devices(config, device)
        if len(devices) > 1:
            if _device_io_ok(image, image_config.process_size, input_size):
                (pw, ph), (hs, ws), flow_mask, mask = outpaint_geometry(image_config)
                fm = _expand_masks(torch.from_numpy(flow_mask[None]).to(device), video_length)
                md = _expand_masks(torch.from_numpy(mask[None]).to(device), video_length)

                def load_slab(rank, lo, hi, dev):
                    return ops.frames_from_image(image[lo:hi].detach().to(dev).contiguous(), (ph, pw), (hs, ws), want_f32=False)[0]
            else:
                frames_u8, flow_masks, masks_dilated = extrapolation(image_to_uint8_frames(image), image_config)
                fm, md = torch.from_numpy(flow_masks).to(device), torch.from_numpy(masks_dilated).to(device)

                def load_slab(rank, lo, hi, dev):
                    return torch.from_numpy(frames_u8[lo:hi]).to(dev)
            tm.mark("input(masks)")
            print(f"\nProcessing  {config.video_length} frames on {len(devices)} GPUs...")
            output_frames, output_masks, _ = _run_sharded(devices, config, load_slab, fm, md, tm)
            output_width, output_height = config.process_size
            return output_frames, output_masks, output_width, output_height
        models = initialize_models(device, config.fp16)
        if _device_io_ok(image, image_config.process_size, input_size):
            # outpaint fast path: the canvas is filled on the device, the two border masks are one static plane each
            (pw, ph), (hs, ws), flow_mask, mask = outpaint_geometry(image_config)
            fr_u8, fr_f32 = ops.frames_from_image(image.detach().to(device).contiguous(), (ph, pw), (hs, ws))
            fm = _expand_masks(torch.from_numpy(flow_mask[None]).to(device), video_length)
            md = _expand_masks(torch.from_numpy(mask[None]).to(device), video_length)
        else:
            frames_u8, flow_masks, masks_dilated = extrapolation(image_to_uint8_frames(image), image_config)
            fr_u8, fr_f32 = torch.from_numpy(frames_u8).to(device), None
            fm, md = torch.from_numpy(flow_masks).to(device), torch.from_numpy(masks_dilated).to(device)
        tm.mark("input(H2D, u8, masks)")
        print(f"\nProcessing  {config.video_length} frames...")
        # the border masks are a function of the canvas geometry alone: the masked-window set of the transformer is cached
        # per geometry across node executions (image_utils.py:200-252 recomputes the planes per call)
        # (canvas size, resized frame size, input size, scales: the resized frame size decides where the frame sits on the canvas --
        # two calls can share every other entry and still have different border planes; mask_dilates does not enter them)
        geometry = ("outpaint", tuple(config.process_size), tuple(image_config.process_size), tuple(input_size),
                    float(width_scale), float(height_scale))
        output_frames, output_masks, _ = _run(models, config, fr_u8, fr_f32, fm, md, tm, static_masks=geometry)
        output_width, output_height = config.process_size
        return output_frames, output_masks, output_width, output_height


NODE_CLASS_MAPPINGS = {
    "ProPainterInpaint": ProPainterInpaint,
    "ProPainterOutpaint": ProPainterOutpaint,
}

NODE_DISPLAY_NAME_MAPPINGS = {
    "ProPainterInpaint": "ProPainter Inpainting",
    "ProPainterOutpaint": "ProPainter Outpainting",
}
