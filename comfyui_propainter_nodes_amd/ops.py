"""Thin host-side wrappers over the C ABI: torch tensors in, kernel launches out.

PyTorch is used only for device memory and streams.  Every activation handled here is
channels-last `[N, H, W, C]`; a view whose last-dim slice is narrower than its pitch
(`t[..., a:b]`) is how channel concatenation is expressed without copies.
"""
from __future__ import annotations

import ctypes
import functools
import threading
from collections import OrderedDict
from dataclasses import dataclass

import math
import os

import torch

from . import lib as _lib

PP_F32, PP_F16 = _lib.CONSTS["PP_F32"], _lib.CONSTS["PP_F16"]
ACT = {
    None: _lib.CONSTS["PP_ACT_NONE"],
    "none": _lib.CONSTS["PP_ACT_NONE"],
    "relu": _lib.CONSTS["PP_ACT_RELU"],
    "leaky": _lib.CONSTS["PP_ACT_LEAKY"],
    "sigmoid": _lib.CONSTS["PP_ACT_SIGMOID"],
    "tanh": _lib.CONSTS["PP_ACT_TANH"],
    "gelu": _lib.CONSTS["PP_ACT_GELU"],
}
EPI = {
    None: _lib.CONSTS["PP_EPI_NONE"],
    "none": _lib.CONSTS["PP_EPI_NONE"],
    "mul": _lib.CONSTS["PP_EPI_MUL_AUX1"],
    "add": _lib.CONSTS["PP_EPI_ADD_AUX1"],
    "add_relu": _lib.CONSTS["PP_EPI_ADD_AUX1_RELU"],
    "gru": _lib.CONSTS["PP_EPI_GRU"],
}
PAD = {"zeros": _lib.CONSTS["PP_PAD_ZEROS"], "replicate": _lib.CONSTS["PP_PAD_REPLICATE"]}


class ConvProfile:
    """Per-launch HIP-event timing of pp_conv2d on the launch stream (bench.py's roofline leg)."""

    def __init__(self, detailed: bool = False):
        self.records = []
        self.detailed = detailed

    def launch(self, key: str, flops: float, fn, nbytes: float = 0.0) -> None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.records.append((key, flops, e0, e1, nbytes))

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out: dict = {}
        for key, flops, e0, e1, nbytes in self.records:
            d = out.setdefault(key, {"ms": 0.0, "flops": 0.0, "n": 0, "bytes": 0.0})
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
            d["n"] += 1
        return out


CONV_PROFILE: ConvProfile | None = None


_INT_CACHE: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()   # bounded LRU: lists no captured graph refers to
_INT_PINNED: dict = {}                                            # lists handed out while a hipGraph was being captured
_INT_EVENTS: dict = {}                                            # upload events of lists that may not have landed yet
_INT_CACHE_MAX = 4096
_INT_LOCK = threading.Lock()
PIN_DEVICE_INTS = 0   # > 0 while graphs.GraphCache warms up / captures a sweep (its replays read these addresses)


def device_ints(values, device, dtype: torch.dtype = torch.int64) -> torch.Tensor:
    """A small integer index tensor on the device, cached by value (window / reference frame ids recur every clip,
    so the steady state issues no pageable H2D copies for them).  Lists requested while a hipGraph is being captured
    (graphs.py bakes their addresses into the graph) are pinned for the life of the process -- one set per captured
    problem shape; everything else lives in an LRU of `_INT_CACHE_MAX` entries, so a long-lived ComfyUI process does not
    grow without bound."""
    key = (tuple(values), str(device), dtype)
    with _INT_LOCK:      # (the rank threads of distributed.run_multi_device share the cache)
        t = _INT_PINNED.get(key)
        if t is None:
            t = _INT_CACHE.get(key)
            if t is not None:
                _INT_CACHE.move_to_end(key)
        ev = _INT_EVENTS.get(key) if t is not None else None
    if t is None:
        # upload OUTSIDE the lock (a host synchronize under the process-wide lock stalled every rank thread behind one
        # rank's stream): the upload is ordered on the creating thread's stream; any OTHER stream that picks the tensor up later
        # waits for the event recorded behind it (below) instead of the host waiting for the stream
        t = torch.tensor(list(values), dtype=dtype, device=device)
        ev = None
        if t.is_cuda and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(t.device))
        with _INT_LOCK:
            t0 = _INT_PINNED.get(key) or _INT_CACHE.get(key)
            if t0 is not None:          # another thread was faster: use its tensor (and its event)
                t, ev = t0, _INT_EVENTS.get(key)
            else:
                if ev is not None:
                    _INT_EVENTS[key] = ev
                if PIN_DEVICE_INTS > 0:
                    _INT_PINNED[key] = t
                else:
                    _INT_CACHE[key] = t
                    while len(_INT_CACHE) > _INT_CACHE_MAX:
                        k0, _ = _INT_CACHE.popitem(last=False)
                        _INT_EVENTS.pop(k0, None)
    elif PIN_DEVICE_INTS > 0:
        # (pin the tensor THIS call hands out, whatever another thread did to the LRU between the lookup above and here
        #  -- an evicted-and-unpinned tensor's address would be baked into the graph being captured)
        with _INT_LOCK:
            _INT_PINNED[key] = t
            _INT_CACHE.pop(key, None)
    if ev is not None and t.is_cuda and not torch.cuda.is_current_stream_capturing():
        if ev.query():
            with _INT_LOCK:
                _INT_EVENTS.pop(key, None)      # landed: later users need no wait
        else:
            torch.cuda.current_stream(t.device).wait_event(ev)   # (a no-op on the stream that uploaded it)
    return t


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return PP_F32
    if dt == torch.float16:
        return PP_F16
    if dt == torch.uint8:
        return _lib.CONSTS["PP_U8"]
    if dt == torch.int32:
        return _lib.CONSTS["PP_I32"]
    raise TypeError(f"unsupported dtype {dt}")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle(t: torch.Tensor) -> int:
    """The HIP stream torch would launch on for this tensor's device (the thread's current stream of that device)."""
    if t.is_cuda:
        if _RAW_STREAM is not None:      # (r05: 0.3 us; torch.cuda.current_stream() builds a Stream object: 4.5 us per launch)
            return _RAW_STREAM(t.device.index)
        return torch.cuda.current_stream(t.device).cuda_stream
    return 0


def check_device(*tensors: torch.Tensor) -> None:
    """The gfx950 library takes device pointers only; the emulator (tests) takes host pointers."""
    L = _lib.current()
    for t in tensors:
        if t is None:
            continue
        if L.is_emulator:
            if t.is_cuda:
                raise RuntimeError("emulator library loaded but a CUDA tensor was passed")
        elif not t.is_cuda:
            raise RuntimeError(
                "libpropainter_mi355 needs tensors on the MI355X (got a CPU tensor); there is no CPU fallback"
            )
        elif t.device.index != torch.cuda.current_device():
            # hipLaunchKernel resolves the kernel for the CURRENT device: launching onto a stream of another device is undefined.
            # A process that drives several GPUs makes the tensor's device current first (distributed.run_multi_device: one thread
            # per device; `with torch.cuda.device(d)` elsewhere).
            raise RuntimeError(f"tensor on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}: "
                               "make the tensor's device current before calling libpropainter_mi355")


def nhwc_view(t: torch.Tensor) -> tuple[int, int, int, int, int]:
    """Validate a channels-last view; return (N, H, W, C, ldc)."""
    if t.dim() != 4:
        raise ValueError(f"expected [N,H,W,C], got {tuple(t.shape)}")
    n, h, w, c = t.shape
    ldc = t.stride(2)
    if c > 1 and t.stride(3) != 1:
        raise ValueError("channel dim must be unit-stride")
    if ldc < c:
        raise ValueError(f"bad channel pitch {ldc} for {c} channels")
    if h > 1 and t.stride(1) != w * ldc:
        raise ValueError(f"rows must be dense: stride {t.stride()} shape {tuple(t.shape)}")
    if n > 1 and t.stride(0) != h * w * ldc:
        raise ValueError(f"images must be dense: stride {t.stride()} shape {tuple(t.shape)}")
    return n, h, w, c, ldc


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def pack_conv_weight(w: torch.Tensor, seg_channels: list[int], dtype: torch.dtype,
                     seg_valid: list[int] | None = None) -> torch.Tensor:
    """Repack a torch conv weight `[Cout, Cin, kh, kw]` to `[Cout][tap][seg][c pad 32]`.

    `seg_channels[s]` is the channel count the kernel will read from segment s (a multiple of
    the 16-byte piece); `seg_valid[s]` (<= seg_channels[s]) is how many of those carry real
    weights, the rest are zero (used when a source tensor is channel-padded).
    """
    cout, cin, kh, kw = w.shape
    seg_valid = seg_valid or seg_channels
    if sum(seg_valid) != cin:
        raise ValueError(f"segments {seg_valid} do not add up to Cin={cin}")
    kp = sum(pad32(c) for c in seg_channels)
    out = torch.zeros(cout, kh * kw, kp, dtype=torch.float32)
    wt = w.detach().float().permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    src = 0
    dst = 0
    for c, v in zip(seg_channels, seg_valid):
        out[:, :, dst:dst + v] = wt[:, :, src:src + v]
        src += v
        dst += pad32(c)
    return out.reshape(cout, kh * kw * kp).to(dtype).contiguous()


def f32_split_enabled() -> bool:
    """PP_F32_GEMM=exact keeps f32 convolutions on the f32 MFMA instructions; default: PP_F32X2 (split) mode."""
    return os.environ.get("PP_F32_GEMM", "split").lower() != "exact"


def split_pack_weight(packed: torch.Tensor) -> tuple[torch.Tensor, float]:
    """PP_F32X2 weight format (r05): every 32-channel chunk [w0..w31] of an f32 packing becomes 32 f16 `h = f16(S w)` followed by
    32 f16 `l = f16(S w - h)` -- the low term UNSCALED relative to the high one, so that the three MFMA products of a multiply-add
    share one accumulator -- with S a power of two per layer that puts max|w| S into [8192, 16384): the low term of every weight
    that matters is then a normal f16 number (unscaled, the low terms of weights below 0.06 would be subnormal: 2-3 bits lost).
    Returns (f32-typed bit container of the same shape, acc_scale = 1 / S: exact, applied to the accumulators by the kernel)."""
    cout, kp = packed.shape
    w = packed.float().reshape(cout, kp // 32, 32)
    wmax = float(w.abs().max())
    k = 0 if not (wmax > 0.0 and math.isfinite(wmax)) else max(-24, min(24, 13 - math.frexp(wmax)[1] + 1))
    w = w * (2.0 ** k)
    h = w.clamp(-65504.0, 65504.0).half()
    l = (w - h.float()).clamp(-65504.0, 65504.0).half()
    return torch.cat([h, l], dim=2).contiguous().view(torch.float32).reshape(cout, kp), 2.0 ** -k


@dataclass
class ConvSpec:
    """Geometry + packed parameters of one convolution / linear layer."""

    weight: torch.Tensor          # packed [Cout_total, Kp]
    bias: torch.Tensor | None     # fp32 [Cout_total]
    seg_channels: list[int]       # per-group channels read from each segment
    cout: int                     # per group
    kh: int = 1
    kw: int = 1
    sh: int = 1
    sw: int = 1
    ph: int = 0
    pw: int = 0
    dh: int = 1
    dw: int = 1
    groups: int = 1
    pad_mode: str = "zeros"
    cin_valid: int = 0            # real (unpadded) input channels per group, for FLOP accounting
    split: bool = False           # f32 tensors on the f16 matrix pipe (PP_F32X2 weight packing)
    acc_scale: float = 0.0        # PP_F32X2: 1 / (the power-of-two scale inside the packed weights); 0 = 1
    many_images: bool = False     # ABI v12: this layer always runs on a batch of many images (never the small-image split-K kernel)
    weight_f32: torch.Tensor | None = None  # Cout <= 4 only: fp32 [tap*chunk][Cout padded to 2 / 4][32] table (conv_direct.hip)
    geometry_key: tuple | None = None       # (conv2d's parameter-block cache: the fields above that a block depends on, built once)

    def to(self, device) -> "ConvSpec":
        self.weight = self.weight.to(device)
        if self.weight_f32 is not None:
            self.weight_f32 = self.weight_f32.to(device)
        if self.bias is not None:
            self.bias = self.bias.to(device)
        return self

    def out_hw(self, h: int, w: int) -> tuple[int, int]:
        ho = (h + 2 * self.ph - self.dh * (self.kh - 1) - 1) // self.sh + 1
        wo = (w + 2 * self.pw - self.dw * (self.kw - 1) - 1) // self.sw + 1
        return ho, wo


def make_conv_spec(w: torch.Tensor, b: torch.Tensor | None, dtype: torch.dtype, *, stride=1, padding=0,
                   dilation=1, groups=1, seg_channels=None, seg_valid=None, pad_mode="zeros",
                   split: bool = False, many_images: bool = False) -> ConvSpec:
    def pair(v):
        return (v, v) if isinstance(v, int) else tuple(v)

    cout, cin_g, kh, kw = w.shape
    sh, sw = pair(stride)
    ph, pw = pair(padding)
    dh, dw = pair(dilation)
    seg_channels = seg_channels or [cin_g]
    packed = pack_conv_weight(w, seg_channels, dtype, seg_valid)
    table = None
    if cout <= 4 and groups == 1 and len(seg_channels) == 1 and os.environ.get("PP_CONV_DIRECT_TABLE") != "0":
        # the values the MFMA path would multiply with (f16 weights stay f16-rounded), as fp32, chunk-major
        t = packed.float().reshape(cout, packed.shape[1] // 32, 32).permute(1, 0, 2)
        table = torch.zeros(t.shape[0], 2 if cout <= 2 else 4, 32)
        table[:, :cout] = t
        table = table.contiguous()
    if split:
        if dtype != torch.float32:
            raise TypeError("split packing applies to f32 convolutions only")
        packed, acc_scale = split_pack_weight(packed)
    else:
        acc_scale = 0.0
    bias = b.detach().float().contiguous() if b is not None else None
    return ConvSpec(packed, bias, list(seg_channels), cout // groups, kh, kw, sh, sw, ph, pw, dh, dw, groups, pad_mode,
                    sum(seg_valid) if seg_valid else cin_g, split, acc_scale, many_images, table)


def _conv2d_params(spec: ConvSpec, inputs: list[torch.Tensor], out: torch.Tensor, *, act=None, act_param=0.0,
                   act2=None, act_split=0, out_scale=0.0, epi=None, aux1=None, aux2=None, pre_add=None, epi_from=0,
                   in_zoff: list[int] | None = None, out_zoff: int | None = None, virtual_input: bool = False):
    """Fill a pp_conv2d_params block.  `virtual_input`: the inputs only describe a shape (meta tensors) -- the block goes
    to pp_deform_conv, which produces the input columns on the fly."""
    check_device(*([] if virtual_input else inputs), out, aux1, aux2, pre_add, spec.weight)
    P = _lib.STRUCTS["pp_conv2d_params"]()
    x0 = inputs[0]
    n, h, w, _, _ = nhwc_view(x0)
    P.dtype = _lib.CONSTS["PP_F32X2"] if spec.split else dtype_code(x0.dtype)
    P.out_dtype = dtype_code(out.dtype)
    if spec.weight.dtype != x0.dtype:
        raise TypeError("weight dtype must match the input dtype")
    P.nseg = len(inputs)
    P.pad_mode = PAD[spec.pad_mode]
    if len(inputs) != len(spec.seg_channels):
        raise ValueError("number of input segments does not match the packed weight")
    g = spec.groups
    for s, t in enumerate(inputs):
        tn, th, tw, tc, ldc = nhwc_view(t)
        if (tn, th, tw) != (n, h, w):
            raise ValueError("all segments must share N,H,W")
        if t.dtype != x0.dtype:
            raise TypeError("all segments must share a dtype")
        cs = spec.seg_channels[s]
        if tc != cs * g:
            raise ValueError(f"segment {s}: expected {cs * g} channels, got {tc}")
        P.in_ptr[s] = None if virtual_input else t.data_ptr()
        P.in_C[s] = cs
        P.in_ldc[s] = ldc
        P.in_zoff[s] = in_zoff[s] if in_zoff is not None else (cs if g > 1 else 0)
    ho, wo = spec.out_hw(h, w)
    on, oh, ow, oc, oldc = nhwc_view(out)
    if (on, oh, ow) != (n, ho, wo) or oc < spec.cout * g:
        raise ValueError(f"bad output view {tuple(out.shape)} for conv result {(n, ho, wo, spec.cout * g)}")
    P.N, P.H, P.W, P.Ho, P.Wo = n, h, w, ho, wo
    P.kh, P.kw, P.sh, P.sw = spec.kh, spec.kw, spec.sh, spec.sw
    P.ph, P.pw, P.dh, P.dw = spec.ph, spec.pw, spec.dh, spec.dw
    P.weight = spec.weight.data_ptr()
    P.w_zoff = spec.cout * spec.weight.shape[1] if g > 1 else 0
    P.bias = spec.bias.data_ptr() if spec.bias is not None else None
    P.bias_zoff = spec.cout if g > 1 else 0
    P.Cout = spec.cout
    P.Z = g
    P.out = out.data_ptr()
    P.out_ldc = oldc
    P.out_zoff = out_zoff if out_zoff is not None else (spec.cout if g > 1 else 0)
    P.act = ACT[act]
    P.act2 = ACT[act2]
    P.act_split = act_split
    P.acc_scale = spec.acc_scale
    P.many_images = int(spec.many_images)
    P.act_param = act_param
    P.out_scale = out_scale
    P.epi = EPI[epi]
    P.epi_from = epi_from
    if aux1 is not None:
        if aux1.dtype != out.dtype:
            raise TypeError("aux1 dtype must match out dtype")
        P.aux1 = aux1.data_ptr()
        P.aux1_ldc = nhwc_view(aux1)[4]
        P.aux1_zoff = spec.cout if g > 1 else 0
    if aux2 is not None:
        if aux2.dtype != out.dtype:
            raise TypeError("aux2 dtype must match out dtype")
        P.aux2 = aux2.data_ptr()
        P.aux2_ldc = nhwc_view(aux2)[4]
        P.aux2_zoff = spec.cout if g > 1 else 0
    if spec.weight_f32 is not None:
        P.weight_f32 = spec.weight_f32.data_ptr()
    if pre_add is not None:
        if pre_add.dtype != out.dtype:
            raise TypeError("pre_add dtype must match out dtype")
        P.pre_add = pre_add.data_ptr()
        P.pre_add_ldc = nhwc_view(pre_add)[4]
    return P


_PARAMS: "OrderedDict[tuple, object]" = OrderedDict()   # filled pp_conv2d_params blocks by launch identity (bounded FIFO)
_PARAMS_MAX = 16384
_PARAMS_STATS = [0, 0]   # [hits, misses]
_PARAMS_LOCK = threading.Lock()


def _tkey(t):
    return None if t is None else (t.data_ptr(), t.shape, t.stride(), t.dtype)


def conv2d(spec: ConvSpec, inputs: list[torch.Tensor], out: torch.Tensor, *, aux1=None, aux2=None, pre_add=None,
           **kw) -> torch.Tensor:
    """Launch pp_conv2d.  `inputs` are channels-last views (the K segments), `out` a
    channels-last view `[N, Ho, Wo, >=Cout*groups]` that receives the result.  Keywords: _conv2d_params."""
    L = _lib.current()
    if CONV_PROFILE is None:
        # r05: the filled parameter block of a launch is kept by (layer, tensor identities, keywords).  A clip repeats the same
        # launches on the same buffers (the caching allocator hands the same addresses back), so in the steady state a convolution
        # costs one key and one dictionary lookup on the host instead of ~45 ctypes field stores and the shape checks behind them:
        # the host side of a clip (bench.py: host_enqueue_ms) matters once N rank threads share the interpreter (PP_GPUS=N).
        # (the key names everything _conv2d_params reads: the layer's geometry and ITS buffers -- not the spec object's id: a test that
        #  builds one layer after another gets the same id, the same weight address and the same tensor addresses back for a
        #  different geometry; tests/test_conv.py caught exactly that on the MI355X)
        geo = spec.geometry_key
        if geo is None:
            geo = spec.geometry_key = (spec.cout, spec.kh, spec.kw, spec.sh, spec.sw, spec.ph, spec.pw, spec.dh, spec.dw, spec.groups,
                                       spec.pad_mode, tuple(spec.seg_channels), spec.split, spec.acc_scale, spec.many_images,
                                       tuple(spec.weight.shape))
        key = (geo, spec.weight.data_ptr(), None if spec.bias is None else spec.bias.data_ptr(),
               None if spec.weight_f32 is None else spec.weight_f32.data_ptr(), id(L), tuple(map(_tkey, inputs)), _tkey(out),
               _tkey(aux1), _tkey(aux2), _tkey(pre_add), tuple((k, tuple(v) if isinstance(v, list) else v) for k, v in kw.items()))
        P = _PARAMS.get(key)
        _PARAMS_STATS[P is None] += 1
        if P is None:
            P = _conv2d_params(spec, inputs, out, aux1=aux1, aux2=aux2, pre_add=pre_add, **kw)
            with _PARAMS_LOCK:
                _PARAMS[key] = P
                while len(_PARAMS) > _PARAMS_MAX:
                    _PARAMS.popitem(last=False)
        L.call("pp_conv2d", stream_handle(out), P)
        return out
    P = _conv2d_params(spec, inputs, out, aux1=aux1, aux2=aux2, pre_add=pre_add, **kw)
    x0 = inputs[0]
    n, h, w, _, _ = nhwc_view(x0)
    ho, wo = spec.out_hw(h, w)
    g = spec.groups
    if CONV_PROFILE is not None and out.is_cuda:
        flops = 2.0 * n * ho * wo * spec.cout * g * spec.cin_valid * spec.kh * spec.kw
        key = "f16" if x0.dtype == torch.float16 else ("f32x2" if spec.split else "f32")
        # (the rule of conv_direct.hip's launcher: these launches are streaming vector-ALU kernels, not MFMA tiles)
        if (spec.cout <= 4 and g == 1 and len(inputs) == 1 and (spec.sh, spec.sw, spec.dh, spec.dw) == (1, 1, 1, 1) and spec.kh <= 3
                and spec.kw <= 3 and spec.pad_mode == "zeros" and n * ho * wo >= 16384 and os.environ.get("PP_CONV_DIRECT") != "0"
                and (x0.dtype == torch.float16 or out.dtype == torch.float32)):
            # (r04: 3x3 f16 layers of this kind run on 16-channel halo MFMA tiles unless PP_CONV_SMALL_HALO=0 -- conv_halo_f16.hip)
            on_mfma = (x0.dtype == torch.float16 and spec.kh == 3 and spec.kw == 3 and sum(spec.seg_channels) > 32
                       and os.environ.get("PP_CONV_SMALL_HALO") != "0"
                       and os.environ.get("PP_CONV_HALO") != "0" and n * (-(-ho // 8)) * (-(-wo // 16)) >= 224)
            if not on_mfma:
                key = "direct"
        if CONV_PROFILE.detailed:
            key += f"|k{spec.kh}x{spec.kw} cin{spec.cin_valid} cout{spec.cout} g{g} M{n * ho * wo}"
        # algorithmic HBM bytes: every input / weight / epilogue operand read once, the output written once
        esz, osz = x0.element_size(), out.element_size()
        nbytes = (n * h * w * sum(spec.seg_channels) * g * esz + spec.weight.numel() * esz
                  + n * ho * wo * spec.cout * g * osz * (1 + sum(t is not None for t in (aux1, aux2, pre_add))))
        CONV_PROFILE.launch(key, flops,
                            lambda: L.call("pp_conv2d", stream_handle(out), P), nbytes)
    else:
        L.call("pp_conv2d", stream_handle(out), P)
    return out


def linear_of_unfold(spec: ConvSpec, x: torch.Tensor, out: torch.Tensor, kernel: int, stride: int, padding: int, *,
                     aux1=None, **kw) -> torch.Tensor:
    """out = linear(unfold(x)) WITHOUT the unfolded matrix (r04, pp_conv2d_params.flat_taps): `spec` is the Linear over the
    tap-major patch vectors (a 1x1 ConvSpec with kernel*kernel*C input channels), x f16 [N,H,W,C] (C % 8 == 0) the map the
    patches are taken from, out [N, ho, wo, Cout].  Bit-identical to conv2d(spec, [unfold(x)], out): same chunks, same order
    (the FusionFeedForward's fc2 reads the folded 40-channel map instead of a 49x copy, sparse_transformer.py:413-433)."""
    n, h, w, c, ldc = nhwc_view(x)
    if spec.kh != 1 or spec.kw != 1 or spec.groups != 1 or len(spec.seg_channels) != 1 or spec.split or x.dtype != torch.float16:
        raise ValueError("linear_of_unfold: needs an f16 single-segment Linear spec")
    if spec.cin_valid != kernel * kernel * c or c % 8:
        raise ValueError(f"linear_of_unfold: the Linear expects {spec.cin_valid} inputs, the patches hold {kernel * kernel * c}")
    ho, wo = (h + 2 * padding - kernel) // stride + 1, (w + 2 * padding - kernel) // stride + 1
    on, oh, ow, oc, oldc = nhwc_view(out)
    if (on, oh, ow) != (n, ho, wo) or oc < spec.cout:
        raise ValueError(f"linear_of_unfold: bad output view {tuple(out.shape)} for {(n, ho, wo, spec.cout)}")
    # the parameter block of the plain Linear on a stand-in of the patch matrix's shape, then the patch geometry
    meta = torch.empty(n, ho, wo, spec.seg_channels[0], device="meta", dtype=x.dtype)
    check_device(x)
    P = _conv2d_params(spec, [meta], out, aux1=aux1, virtual_input=True, **kw)
    P.in_ptr[0] = x.data_ptr()
    P.in_C[0], P.in_ldc[0] = c, ldc
    P.H, P.W = h, w
    P.kh = P.kw = kernel
    P.sh = P.sw = stride
    P.ph = P.pw = padding
    P.flat_taps = 1
    L = _lib.current()
    if CONV_PROFILE is not None and out.is_cuda:
        flops = 2.0 * n * ho * wo * spec.cout * spec.cin_valid
        key = "f16" + (f"|unfold{kernel}x{kernel}s{stride} cin{c} cout{spec.cout} g1 M{n * ho * wo}" if CONV_PROFILE.detailed else "")
        nbytes = (x.numel() + spec.weight.numel() + n * ho * wo * spec.cout * (2 if aux1 is not None else 1)) * 2.0
        CONV_PROFILE.launch(key, flops, lambda: L.call("pp_conv2d", stream_handle(out), P), nbytes)
    else:
        L.call("pp_conv2d", stream_handle(out), P)
    return out


def patch_conv_enabled() -> bool:
    """PP_CONV_PATCH=0: RAFT's 7x7 convolutions on the flow / the frames as pp_im2col + 1x1 PP_F32X2 again (the r01-r05 form)."""
    return os.environ.get("PP_CONV_PATCH", "1") != "0"


def conv2d_patch(spec: ConvSpec, x: torch.Tensor, out: torch.Tensor, kh: int, kw: int, *, stride: int = 1, padding: int = 0,
                 **kw_) -> torch.Tensor:
    """out = conv(x) for an f32 input of at most 4 channels WITHOUT the im2col tensor (r06, conv_patch.hip: PP_F32X2 + flat_taps).
    `spec` is the PP_F32X2 1x1 ConvSpec over the (ky, kx, c)-ordered patch vector that conv2d(spec, [im2col(x)]) used -- same
    weights, same three products per multiply-add; x fp32 [N,H,W,C] view (C <= 4, any pitch), out fp32 [N,Ho,Wo,Cout]
    (Cout 64 / 128).  RAFT: convf1 on the flow (update.py:100-106) and the encoder stems (extractor.py:130-136)."""
    n, h, w, c = x.shape
    if x.dtype != torch.float32 or out.dtype != torch.float32 or not spec.split or x.dim() != 4:
        raise ValueError("conv2d_patch: PP_F32X2 layers on fp32 tensors only")
    if x.stride(3) != 1 and c > 1:
        raise ValueError("conv2d_patch: channel dim must be unit-stride")
    ldc = x.stride(2)
    if (h > 1 and x.stride(1) != w * ldc) or (n > 1 and x.stride(0) != h * w * ldc):
        raise ValueError("conv2d_patch: rows / images must be dense")
    if spec.kh != 1 or spec.kw != 1 or spec.groups != 1 or len(spec.seg_channels) != 1 or spec.cin_valid != kh * kw * c:
        raise ValueError(f"conv2d_patch: the layer expects {spec.cin_valid} patch elements, the patches hold {kh * kw * c}")
    ho, wo = (h + 2 * padding - kh) // stride + 1, (w + 2 * padding - kw) // stride + 1
    on, oh, ow, oc, oldc = nhwc_view(out)
    if (on, oh, ow) != (n, ho, wo) or oc < spec.cout:
        raise ValueError(f"conv2d_patch: bad output view {tuple(out.shape)} for {(n, ho, wo, spec.cout)}")
    meta = torch.empty(n, ho, wo, spec.seg_channels[0], device="meta", dtype=x.dtype)
    check_device(x)
    P = _conv2d_params(spec, [meta], out, virtual_input=True, **kw_)
    P.in_ptr[0] = x.data_ptr()
    P.in_C[0], P.in_ldc[0] = c, ldc
    P.H, P.W = h, w
    P.kh, P.kw = kh, kw
    P.sh = P.sw = stride
    P.ph = P.pw = padding
    P.flat_taps = 1
    L = _lib.current()
    if CONV_PROFILE is not None and out.is_cuda:
        flops = 2.0 * n * ho * wo * spec.cout * spec.cin_valid
        key = "f32x2" + (f"|patch{kh}x{kw}s{stride} cin{c} cout{spec.cout} g1 M{n * ho * wo}" if CONV_PROFILE.detailed else "")
        nbytes = (x.numel() + spec.weight.numel() + n * ho * wo * spec.cout) * 4.0
        CONV_PROFILE.launch(key, flops, lambda: L.call("pp_conv2d", stream_handle(out), P), nbytes)
    else:
        L.call("pp_conv2d", stream_handle(out), P)
    return out


def split_pack(b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """fp32 [..., K] (K % 32 == 0, dense) -> the PP_F32X2 weight packing of the same shape (an f32-typed bit container; `out`: an
    existing dense tensor of that shape to fill)."""
    check_device(b, out)
    if b.dtype != torch.float32 or not b.is_contiguous() or b.shape[-1] % 32:
        raise ValueError("split_pack: expected a dense fp32 tensor with K % 32 == 0")
    if out is None:
        out = torch.empty_like(b)
    elif out.shape != b.shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("split_pack: bad output tensor")
    P = _lib.STRUCTS["pp_split_pack_params"]()
    setattr(P, "in", b.data_ptr())
    P.out, P.rows, P.K = out.data_ptr(), b.numel() // b.shape[-1], b.shape[-1]
    _call("pp_split_pack", out, P)
    return out


def batched_gemm_nt(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, scale: float = 0.0, split: bool = False,
                    b_packed: bool = False) -> torch.Tensor:
    """out[z, 0, m, n] = scale * sum_k a[z, 0, m, k] * b[z, n, k]  (pp_conv2d with gridDim.z = batch).

    a: [Z, 1, M, K] channels-last "image" of M pixels, b: [Z, N, K] per-batch "weights"
    (K a multiple of 32), out: [Z, 1, M, N].  Used for the RAFT all-pairs volume (corr.py:52-60).
    `split` (fp32 only): PP_F32X2 products -- b is packed on the device (pp_split_pack; `b_packed`: the caller already did),
    a is split inside the kernel.  a / out may be views with a dense [M, K] / [M, N] image per batch entry.
    """
    L = _lib.current()
    check_device(a, b, out)
    if split and a.dtype == torch.float32 and not b_packed:
        b = split_pack(b)
    z, one, m, k = a.shape
    zb, n, kb = b.shape
    if one != 1 or zb != z or kb != k or k % 32 != 0 or not b.is_contiguous():
        raise ValueError("batched_gemm_nt: bad shapes")
    if tuple(out.shape) != (z, 1, m, n):
        raise ValueError("batched_gemm_nt: bad output shape")
    P = _lib.STRUCTS["pp_conv2d_params"]()
    P.dtype = _lib.CONSTS["PP_F32X2"] if (split and a.dtype == torch.float32) else dtype_code(a.dtype)
    P.out_dtype = dtype_code(out.dtype)
    P.nseg = 1
    P.in_ptr[0] = a.data_ptr()
    P.in_C[0] = k
    P.in_ldc[0] = a.stride(2)
    P.in_zoff[0] = a.stride(0)
    P.N, P.H, P.W, P.Ho, P.Wo = 1, 1, m, 1, m
    P.kh = P.kw = P.sh = P.sw = P.dh = P.dw = 1
    P.weight = b.data_ptr()
    P.w_zoff = n * k
    P.Cout = n
    P.Z = z
    P.out = out.data_ptr()
    P.out_ldc = out.stride(2)
    P.out_zoff = out.stride(0)
    P.out_scale = scale
    if CONV_PROFILE is not None and out.is_cuda:
        key = "f16" if a.dtype == torch.float16 else ("f32x2" if split else "f32")
        CONV_PROFILE.launch(key, 2.0 * z * m * n * k, lambda: L.call("pp_conv2d", stream_handle(out), P),
                            float(a.numel() + b.numel() + out.numel()) * a.element_size())
    else:
        L.call("pp_conv2d", stream_handle(out), P)
    return out


# --------------------------------------------------------------------------------------------
# RAFT-stage kernels
# --------------------------------------------------------------------------------------------
def _call(name: str, ref: torch.Tensor, P) -> None:
    _lib.current().call(name, stream_handle(ref), P)


def im2col(x: torch.Tensor, out: torch.Tensor, kh: int, kw: int, *, stride=1, padding=0, pad_mode="zeros") -> torch.Tensor:
    """x [N,H,W,C] (tiny C) -> out [N,Ho,Wo,Kpad] patch matrix, k = (ky*kw+kx)*C + c."""
    check_device(x, out)
    n, h, w, c, ldc = nhwc_view(x)
    on, ho, wo, kpad, oldc = nhwc_view(out)
    if oldc != kpad or not out.is_contiguous():
        raise ValueError("im2col output must be dense")
    P = _lib.STRUCTS["pp_im2col_params"]()
    P.dtype, P.out_dtype, P.pad_mode = dtype_code(x.dtype), dtype_code(out.dtype), PAD[pad_mode]
    P.kh, P.kw, P.sh, P.sw, P.ph, P.pw = kh, kw, stride, stride, padding, padding
    P.in_ldc = ldc
    setattr(P, "in", x.data_ptr())
    P.N, P.H, P.W, P.C, P.Ho, P.Wo = n, h, w, c, ho, wo
    P.out, P.Kpad = out.data_ptr(), kpad
    _call("pp_im2col", out, P)
    return out


def instnorm(x: torch.Tensor, y: torch.Tensor, *, relu_pre=False, relu_post=False, skip: torch.Tensor | None = None,
             eps: float = 1e-5, scratch: torch.Tensor | None = None) -> torch.Tensor:
    """Instance norm over H*W per (n, c) of a channels-last fp32 tensor, fused pre-ReLU / skip / post-ReLU."""
    check_device(x, y, skip)
    n, h, w, c, ldc = nhwc_view(x)
    nchunks = max(1, min(256, (h * w) // 512))
    need = n * nchunks * c * 2
    if scratch is None or scratch.numel() < need:
        scratch = torch.empty(need, dtype=torch.float64, device=x.device)
    P = _lib.STRUCTS["pp_instnorm_params"]()
    P.x, P.x_ldc = x.data_ptr(), ldc
    P.y, P.y_ldc = y.data_ptr(), nhwc_view(y)[4]
    if skip is not None:
        P.skip, P.skip_ldc = skip.data_ptr(), nhwc_view(skip)[4]
    P.N, P.HW, P.C = n, h * w, c
    P.relu_pre, P.relu_post = int(relu_pre), int(relu_post)
    P.partials, P.nchunks, P.eps = scratch.data_ptr(), nchunks, eps
    _call("pp_instnorm", y, P)
    return y


def tiled_pitch(h: int, w: int) -> int:
    """Floats of an h x w correlation plane stored in 4 x 8 tiles of 128 bytes (pp_corr_lookup)."""
    return -(-h // 4) * -(-w // 8) * 32


@functools.lru_cache(maxsize=64)
def _tiled_order(h: int, w: int) -> tuple:
    th, tw = -(-h // 4), -(-w // 8)
    out = []
    for ty in range(th):
        for tx in range(tw):
            for r in range(4):
                for c in range(8):
                    y, x = 4 * ty + r, 8 * tx + c
                    out.append(y * w + x if y < h and x < w else h * w)
    return tuple(out)


def tiled_order(h: int, w: int) -> list[int]:
    """Source pixel (y*w + x) of every position of a tiled h x w plane; the zero padding is index h*w."""
    return list(_tiled_order(h, w))


_TILED_INDEX: dict = {}


def tiled_order_index(h: int, w: int, device) -> torch.Tensor:
    """tiled_order(h, w) as an int64 device tensor, built once per (h, w, device) (RAFT asks for it per chunk: the list is
    ~15k integers at 90x160)."""
    # (same hand-over as device_ints -- cached by value, uploaded outside any lock, a later user on another stream
    #  waits for the upload's event)
    # (device_ints keys on the VALUES -- re-hashing ~15k integers per call; this front cache keys on the geometry.
    #  Entries are only ever handed out pinned-or-cached tensors of device_ints, so its hand-over rules still apply.)
    key = (h, w, str(device))
    t = _TILED_INDEX.get(key)
    if t is None or torch.cuda.is_current_stream_capturing() or PIN_DEVICE_INTS > 0:
        t = device_ints(_tiled_order_cached(h, w), device)
        if len(_TILED_INDEX) > 32:
            _TILED_INDEX.clear()
        _TILED_INDEX[key] = t
    return t


@functools.lru_cache(maxsize=16)
def _tiled_order_cached(h: int, w: int) -> tuple:
    return tuple(_tiled_order(h, w))


def avgpool2x2(x: torch.Tensor, out: torch.Tensor, hw: tuple[int, int] | None = None, in_tiled: bool = False,
               out_tiled: bool = False) -> torch.Tensor:
    """x [B,H,W] fp32 dense -> out [B,H//2,W//2]; with `hw` = (H, W) the planes are flat [B, pitch] tensors in the
    row-major or the tiled layout (in_tiled / out_tiled; tiled_pitch())."""
    check_device(x, out)
    if hw is None:
        b, h, w = x.shape
        ok = tuple(out.shape) == (b, h // 2, w // 2)
    else:
        h, w = hw
        b = x.shape[0]
        ok = (x.numel() == b * (tiled_pitch(h, w) if in_tiled else h * w)
              and out.numel() == b * (tiled_pitch(h // 2, w // 2) if out_tiled else (h // 2) * (w // 2)))
    if not x.is_contiguous() or not out.is_contiguous() or not ok:
        raise ValueError("avgpool2x2: bad shapes")
    P = _lib.STRUCTS["pp_avgpool2x2_params"]()
    setattr(P, "in", x.data_ptr())
    P.out, P.B, P.H, P.W = out.data_ptr(), b, h, w
    P.in_tiled, P.out_tiled = int(in_tiled), int(out_tiled)
    _call("pp_avgpool2x2", out, P)
    return out


def corr_lookup(pyramid: list, flow: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """pyramid[l]: [N, h*w, h_l, w_l] fp32 (row-major planes) or a tuple (flat [N, h*w, pitch] tensor, h_l, w_l, tiled);
    flow [N,h,w,2] view; out [N,h,w,324] view."""
    levels = [(t, t.shape[2], t.shape[3], False) if torch.is_tensor(t) else t for t in pyramid]
    check_device(*[t for t, _, _, _ in levels], flow, out)
    n, h, w, _, fl = nhwc_view(flow)
    P = _lib.STRUCTS["pp_corr_lookup_params"]()
    for l, (t, ph, pw, tiled) in enumerate(levels):
        pitch = tiled_pitch(ph, pw) if tiled else ph * pw
        if not t.is_contiguous() or t.shape[0] != n or t.numel() != n * h * w * pitch:
            raise ValueError("corr_lookup: bad pyramid level")
        P.pyr[l], P.ph[l], P.pw[l], P.tiled[l] = t.data_ptr(), ph, pw, int(tiled)
    P.flow, P.flow_ldc = flow.data_ptr(), fl
    P.out, P.out_ldc = out.data_ptr(), nhwc_view(out)[4]
    P.N, P.h, P.w = n, h, w
    if CONV_PROFILE is not None and out.is_cuda:
        # algorithmic bytes: per pixel the 10x10 fp32 footprint of the 9x9 bilinear samples on each level
        # + the 324 fp32 outputs
        nbytes = float(n * h * w) * (len(levels) * 100 * 4 + 324 * 4)
        CONV_PROFILE.launch("corr_lookup", 0.0, lambda: _call("pp_corr_lookup", out, P), nbytes)
    else:
        _call("pp_corr_lookup", out, P)
    return out


def lookup_fused_enabled() -> bool:
    """PP_LOOKUP_FUSED=1: RAFT's lookup + convc1 as ONE launch (pp_corr_lookup_conv).  Off by default: measured on the MI355X at
    158 x 45 x 80 pixels the fused kernel is bit-identical to the two launches and 1.5x SLOWER (1691 vs 647 + 467 us,
    tools/bench_lookup.py, profiles/r06_lookup_fusion.md) -- a CU cannot hold what the fusion needs at once: the operand fragments
    of a pixel tile (44 KB per 32 pixels), >= 12 pixels' windows in flight to cover the gather latency (3 KB each; the two-launch
    lookup runs 32 waves per CU), and a 360 KB weight stream per tile that fits neither registers nor L1."""
    return os.environ.get("PP_LOOKUP_FUSED", "0") == "1"


def corr_lookup_conv(pyramid: list, flow: torch.Tensor, spec: ConvSpec, out: torch.Tensor, **kw) -> torch.Tensor:
    """out = act(conv1x1(corr_lookup(pyramid, flow))) in ONE launch (r06, pp_corr_lookup_conv): `spec` is the PP_F32X2 1x1
    convolution 324 -> 256 (RAFT's convc1, update.py:94-112); pyramid / flow as for corr_lookup(); out fp32 [N,h,w,256] view.
    The 324-channel lookup tensor is never written."""
    levels = [(t, t.shape[2], t.shape[3], False) if torch.is_tensor(t) else t for t in pyramid]
    check_device(*[t for t, _, _, _ in levels], flow)
    n, h, w, _, fl = nhwc_view(flow)
    if not spec.split or spec.cin_valid != 324 or spec.cout != 256 or (spec.kh, spec.kw, spec.groups) != (1, 1, 1):
        raise ValueError("corr_lookup_conv: needs the PP_F32X2 1x1 convolution 324 -> 256")
    S = _lib.STRUCTS["pp_corr_lookup_params"]()
    for l, (t, ph, pw, tiled) in enumerate(levels):
        pitch = tiled_pitch(ph, pw) if tiled else ph * pw
        if not t.is_contiguous() or t.shape[0] != n or t.numel() != n * h * w * pitch:
            raise ValueError("corr_lookup_conv: bad pyramid level")
        S.pyr[l], S.ph[l], S.pw[l], S.tiled[l] = t.data_ptr(), ph, pw, int(tiled)
    S.flow, S.flow_ldc = flow.data_ptr(), fl
    S.N, S.h, S.w = n, h, w
    meta = torch.empty(n, h, w, 324, device="meta", dtype=torch.float32)
    G = _conv2d_params(spec, [meta], out, virtual_input=True, **kw)
    L = _lib.current()
    if CONV_PROFILE is not None and out.is_cuda:
        # one launch, two roofline entries: the lookup's algorithmic bytes WITHOUT its output tensor, the projection's flops
        nbytes = float(n * h * w) * (len(levels) * 100 * 4 + 256 * 4) + spec.weight.numel() * 4.0
        flops = 2.0 * n * h * w * 256 * 324
        CONV_PROFILE.launch("lookup_conv", flops, lambda: L.call2("pp_corr_lookup_conv", stream_handle(out), S, G), nbytes)
    else:
        L.call2("pp_corr_lookup_conv", stream_handle(out), S, G)
    return out


def convex_upsample(mask: torch.Tensor, flow: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """mask [N,h,w,576], flow [N,h,w,2] view -> out [N,8h,8w,2] dense fp32."""
    check_device(mask, flow, out)
    n, h, w, _, ml = nhwc_view(mask)
    P = _lib.STRUCTS["pp_convex_upsample_params"]()
    P.mask, P.mask_ldc = mask.data_ptr(), ml
    P.flow, P.flow_ldc = flow.data_ptr(), nhwc_view(flow)[4]
    if not out.is_contiguous() or tuple(out.shape) != (n, 8 * h, 8 * w, 2):
        raise ValueError("convex_upsample: bad output")
    P.out, P.N, P.h, P.w = out.data_ptr(), n, h, w
    _call("pp_convex_upsample", out, P)
    return out


# --------------------------------------------------------------------------------------------
# sampling kernels
# --------------------------------------------------------------------------------------------
def deform_cols(x0: torch.Tensor, x1: torch.Tensor | None, om: torch.Tensor, cols: torch.Tensor, *, dg: int = 16,
                flow: torch.Tensor | None = None) -> torch.Tensor:
    """Sampling half of the modulated deformable 3x3 conv: cols [N,H,W,9*Cin] (tap-major)."""
    check_device(x0, x1, om, cols, flow)
    P, cin = _deform_cols_params(x0, x1, om, dg, flow)
    if cols.dtype != x0.dtype or not cols.is_contiguous() or tuple(cols.shape) != (*x0.shape[:3], 9 * cin):
        raise ValueError("deform_cols: bad cols tensor")
    P.cols = cols.data_ptr()
    _call("pp_deform_cols", cols, P)
    return cols


def deform_fused(h: int, w: int) -> bool:
    """Whether a recurrence runs its modulated deformable convolution on h x w images as one launch (pp_deform_conv) or as
    pp_deform_cols + 1x1 pp_conv2d.  Same results bit for bit either way; measured on the MI355X (profiles/r03_deform_fusion.md)
    the one-launch form wins where the 1x1 convolution would be the in-work-group split-K kernel (small images: flow
    completion's 45 x 80, 46.9 vs 54.3 us) and loses on large ones (feature propagation's 90 x 160: 378 vs 358 us), so the
    default follows that kernel's rule (conv_ksplit.hip: at most 160 32-pixel tiles per image).  PP_DEFORM_FUSED=0 / force:
    never / always."""
    mode = os.environ.get("PP_DEFORM_FUSED", "1")
    if mode == "0":
        return False
    return mode == "force" or (h * w + 31) // 32 <= 160


def _deform_cols_params(x0, x1, om, dg, flow):
    n, h, w, c0, l0 = nhwc_view(x0)
    P = _lib.STRUCTS["pp_deform_cols_params"]()
    P.dtype, P.dg = dtype_code(x0.dtype), dg
    P.x0, P.x0_C, P.x0_ldc = x0.data_ptr(), c0, l0
    c1 = 0
    if x1 is not None:
        if x1.dtype != x0.dtype or tuple(x1.shape[:3]) != (n, h, w):
            raise ValueError("deform: the two inputs must share dtype and N,H,W")
        _, _, _, c1, l1 = nhwc_view(x1)
        P.x1, P.x1_C, P.x1_ldc = x1.data_ptr(), c1, l1
    if om.dtype != torch.float32 or om.shape[3] != 27 * dg or tuple(om.shape[:3]) != (n, h, w):
        raise ValueError("deform: om must be fp32 [N,H,W,27*dg]")
    P.om, P.om_ldc = om.data_ptr(), nhwc_view(om)[4]
    if flow is not None:
        if flow.dtype != torch.float32 or tuple(flow.shape[:3]) != (n, h, w):
            raise TypeError("deform: flow must be fp32 [N,H,W,>=2]")
        P.flow, P.flow_ldc = flow.data_ptr(), nhwc_view(flow)[4]
    P.N, P.H, P.W = n, h, w
    return P, c0 + c1


def deform_conv(spec: ConvSpec, x0: torch.Tensor, x1: torch.Tensor | None, om: torch.Tensor, out: torch.Tensor, *,
                dg: int = 16, flow: torch.Tensor | None = None, **kw) -> torch.Tensor:
    """The modulated deformable 3x3 convolution in one launch (pp_deform_conv): deform_cols's sampling feeds the MFMA operand
    of the 1x1 convolution `spec` over the 9*Cin columns directly; the column tensor is never written.  f16 tensors."""
    check_device(x0, x1, om, flow)
    S, cin = _deform_cols_params(x0, x1, om, dg, flow)
    n, h, w = x0.shape[:3]
    cols = torch.empty(n, h, w, 9 * cin, dtype=x0.dtype, device="meta")   # shape / dtype only
    G = _conv2d_params(spec, [cols], out, virtual_input=True, **kw)
    L = _lib.current()
    if CONV_PROFILE is not None and out.is_cuda:
        key = "f16"
        if CONV_PROFILE.detailed:
            key += f"|deform3x3 cin{cin} cout{spec.cout} g1 M{n * h * w}"
        esz = x0.element_size()
        nbytes = n * h * w * (cin * esz + 27 * dg * 4 + spec.cout * out.element_size()) + spec.weight.numel() * esz
        CONV_PROFILE.launch(key, 2.0 * n * h * w * spec.cout * 9 * cin, lambda: L.call2("pp_deform_conv", stream_handle(out), S, G), nbytes)
    else:
        L.call2("pp_deform_conv", stream_handle(out), S, G)
    return out


def upsample2x(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    check_device(x, out)
    n, h, w, c, ldc = nhwc_view(x)
    on, oh, ow, oc, oldc = nhwc_view(out)
    if (on, oh, ow, oc) != (n, 2 * h, 2 * w, c) or out.dtype != x.dtype:
        raise ValueError("upsample2x: bad output")
    P = _lib.STRUCTS["pp_upsample2x_params"]()
    P.dtype = dtype_code(x.dtype)
    setattr(P, "in", x.data_ptr())
    P.in_ldc, P.out, P.out_ldc = ldc, out.data_ptr(), oldc
    P.N, P.H, P.W, P.C = n, h, w, c
    _call("pp_upsample2x", out, P)
    return out


def rfc_prep(flows: torch.Tensor, masks_u8: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """flows fp32 [2,T,H,W,2], masks u8 [T+1,H,W] -> out f16 / f32 [T,2,H,W,4] (direction 1 time-flipped)."""
    check_device(flows, masks_u8, out)
    _, t, h, w, _ = flows.shape
    if not (flows.is_contiguous() and masks_u8.is_contiguous() and out.is_contiguous()):
        raise ValueError("rfc_prep: tensors must be dense")
    if masks_u8.dtype != torch.uint8 or tuple(masks_u8.shape) != (t + 1, h, w) or tuple(out.shape) != (t, 2, h, w, 4):
        raise ValueError("rfc_prep: bad shapes")
    P = _lib.STRUCTS["pp_rfc_prep_params"]()
    P.flows, P.masks, P.out, P.T, P.H, P.W = flows.data_ptr(), masks_u8.data_ptr(), out.data_ptr(), t, h, w
    P.out_dtype = dtype_code(out.dtype)
    _call("pp_rfc_prep", out, P)
    return out


def flow_combine(pred: torch.Tensor, flows: torch.Tensor, masks_u8: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """pred f16 / f32 [T,2,H,W,>=2] (direction 1 time-flipped) + gt flows fp32 [2,T,H,W,2] -> out fp32 [2,T,H,W,2]."""
    check_device(pred, flows, masks_u8, out)
    _, t, h, w, _ = flows.shape
    if pred.dtype not in (torch.float16, torch.float32) or tuple(pred.shape[:4]) != (t, 2, h, w):
        raise ValueError("flow_combine: bad pred")
    P = _lib.STRUCTS["pp_flow_combine_params"]()
    P.pred, P.pred_ldc, P.pred_dtype = pred.data_ptr(), pred.stride(3), dtype_code(pred.dtype)
    P.flows, P.masks, P.out, P.T, P.H, P.W = flows.data_ptr(), masks_u8.data_ptr(), out.data_ptr(), t, h, w
    _call("pp_flow_combine", out, P)
    return out


# --------------------------------------------------------------------------------------------
# flow-guided propagation kernels
# --------------------------------------------------------------------------------------------
def img_prop_step(x_cur, m_cur, f_new, m_new, *, f_prev=None, m_prev=None, flow_prop=None, flow_check=None,
                  mask_input=False) -> None:
    """One image-propagation step on fp32 [H,W,3] frames / u8 [H,W] masks / fp32 [H,W,2] flows."""
    check_device(x_cur, m_cur, f_new, m_new, f_prev, m_prev, flow_prop, flow_check)
    h, w = m_cur.shape
    P = _lib.STRUCTS["pp_img_prop_step_params"]()
    first = f_prev is None
    for name, t in (("x_cur", x_cur), ("m_cur", m_cur), ("f_new", f_new), ("m_new", m_new), ("f_prev", f_prev),
                    ("m_prev", m_prev), ("flow_prop", flow_prop), ("flow_check", flow_check)):
        if t is not None:
            if not t.is_contiguous():
                raise ValueError(f"img_prop_step: {name} must be dense")
            setattr(P, name, t.data_ptr())
    P.H, P.W, P.first, P.mask_input = h, w, int(first), int(mask_input)
    _call("pp_img_prop_step", f_new, P)


def pack_encoder_input(frames, prop, m_in, m_upd, out, updated=None) -> torch.Tensor:
    """frames/prop fp32 [T,H,W,3], masks u8 [T,H,W] -> out f16 / f32 [T,H,W,8] (+ updated fp32 [T,H,W,3])."""
    check_device(frames, prop, m_in, m_upd, out, updated)
    for t in (frames, prop, m_in, m_upd, out):
        if not t.is_contiguous():
            raise ValueError("pack_encoder_input: tensors must be dense")
    P = _lib.STRUCTS["pp_pack_encoder_input_params"]()
    P.frames, P.prop, P.m_in, P.m_upd, P.out = (frames.data_ptr(), prop.data_ptr(), m_in.data_ptr(), m_upd.data_ptr(),
                                                out.data_ptr())
    if updated is not None:
        P.updated = updated.data_ptr()
    P.out_dtype = dtype_code(out.dtype)
    P.total_pixels = m_in.numel()
    _call("pp_pack_encoder_input", out, P)
    return out


def flow_down4(flows: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """fp32 [N,H,W,2] -> fp32 [N,H/4,W/4,2] (bilinear x1/4, align_corners=False, then /4)."""
    check_device(flows, out)
    n, h, w, _ = flows.shape
    if not flows.is_contiguous() or not out.is_contiguous() or tuple(out.shape) != (n, h // 4, w // 4, 2):
        raise ValueError("flow_down4: bad shapes")
    P = _lib.STRUCTS["pp_flow_down4_params"]()
    setattr(P, "in", flows.data_ptr())
    P.out, P.N, P.H, P.W = out.data_ptr(), n, h, w
    _call("pp_flow_down4", out, P)
    return out


def featprop_aux(flow_prop, flow_check, maskpair, out) -> torch.Tensor:
    """(flow.x, flow.y, fb-valid, m_in, m_updated, 0,0,0) f16 planes for the learnable propagation."""
    check_device(flow_prop, flow_check, maskpair, out)
    n, h, w, _ = flow_prop.shape
    for t in (flow_prop, flow_check, maskpair, out):
        if not t.is_contiguous():
            raise ValueError("featprop_aux: tensors must be dense")
    P = _lib.STRUCTS["pp_featprop_aux_params"]()
    P.flow_prop, P.flow_check, P.maskpair, P.out = flow_prop.data_ptr(), flow_check.data_ptr(), maskpair.data_ptr(), out.data_ptr()
    if maskpair.dtype != out.dtype:
        raise TypeError("featprop_aux: maskpair and out must share a dtype")
    P.dtype = dtype_code(out.dtype)
    P.N, P.H, P.W = n, h, w
    _call("pp_featprop_aux", out, P)
    return out


def flow_warp(x: torch.Tensor, flow: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """Bilinear flow warp (zeros padding, align_corners=True) of channels-last features."""
    check_device(x, flow, out)
    n, h, w, c, ldc = nhwc_view(x)
    if not flow.is_contiguous() or flow.dtype != torch.float32 or tuple(flow.shape) != (n, h, w, 2):
        raise ValueError("flow_warp: bad flow")
    P = _lib.STRUCTS["pp_flow_warp_params"]()
    P.dtype, P.x, P.x_ldc, P.flow = dtype_code(x.dtype), x.data_ptr(), ldc, flow.data_ptr()
    P.out, P.out_ldc = out.data_ptr(), nhwc_view(out)[4]
    P.N, P.H, P.W, P.C = n, h, w, c
    _call("pp_flow_warp", out, P)
    return out


# --------------------------------------------------------------------------------------------
# transformer kernels
# --------------------------------------------------------------------------------------------
def layernorm(x: torch.Tensor, out: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """x f16 [T,fh,fw,512] dense -> out f16 [T,Hp,Wp,512] dense (Hp>=fh, Wp>=fw; pad region untouched)."""
    check_device(x, out, gamma, beta)
    t, fh, fw, c = x.shape
    ot, hp, wp, oc = out.shape
    if not (x.is_contiguous() and out.is_contiguous()) or ot != t or oc != c:
        raise ValueError("layernorm: bad tensors")
    P = _lib.STRUCTS["pp_layernorm_params"]()
    P.x, P.out, P.gamma, P.beta = x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    if x.dtype != out.dtype:
        raise TypeError("layernorm: x and out must share a dtype")
    P.dtype = dtype_code(x.dtype)
    P.T, P.fh, P.fw, P.Hp, P.Wp, P.C, P.eps = t, fh, fw, hp, wp, c, eps
    _call("pp_layernorm", out, P)
    return out


def pool_tokens(x: torch.Tensor, out: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    check_device(x, out, weight, bias)
    t, hp, wp, c = x.shape
    if not (x.is_contiguous() and out.is_contiguous()) or tuple(out.shape) != (t, hp // 4, wp // 4, c):
        raise ValueError("pool_tokens: bad tensors")
    P = _lib.STRUCTS["pp_pool_tokens_params"]()
    P.x, P.out, P.weight, P.bias = x.data_ptr(), out.data_ptr(), weight.data_ptr(), bias.data_ptr()
    if x.dtype != out.dtype:
        raise TypeError("pool_tokens: x and out must share a dtype")
    P.dtype = dtype_code(x.dtype)
    P.T, P.Hp, P.Wp, P.C = t, hp, wp, c
    _call("pp_pool_tokens", out, P)
    return out


def attention_exact_enabled() -> bool:
    """fp32 storage only: q, k, v and the probabilities stay fp32 (ABI v11 `exact`) when the f32 convolutions are on the f32 MFMA
    instructions too (PP_F32_GEMM=exact: the whole fp16 "disable" generator at fp32 level), or on its own with PP_ATTN_F32=exact;
    PP_ATTN_F32=f16 keeps the f16 MFMA operands whatever PP_F32_GEMM says."""
    mode = os.environ.get("PP_ATTN_F32", "").lower()
    if mode in ("exact", "f16"):
        return mode == "exact"
    return not f32_split_enabled()


def window_attention(qkv: torch.Tensor, pkv: torch.Tensor, win_masked: torch.Tensor, t_ind: torch.Tensor,
                     out: torch.Tensor, exact: bool | None = None) -> torch.Tensor:
    """qkv f16 [t,Hp,Wp,1536], pkv f16 [t,npool,1024], win_masked i32 [nwin], t_ind i32 [nt] -> out f16 [t,fh,fw,512]
    (or all three fp32; `exact`: see attention_exact_enabled(), fp32 storage only)."""
    check_device(qkv, pkv, win_masked, t_ind, out)
    t, hp, wp, c3 = qkv.shape
    _, fh, fw, c = out.shape
    for tt in (qkv, pkv, out, win_masked, t_ind):
        if not tt.is_contiguous():
            raise ValueError("window_attention: tensors must be dense")
    if c3 != 1536 or c != 512 or pkv.shape[2] != 1024 or win_masked.dtype != torch.int32 or t_ind.dtype != torch.int32:
        raise ValueError("window_attention: bad tensors")
    P = _lib.STRUCTS["pp_window_attention_params"]()
    P.qkv, P.pkv, P.win_masked, P.t_ind, P.out = (qkv.data_ptr(), pkv.data_ptr(), win_masked.data_ptr(), t_ind.data_ptr(),
                                                  out.data_ptr())
    if not (qkv.dtype == pkv.dtype == out.dtype):
        raise TypeError("window_attention: qkv, pkv and out must share a dtype")
    P.dtype = dtype_code(qkv.dtype)
    if exact and qkv.dtype != torch.float32:
        raise ValueError("window_attention: the exact (fp32-operand) core needs fp32 storage")
    P.exact = int(qkv.dtype == torch.float32 and (attention_exact_enabled() if exact is None else bool(exact)))
    P.t, P.nt, P.Hp, P.Wp, P.fh, P.fw, P.npool = t, t_ind.numel(), hp, wp, fh, fw, pkv.shape[1]
    P.scale = 1.0 / (128 ** 0.5)
    if CONV_PROFILE is not None and out.is_cuda:
        # algorithmic work: masked window = 45 t queries x nt (45 + 148 + npool) keys per head, unmasked = 45 x 45 per
        # frame and head; 4 flops per (query, key, channel).  (the masked-window count is read back: profiling only)
        nwin = win_masked.numel()
        nm = int(win_masked.ne(0).sum())
        nk = t_ind.numel() * (45 + 148 + pkv.shape[1])
        flops = 4.0 * 128 * 4 * (nm * 45.0 * t * nk + (nwin - nm) * t * 45.0 * 45.0)
        nbytes = float(qkv.numel() + pkv.numel() + out.numel()) * qkv.element_size()
        CONV_PROFILE.launch("attention", flops, lambda: _call("pp_window_attention", out, P), nbytes)
    else:
        _call("pp_window_attention", out, P)
    return out


def fold(x: torch.Tensor, out: torch.Tensor, fh: int, fw: int, normalize: bool, gelu: bool = False) -> torch.Tensor:
    """x f16 [T, fh*fw, 49*C] (tap-major) -> out f16 [T,H,W,C] overlap-add (optionally averaged; `gelu`: followed by the
    exact GELU on the storage-rounded value, for unfold_gelu(..., pre_activated=True))."""
    check_device(x, out)
    t, h, w, c = out.shape
    if not (x.is_contiguous() and out.is_contiguous()) or tuple(x.shape) != (t, fh * fw, 49 * c):
        raise ValueError("fold: bad tensors")
    P = _lib.STRUCTS["pp_fold_params"]()
    setattr(P, "in", x.data_ptr())
    if x.dtype != out.dtype:
        raise TypeError("fold: x and out must share a dtype")
    P.dtype = dtype_code(x.dtype)
    P.out, P.T, P.H, P.W, P.C, P.fh, P.fw, P.normalize = out.data_ptr(), t, h, w, c, fh, fw, int(normalize)
    P.gelu = int(gelu)
    _call("pp_fold", out, P)
    return out


def unfold_gelu(x: torch.Tensor, out: torch.Tensor, fh: int, fw: int, pre_activated: bool = False) -> torch.Tensor:
    """x f16 [T,H,W,C] -> out f16 [T, fh*fw, 49*C] (tap-major) with exact GELU applied (`pre_activated`: x already holds
    GELU(value) -- fold(..., gelu=True) -- and is only re-extracted)."""
    check_device(x, out)
    t, h, w, c = x.shape
    if not (x.is_contiguous() and out.is_contiguous()) or tuple(out.shape) != (t, fh * fw, 49 * c):
        raise ValueError("unfold_gelu: bad tensors")
    P = _lib.STRUCTS["pp_unfold_gelu_params"]()
    setattr(P, "in", x.data_ptr())
    if x.dtype != out.dtype:
        raise TypeError("unfold_gelu: x and out must share a dtype")
    P.dtype = dtype_code(x.dtype)
    P.out, P.T, P.H, P.W, P.C, P.fh, P.fw = out.data_ptr(), t, h, w, c, fh, fw
    P.pre_activated = int(pre_activated)
    _call("pp_unfold_gelu", out, P)
    return out


def compose_u8(pred: torch.Tensor, frame_ids: torch.Tensor, first: torch.Tensor, masks_u8: torch.Tensor,
               orig_u8: torch.Tensor, comp_u8: torch.Tensor) -> torch.Tensor:
    """pred f16 [L,H,W,>=3]; frame_ids/first i32 [L]; masks u8 [T,H,W]; orig/comp u8 [T,H,W,3] (comp updated in place)."""
    check_device(pred, frame_ids, first, masks_u8, orig_u8, comp_u8)
    l, h, w, _ = pred.shape
    P = _lib.STRUCTS["pp_compose_u8_params"]()
    P.pred, P.pred_ldc, P.pred_dtype = pred.data_ptr(), pred.stride(2), dtype_code(pred.dtype)
    P.frame_ids, P.first, P.masks, P.orig, P.comp = (frame_ids.data_ptr(), first.data_ptr(), masks_u8.data_ptr(),
                                                     orig_u8.data_ptr(), comp_u8.data_ptr())
    P.L, P.H, P.W = l, h, w
    _call("pp_compose_u8", comp_u8, P)
    return comp_u8


# --------------------------------------------------------------------------------------------
# device-side pre / post-processing (the node's byte plumbing)
# --------------------------------------------------------------------------------------------
def frames_from_image(image: torch.Tensor, canvas_hw: tuple[int, int] | None = None, offset: tuple[int, int] = (0, 0),
                      want_f32: bool = True):
    """IMAGE fp32 [T,H,W,3] on the device -> (uint8 frames [T,Ho,Wo,3], fp32 frames in [-1,1] or None); with a
    canvas the frames are placed at `offset` = (oy, ox) inside zeros (the outpaint canvas)."""
    check_device(image)
    if image.dtype != torch.float32 or image.dim() != 4 or image.shape[3] != 3 or not image.is_contiguous():
        raise ValueError("frames_from_image: expected a dense fp32 [T,H,W,3] image")
    t, h, w, _ = image.shape
    ho, wo = canvas_hw or (h, w)
    u8 = torch.empty(t, ho, wo, 3, dtype=torch.uint8, device=image.device)
    f32 = torch.empty(t, ho, wo, 3, dtype=torch.float32, device=image.device) if want_f32 else None
    P = _lib.STRUCTS["pp_frames_from_image_params"]()
    P.image, P.out_u8 = image.data_ptr(), u8.data_ptr()
    if f32 is not None:
        P.out_f32 = f32.data_ptr()
    P.T, P.H, P.W, P.Ho, P.Wo, P.oy, P.ox = t, h, w, ho, wo, offset[0], offset[1]
    _call("pp_frames_from_image", u8, P)
    return u8, f32


def frames_from_u8(frames_u8: torch.Tensor) -> torch.Tensor:
    """uint8 frames [T,H,W,3] on the device -> fp32 frames in [-1,1]: (u8/255)*2-1 (image_utils.py:178-191)."""
    check_device(frames_u8)
    if frames_u8.dtype != torch.uint8 or not frames_u8.is_contiguous() or frames_u8.dim() != 4:
        raise ValueError("frames_from_u8: expected dense uint8 [T,H,W,3] frames")
    t, h, w, _ = frames_u8.shape
    f32 = torch.empty(t, h, w, 3, dtype=torch.float32, device=frames_u8.device)
    P = _lib.STRUCTS["pp_frames_from_image_params"]()
    P.in_u8, P.out_u8, P.out_f32 = frames_u8.data_ptr(), frames_u8.data_ptr(), f32.data_ptr()
    P.T, P.H, P.W, P.Ho, P.Wo = t, h, w, h, w
    _call("pp_frames_from_image", f32, P)
    return f32


def image_from_u8(u8: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """uint8 tensor -> fp32 k/255 of the same shape (handle_output, image_utils.py:276-290)."""
    check_device(u8, out)
    if u8.dtype != torch.uint8 or not u8.is_contiguous() or u8.numel() % 4:
        raise ValueError("image_from_u8: expected a dense uint8 tensor with a multiple of 4 elements")
    if out is None:
        out = torch.empty(u8.shape, dtype=torch.float32, device=u8.device)
    P = _lib.STRUCTS["pp_image_from_u8_params"]()
    setattr(P, "in", u8.data_ptr())
    P.out, P.total = out.data_ptr(), u8.numel()
    _call("pp_image_from_u8", out, P)
    return out


def mask_dilate(mask: torch.Tensor, iterations: int) -> torch.Tensor:
    """MASK fp32 (ComfyUI) or uint8 [N,H,W] on the device -> uint8 {0,1} [N,H,W] (read_masks, image_utils.py:142-175)."""
    check_device(mask)
    if mask.dim() != 3 or not mask.is_contiguous() or mask.dtype not in (torch.float32, torch.uint8):
        raise ValueError("mask_dilate: expected a dense fp32 / uint8 [N,H,W] mask")
    n, h, w = mask.shape
    out = torch.empty(n, h, w, dtype=torch.uint8, device=mask.device)
    scratch = torch.empty(n, h, w, dtype=torch.uint8, device=mask.device)
    P = _lib.STRUCTS["pp_mask_dilate_params"]()
    P.dtype, P.iterations = dtype_code(mask.dtype), int(iterations)
    setattr(P, "in", mask.data_ptr())
    P.out, P.scratch, P.N, P.H, P.W = out.data_ptr(), scratch.data_ptr(), n, h, w
    _call("pp_mask_dilate", out, P)
    return out


def clip_masks(m_in: torch.Tensor, m_upd: torch.Tensor, fh: int, fw: int, dtype: torch.dtype = torch.float16):
    """u8 masks [T,H,W] -> (maskpair f16 / f32 [T,H/4,W/4,8], tokmask u8 [T,fh,fw]) (propainter.py:409-428)."""
    check_device(m_in, m_upd)
    t, h, w = m_in.shape
    if not (m_in.is_contiguous() and m_upd.is_contiguous()) or m_in.dtype != torch.uint8 or m_upd.dtype != torch.uint8:
        raise ValueError("clip_masks: expected dense uint8 masks")
    maskpair = torch.empty(t, h // 4, w // 4, 8, dtype=dtype, device=m_in.device)
    tok = torch.empty(t, fh, fw, dtype=torch.uint8, device=m_in.device)
    P = _lib.STRUCTS["pp_clip_masks_params"]()
    P.m_in, P.m_upd, P.maskpair, P.tokmask = m_in.data_ptr(), m_upd.data_ptr(), maskpair.data_ptr(), tok.data_ptr()
    P.T, P.H, P.W, P.fh, P.fw, P.dtype = t, h, w, fh, fw, dtype_code(dtype)
    _call("pp_clip_masks", tok, P)
    return maskpair, tok


def window_flags(tokmask: torch.Tensor, g0: int, lt: int, window: tuple[int, int]) -> torch.Tensor:
    """tokmask u8 [T,fh,fw] -> i32 [nwin] 'window is masked' flags for local frames [g0, g0+lt) (sparse_transformer.py:321-326)."""
    check_device(tokmask)
    t, fh, fw = tokmask.shape
    nwin = -(-fh // window[0]) * -(-fw // window[1])
    flags = torch.empty(nwin, dtype=torch.int32, device=tokmask.device)
    P = _lib.STRUCTS["pp_window_flags_params"]()
    P.tokmask, P.flags = tokmask.data_ptr(), flags.data_ptr()
    P.T, P.fh, P.fw, P.g0, P.lt, P.wh, P.ww = t, fh, fw, g0, lt, window[0], window[1]
    _call("pp_window_flags", flags, P)
    return flags
