"""pp_conv2d (implicit-GEMM MFMA convolution) against torch.nn.functional.conv2d in fp32.

Tolerances: f32 path -> 2e-5 relative to the output scale (exact-f32 MFMA, different summation
order); f16 path -> 4e-3 relative (f16 operands/outputs, fp32 accumulate)."""
import pytest
import torch
import torch.nn.functional as F

from comfyui_propainter_nodes_amd import ops

CASES = [
    # dtype, N, H, W, segC, Cout, k, stride, pad, dil, groups, act
    (torch.float32, 1, 9, 11, [8], 20, 3, 1, 1, 1, 1, None),
    (torch.float32, 2, 9, 11, [8, 36], 70, 3, 2, 1, 1, 1, "leaky"),
    (torch.float32, 1, 7, 13, [4], 130, (1, 5), 1, (0, 2), 1, 1, "sigmoid"),
    (torch.float16, 2, 9, 11, [8, 40], 130, 3, 1, 2, 2, 1, None),
    (torch.float16, 1, 12, 10, [16], 6, 1, 1, 0, 1, 1, "relu"),
    (torch.float16, 1, 12, 10, [16, 8], 24, 3, 1, 1, 1, 2, "tanh"),
    (torch.float16, 1, 20, 21, [128, 128, 8], 40, 3, 1, 1, 1, 1, "leaky"),
    # 96-channel-wide tiles (weight tile padded to whole DMA passes)
    (torch.float32, 1, 14, 13, [16], 180, 3, 1, 1, 1, 1, "relu"),
    (torch.float16, 1, 14, 13, [24], 96, 3, 1, 1, 1, 1, None),
    # 49 taps: the general (per-tap bounds test) gather instead of the 32-bit tap-mask fast path
    (torch.float16, 1, 20, 23, [16], 40, 7, 3, 3, 1, 1, None),
    (torch.float32, 1, 11, 12, [8], 24, 7, 1, 3, 1, 1, "relu"),
    ("f32x2", 1, 11, 12, [8], 24, 7, 1, 3, 1, 1, "relu"),
    # Cout % 256 == 0: the 8-wave 256 x 128 tiles under PP_CONV_TILE=xlforce
    (torch.float16, 1, 13, 12, [32, 8], 256, 3, 1, 1, 1, 1, "leaky"),
    (torch.float32, 1, 9, 10, [8], 256, 3, 1, 1, 1, 1, None),
    ("f32x2", 1, 13, 12, [16, 8], 256, (1, 5), 1, (0, 2), 1, 1, "tanh"),
    (torch.float16, 1, 6, 7, [32], 472, 1, 1, 0, 1, 1, None),      # partial last 256-channel tile
    ("f32x2", 1, 6, 7, [16], 472, 3, 1, 1, 1, 1, "relu"),
    # PP_F32X2: f32 tensors on the f16 matrix pipe (two-term operand split), every tile family
    ("f32x2", 1, 9, 11, [8], 20, 3, 1, 1, 1, 1, None),
    ("f32x2", 2, 9, 11, [8, 36], 70, 3, 2, 1, 1, 1, "leaky"),
    ("f32x2", 1, 7, 13, [4], 130, (1, 5), 1, (0, 2), 1, 1, "sigmoid"),
    ("f32x2", 1, 14, 13, [16], 180, 3, 1, 1, 1, 1, "relu"),
    ("f32x2", 1, 12, 10, [16], 6, 1, 1, 0, 1, 1, "tanh"),
    ("f32x2", 1, 12, 10, [16, 8], 24, 3, 1, 1, 1, 2, None),
    # halo-tile kernel (PP_CONV_HALO=force): several 8x16 output tiles with partial ones in both directions, two images,
    # segments with a partial 32-channel chunk, every tile family (128 / 96 / 64 channels), 3x3 / 1x5 / 5x1 / 2x3 taps
    ("f32x2", 2, 19, 37, [40, 8], 130, 3, 1, 1, 1, 1, "relu"),
    ("f32x2", 1, 17, 33, [36], 192, (1, 5), 1, (0, 2), 1, 1, "sigmoid"),
    ("f32x2", 1, 21, 18, [64, 4], 64, (5, 1), 1, (2, 0), 1, 1, "tanh"),
    ("f32x2", 1, 10, 40, [32], 126, (2, 3), 1, (1, 0), 1, 2, None),
    ("f32x2", 1, 16, 32, [8], 40, 3, 1, 0, 1, 1, "leaky"),
    # taller problems on the halo tiles: 128- and 96-channel tiles, 3x3 / 1x5 / 5x1, several chunks and segments (the
    # weight ring wraps), partial row / column tiles
    ("f32x2", 1, 35, 20, [72], 128, 3, 1, 1, 1, 1, "tanh"),
    ("f32x2", 2, 18, 17, [32, 32], 256, (1, 5), 1, (0, 2), 1, 1, "sigmoid"),
    ("f32x2", 1, 20, 16, [36, 4], 100, (5, 1), 1, (2, 0), 1, 1, None),
    # f16 halo-tile kernel: several tiles, segments with a partial chunk, dilation 2 (the 5-pass pixel stage), 96 / 64 tiles
    (torch.float16, 2, 19, 37, [40, 24], 130, 3, 1, 1, 1, 1, "relu"),
    (torch.float16, 1, 17, 33, [32], 192, 3, 1, 2, 2, 1, "leaky"),
    (torch.float16, 1, 12, 20, [64, 8], 64, (3, 1), 1, (2, 0), (2, 1), 1, None),
    # ... its compile-time 3x3 form (rows of three taps per weight stage): 128- and 64-channel tiles, several chunks / segments
    (torch.float16, 1, 19, 21, [72], 128, 3, 1, 1, 1, 1, "leaky"),
    (torch.float16, 2, 9, 33, [32, 40], 64, 3, 1, 1, 1, 1, None),
    # in-work-group split-K kernel (PP_CONV_KSPLIT=force): reductions of 9 / 27 / 45 / 10 chunks over 4 groups, segments,
    # a last group with fewer (or no) chunks, partial 32-pixel and 128-channel tiles
    (torch.float16, 1, 7, 9, [32], 128, 3, 1, 1, 1, 1, "leaky"),
    (torch.float16, 2, 6, 5, [40, 24, 16], 130, 3, 1, 1, 1, 1, None),
    (torch.float16, 1, 9, 8, [64, 96], 96, 3, 1, 1, 1, 1, "relu"),
    (torch.float16, 1, 5, 13, [320], 140, 1, 1, 0, 1, 1, "tanh"),
    (torch.float16, 1, 8, 8, [16], 72, (1, 5), 1, (0, 2), 1, 1, None),
]


def _ref_input(x, segC, groups):
    if groups == 1:
        return torch.cat([t.float() for t in x], 3).permute(0, 3, 1, 2)
    parts = []
    for g in range(groups):
        for t_, c in zip(x, segC):
            parts.append(t_.float()[..., g * c:(g + 1) * c])
    return torch.cat(parts, 3).permute(0, 3, 1, 2)


# ("xlforce" = the experimental 8-wave 256-channel tiles: emulator only until they have been measured on the MI355X)
@pytest.mark.parametrize("be,tile", [("emu", "large"), ("emu", "small"), ("emu", "xlforce"), ("emu", "tiny"), ("emu", "halo"), ("emu", "halo_rt"),
                                     ("emu", "ksplit"),
                                     pytest.param("hip", "large", marks=pytest.mark.gpu),
                                     pytest.param("hip", "small", marks=pytest.mark.gpu),
                                     pytest.param("hip", "xlforce", marks=pytest.mark.gpu),
                                     pytest.param("hip", "tiny", marks=pytest.mark.gpu),
                                     pytest.param("hip", "halo", marks=pytest.mark.gpu),
                                     pytest.param("hip", "halo_rt", marks=pytest.mark.gpu),
                                     pytest.param("hip", "ksplit", marks=pytest.mark.gpu),
                                     ("emu", "launch_order"), pytest.param("hip", "launch_order", marks=pytest.mark.gpu)])
def test_conv2d_matches_torch(be, tile):
    """Every case on both tile families (128-pixel tiles / 32-pixel tiles for small problems).  The tile
    choice is read once per process (PP_CONV_TILE), so each family runs in a fresh interpreter."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    import torch as _t

    if be == "hip" and not _t.cuda.is_available():
        pytest.skip("no GPU visible")
    root = str(Path(__file__).resolve().parent.parent)
    env = dict(os.environ, PP_CONV_TILE=tile, PP_TEST_BACKEND=be, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env["PP_CONV_HALO"] = "0"      # the flat-tile kernels ...
    env["PP_CONV_KSPLIT"] = "0"
    if tile == "halo":             # ... or the halo-tile kernel for every eligible PP_F32X2 geometry, whatever its size
        env.update(PP_CONV_TILE="large", PP_CONV_HALO="force")
    if tile == "halo_rt":          # ... with the runtime-tap f16 kernels also where a compile-time-tap form exists (PP_F32X2: flat kernel)
        env.update(PP_CONV_TILE="large", PP_CONV_HALO="force", PP_CONV_HALO_CT="0")
    if tile == "launch_order":     # ... flat tiles walked in launch order instead of the XCD-contiguous, channel-adjacent default
        env.update(PP_CONV_TILE="large", PP_CONV_ORDER="launch")
    if tile == "ksplit":           # ... or the in-work-group split-K kernel for every f16 problem with >= 4 chunks
        env.update(PP_CONV_TILE="large", PP_CONV_KSPLIT="force")
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def _random_cases(seed, n):
    """Seeded random geometries: segments, groups, strides, dilations, asymmetric kernels and pads, channel counts that
    leave partial chunks and partial tiles -- the implicit-GEMM gather and K iterator see combinations the hand-written
    cases do not list."""
    import random

    rnd = random.Random(seed)
    out = []
    while len(out) < n:
        dt = rnd.choice([torch.float32, torch.float16, "f32x2"])
        epp = 8 if dt == torch.float16 else 4
        groups = rnd.choice([1, 1, 1, 2])
        nseg = rnd.choice([1, 1, 2, 3])
        segC = [epp * rnd.randint(1, 9) for _ in range(nseg)]
        cout = rnd.choice([3, 16, 24, 40, 64, 70, 96, 130, 180, 256, 300])
        kh, kw = rnd.choice([(1, 1), (3, 3), (1, 5), (5, 1), (3, 2), (5, 5)])
        s_ = rnd.choice([1, 1, 2, 3])
        d = rnd.choice([1, 1, 2])
        ph, pw = rnd.randint(0, d * (kh - 1)), rnd.randint(0, d * (kw - 1))
        N, H, W = rnd.randint(1, 2), rnd.randint(6, 13), rnd.randint(6, 14)
        if (H + 2 * ph - d * (kh - 1) - 1) < 0 or (W + 2 * pw - d * (kw - 1) - 1) < 0:
            continue
        act = rnd.choice([None, "relu", "leaky", "sigmoid", "tanh"])
        out.append((dt, N, H, W, segC, cout, (kh, kw), s_, (ph, pw), d, groups, act))
    return out


def _run_case(backend, case):
    dt, N, H, W, segC, Cout, k, s, p, d, groups, act = case
    split = dt == "f32x2"
    if split:
        dt = torch.float32
    dev = backend
    g = torch.Generator().manual_seed(1234)
    x = [torch.randn(N, H, W, c * groups, generator=g).to(dt) for c in segC]
    if split:  # wide dynamic range: tiny values exercise the scaled low term
        x = [t * torch.logspace(-6, 2, t.shape[-1])[torch.randperm(t.shape[-1], generator=g)] for t in x]
    w = torch.randn(Cout * groups, sum(segC), *((k, k) if isinstance(k, int) else k), generator=g) * 0.1
    b = torch.randn(Cout * groups, generator=g)
    spec = ops.make_conv_spec(w, b, dt, stride=s, padding=p, dilation=d, groups=groups, seg_channels=segC,
                              split=split).to(dev)
    ho, wo = spec.out_hw(H, W)
    # output is a channel slice of a wider buffer: exercises the ldc / slice-view path
    buf = torch.full((N, ho, wo, Cout * groups + 8), 7.0, dtype=dt, device=dev)
    out = buf[..., 4:4 + Cout * groups]
    ops.conv2d(spec, [t.to(dev) for t in x], out, act=act, act_param=0.2)
    # float64 reference; the tolerance of the f32 paths scales with the largest pre-activation (fp32 accumulation of
    # terms that cancel: torch's own f32 convolution is no closer to float64 than that)
    pre = F.conv2d(_ref_input(x, segC, groups).double(), w.to(dt).double(), b.double(), stride=s, padding=p, dilation=d,
                   groups=groups)
    ref = {None: lambda v: v, "leaky": lambda v: F.leaky_relu(v, 0.2), "relu": F.relu,
           "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](pre).permute(0, 2, 3, 1)
    got = out.double().cpu()
    if dt == torch.float32:
        tol = 2e-5 * max(1.0, ref.abs().max().item()) + 5e-7 * pre.abs().max().item()
    else:
        tol = 4e-3 * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() <= tol, ((got - ref).abs().max().item(), tol)
    # untouched pad channels
    assert torch.all(buf[..., :4].float().cpu() == 7.0) and torch.all(buf[..., 4 + Cout * groups:].float().cpu() == 7.0)


@pytest.mark.parametrize("family", ["flat", "halo", "ksplit"])
def test_f16_conv_with_f32_output(backend, family, pp_knobs):
    """f16 operands, fp32 output (the deformable offsets / masks `om`) with a two-activation split, every kernel family."""
    pp_knobs(PP_CONV_HALO="force" if family == "halo" else "0", PP_CONV_KSPLIT="force" if family == "ksplit" else "0")
    g = torch.Generator().manual_seed(21)
    N, H, W, C, Cout = 1, 11, 19, 72, 136
    x = torch.randn(N, H, W, C, generator=g).half()
    w = torch.randn(Cout, C, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    spec = ops.make_conv_spec(w, b, torch.float16, padding=1).to(backend)
    out = torch.empty(N, H, W, Cout, device=backend, dtype=torch.float32)
    ops.conv2d(spec, [x.to(backend)], out, act="tanh", out_scale=5.0, act2="sigmoid", act_split=96)
    pre = F.conv2d(x.float().permute(0, 3, 1, 2).double(), w.half().double(), b.double(), padding=1).permute(0, 2, 3, 1)
    ref = torch.cat([5.0 * torch.tanh(pre[..., :96]), torch.sigmoid(pre[..., 96:])], -1)
    assert (out.double().cpu() - ref).abs().max().item() < 2e-3


@pytest.mark.parametrize("order", ["default", "launch"])
def test_many_channel_tiles(backend, order, pp_knobs):
    """The flat-tile order (conv_common.h: flat_tile_of) on layers with MANY channel tiles: 2 200 and 4 170 output channels
    = 18 and 33 tiles of 128 (the soft-composite's 512 -> 6272 has 49), i.e. whole groups of 16 plus a remainder group, with a
    partial last pixel tile; every output element must be written exactly once, in both orders, f16 and PP_F32X2."""
    if order == "launch":
        pp_knobs(PP_CONV_ORDER="launch")
    g = torch.Generator().manual_seed(31)
    for dt, split, cout, (n, h, w), cin in ((torch.float16, False, 2200, (1, 9, 31), 40), (torch.float32, True, 4170, (2, 5, 13), 12)):
        x = torch.randn(n, h, w, cin, generator=g).to(dt)
        wgt = torch.randn(cout, cin, 1, 1, generator=g) * 0.1
        b = torch.randn(cout, generator=g)
        spec = ops.make_conv_spec(wgt, b, dt, split=split).to(backend)
        out = torch.full((n, h, w, cout), float("nan"), device=backend, dtype=dt)
        ops.conv2d(spec, [x.to(backend)], out)
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), wgt.to(dt).double(), b.double()).permute(0, 2, 3, 1)
        err = (out.double().cpu() - ref).abs().max().item()   # (NaN if a tile was never written)
        assert err <= (4e-3 if dt == torch.float16 else 2e-5) * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("cfg", ["1", "2", "3", "4", "5", "6"])
def test_gemm_kernel_equals_flat_kernel(backend, cfg, pp_knobs):
    """r04: the GEMM kernel of the 1x1 f16 layers (conv_gemm_f16.hip: the transformer's Linear layers) issues the flat kernel's
    MFMAs in the flat kernel's order, so it must reproduce it BIT FOR BIT -- every tile configuration, on shapes that exercise
    the channel tail of the last chunk (C % 32 = 8, the 1960-channel fc2), a partial last stage (odd chunk count with two
    chunks per barrier), pixel / channel tiles that end inside the problem, one chunk only, fp32 output and a fused epilogue."""
    g = torch.Generator().manual_seed(41)
    shapes = [  # N, H, W, C, Cout, out dtype, epilogue
        (1, 13, 23, 104, 200, torch.float16, None),          # 4 chunks (3.25), partial tiles both ways
        (2, 9, 17, 160, 264, torch.float16, "add"),          # 5 chunks: partial last stage when KC = 2
        (1, 6, 11, 24, 136, torch.float32, None),            # one (partial) chunk, fp32 output
        (1, 20, 33, 328, 96, torch.float16, None),           # 10.25 chunks: the ring wraps several times
    ]
    for N, H, W, C, Cout, odt, epi in shapes:
        x = torch.randn(N, H, W, C, generator=g).half().to(backend)
        wgt = torch.randn(Cout, C, 1, 1, generator=g) * 0.1
        b = torch.randn(Cout, generator=g)
        aux = torch.randn(N, H, W, Cout, generator=g).to(odt).to(backend)
        spec = ops.make_conv_spec(wgt, b, torch.float16).to(backend)
        outs = []
        for knobs in (dict(PP_CONV_GEMM="0", PP_CONV_GEMM_CFG="0"), dict(PP_CONV_GEMM="force", PP_CONV_GEMM_CFG=cfg)):
            pp_knobs(PP_CONV_KSPLIT="0", **knobs)
            out = torch.full((N, H, W, Cout), float("nan"), device=backend, dtype=odt)
            ops.conv2d(spec, [x], out, act="gelu", **({"epi": "add", "aux1": aux} if epi else {}))
            outs.append(out.cpu())
        assert torch.equal(outs[0], outs[1]), (N, H, W, C, Cout, (outs[0].float() - outs[1].float()).abs().max())
        ref = F.conv2d(x.cpu().double().permute(0, 3, 1, 2), wgt.half().double(), b.double()).permute(0, 2, 3, 1)
        ref = torch.nn.functional.gelu(ref) + (aux.cpu().double() if epi else 0)
        assert (outs[1].double() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


def test_many_images_hint_skips_the_split_k_kernel(backend, pp_knobs):
    """ABI v12: a small-image / long-K layer (20 x 40 pixels, 36 chunks, 128 channels: the split-K kernel's territory) that its
    caller marks `many_images` runs on the kernels PP_CONV_KSPLIT=0 would select -- bit for bit -- whatever the batch; without the
    mark the split-K kernel sums in four K groups (close, not equal)."""
    g = torch.Generator().manual_seed(71)
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    b = torch.randn(128, generator=g)
    outs = {}
    for n in (2, 5):
        x = torch.randn(n, 20, 40, 128, generator=g).half().to(backend)
        for name, hint, knobs in (("hint", True, {}), ("no ksplit", False, dict(PP_CONV_KSPLIT="0")), ("default", False, {})):
            pp_knobs(PP_CONV_KSPLIT=knobs.get("PP_CONV_KSPLIT", "1"))
            spec = ops.make_conv_spec(w, b, torch.float16, padding=2, dilation=2, many_images=hint).to(backend)
            out = torch.full((n, 20, 40, 128), float("nan"), device=backend, dtype=torch.float16)
            ops.conv2d(spec, [x], out, act="leaky", act_param=0.2)
            outs[name] = out.cpu()
        assert torch.equal(outs["hint"], outs["no ksplit"]), n
        assert not torch.equal(outs["hint"], outs["default"]), "the unmarked layer should have taken the split-K kernel"
        ref = F.leaky_relu(F.conv2d(x.cpu().double().permute(0, 3, 1, 2), w.half().double(), b.double(), padding=2, dilation=2), 0.2)
        for o in outs.values():
            assert (o.double() - ref.permute(0, 2, 3, 1)).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


def test_halo_c64_knob_is_bit_identical(emu_lib, pp_knobs):
    """r04: PP_CONV_HALO_C64=1 sends every 3x3 f16 compile-time-tap halo layer to the 64-channel tile (three work-groups per CU;
    the occupancy A/B of profiles/r04_conv_counters.md, default off until timed).  A tile shape changes which work-group computes
    an output, never the order of its K sum: bit-identical to the default selection for 96- and 128-channel-tile layers, partial
    channel tiles, two input segments, f16 and fp32 outputs, a fused epilogue.  (Emulator only: the knob was added after the round's
    last GPU call; tools/gpu_r5_first.sh times it and checks the whole step's parity with it on the MI355X.)"""
    backend = torch.device("cpu")
    g = torch.Generator().manual_seed(67)
    for N, H, W, segC, Cout, odt, epi in ((1, 17, 35, [64], 128, torch.float16, None), (2, 9, 20, [32, 40], 192, torch.float16, "add"),
                                          (1, 16, 16, [96], 200, torch.float32, None)):
        x = [torch.randn(N, H, W, c, generator=g).half().to(backend) for c in segC]
        w = torch.randn(Cout, sum(segC), 3, 3, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        aux = torch.randn(N, H, W, Cout, generator=g).to(odt).to(backend)
        spec = ops.make_conv_spec(w, b, torch.float16, padding=1, seg_channels=segC).to(backend)
        outs = []
        for c64 in ("0", "1"):
            pp_knobs(PP_CONV_HALO="force", PP_CONV_KSPLIT="0", PP_CONV_HALO_C64=c64)
            out = torch.full((N, H, W, Cout), float("nan"), device=backend, dtype=odt)
            ops.conv2d(spec, x, out, act="relu", **({"epi": "add", "aux1": aux} if epi else {}))
            outs.append(out.cpu())
        assert torch.equal(outs[0], outs[1]), (Cout, (outs[0].float() - outs[1].float()).abs().max())
        xin = torch.cat([t.cpu() for t in x], -1).double().permute(0, 3, 1, 2)
        ref = torch.relu(F.conv2d(xin, w.half().double(), b.double(), padding=1)).permute(0, 2, 3, 1) + (aux.cpu().double() if epi else 0)
        assert (outs[1].double() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


def test_linear_of_unfold_equals_unfold_then_linear(backend, pp_knobs):
    """r04 (ABI v8, pp_conv2d_params.flat_taps): the Linear over F.unfold()'s tap-major patch vectors with the patches gathered
    inside the GEMM kernel must equal unfold (a copy) + the same Linear BIT FOR BIT -- the FusionFeedForward's fc2 on the folded
    40-channel map (7x7 / stride 3 / padding 3: patches that hang over every image border, K = 1960 with a partial last chunk),
    wide and narrow output tiles, fp32 output, and a residual epilogue.  (The comparison Linear is pinned to the flat / GEMM
    kernels, which walk the chunks in order like the patch kernel does; the split-K kernel these tiny problems would select sums
    in four groups.)"""
    pp_knobs(PP_CONV_KSPLIT="0")
    g = torch.Generator().manual_seed(51)
    for N, H, W, C, Cout, odt in ((2, 13, 17, 40, 264, torch.float16), (1, 9, 22, 40, 100, torch.float32), (3, 7, 7, 16, 520, torch.float16)):
        fh, fw = (H + 6 - 7) // 3 + 1, (W + 6 - 7) // 3 + 1
        x = torch.randn(N, H, W, C, generator=g).half().to(backend)
        wgt = torch.randn(Cout, 49 * C, 1, 1, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        res = torch.randn(N, fh, fw, Cout, generator=g).to(odt).to(backend)
        spec = ops.make_conv_spec(wgt, b, torch.float16).to(backend)
        un = torch.empty(N, fh * fw, 49 * C, dtype=torch.float16, device=backend)
        ops.unfold_gelu(x, un, fh, fw, pre_activated=True)          # tap-major patch matrix (copy only)
        want = torch.full((N, fh, fw, Cout), float("nan"), device=backend, dtype=odt)
        ops.conv2d(spec, [un.view(N, fh, fw, 49 * C)], want, epi="add", aux1=res)
        got = torch.full((N, fh, fw, Cout), float("nan"), device=backend, dtype=odt)
        ops.linear_of_unfold(spec, x, got, 7, 3, 3, epi="add", aux1=res)
        assert torch.equal(got.cpu(), want.cpu()), (N, H, W, C, Cout, (got.float() - want.float()).abs().max())
        ref = F.unfold(x.cpu().float().permute(0, 3, 1, 2), kernel_size=7, stride=3, padding=3)          # [N, C*49, L], c-major
        ref = ref.view(N, C, 49, fh * fw).permute(0, 3, 2, 1).reshape(N, fh * fw, 49 * C).double()      # tap-major
        ref = (ref @ wgt.half().double().view(Cout, -1).t() + b.double()).view(N, fh, fw, Cout) + res.cpu().double()
        assert (got.cpu().double() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_f16_kernels_are_bit_stable_next_to_a_busy_stream(hip_lib, pp_knobs):
    """r04 regression: conv_halo_f16_ct_kernel restaged a weight buffer one barrier after LDS reads hipcc had let slip below that
    barrier (pp_device.h: pp_barrier now retires the wave's LDS reads first).  Alone on the chip the race never showed; with
    another stream keeping the LDS queues busy ~0.2 % of the launches computed with stale weight rows (found when r04 put RAFT
    of the next sub-video under the flow completion of the current one).  Here the 3x3 convolutions of the flow-completion step run
    on a side stream next to a large PP_F32X2 convolution: every launch must reproduce the quiet result bit for bit."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    B, h, w = 2, 45, 80
    x = torch.randn(B, h, w, 128, device=dev, generator=g).half()
    big = torch.randn(64, 90, 160, 128, device=dev)
    lspec = ops.make_conv_spec(torch.randn(128, 128, 3, 3) * 0.05, torch.zeros(128), torch.float32, padding=1, split=True).to(dev)
    lout = torch.empty(64, 90, 160, 128, device=dev)
    side = torch.cuda.Stream(dev)
    pp_knobs(PP_CONV_KSPLIT="0", PP_CONV_HALO="force")           # the halo-tile kernels also for the 128 -> 128 layer
    for cout, odt in ((432, torch.float32), (128, torch.float16)):
        spec = ops.make_conv_spec(torch.randn(cout, 128, 3, 3) * 0.05, torch.randn(cout), torch.float16, padding=1).to(dev)
        ref = torch.empty(B, h, w, cout, device=dev, dtype=odt)
        ops.conv2d(spec, [x], ref, act="leaky", act_param=0.1)
        torch.cuda.synchronize()
        bad = 0
        for rep in range(300):
            out = torch.full_like(ref, float("nan"))
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                for _ in range(5):
                    ops.conv2d(spec, [x], out, act="leaky", act_param=0.1)
            ops.conv2d(lspec, [big], lout)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            bad += 0 if torch.equal(out, ref) else 1
        assert bad == 0, f"3x3 128->{cout}: {bad} of 300 runs next to a busy stream differ from the quiet run"


def test_small_cout_on_mfma_tiles(backend, pp_knobs):
    """r04: 3x3 f16 layers with at most 4 output channels and at least two channel chunks (the generator's 64 -> 3 output layer) run on
    16-channel tiles of the compile-time-tap halo kernel (conv_halo_f16.hip: launch_halo_f16_small_cout) instead of the vector-ALU
    kernel: against torch, f16 and fp32 output, into a wider output view (the node's [.., 4] buffer: the padding channel must stay
    untouched), partial tiles both ways; PP_CONV_SMALL_HALO=0 and single-chunk layers keep the vector-ALU kernel."""
    g = torch.Generator().manual_seed(61)
    for small_halo in ("1", "0"):
        pp_knobs(PP_CONV_SMALL_HALO=small_halo, PP_CONV_HALO="force", PP_CONV_DIRECT="force")
        for N, H, W, C, Cout, odt in ((2, 19, 37, 64, 3, torch.float16), (1, 9, 33, 40, 2, torch.float32), (1, 16, 16, 32, 4, torch.float16)):
            x = torch.randn(N, H, W, C, generator=g).half()
            w = torch.randn(Cout, C, 3, 3, generator=g) * 0.05
            b = torch.randn(Cout, generator=g)
            spec = ops.make_conv_spec(w, b, torch.float16, padding=1).to(backend)
            out = torch.full((N, H, W, 4), float("nan"), dtype=odt, device=backend)
            ops.conv2d(spec, [x.to(backend)], out[..., :Cout], act="tanh")
            ref = torch.tanh(F.conv2d(x.float().permute(0, 3, 1, 2).double(), w.half().double(), b.double(), padding=1)).permute(0, 2, 3, 1)
            got = out.cpu()
            assert (got[..., :Cout].double() - ref).abs().max().item() < (2e-3 if odt == torch.float16 else 1e-5)
            assert Cout == 4 or bool(torch.isnan(got[..., Cout:]).all())


@pytest.mark.parametrize("halo", ["0", "force"])
def test_epilogue_from_a_channel(backend, halo, pp_knobs):
    """`epi_from` (ABI v6): RAFT's GRU computes the z and r gates (update.py:41-43) in ONE 256-channel PP_F32X2 convolution
    over [h | motion] -- sigmoid on all channels, r * h (PP_EPI_MUL_AUX1) on channels 128..255 only, read at channel
    c - 128 of `h` -- into one buffer whose halves the q convolution then uses as views.  1x5 taps, a context pre-addend,
    partial tiles; flat and halo kernel families; plus the GRU blend epilogue from channel 64 on."""
    pp_knobs(PP_CONV_HALO=halo)
    g = torch.Generator().manual_seed(23)
    N, H, W = 2, 9, 21
    h = torch.randn(N, H, W, 128, generator=g)
    mf = torch.randn(N, H, W, 128, generator=g)
    w = torch.randn(256, 256, 1, 5, generator=g) * 0.03
    b = torch.randn(256, generator=g) * 0.1
    pre = torch.randn(N, H, W, 256, generator=g) * 0.2
    spec = ops.make_conv_spec(w, b, torch.float32, padding=(0, 2), seg_channels=[128, 128], split=True).to(backend)
    zr = torch.empty(N, H, W, 256, device=backend)
    ops.conv2d(spec, [h.to(backend), mf.to(backend)], zr, act="sigmoid", epi="mul", aux1=h.to(backend), epi_from=128,
               pre_add=pre.to(backend))
    x = torch.cat([h, mf], -1).permute(0, 3, 1, 2).double()
    lin = F.conv2d(x, w.double(), b.double(), padding=(0, 2)).permute(0, 2, 3, 1) + pre.double()
    ref = torch.sigmoid(lin)
    ref[..., 128:] = ref[..., 128:] * h.double()
    assert (zr.double().cpu() - ref).abs().max().item() < 2e-5
    # GRU blend from channel 64 on: y = (1 - z) * h + z * v for c >= 64, plain tanh below
    z = torch.rand(N, H, W, 64, generator=g)
    hh = torch.randn(N, H, W, 64, generator=g)
    w2 = torch.randn(128, 256, 1, 5, generator=g) * 0.03
    spec2 = ops.make_conv_spec(w2, None, torch.float32, padding=(0, 2), seg_channels=[128, 128], split=True).to(backend)
    out = torch.empty(N, H, W, 128, device=backend)
    ops.conv2d(spec2, [h.to(backend), mf.to(backend)], out, act="tanh", epi="gru", aux1=z.to(backend), aux2=hh.to(backend), epi_from=64)
    v = torch.tanh(F.conv2d(x, w2.double(), None, padding=(0, 2)).permute(0, 2, 3, 1))
    ref2 = v.clone()
    ref2[..., 64:] = (1 - z.double()) * hh.double() + z.double() * v[..., 64:]
    assert (out.double().cpu() - ref2).abs().max().item() < 2e-5
    with pytest.raises(RuntimeError):
        ops.conv2d(spec2, [h.to(backend), mf.to(backend)], out, act="tanh", epi="gru", aux1=z.to(backend), aux2=hh.to(backend), epi_from=62)


@pytest.mark.parametrize("halo", ["0", "force"])
def test_f32x2_operand_range(backend, halo, pp_knobs):
    """PP_F32X2 at the edge of the f16 range (silent failure for |v| >= 32752).  Both terms saturate (r05: the
    low term is the unscaled remainder, f16_rtz(v - h)): inputs up to 65504 stay within fp32-GEMM-like accuracy (absolute operand
    error <= 0.016), larger inputs saturate at +-131008 (h = l = 65504) -- the result is finite, never Inf/NaN.  Both kernel
    families (flat 128-pixel tiles / halo tiles)."""
    pp_knobs(PP_CONV_HALO=halo)
    g = torch.Generator().manual_seed(5)
    N, H, W, C, Cout = 1, 9, 17, 8, 40
    x = torch.randn(N, H, W, C, generator=g)
    x[0, 2, 3, 0], x[0, 4, 5, 1], x[0, 6, 7, 2], x[0, 1, 9, 3] = 32751.9, 40007.77, -65503.99, 32783.97
    w = torch.randn(Cout, C, 3, 3, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    spec = ops.make_conv_spec(w, b, torch.float32, padding=1, split=True).to(backend)
    out = torch.empty(N, H, W, Cout, device=backend)
    ops.conv2d(spec, [x.to(backend)], out)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    err = (out.double().cpu() - ref).abs().max().item()
    # operand error <= 0.016 per large input, |w| <= ~0.5, at most a few large inputs under one window
    assert err <= 0.05 + 2e-5 * ref.abs().max().item(), err
    x2 = x.clone()
    x2[0, 3, 3, 0], x2[0, 5, 5, 1] = 1.0e6, -3.0e38
    ops.conv2d(spec, [x2.to(backend)], out)
    assert torch.isfinite(out).all()
    sat = x2.clamp(-131008.0, 131008.0)
    ref2 = F.conv2d(sat.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    # (beyond the f16 range the low term is as large as the high one, so the dropped wl * xl product is 2^-11 of those terms)
    assert (out.double().cpu() - ref2).abs().max().item() <= 0.05 + 6e-4 * ref2.abs().max().item()


@pytest.mark.parametrize("tile,seed", [("large", 11), ("small", 12), ("xlforce", 13), ("tiny", 14), ("halo", 15), ("ksplit", 16)])
def test_conv2d_random_geometries_under_emulation(tile, seed):
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parent.parent)
    env = dict(os.environ, PP_CONV_TILE=tile, PP_TEST_BACKEND="emu", PP_CONV_RANDOM=str(seed), PP_CONV_HALO="0",
               PP_CONV_KSPLIT="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if tile == "halo":
        env.update(PP_CONV_TILE="large", PP_CONV_HALO="force")
    if tile == "ksplit":
        env.update(PP_CONV_TILE="large", PP_CONV_KSPLIT="force")
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_conv2d_random_epilogues(emu_lib):
    """(Emulator only this round: not yet run on the MI355X; switch the fixture to `backend` once it has been.)
    Seeded random epilogue configurations: channel counts that leave partial quads, output / aux / pre_add views at
    channel offsets that break the 16-byte alignment of the vector path, every fused op, two-activation splits that
    straddle a quad, out_scale -- against float64 torch."""
    import os
    import random

    dev = torch.device("cpu")
    rnd = random.Random(int(os.environ.get("PP_EPI_FUZZ_SEED", "2024")))
    g = torch.Generator().manual_seed(99)
    acts = {None: lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.2), "sigmoid": torch.sigmoid, "tanh": torch.tanh}
    ncase = int(os.environ.get("PP_EPI_FUZZ_N", "24" if dev.type == "cpu" else "60"))
    for _ in range(ncase):
        mode = rnd.choice(["f32", "f16", "f32x2"])
        dt = torch.float16 if mode == "f16" else torch.float32
        cin = (8 if mode == "f16" else 4) * rnd.randint(1, 5)
        cout = rnd.choice([3, 5, 16, 22, 37, 64, 70, 129])
        H, W = rnd.randint(4, 9), rnd.randint(4, 11)
        x = torch.randn(1, H, W, cin, generator=g).to(dt)
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
        b = torch.randn(cout, generator=g) if rnd.random() < 0.8 else None
        spec = ops.make_conv_spec(w, b, dt, padding=1, split=(mode == "f32x2")).to(dev)
        odt = dt

        def view(off, extra):  # a channel slice of a wider buffer: ldc = cout + extra, first channel at `off`
            buf = torch.full((1, H, W, cout + extra), 3.0, dtype=odt)
            return buf, buf[..., off:off + cout]

        ooff, oext = rnd.choice([(0, 0), (4, 8), (1, 3), (2, 5), (0, 4)])
        obuf, _ = view(ooff, oext)
        obuf = obuf.to(dev)
        out = obuf[..., ooff:ooff + cout]
        act = rnd.choice(list(acts))
        act2, split = None, 0
        if rnd.random() < 0.35 and cout > 8:
            act2, split = rnd.choice(["sigmoid", "relu", "tanh"]), rnd.randint(1, cout - 1)
        scale = rnd.choice([0.0, 0.0, 5.0, 0.25])
        epi = rnd.choice([None, None, "mul", "add", "add_relu", "gru"])
        a1 = a2 = pre = None
        aoff, aext = rnd.choice([(0, 0), (4, 4), (3, 5)])
        if epi:
            a1 = torch.rand(1, H, W, cout + aext, generator=g).to(odt)[..., aoff:aoff + cout]
            if epi == "gru":
                a2 = torch.randn(1, H, W, cout + aext, generator=g).to(odt)[..., aoff:aoff + cout]
        if rnd.random() < 0.4:
            pre = torch.randn(1, H, W, cout + aext, generator=g).to(odt)[..., aoff:aoff + cout]
        ops.conv2d(spec, [x.to(dev)], out, act=act, act_param=0.2, act2=act2, act_split=split, out_scale=scale, epi=epi,
                   aux1=None if a1 is None else a1.to(dev), aux2=None if a2 is None else a2.to(dev),
                   pre_add=None if pre is None else pre.to(dev))
        v = F.conv2d(x.double().permute(0, 3, 1, 2), w.to(dt).double(), None if b is None else b.double(), padding=1).permute(0, 2, 3, 1)
        if pre is not None:
            v = v + pre.double()
        if split > 0:
            lo = acts[act](v[..., :split])
            lo = lo * scale if scale != 0.0 else lo
            v = torch.cat([lo, acts[act2](v[..., split:])], -1)
        else:
            v = acts[act](v)
            v = v * scale if scale != 0.0 else v
        if epi == "mul":
            v = v * a1.double()
        elif epi == "add":
            v = v + a1.double()
        elif epi == "add_relu":
            v = F.relu(v + a1.double())
        elif epi == "gru":
            v = (1 - a1.double()) * a2.double() + a1.double() * v
        got = out.double().cpu()
        tol = (3e-5 if dt == torch.float32 else 6e-3) * max(1.0, v.abs().max().item())
        cfg = (mode, cin, cout, act, act2, split, scale, epi, ooff, oext, aoff, aext, pre is not None)
        assert (got - v).abs().max().item() <= tol, (cfg, (got - v).abs().max().item())
        full = obuf.double().cpu()
        assert torch.all(full[..., :ooff] == 3.0) and torch.all(full[..., ooff + cout:] == 3.0), cfg  # neighbours untouched


def test_conv2d_epilogues(backend):
    dev = backend
    g = torch.Generator().manual_seed(7)
    dt = torch.float32
    x = torch.randn(1, 6, 10, 16, generator=g)
    w = torch.randn(32, 16, 3, 3, generator=g) * 0.1
    b = torch.randn(32, generator=g)
    a1 = torch.rand(1, 6, 10, 32, generator=g)
    a2 = torch.randn(1, 6, 10, 32, generator=g)
    spec = ops.make_conv_spec(w, b, dt, padding=1).to(dev)
    base = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    out = torch.empty(1, 6, 10, 32, device=dev)
    xs = [x.to(dev)]
    ops.conv2d(spec, xs, out, act="sigmoid", epi="mul", aux1=a1.to(dev))
    assert torch.allclose(out.cpu(), torch.sigmoid(base) * a1, atol=1e-5)
    ops.conv2d(spec, xs, out, act="relu", epi="add_relu", aux1=a2.to(dev))
    assert torch.allclose(out.cpu(), F.relu(F.relu(base) + a2), atol=1e-5)
    ops.conv2d(spec, xs, out, act="tanh", epi="gru", aux1=a1.to(dev), aux2=a2.to(dev))
    assert torch.allclose(out.cpu(), (1 - a1) * a2 + a1 * torch.tanh(base), atol=1e-5)
    ops.conv2d(spec, xs, out, act="tanh", act2="sigmoid", act_split=20, out_scale=5.0)
    ref = torch.cat([5 * torch.tanh(base[..., :20]), torch.sigmoid(base[..., 20:])], -1)
    assert torch.allclose(out.cpu(), ref, atol=1e-5)


def test_conv2d_replicate_pad_and_batched_gemm(backend):
    dev = backend
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 9, 8, generator=g)
    w = torch.randn(16, 8, 5, 5, generator=g) * 0.1
    spec = ops.make_conv_spec(w, None, torch.float32, stride=2, padding=2, pad_mode="replicate").to(dev)
    out = torch.empty(2, *spec.out_hw(8, 9), 16, device=dev)
    ops.conv2d(spec, [x.to(dev)], out)
    ref = F.conv2d(F.pad(x.permute(0, 3, 1, 2), (2, 2, 2, 2), mode="replicate"), w, stride=2).permute(0, 2, 3, 1)
    assert torch.allclose(out.cpu(), ref, atol=1e-5)
    # batched GEMM: the RAFT all-pairs volume form (corr.py:52-60): out[b,p1,p2] = <f1[b,p1], f2[b,p2]>
    f1 = torch.randn(3, 1, 40, 32, generator=g)
    f2 = torch.randn(3, 40, 32, generator=g)
    vol = torch.empty(3, 1, 40, 40, device=dev)
    ops.batched_gemm_nt(f1.to(dev), f2.to(dev), vol, scale=1.0 / 16)
    ref = torch.einsum("bpc,bqc->bpq", f1[:, 0], f2) / 16
    assert torch.allclose(vol.cpu()[:, 0], ref, atol=1e-5)
    # ... with PP_F32X2 products: f2 is packed on the device into the layout the host gives constant weights
    # (32 h | 32 l per chunk; h rounded toward zero here, to nearest on the host: both reconstruct v to 2^-22)
    f1w, f2w = f1 * torch.logspace(-3, 2, 32), f2 * torch.logspace(2, -3, 32)   # wide dynamic range
    pk = ops.split_pack(f2w.contiguous().to(dev)).cpu().view(torch.float16).view(3, 40, 64).float()
    # the operand contract of the r05 split (pp_device.h: split_pair): |v - h - l| <= max(2^-20 |v|, 2^-24)
    eps = lambda v: torch.maximum(v.abs() * 2.0 ** -20, torch.full_like(v, 2.0 ** -24))
    assert (((pk[..., :32] + pk[..., 32:]) - f2w).abs() <= eps(f2w)).all()
    ops.batched_gemm_nt(f1w.contiguous().to(dev), f2w.contiguous().to(dev), vol, scale=1.0 / 16, split=True)
    a, b = f1w[:, 0].double(), f2w.double()
    ref = torch.einsum("bpc,bqc->bpq", a, b) / 16
    # products: |a| eps(b) + |b| eps(a) + the dropped low x low term + fp32 accumulation of 32 terms
    bound = (torch.einsum("bpc,bqc->bpq", a.abs(), eps(b)) + torch.einsum("bpc,bqc->bpq", eps(a), b.abs())) * 1.01 / 16 \
        + 4e-6 * torch.einsum("bpc,bqc->bpq", a.abs(), b.abs()) / 16
    assert ((vol.cpu()[:, 0].double() - ref).abs() <= bound).all()
    f1n, f2n = f1.contiguous(), f2.contiguous()        # operands of one magnitude (the all-pairs volume's case): fp32-GEMM-like
    ops.batched_gemm_nt(f1n.to(dev), f2n.to(dev), vol, scale=1.0 / 16, split=True)
    ref = torch.einsum("bpc,bqc->bpq", f1n[:, 0].double(), f2n.double()) / 16
    assert (vol.cpu()[:, 0].double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()


def test_parameter_block_cache_keys_on_the_layer_not_on_the_object(emu_lib):
    """r05 (ops.conv2d caches the filled pp_conv2d_params block of a launch): two layers that share EVERY buffer -- weights, bias,
    input, output -- and differ in geometry only must not share a block.  (The first key carried id(spec) + the weight address: a
    test that builds layers one after the other got the same id and the same addresses back for another geometry, on the MI355X.)"""
    import dataclasses

    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 9, 10, 32, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
    b = torch.randn(32, generator=g)
    spec = ops.make_conv_spec(w, b, torch.float32, padding=1, split=True)
    out = torch.empty(1, 9, 10, 32)
    ops.conv2d(spec, [x], out)
    zeros = out.clone()
    twin = dataclasses.replace(spec, pad_mode="replicate", geometry_key=None)      # same tensors, other padding
    ops.conv2d(twin, [x], out)
    ref = F.conv2d(F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="replicate"), w, b).permute(0, 2, 3, 1)
    assert (out - ref).abs().max().item() < 1e-4 and (out - zeros).abs().max().item() > 1e-2
    ops.conv2d(spec, [x], out)
    assert torch.equal(out, zeros)


if __name__ == "__main__":  # child process of test_conv2d_matches_torch
    import os
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    sys.path.insert(0, str(Path(__file__).resolve().parent / "emu"))
    from comfyui_propainter_nodes_amd import lib

    if os.environ["PP_TEST_BACKEND"] == "emu":
        import emu_loader

        emu_loader.load_emulator()
        dev = torch.device("cpu")
    else:
        lib.load()
        dev = torch.device("cuda:0")
    cases = CASES
    if os.environ.get("PP_CONV_RANDOM"):
        cases = _random_cases(int(os.environ["PP_CONV_RANDOM"]), 14)
    for i, case in enumerate(cases):
        try:
            _run_case(dev, case)
        except Exception:
            print("FAILED CASE", case, flush=True)
            raise
        print("case", i, "ok", flush=True)


def test_direct_small_cout_kernel(backend, pp_knobs):
    """conv_direct.hip (at most 4 output channels, fp32 FMAs on the vector ALU) against float64 torch: the three layers it
    exists for (RAFT flow head 256 -> 2 with the in-place `coords += delta` epilogue, generator output 64 -> 3 tanh into a
    channel view, flow-completion output 32 -> 2) at sizes with partial 16 x 16 tiles and a partial 32-channel chunk, every
    weight packing (PP_F32X2 / f32 / f16), both output types, then seeded random epilogues."""
    import random

    pp_knobs(PP_CONV_DIRECT="force")
    dev = backend
    g = torch.Generator().manual_seed(31)
    acts = {None: lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.2), "sigmoid": torch.sigmoid, "tanh": torch.tanh}

    # RAFT flow head: f32 tensors, PP_F32X2 and exact packings, delta added onto the flow in place
    # (r06, ADVICE r05: also WITHOUT the optional fp32 weight table -- PP_CONV_DIRECT_TABLE=0, or a C-ABI caller that leaves
    #  weight_f32 NULL: the kernel then decodes the packed weights itself, for PP_F32X2 the ABI v9 form h + l with acc_scale)
    for split, table in ((True, True), (False, True), (True, False), (False, False)):
        x = torch.randn(2, 19, 37, 256, generator=g)
        w = torch.randn(2, 256, 3, 3, generator=g) * 0.05
        b = torch.randn(2, generator=g)
        pp_knobs(PP_CONV_DIRECT_TABLE="1" if table else "0")
        spec = ops.make_conv_spec(w, b, torch.float32, padding=1, split=split).to(dev)
        assert (spec.weight_f32 is not None) == table
        pp_knobs(PP_CONV_DIRECT_TABLE="1")
        flow0 = torch.randn(2, 19, 37, 2, generator=g)
        flow = flow0.clone().to(dev)
        ops.conv2d(spec, [x.to(dev)], flow, epi="add", aux1=flow)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + flow0.double()
        assert (flow.double().cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())

    # f16 layers: 64 -> 3 tanh into [..., 0:3] of a 4-channel buffer; 40 -> 2 (partial chunk), f16 and f32 outputs; 1x3 taps
    for cin, cout, k, pad, act, odt in ((64, 3, (3, 3), (1, 1), "tanh", torch.float16), (40, 2, (3, 3), (1, 1), None, torch.float16),
                                        (40, 2, (3, 3), (1, 1), "leaky", torch.float32), (32, 4, (1, 3), (0, 1), "relu", torch.float16)):
        x = torch.randn(1, 21, 34, cin, generator=g).half()
        w = torch.randn(cout, cin, *k, generator=g) * 0.1
        b = torch.randn(cout, generator=g)
        spec = ops.make_conv_spec(w, b, torch.float16, padding=pad).to(dev)
        buf = torch.full((1, 21, 34, 4), 9.0, dtype=odt, device=dev)
        ops.conv2d(spec, [x.to(dev)], buf[..., 0:cout], act=act, act_param=0.2)
        ref = acts[act](F.conv2d(x.double().permute(0, 3, 1, 2), w.half().double(), b.double(), padding=pad)).permute(0, 2, 3, 1)
        tol = (4e-3 if odt == torch.float16 else 2e-5) * max(1.0, ref.abs().max().item())
        assert (buf[..., 0:cout].double().cpu() - ref).abs().max().item() <= tol, (cin, cout, act, odt)
        assert torch.all(buf[..., cout:].float().cpu() == 9.0)

    # random epilogues on the direct path
    rnd = random.Random(77)
    for _ in range(10):
        cout = rnd.randint(1, 4)
        cin = 4 * rnd.randint(1, 12)
        H, W = rnd.randint(5, 20), rnd.randint(5, 20)
        x = torch.randn(1, H, W, cin, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
        b = torch.randn(cout, generator=g) if rnd.random() < 0.8 else None
        spec = ops.make_conv_spec(w, b, torch.float32, padding=1, split=rnd.random() < 0.5).to(dev)
        ext = rnd.choice([0, 1, 4])
        obuf = torch.full((1, H, W, cout + ext), 3.0, device=dev)
        act = rnd.choice(list(acts))
        act2, asplit = (rnd.choice(["sigmoid", "relu"]), rnd.randint(1, cout - 1)) if cout > 1 and rnd.random() < 0.4 else (None, 0)
        scale = rnd.choice([0.0, 5.0, 0.25])
        epi = rnd.choice([None, "mul", "add", "add_relu", "gru"])
        a1 = torch.rand(1, H, W, cout, generator=g) if epi else None
        a2 = torch.randn(1, H, W, cout, generator=g) if epi == "gru" else None
        pre = torch.randn(1, H, W, cout, generator=g) if rnd.random() < 0.4 else None
        ops.conv2d(spec, [x.to(dev)], obuf[..., :cout], act=act, act_param=0.2, act2=act2, act_split=asplit, out_scale=scale, epi=epi,
                   aux1=None if a1 is None else a1.to(dev), aux2=None if a2 is None else a2.to(dev),
                   pre_add=None if pre is None else pre.to(dev))
        v = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None if b is None else b.double(), padding=1).permute(0, 2, 3, 1)
        if pre is not None:
            v = v + pre.double()
        if asplit > 0:
            lo = acts[act](v[..., :asplit])
            v = torch.cat([lo * scale if scale != 0.0 else lo, acts[act2](v[..., asplit:])], -1)
        else:
            v = acts[act](v)
            v = v * scale if scale != 0.0 else v
        if epi == "mul":
            v = v * a1.double()
        elif epi == "add":
            v = v + a1.double()
        elif epi == "add_relu":
            v = F.relu(v + a1.double())
        elif epi == "gru":
            v = (1 - a1.double()) * a2.double() + a1.double() * v
        cfg = (cin, cout, act, act2, asplit, scale, epi, ext, pre is not None)
        assert (obuf[..., :cout].double().cpu() - v).abs().max().item() <= 3e-5 * max(1.0, v.abs().max().item()), cfg
        assert torch.all(obuf[..., cout:].cpu() == 3.0), cfg


def test_patch_conv_f32x2(backend):
    """r06, conv_patch.hip (pp_conv2d, PP_F32X2 + flat_taps): RAFT's 7x7 convolutions on the 2-channel flow (a channel VIEW of the
    128-channel motion buffer: pitch 128, 8-byte aligned base) and on the 3-channel frames (stride 2, 64 channels), partial 8 x 16
    tiles both ways, two images, against float64 torch and against the pp_im2col + 1x1 form it replaces (same weights, same
    three-product arithmetic: equal to fp32 summation-order noise); then a fused epilogue (relu, in-place add) and the refusals."""
    dev = backend
    g = torch.Generator().manual_seed(5)
    for (n, h, w, c, cout, k, stride, pad, act, view) in ((2, 19, 37, 2, 128, 7, 1, 3, "relu", True), (2, 29, 41, 3, 64, 7, 2, 3, None, False),
                                                          (1, 8, 16, 2, 128, 7, 1, 3, None, False), (1, 21, 18, 4, 64, 5, 1, 2, "leaky", False),
                                                          (1, 17, 23, 1, 128, 3, 2, 1, "relu", False)):
        x = torch.randn(n, h, w, c, generator=g) * 3.0
        wt = torch.randn(cout, c, k, k, generator=g) * 0.1
        b = torch.randn(cout, generator=g)
        kv = k * k * c
        kpad = ops.pad32(kv)
        spec = ops.make_conv_spec(wt.permute(0, 2, 3, 1).reshape(cout, kv, 1, 1), b, torch.float32, seg_channels=[kpad],
                                  seg_valid=[kv], split=True).to(dev)
        if view:   # the flow as RAFT holds it: channels 126..127 of the motion buffer
            buf = torch.zeros(n, h, w, 128)
            buf[..., 126:128] = x
            xd = buf.to(dev)[..., 126:128]
        else:
            xd = x.to(dev)
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        out = torch.full((n, ho, wo, cout), 7.0, device=dev)
        ops.conv2d_patch(spec, xd, out, k, k, stride=stride, padding=pad, act=act, act_param=0.2)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
        ref = {None: lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.2)}[act](ref)
        err = (out.double().cpu() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (n, h, w, c, cout, err)
        cols = torch.empty(n, ho, wo, kpad, device=dev)
        ops.im2col(xd, cols, k, k, stride=stride, padding=pad)
        old = torch.empty(n, ho, wo, cout, device=dev)
        ops.conv2d(spec, [cols], old, act=act, act_param=0.2)
        assert (out - old).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    # fused epilogue on the patch path: out = relu(conv + bias) + aux
    aux = torch.randn(n, ho, wo, cout, generator=g)
    out2 = torch.empty(n, ho, wo, cout, device=dev)
    ops.conv2d_patch(spec, xd, out2, k, k, stride=stride, padding=pad, act="relu", epi="add", aux1=aux.to(dev))
    assert (out2.double().cpu() - (ref + aux.double())).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # refused: more than 4 input channels, Cout other than 64 / 128
    x5 = torch.randn(1, 9, 9, 5, generator=g).to(dev)
    sp5 = ops.make_conv_spec(torch.randn(64, 45, 1, 1), None, torch.float32, seg_channels=[64], seg_valid=[45], split=True).to(dev)
    with pytest.raises(RuntimeError, match="flat_taps"):
        ops.conv2d_patch(sp5, x5, torch.empty(1, 9, 9, 64, device=dev), 3, 3, padding=1)
    sp32 = ops.make_conv_spec(torch.randn(32, 18, 1, 1), None, torch.float32, seg_channels=[32], seg_valid=[18], split=True).to(dev)
    with pytest.raises(RuntimeError, match="flat_taps"):
        ops.conv2d_patch(sp32, x5[..., :2], torch.empty(1, 9, 9, 32, device=dev), 3, 3, padding=1)
