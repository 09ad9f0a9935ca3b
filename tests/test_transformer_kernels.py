"""Transformer-side kernels against the oracle's torch formulation: f16 storage (fp16 "enable": tol 4e-3 rel) and fp32
storage (fp16 "disable": fp32 rounding noise, the attention core 2e-3 for its f16 MFMA operands); fp32 math in both."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from comfyui_propainter_nodes_amd import ops, weights
from oracle import generator as OG

H16 = torch.float16


def _close(got, ref, tol=4e-3):
    if got.dtype == torch.float32 and tol == 4e-3:
        tol = 2e-6
    err = (got.float().cpu() - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err


DTYPES = pytest.mark.parametrize("dt", [torch.float16, torch.float32], ids=["f16", "f32"])


@DTYPES
def test_layernorm_into_padded_grid(backend, dt):
    dev = backend
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(2, 6, 8, 512, generator=g) * 2 + 0.5).to(dt)
    gam, bet = 1 + 0.1 * torch.randn(512, generator=g), 0.1 * torch.randn(512, generator=g)
    out = torch.zeros(2, 10, 9, 512, dtype=dt, device=dev)
    ops.layernorm(x.to(dev), out, gam.to(dev), bet.to(dev))
    ref = F.layer_norm(x.float(), (512,), gam, bet)
    _close(out[:, :6, :8], ref)
    assert torch.all(out[:, 6:].float().cpu() == 0) and torch.all(out[:, :, 8:].float().cpu() == 0)


@DTYPES
def test_pool_tokens(backend, dt):
    dev = backend
    g = torch.Generator().manual_seed(32)
    x = torch.randn(2, 8, 12, 64, generator=g).to(dt)
    w = torch.randn(64, 1, 4, 4, generator=g) * 0.2
    b = torch.randn(64, generator=g) * 0.1
    out = torch.empty(2, 2, 3, 64, dtype=dt, device=dev)
    ops.pool_tokens(x.to(dev), out, w.view(64, 16).t().contiguous().to(dev), b.to(dev))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, b, stride=4, groups=64).permute(0, 2, 3, 1)
    _close(out, ref)


@DTYPES
def test_fold_unfold_roundtrip_against_torch(backend, dt):
    dev = backend
    g = torch.Generator().manual_seed(33)
    T, h, w, C = 2, 13, 17, 16
    fh, fw = (h + 6 - 7) // 3 + 1, (w + 6 - 7) // 3 + 1
    tok = torch.randn(T, fh * fw, C * 49, generator=g).to(dt)  # torch order: c*49 + tap
    tap_major = tok.view(T, fh * fw, C, 49).permute(0, 1, 3, 2).reshape(T, fh * fw, 49 * C).contiguous()
    kw = dict(output_size=(h, w), kernel_size=7, stride=3, padding=3)
    for normalize in (True, False):
        out = torch.empty(T, h, w, C, dtype=dt, device=dev)
        ops.fold(tap_major.to(dev), out, fh, fw, normalize)
        ref = F.fold(tok.float().permute(0, 2, 1), **kw)
        if normalize:
            ref = ref / F.fold(torch.ones(1, 49, fh * fw), **kw)
        _close(out, ref.permute(0, 2, 3, 1))
    x = torch.randn(T, h, w, C, generator=g).to(dt)
    un = torch.empty(T, fh * fw, 49 * C, dtype=dt, device=dev)
    ops.unfold_gelu(x.to(dev), un, fh, fw)
    ref = F.gelu(F.unfold(x.float().permute(0, 3, 1, 2), kernel_size=7, stride=3, padding=3)).permute(0, 2, 1)
    ref = ref.view(T, fh * fw, C, 49).permute(0, 1, 3, 2).reshape(T, fh * fw, 49 * C)
    _close(un, ref)
    # r04 (ABI v8): the GELU once per folded value in pp_fold + a copy-only unfold must equal fold + unfold_gelu BIT FOR BIT
    # (the fused-feed-forward path of the generator, sparse_transformer.py:413-433)
    folded, folded_g = (torch.empty(T, h, w, C, dtype=dt, device=dev) for _ in range(2))
    ops.fold(tap_major.to(dev), folded, fh, fw, True)
    two_step = torch.empty(T, fh * fw, 49 * C, dtype=dt, device=dev)
    ops.unfold_gelu(folded, two_step, fh, fw)
    ops.fold(tap_major.to(dev), folded_g, fh, fw, True, gelu=True)
    fused = torch.empty(T, fh * fw, 49 * C, dtype=dt, device=dev)
    ops.unfold_gelu(folded_g, fused, fh, fw, pre_activated=True)
    assert torch.equal(fused.cpu(), two_step.cpu())


@pytest.mark.parametrize("fh,fw,t,lt,dt", [(11, 12, 4, 3, H16), (5, 18, 4, 3, H16), (30, 54, 18, 11, H16), (15, 27, 7, 5, H16),
                                            (11, 12, 4, 3, torch.float32)])
def test_window_attention_matches_oracle(backend, fh, fw, t, lt, dt):
    """Masked + unmasked windows, window padding, circular rolled neighbours and pooled tokens.  (30, 54, 18, 11) is the
    BASELINE.json configs[1] geometry (640x360: 6x6 windows, 91 pooled keys, 18 frames, 810 queries per masked window:
    several 128-query blocks with a ragged last one) -- MI355X only, the emulator would need minutes; (15, 27, 7, 5) has
    315 queries per masked window: full and partial 128-query blocks under the emulator."""
    dev = backend
    if backend.type == "cpu" and fh * fw * t > 5000:
        pytest.skip("BASELINE-size case runs on the MI355X only")
    g = torch.Generator().manual_seed(34)
    sd = weights.synth_state_dicts(1)["gen"]
    pre = "transformers.transformer.0.attention."
    x = torch.randn(1, t, fh, fw, 512, generator=g).half().float()  # "LayerNorm-ed" tokens
    mask = torch.zeros(1, lt, fh, fw, 1)
    mask[0, 1, 1:3, 2:5] = 1  # only the first window (and maybe its neighbour) is masked
    t_ind = torch.arange(1, t, 2)
    with torch.no_grad():
        ref = OG.window_attention(sd, pre, x, mask, t_ind)
    # product path: padded grid, fused qkv / pooled kv GEMMs, fused attention, proj
    Hp, Wp = math.ceil(fh / 5) * 5, math.ceil(fw / 9) * 9
    xn = torch.zeros(t, Hp, Wp, 512, dtype=dt, device=dev)
    xn[:, :fh, :fw] = x[0].to(dt).to(dev)

    def lin(names):
        w = torch.cat([sd[pre + n + ".weight"] for n in names], 0)
        b = torch.cat([sd[pre + n + ".bias"] for n in names], 0)
        return ops.make_conv_spec(w.reshape(w.shape[0], 512, 1, 1), b, dt, split=dt == torch.float32).to(dev)

    qkv = torch.empty(t, Hp, Wp, 1536, dtype=dt, device=dev)
    ops.conv2d(lin(["query", "key", "value"]), [xn], qkv)
    pooled = torch.empty(t, Hp // 4, Wp // 4, 512, dtype=dt, device=dev)
    ops.pool_tokens(xn, pooled, sd[pre + "pool_layer.weight"].view(512, 16).t().contiguous().to(dev),
                    sd[pre + "pool_layer.bias"].to(dev))
    pkv = torch.empty(t, Hp // 4, Wp // 4, 1024, dtype=dt, device=dev)
    ops.conv2d(lin(["key", "value"]), [pooled], pkv)
    pm = F.pad(mask[0, :, :, :, 0], (0, Wp - fw, 0, Hp - fh))
    flags = (F.max_pool2d(pm, (5, 9), (5, 9)).sum(0) > 0).flatten().to(torch.int32)
    assert 0 < int(flags.sum()) < flags.numel(), "test must cover both window kinds"
    att = torch.empty(t, fh, fw, 512, dtype=dt, device=dev)
    ops.window_attention(qkv, pkv.view(t, -1, 1024), flags.to(dev), t_ind.to(torch.int32).to(dev), att)
    out = torch.empty(t, fh, fw, 512, dtype=dt, device=dev)
    ops.conv2d(lin(["proj"]), [att], out)
    _close(out, ref[0], tol=6e-3 if dt == H16 else 2e-3)


@DTYPES
def test_compose_u8_bit_exact(backend, dt):
    """uint8 compose incl. the order-dependent 0.5/0.5 blend with truncation (propainter_inference.py:283-307)."""
    from oracle import pipeline as OP

    dev = backend
    g = torch.Generator().manual_seed(35)
    T, H, W = 5, 12, 16
    orig = torch.randint(0, 256, (T, H, W, 3), generator=g, dtype=torch.uint8)
    masks = (torch.rand(T, H, W, generator=g) > 0.5).to(torch.uint8)
    comp = torch.zeros(T, H, W, 3, dtype=torch.uint8, device=dev)
    ref_comp = [None] * T
    seen = [False] * T
    for nb in ([0, 1, 2], [1, 2, 3, 4], [2, 3, 4]):
        pred = (torch.rand(len(nb), H, W, 4, generator=g) * 2 - 1).half().to(dt)   # f16-representable values in both modes
        ids = torch.tensor(nb, dtype=torch.int32)
        first = torch.tensor([0 if seen[i] else 1 for i in nb], dtype=torch.int32)
        ops.compose_u8(pred.to(dev), ids.to(dev), first.to(dev), masks.to(dev), orig.to(dev), comp)
        for i in nb:
            seen[i] = True
        OP.compose_window(ref_comp, pred[..., :3].float().permute(0, 3, 1, 2), masks.float()[None, :, None],
                          [o.numpy() for o in orig], nb)
    assert np.array_equal(comp.cpu().numpy(), np.stack(ref_comp, 0))


def test_static_mask_window_flags_are_one_constant_of_the_clip(backend):
    """with one MASK frame replicated over the clip (or the border planes of an outpaint canvas) the
    'window is masked' flags (sparse_transformer.py:321-326) do not depend on the window's local frames; the generator
    computes them once per clip / geometry (generator.window_mask_flags) -- here: that constant equals the per-window
    computation for every (first frame, count), and differs for a moving mask."""
    dev = backend
    g = torch.Generator().manual_seed(36)
    fh, fw, T = 11, 20, 9
    plane = (torch.rand(fh, fw, generator=g) > 0.93).to(torch.uint8)
    static = plane[None].expand(T, -1, -1).contiguous().to(dev)
    const = ops.window_flags(static, 0, 1, (5, 9)).cpu()
    assert 0 < int(const.sum()) < const.numel()
    for g0, lt in ((0, 5), (3, 6), (8, 1), (2, 7)):
        assert torch.equal(ops.window_flags(static, g0, lt, (5, 9)).cpu(), const)
    moving = static.clone()
    moving[4:] = 0
    assert not torch.equal(ops.window_flags(moving, 4, 3, (5, 9)).cpu(), const)


def _attention_reference(qkv, pkv, flags, t_ind, fh, fw):
    """Brute-force torch fp32 restatement of the key sets of sparse_transformer.py:218-385 on given q|k|v tensors
    (qkv [t,Hp,Wp,1536], pkv [t,npool,1024] as the kernel sees them): masked window = all queries of the window against, per
    key frame, the window's 45 tokens + the 148 rolled neighbours + all pooled tokens; unmasked = per frame 45 x 45."""
    t, Hp, Wp, _ = qkv.shape
    out = torch.zeros(t, fh, fw, 512)
    q_all, k_all, v_all = qkv[..., :512].float(), qkv[..., 512:1024].float(), qkv[..., 1024:].float()
    rows = [r - 3 for r in range(5)] + [r + 3 for r in range(5)]
    cols = [c - 5 for c in range(9)] + [c + 5 for c in range(9)]
    nww = Wp // 9
    for win in range((Hp // 5) * nww):
        wi, wj = win // nww, win % nww
        ys, xs = slice(5 * wi, 5 * wi + 5), slice(9 * wj, 9 * wj + 9)
        for head in range(4):
            hs = slice(128 * head, 128 * head + 128)
            if int(flags[win]):
                q = q_all[:, ys, xs, hs].reshape(-1, 128)
                ks, vs = [], []
                for fr in t_ind.tolist():
                    ks.append(k_all[fr, ys, xs, hs].reshape(-1, 128)); vs.append(v_all[fr, ys, xs, hs].reshape(-1, 128))
                    for dr in rows:
                        for dc in cols:
                            if 0 <= dr < 5 and 0 <= dc < 9:
                                continue
                            y, x = (5 * wi + dr) % Hp, (9 * wj + dc) % Wp
                            ks.append(k_all[fr, y, x, hs][None]); vs.append(v_all[fr, y, x, hs][None])
                    ks.append(pkv[fr, :, hs].float()); vs.append(pkv[fr, :, 512 + 128 * head:512 + 128 * head + 128].float())
                K, V = torch.cat(ks), torch.cat(vs)
                o = (torch.softmax(q @ K.t() / math.sqrt(128), -1) @ V).view(t, 5, 9, 128)
            else:
                q = q_all[:, ys, xs, hs].reshape(t, 45, 128)
                K = k_all[:, ys, xs, hs].reshape(t, 45, 128)
                V = v_all[:, ys, xs, hs].reshape(t, 45, 128)
                o = (torch.softmax(q @ K.transpose(1, 2) / math.sqrt(128), -1) @ V).view(t, 5, 9, 128)
            y1, x1 = min(5 * wi + 5, fh), min(9 * wj + 9, fw)
            out[:, 5 * wi:y1, 9 * wj:x1, hs] = o[:, :y1 - 5 * wi, :x1 - 9 * wj]
    return out


def test_window_attention_deferred_rescale_is_exact_on_score_spikes(backend):
    """The f16 kernel rescales O / l only when a row maximum of a wave grows by more than 2^8 in the exponent (deferred
    rescale, csrc/window_attention.hip).  Random data almost never takes that branch after the first tile, so this test
    forces it: single keys in LATE tiles (spatial, rolled-neighbour and pooled ones, in both key frames) are aligned with
    single queries so that their raw scores exceed everything before them by far more than the threshold, other rows of the
    same wave stay ordinary, and some spikes stay just BELOW the threshold (probabilities up to 2^8 against the old
    reference).  Compared with a brute-force fp32 softmax over the same q|k|v."""
    dev = backend
    g = torch.Generator().manual_seed(41)
    t, fh, fw = 4, 10, 18                          # 2 x 2 windows; npool = 2 * 4 = 8; nk = 2 * (193 + 8) = 402: 13 tiles
    qkv = (torch.randn(t, fh, fw, 1536, generator=g) * 0.6).half()
    pkv = (torch.randn(t, 8, 1024, generator=g) * 0.6).half()
    flags = torch.tensor([1, 0, 1, 1], dtype=torch.int32)
    t_ind = torch.arange(1, t, 2, dtype=torch.int32)

    def spike(kvec_setter, q_pos, head, gain):
        tq, yq, xq = q_pos
        q = qkv[tq, yq, xq, 128 * head:128 * head + 128].float()
        kvec_setter((gain * q / q.norm()).half())

    # window 0 (rows 0..4, cols 0..8), head 0: a huge spike on a rolled-neighbour key of key frame 3 (late tiles)
    spike(lambda v: qkv[3, 6, 10].__setitem__(slice(512, 640), v), (2, 1, 3), 0, 60.0)
    # ... a pooled key of key frame 1 for another query of the same wave, just below the threshold (2^8 ~ raw score gap 63)
    spike(lambda v: pkv[1, 5].__setitem__(slice(0, 128), v), (2, 2, 4), 0, 9.0)
    # window 2 (rows 5..9, cols 0..8), head 3: a spike on one of the window's own tokens in key frame 3
    spike(lambda v: qkv[3, 7, 2].__setitem__(slice(512 + 384, 512 + 512), v), (0, 9, 8), 3, 45.0)
    # window 3, head 1: two spikes for the same query in different tiles, the later one larger (two rescales)
    spike(lambda v: qkv[1, 9, 17].__setitem__(slice(512 + 128, 512 + 256), v), (3, 6, 12), 1, 30.0)
    spike(lambda v: pkv[3, 7].__setitem__(slice(128, 256), v), (3, 6, 12), 1, 70.0)
    ref = _attention_reference(qkv, pkv, flags, t_ind, fh, fw)
    out = torch.empty(t, fh, fw, 512, dtype=torch.float16, device=dev)
    ops.window_attention(qkv.to(dev), pkv.to(dev), flags.to(dev), t_ind.to(dev), out)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), err
    # the spiked queries put (almost) all their weight on the spiked key: the output row is that key's value row
    assert (out[2, 1, 3, :128].float().cpu() - qkv[3, 6, 10, 1024:1152].float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("fh,fw,tiles", [(20, 18, 14), (20, 36, 15)])
def test_window_attention_every_remainder_of_the_tile_loop(backend, fh, fw, tiles):
    """The f16 kernel walks the keys four 32-key steps per loop trip (static ring slots) and finishes with 0..3 remainder
    steps plus a masked last tile at ring position (tiles - 1) % 4.  The other attention cases all have tiles % 4 == 1 or 0;
    these two grids give 14 and 15 tiles (remainders 1 and 2) -- against the brute-force fp32 softmax over the same q|k|v."""
    dev = backend
    g = torch.Generator().manual_seed(43)
    t = 4
    npool = (fh // 4) * (fw // 4)
    assert -(-(2 * (193 + npool)) // 32) == tiles
    qkv = (torch.randn(t, fh, fw, 1536, generator=g) * 0.7).half()
    pkv = (torch.randn(t, npool, 1024, generator=g) * 0.7).half()
    nwin = (fh // 5) * (fw // 9)
    flags = torch.zeros(nwin, dtype=torch.int32)
    flags[1] = flags[nwin - 2] = 1
    t_ind = torch.arange(0, t, 2, dtype=torch.int32)
    ref = _attention_reference(qkv, pkv, flags, t_ind, fh, fw)
    out = torch.empty(t, fh, fw, 512, dtype=torch.float16, device=dev)
    ops.window_attention(qkv.to(dev), pkv.to(dev), flags.to(dev), t_ind.to(dev), out)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 4e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("fh,fw,t", [(10, 18, 4), (15, 27, 7)])
def test_window_attention_exact_fp32_core(backend, fh, fw, t):
    """ABI v11 `exact` (fp32 storage): q, k, v and the probabilities stay fp32, both products on v_mfma_f32_16x16x4_f32 -- the
    reference's fp32 attention (sparse_transformer.py:366-393 under fp16 "disable") at fp32 rounding level.  fp32 inputs that are
    NOT f16-representable (the default core rounds them: its error on the same data is two orders larger), masked and unmasked
    windows, partial 128-query blocks ((15, 27, 7): 315 queries), a ragged last key tile, one score spike in a late tile."""
    dev = backend
    g = torch.Generator().manual_seed(47)
    Hp, Wp = math.ceil(fh / 5) * 5, math.ceil(fw / 9) * 9
    npool = (Hp // 4) * (Wp // 4)
    qkv = torch.zeros(t, Hp, Wp, 1536)
    qkv[:, :fh, :fw] = torch.randn(t, fh, fw, 1536, generator=g) * 0.7
    pkv = torch.randn(t, npool, 1024, generator=g) * 0.7
    nwin = (Hp // 5) * (Wp // 9)
    flags = torch.zeros(nwin, dtype=torch.int32)
    flags[0] = flags[nwin - 1] = 1
    t_ind = torch.arange(1, t, 2, dtype=torch.int32)
    q = qkv[2, 1, 3, :128].clone()
    pkv[int(t_ind[-1]), npool - 1, :128] = 25.0 * q / q.norm()       # a late pooled key that dominates one query's row
    ref = _attention_reference(qkv, pkv, flags, t_ind, fh, fw)
    scale = max(1.0, ref.abs().max().item())
    out = torch.empty(t, fh, fw, 512, dtype=torch.float32, device=dev)
    ops.window_attention(qkv.to(dev), pkv.to(dev), flags.to(dev), t_ind.to(dev), out, exact=True)
    err = (out.cpu() - ref).abs().max().item()
    assert err < 1e-5 * scale, err
    rounded = torch.empty_like(out)
    ops.window_attention(qkv.to(dev), pkv.to(dev), flags.to(dev), t_ind.to(dev), rounded, exact=False)
    err16 = (rounded.cpu() - ref).abs().max().item()
    assert 20 * err < err16 < 4e-3 * scale, (err, err16)             # the flag selects a different, much tighter core
    # fp16 storage has no exact form
    with pytest.raises(Exception):
        ops.window_attention(qkv.half().to(dev), pkv.half().to(dev), flags.to(dev), t_ind.to(dev),
                             torch.empty(t, fh, fw, 512, dtype=torch.float16, device=dev), exact=True)
