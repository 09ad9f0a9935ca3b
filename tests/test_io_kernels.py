"""Device-side pre / post-processing (csrc/io_kernels.hip) against the reference-minted host fixtures
(tests/golden/host_cases.npz: arrays dumped from the reference's own utils/image_utils.py functions) and against
scipy / numpy restatements.  Everything here is integer / byte work: bit-exact."""
from pathlib import Path

import numpy as np
import pytest
import scipy.ndimage
import torch

from comfyui_propainter_nodes_amd import image_utils, ops

GOLD = np.load(Path(__file__).parent / "golden" / "host_cases.npz")


def test_frames_from_image_matches_reference_conversion(backend):
    img = torch.from_numpy(GOLD["frames_in"])        # values outside [0,1] exercise the clip
    u8, f32 = ops.frames_from_image(img.to(backend).contiguous())
    assert np.array_equal(u8.cpu().numpy(), GOLD["frames_u8"])
    want = (torch.from_numpy(GOLD["frames_u8"]).float().div(255) * 2 - 1)
    assert torch.equal(f32.cpu(), want)
    assert torch.equal(ops.frames_from_u8(u8).cpu(), want)


def test_outpaint_canvas_on_device(backend):
    """ops.frames_from_image with a canvas == extrapolation's canvas (image_utils.py:200-252), no-resize case."""
    img = torch.from_numpy(GOLD["frames_in"])
    T, H, W, _ = img.shape
    cfg = image_utils.ImageOutpaintConfig(W - W % 8, H - H % 8, 5, 8, (W - W % 8, H - H % 8), T, 1.6, 1.9)
    crop = img[:, :H - H % 8, :W - W % 8].contiguous()
    (pw, ph), (hs, ws), fmask, mask = image_utils.outpaint_geometry(cfg)
    u8, f32 = ops.frames_from_image(crop.to(backend), (ph, pw), (hs, ws))
    canvas, fms, mds = image_utils.extrapolation(image_utils.image_to_uint8_frames(crop), cfg)
    assert np.array_equal(u8.cpu().numpy(), canvas)
    assert np.array_equal(fms[0], fmask) and np.array_equal(mds[0], mask)
    assert torch.equal(f32.cpu(), torch.from_numpy(canvas).float().div(255) * 2 - 1)


@pytest.mark.parametrize("case", [0, 2])
def test_mask_dilate_matches_reference_read_masks(backend, case):
    """Reference-minted (read_masks, image_utils.py:142-175) flow / dilated masks for the no-resize cases."""
    m = torch.from_numpy(GOLD[f"mask_in_{case}"]).float().contiguous()
    w, h, md, fmd, T = [int(v) for v in GOLD[f"mask_par_{case}"]]
    assert (w, h) == (m.shape[2], m.shape[1])
    got_f = ops.mask_dilate(m.to(backend), fmd).cpu().numpy()
    got_d = ops.mask_dilate(m.to(backend), md).cpu().numpy()
    want_f, want_d = GOLD[f"mask_flow_{case}"], GOLD[f"mask_dil_{case}"]
    if m.shape[0] == 1:
        got_f, got_d = np.repeat(got_f, T, 0), np.repeat(got_d, T, 0)
    assert np.array_equal(got_f, want_f) and np.array_equal(got_d, want_d)


@pytest.mark.parametrize("k", [0, 1, 5, 8, 31])
def test_mask_dilate_matches_scipy(backend, k):
    g = torch.Generator().manual_seed(7 + k)
    m = (torch.rand(3, 37, 53, generator=g) > 0.985).to(torch.uint8) * 255
    m[1, 0, 0] = 7          # corner, small non-zero value
    m[2, -1, -1] = 1
    got = ops.mask_dilate(m.contiguous().to(backend), k).cpu().numpy()
    for i in range(3):
        a = m[i].numpy()
        want = scipy.ndimage.binary_dilation(a, iterations=k).astype(np.uint8) if k > 0 else (a > 0.1).astype(np.uint8)
        assert np.array_equal(got[i], want), (k, i)
    # float MASK input: trunc(clamp(m*255)) decides "non-zero" (0.003*255 < 1 -> zero, 0.004*255 > 1 -> set)
    mf = torch.zeros(1, 16, 16)
    mf[0, 3, 3], mf[0, 10, 10], mf[0, 12, 2] = 0.003, 0.004, -4.0
    got = ops.mask_dilate(mf.to(backend), 1).cpu().numpy()[0]
    want = scipy.ndimage.binary_dilation((mf[0] * 255).clamp(0, 255).byte().numpy(), iterations=1).astype(np.uint8)
    assert np.array_equal(got, want) and got.sum() == 5


def test_image_from_u8(backend):
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (2, 5, 8, 3), generator=g, dtype=torch.uint8)
    got = ops.image_from_u8(u8.to(backend)).cpu()
    assert torch.equal(got, torch.from_numpy(u8.numpy().astype(np.float32) / 255.0))


def test_clip_masks_and_window_flags(backend):
    """maskpair / token masks / window flags against torch restatements of propainter.py:409-428 and
    sparse_transformer.py:212-216,321-326."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(11)
    T, H, W = 5, 128, 144
    m_in = (torch.rand(T, H, W, generator=g) > 0.9995).to(torch.uint8)
    m_in[2, 40:60, 100:130] = 1
    m_upd = (torch.rand(T, H, W, generator=g) > 0.7).to(torch.uint8)
    h, w = H // 4, W // 4
    fh, fw = (h + 6 - 7) // 3 + 1, (w + 6 - 7) // 3 + 1
    mp, tok = ops.clip_masks(m_in.to(backend), m_upd.to(backend), fh, fw)
    assert torch.equal(mp.cpu()[..., 0].float(), m_in[:, ::4, ::4].float())
    assert torch.equal(mp.cpu()[..., 1].float(), m_upd[:, ::4, ::4].float())
    assert float(mp.cpu()[..., 2:].abs().max()) == 0
    want_tok = F.max_pool2d(m_in[:, ::4, ::4].float().unsqueeze(1), 7, 3, 3)[:, 0] > 0
    assert torch.equal(tok.cpu().bool(), want_tok)
    for g0, lt in ((0, 5), (1, 2), (3, 1)):
        flags = ops.window_flags(tok, g0, lt, (5, 9)).cpu()
        tm = want_tok[g0:g0 + lt].any(0)
        Hp, Wp = -(-fh // 5) * 5, -(-fw // 9) * 9
        pad = torch.zeros(Hp, Wp, dtype=torch.bool)
        pad[:fh, :fw] = tm
        want = pad.view(Hp // 5, 5, Wp // 9, 9).permute(0, 2, 1, 3).reshape(-1, 45).any(1)
        assert torch.equal(flags.bool(), want)
