"""Node-level drop-in behaviour on the MI355X: the two ComfyUI node methods, called the way ComfyUI
calls them (keyword arguments named like the INPUT_TYPES), against the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

from comfyui_propainter_nodes_amd import image_utils, nodes, pipeline, synth, weights
from oracle import pipeline as OP


def _oracle(frames_u8, fm, md, sds, **kw):
    frames = (torch.from_numpy(frames_u8).float().div(255) * 2 - 1).permute(0, 3, 1, 2)[None]
    out = OP.run(sds, frames, torch.from_numpy(fm).float()[None, :, None], torch.from_numpy(md).float()[None, :, None],
                 [f for f in frames_u8], **kw)
    return np.stack(out, 0)


def _psnr(a, b):
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean())
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


@pytest.fixture()
def seeded_models(hip_lib, monkeypatch):
    """The nodes load weights through pipeline.initialize_models; pin it to the seeded synthetic weights."""
    sds = weights.synth_state_dicts(0)
    dev = torch.device("cuda:0")
    models = pipeline.models_from_state_dicts(sds, dev)
    monkeypatch.setattr(nodes, "initialize_models", lambda device, fp16: models)
    return sds


@pytest.mark.gpu
def test_inpaint_node_with_resize_matches_oracle(seeded_models):
    T = 5
    image, mask = synth.synthetic_clip(T, 150, 170)            # input size != process size -> PIL bicubic resize path
    node = nodes.ProPainterInpaint()
    kw = dict(raft_iter=2, neighbor_length=4, ref_stride=2, subvideo_length=80)
    out_img, flow_mask, mask_dil = getattr(node, node.FUNCTION)(image=image, mask=mask, width=144, height=128, mask_dilates=3,
                                                                flow_mask_dilates=4, fp16="enable", **kw)
    assert out_img.shape == (T, 128, 144, 3) and out_img.dtype == torch.float32
    assert flow_mask.shape == (T, 128, 144) and mask_dil.shape == (T, 128, 144) and flow_mask.is_cuda
    assert set(torch.unique(mask_dil).tolist()) <= {0.0, 1.0}
    icfg = image_utils.ImageConfig(144, 128, 3, 4, (170, 150), T)
    fr, fm, md = image_utils.prepare_frames_and_masks(image_utils.image_to_uint8_frames(image), mask, icfg)
    ref = _oracle(fr, fm, md, seeded_models, **kw)
    got = (out_img.numpy() * 255 + 0.5).astype(np.uint8)
    assert _psnr(got, ref) >= 40.0
    assert np.array_equal(mask_dil.cpu().numpy().astype(np.uint8), md)


@pytest.mark.gpu
def test_streamed_output_equals_blocking_output(seeded_models, monkeypatch):
    """The node streams finished frame ranges to the host under the remaining windows (nodes._HostImageSink): same IMAGE,
    bit for bit, as the blocking copy + conversion after the last kernel, also when the clip spans several windows and the
    pinned staging buffer is reused by a second call."""
    T = 13
    image, mask = synth.synthetic_clip(T, 128, 144)
    node = nodes.ProPainterInpaint()
    kw = dict(image=image, mask=mask, width=144, height=128, mask_dilates=3, flow_mask_dilates=4, fp16="enable", raft_iter=2,
              neighbor_length=4, ref_stride=3, subvideo_length=80)
    outs = {}
    for mode in ("host", "device", "stream", "stream"):
        monkeypatch.setenv("PP_OUTPUT", mode)
        outs.setdefault(mode, []).append(getattr(node, node.FUNCTION)(**kw)[0].clone())
    assert outs["host"][0].dtype == torch.float32 and not outs["stream"][0].is_cuda
    assert torch.equal(outs["host"][0], outs["stream"][0]) and torch.equal(outs["host"][0], outs["stream"][1])
    assert torch.equal(outs["host"][0], outs["device"][0])    # float32(k) / 255 on the GPU (the r03 default) is the same division


@pytest.mark.gpu
def test_outpaint_node_matches_oracle(seeded_models):
    T = 4
    image, _ = synth.synthetic_clip(T, 128, 128)
    node = nodes.ProPainterOutpaint()
    kw = dict(raft_iter=2, neighbor_length=4, ref_stride=2, subvideo_length=80)
    out_img, out_mask, w, h = getattr(node, node.FUNCTION)(image=image, width=128, height=128, width_scale=1.3, height_scale=1.0,
                                                          mask_dilates=5, flow_mask_dilates=8, fp16="enable", **kw)
    assert (w, h) == (160, 128) and out_img.shape == (T, 128, 160, 3) and out_mask.shape == (T, 128, 160)
    ocfg = image_utils.ImageOutpaintConfig(128, 128, 5, 8, (128, 128), T, 1.3, 1.0)
    fr, fm, md = image_utils.extrapolation(image_utils.image_to_uint8_frames(image), ocfg)
    ref = _oracle(fr, fm, md, seeded_models, **kw)
    got = (out_img.numpy() * 255 + 0.5).astype(np.uint8)
    assert _psnr(got, ref) >= 40.0
    # the pasted input region is returned untouched
    assert np.array_equal(got[:, :, 16:144], image_utils.image_to_uint8_frames(image))


@pytest.mark.gpu
def test_pretrained_weights_psnr_when_available(hip_lib):
    """Auto-enables when the three ProPainter checkpoints are present in weights/ (none are shipped offline)."""
    if not weights.weights_available():
        pytest.skip("no pretrained checkpoints in weights/ (no network in this environment)")
    sds = weights.load_state_dicts()
    T, H, W = 8, 240, 432
    image, mask = synth.synthetic_clip(T, H, W)
    icfg = image_utils.ImageConfig(W, H, 5, 8, (W, H), T)
    fr, fm, md = image_utils.prepare_frames_and_masks(image_utils.image_to_uint8_frames(image), mask, icfg)
    dev = torch.device("cuda:0")
    cfg = pipeline.ProPainterConfig(10, 10, 80, 20, "enable", T, dev, (W, H))
    got = pipeline.run_inpainting(pipeline.models_from_state_dicts(sds, dev), fr, fm, md, cfg).numpy()
    ref = _oracle(fr, fm, md, sds, raft_iter=20, neighbor_length=10, ref_stride=10, subvideo_length=80)
    assert _psnr(got, ref) >= 40.0
