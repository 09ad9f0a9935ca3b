"""The oracle pin: oracle/ must reproduce the fixtures minted from the reference itself
(tests/golden/make_golden.py ran the reference on CPU fp32 and dumped these tensors)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from comfyui_propainter_nodes_amd import weights
from oracle import pipeline as OP

GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("case", ["e2e_small", "e2e_chunked"])
def test_oracle_reproduces_reference_fixture(case):
    g = np.load(GOLD / f"{case}.npz")
    T, H, W, iters, nl, rs, sv, _, _, seed = [int(v) for v in g["params"]]
    sds = weights.synth_state_dicts(seed)
    torch.set_num_threads(8)
    frames_u8 = g["frames_u8"]
    frames = (torch.from_numpy(frames_u8).float().div(255) * 2 - 1).permute(0, 3, 1, 2)[None]
    fm = torch.from_numpy(g["flow_masks"]).float()[None, :, None]
    md = torch.from_numpy(g["masks_dilated"]).float()[None, :, None]
    comp, tr = OP.run(sds, frames, fm, md, [f for f in frames_u8], raft_iter=iters, neighbor_length=nl, ref_stride=rs,
                      subvideo_length=sv, return_trace=True)
    assert torch.allclose(tr["gt_flows"][0][0], torch.from_numpy(g["gt_flow_f"]), atol=1e-5)
    assert torch.allclose(tr["gt_flows"][1][0], torch.from_numpy(g["gt_flow_b"]), atol=1e-5)
    assert torch.allclose(tr["pred_flows"][0][0], torch.from_numpy(g["pred_flow_f"]).float(), atol=8e-3)  # f16 fixture
    assert torch.equal(tr["updated_masks"][0, :, 0], torch.from_numpy(g["updated_masks"]).float())
    assert torch.allclose(tr["updated_frames"][0], torch.from_numpy(g["updated_frames"]).float(), atol=1e-3)
    pi = torch.cat(tr["pred_imgs"], 0)
    assert torch.allclose(pi, torch.from_numpy(g["pred_imgs"]).float(), atol=2e-3)
    out = np.stack(comp, 0)
    assert np.abs(out.astype(np.int32) - g["out_image"].astype(np.int32)).max() <= 1


@pytest.mark.parametrize("case", __import__("node_case").EDGE)
def test_oracle_reproduces_reference_node_fixture_at_the_parameter_corners(case):
    """The oracle on the 15 edge-case fixtures minted through the reference's NODE methods (2-frame clip, 1-frame sub-videos,
    neighbor_length 300, empty / full masks, outpaint scales ...): RAFT flows, completed flows, updated masks and the composed
    frames inside the mask as the reference produced them.  The prepared frames / masks come from this repo's host plumbing,
    whose masks are first checked against the fixture's bit for bit."""
    import json

    from comfyui_propainter_nodes_amd import image_utils
    from node_case import _unpack, clip_of

    g = np.load(GOLD / f"{case}.npz")
    P = json.loads(str(g["params_json"]))
    image, mask = clip_of(P)
    u8 = image_utils.image_to_uint8_frames(image)
    T = P["T"]
    if str(g["kind"]) == "inpaint":
        cfg = image_utils.ImageConfig(P["width"], P["height"], P["mask_dilates"], P["flow_mask_dilates"], (P["W"], P["H"]), T)
        frames_u8, fm, md = image_utils.prepare_frames_and_masks(u8, mask, cfg)
    else:
        cfg = image_utils.ImageOutpaintConfig(P["width"], P["height"], P["mask_dilates"], P["flow_mask_dilates"], (P["W"], P["H"]), T,
                                              P["width_scale"], P["height_scale"])
        frames_u8, fm, md = image_utils.extrapolation(u8, cfg)
        fm, md = np.array(np.broadcast_to(fm, (T, *fm.shape[1:]))), np.array(np.broadcast_to(md, (T, *md.shape[1:])))
    h, w = [int(v) for v in g["hw"]]
    assert np.array_equal(fm, _unpack(g["flow_masks"], (T, h, w))) and np.array_equal(md, _unpack(g["masks_dilated"], (T, h, w)))
    frames = (torch.from_numpy(np.ascontiguousarray(frames_u8)).float().div(255) * 2 - 1).permute(0, 3, 1, 2)[None]
    torch.set_num_threads(8)
    comp, tr = OP.run(weights.synth_state_dicts(P["seed"]), frames, torch.from_numpy(np.ascontiguousarray(fm)).float()[None, :, None],
                      torch.from_numpy(np.ascontiguousarray(md)).float()[None, :, None], [f for f in frames_u8],
                      raft_iter=P["raft_iter"], neighbor_length=P["neighbor_length"], ref_stride=P["ref_stride"],
                      subvideo_length=P["subvideo_length"], return_trace=True)
    s = P["flow_stride"]
    for d in (0, 1):
        assert np.abs(tr["gt_flows"][d][0, :, :, ::2 * s, ::2 * s].numpy() - g["gt_flow"][d]).max() < 1e-5
        pf = tr["pred_flows"][d][0, :, :, ::s, ::s].numpy()
        assert np.abs(pf - g["pred_flow"][d].astype(np.float32)).max() <= 2e-3 + np.abs(pf).max() * 2.0 ** -10   # f16 fixture
    assert np.array_equal(tr["updated_masks"][0, :, 0].numpy().astype(np.uint8), _unpack(g["updated_masks"], (T, h, w)))
    out = np.stack(comp, 0)
    sel = md.astype(bool)
    assert np.array_equal(out[~sel], np.asarray(frames_u8)[~sel])
    if sel.any():
        assert np.abs(out[sel].astype(np.int32) - g["out_masked"].astype(np.int32)).max() <= 1


def test_oracle_with_empty_masks_returns_the_input_frames():
    """Size-independent property of the path (propainter_inference.py:331-340: comp = pred*mask + frame*(1-mask)):
    with nothing to inpaint the composed uint8 frames are the input frames, whatever the networks predict."""
    from comfyui_propainter_nodes_amd import synth

    T, H, W = 3, 64, 64
    image, _ = synth.synthetic_clip(T, H, W, 7)
    frames_u8 = (image.numpy() * 255).clip(0, 255).astype(np.uint8)
    frames = (torch.from_numpy(frames_u8).float().div(255) * 2 - 1).permute(0, 3, 1, 2)[None]
    zeros = torch.zeros(1, T, 1, H, W)
    sds = weights.synth_state_dicts(0)
    torch.set_num_threads(8)
    comp = OP.run(sds, frames, zeros, zeros, [f for f in frames_u8], raft_iter=2, neighbor_length=2, ref_stride=2,
                  subvideo_length=80)
    assert np.array_equal(np.stack(comp, 0), frames_u8)
