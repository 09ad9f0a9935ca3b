"""Host-side integer logic: bit-exact against fixtures minted from the reference (tests/golden/host_cases.npz)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from comfyui_propainter_nodes_amd import image_utils, nodes, pipeline, synth

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def host_cases():
    return np.load(GOLD / "host_cases.npz", allow_pickle=False)


def test_node_api_is_the_reference_api():
    assert set(nodes.NODE_CLASS_MAPPINGS) == {"ProPainterInpaint", "ProPainterOutpaint"}
    assert nodes.NODE_DISPLAY_NAME_MAPPINGS == {"ProPainterInpaint": "ProPainter Inpainting",
                                                "ProPainterOutpaint": "ProPainter Outpainting"}
    I, O = nodes.ProPainterInpaint, nodes.ProPainterOutpaint
    assert I.RETURN_TYPES == ("IMAGE", "MASK", "MASK") and I.RETURN_NAMES == ("IMAGE", "FLOW_MASK", "MASK_DILATE")
    assert I.FUNCTION == "propainter_inpainting" and I.CATEGORY == "ProPainter"
    assert O.RETURN_TYPES == ("IMAGE", "MASK", "INT", "INT")
    assert O.RETURN_NAMES == ("IMAGE", "OUTPAINT_MASK", "output_width", "output_height")
    assert O.FUNCTION == "propainter_outpainting"
    req = I.INPUT_TYPES()["required"]
    assert list(req) == ["image", "mask", "width", "height", "mask_dilates", "flow_mask_dilates", "ref_stride",
                         "neighbor_length", "subvideo_length", "raft_iter", "fp16"]
    assert req["width"] == ("INT", {"default": 640, "min": 0, "max": 2560}) and req["raft_iter"][1]["default"] == 20
    assert req["fp16"] == (["enable", "disable"],)
    oreq = O.INPUT_TYPES()["required"]
    assert list(oreq)[:5] == ["image", "width", "height", "width_scale", "height_scale"]
    assert oreq["width_scale"] == ("FLOAT", {"default": 1.2, "min": 0.0, "max": 10.0, "step": 0.01})


def test_node_api_matches_reference_fixture(host_cases):
    """INPUT_TYPES dumped from the reference classes themselves."""
    import json

    ref = json.loads(str(host_cases["api_json"]))
    assert json.loads(json.dumps(nodes.ProPainterInpaint.INPUT_TYPES())) == ref["inpaint_inputs"]
    assert json.loads(json.dumps(nodes.ProPainterOutpaint.INPUT_TYPES())) == ref["outpaint_inputs"]


def test_check_inputs_errors():
    with pytest.raises(Exception, match="Image length must be greater than 1"):
        nodes.check_inputs(torch.zeros(1, 8, 8, 3), torch.zeros(1, 8, 8))
    with pytest.raises(Exception, match="same length"):
        nodes.check_inputs(torch.zeros(4, 8, 8, 3), torch.zeros(3, 8, 8))
    with pytest.raises(Exception, match="same dimensions"):
        nodes.check_inputs(torch.zeros(4, 8, 8, 3), torch.zeros(1, 8, 9))
    nodes.check_inputs(torch.zeros(4, 8, 8, 3), torch.zeros(1, 8, 8))


def test_zero_size_fails_like_the_reference():
    """width / height 0 are inside INPUT_TYPES' range (min 0, propainter_nodes.py:50-51) and the reference fails on them in
    PIL's resize (utils/image_utils.py:184 -> "height and width must be > 0", observed by running its node); so does the
    drop-in -- also when only one side rounds down to 0 (7 -> 0)."""
    image, mask = synth.synthetic_clip(3, 132, 150)
    for wh in ((0, 0), (7, 300)):
        cfg = image_utils.ImageConfig(*wh, 2, 3, (150, 132), 3)
        with pytest.raises(ValueError, match="height and width must be > 0"):
            image_utils.prepare_frames_and_masks(image_utils.image_to_uint8_frames(image), mask, cfg)


def test_sizes_round_down_to_multiples_of_8():
    c = image_utils.ImageConfig(320, 180, 5, 8, (320, 180), 16)
    assert c.process_size == (320, 176)
    o = image_utils.ImageOutpaintConfig(640, 360, 5, 8, (640, 360), 80, 1.2, 1.0)
    assert o.process_size == (640, 360) and o.outpaint_size == (768, 360)


def test_read_masks_bit_exact(host_cases):
    for i in range(int(host_cases["n_mask_cases"])):
        mask = torch.from_numpy(host_cases[f"mask_in_{i}"])
        w, h, md, fmd, T = [int(v) for v in host_cases[f"mask_par_{i}"]]
        cfg = image_utils.ImageConfig(w, h, md, fmd, (mask.shape[2], mask.shape[1]), T)
        fm, dm = image_utils.read_masks(mask, cfg)
        assert np.array_equal(fm, host_cases[f"mask_flow_{i}"]), i
        assert np.array_equal(dm, host_cases[f"mask_dil_{i}"]), i


def test_frame_conversion_and_resize_bit_exact(host_cases):
    img = torch.from_numpy(host_cases["frames_in"])
    u8 = image_utils.image_to_uint8_frames(img)
    assert np.array_equal(u8, host_cases["frames_u8"])
    w, h = [int(v) for v in host_cases["frames_resize_to"]]
    assert np.array_equal(image_utils.resize_frames(u8, (w, h)), host_cases["frames_resized"])


def test_extrapolation_bit_exact(host_cases):
    img = torch.from_numpy(host_cases["frames_in"])
    u8 = image_utils.image_to_uint8_frames(img)
    for i in range(int(host_cases["n_out_cases"])):
        w, h, T = [int(v) for v in host_cases[f"out_par_{i}"][:3]]
        ws, hs = [float(v) for v in host_cases[f"out_scale_{i}"]]
        cfg = image_utils.ImageOutpaintConfig(w, h, 5, 8, (u8.shape[2], u8.shape[1]), T, ws, hs)
        canvas, fm, dm = image_utils.extrapolation(u8, cfg)
        assert np.array_equal(canvas, host_cases[f"out_canvas_{i}"])
        assert np.array_equal(fm[0], host_cases[f"out_flow_{i}"]) and np.array_equal(dm[0], host_cases[f"out_dil_{i}"])


def test_window_schedules_match_reference(host_cases):
    import json

    for key, ref in json.loads(str(host_cases["schedules_json"])).items():
        T, nl, rs, sv = [int(v) for v in key.split(",")]
        cfg = pipeline.ProPainterConfig(rs, nl, sv, 20, "enable", T, torch.device("cuda"), (640, 360))
        got = [[nb, refs] for nb, refs in pipeline.window_schedule(cfg)]
        assert got == ref, key


def test_window_schedules_match_reference_on_random_parameters():
    """400 seeded random (video_length, neighbor_length, ref_stride, subvideo_length) from the node's parameter ranges -- global
    and local reference mode, odd strides, windows longer than the clip, 1-frame sub-videos: the schedule equals the one the
    reference's own get_ref_index loop produced (tests/golden/make_golden.py: schedule_cases, one SHA-1 per combination)."""
    import hashlib
    import json

    g = np.load(Path(__file__).parent / "golden" / "schedule_cases.npz")
    for (T, nl, rs, sv), want, cnt in zip(g["keys"].tolist(), g["digests"].tolist(), g["counts"].tolist()):
        cfg = pipeline.ProPainterConfig(rs, nl, sv, 20, "enable", T, torch.device("cuda"), (640, 360))
        rows = [[nb, refs] for nb, refs in pipeline.window_schedule(cfg)]
        assert [len(rows), sum(len(a) for a, _ in rows), sum(len(b) for _, b in rows)] == cnt, (T, nl, rs, sv)
        assert hashlib.sha1(json.dumps(rows).encode()).hexdigest() == want, (T, nl, rs, sv)


def test_subvideo_plans_match_reference(monkeypatch):
    """Which sub-video every completed flow / propagated frame is computed in, and at which position of it (flow completion:
    chunks of subvideo_length flows with 5-frame halos; image propagation: chunks of min(100, subvideo_length) frames with
    10-frame halos), for 80 seeded random (video_length, subvideo_length): equal to what the REFERENCE's complete_flow /
    image_propagation did with stand-in models that tag their output (tests/golden/make_golden.py: chunk_plan_cases)."""
    g = np.load(Path(__file__).parent / "golden" / "chunk_plans.npz")

    def fake_rfc(flows, masks):              # [2,n,1,1,2] -> frame j of the chunk = 1000 * (first flow of the chunk) + j
        n = flows.shape[1]
        return (flows[:1, :1] * 1000 + torch.arange(n, dtype=torch.float32).view(1, n, 1, 1, 1)).expand(2, n, 1, 1, 2).clone()

    def fake_imgprop(frames, masks_u8, flows):
        t = frames.shape[0]
        tag = flows[0, 0, 0, 0, 0] * 1000 + torch.arange(t, dtype=torch.float32)
        return tag.view(t, 1, 1, 1).expand(t, 1, 1, 3).clone(), (tag % 200).to(torch.uint8).view(t, 1, 1)

    monkeypatch.setattr(pipeline.imgprop, "image_propagation", fake_imgprop)
    of = oi = 0
    for T, sv in g["keys"].tolist():
        nf = T - 1
        idx = torch.arange(nf, dtype=torch.float32).view(1, nf, 1, 1, 1).expand(2, nf, 1, 1, 2).contiguous()
        got = pipeline.complete_flow(fake_rfc, idx, torch.zeros(T, 1, 1, dtype=torch.uint8), sv)[0, :, 0, 0, 0]
        assert got.long().tolist() == g["flow"][of:of + nf].tolist(), (T, sv)
        cfg = pipeline.ProPainterConfig(10, 10, sv, 20, "enable", T, "cpu", (1, 1))
        prop, upd = pipeline.image_propagation(torch.zeros(T, 1, 1, 3), torch.ones(T, 1, 1, dtype=torch.uint8), idx, cfg)
        assert prop[:, 0, 0, 0].long().tolist() == g["img"][oi:oi + T].tolist(), (T, sv)
        assert upd[:, 0, 0].long().tolist() == g["img_mask"][oi:oi + T].tolist(), (T, sv)
        of += nf
        oi += T


def test_survey_window_counts():
    """SURVEY.md section 3 table (computed there with the reference's get_ref_index)."""
    for (T, nl, rs, sv), (nwin, sum_lt, sum_t) in {(16, 10, 10, 80): (4, 34, 37), (80, 10, 10, 80): (16, 170, 275),
                                                   (640, 10, 10, 80): (128, 1402, 2266), (160, 20, 10, 80): (16, 325, 391)}.items():
        cfg = pipeline.ProPainterConfig(rs, nl, sv, 20, "enable", T, torch.device("cuda"), (640, 360))
        s = pipeline.window_schedule(cfg)
        assert (len(s), sum(len(a) for a, _ in s), sum(len(a) + len(b) for a, b in s)) == (nwin, sum_lt, sum_t)


@pytest.mark.parametrize("T,nl,rs,sv", [(80, 10, 10, 80), (13, 4, 3, 80), (2, 2, 2, 80), (31, 6, 5, 10), (100, 20, 10, 80), (7, 300, 1, 80)])
def test_final_ranges_tile_the_clip_and_are_really_final(T, nl, rs, sv):
    """The node streams frames out as soon as pipeline.final_ranges says no later window blends into them: the ranges
    must tile [0, T) in order, and no frame of a range may be a local frame of a LATER window (checked by brute force
    on the reference's own window schedule)."""
    from comfyui_propainter_nodes_amd import pipeline

    cfg = pipeline.ProPainterConfig(rs, nl, sv, 1, "enable", T, "cpu", (64, 64))
    sched = pipeline.window_schedule(cfg)
    fin = pipeline.final_ranges(sched, T)
    assert len(fin) == len(sched) and fin[0][0] == 0 and fin[-1][1] == T
    assert all(a <= b for a, b in fin) and all(fin[i][1] == fin[i + 1][0] for i in range(len(fin) - 1))
    for wi, (lo, hi) in enumerate(fin):
        later = {f for nb, _ in sched[wi + 1:] for f in nb}
        assert not (set(range(lo, hi)) & later), (wi, lo, hi)
        # ... and nothing is held back longer than necessary: the first frame after the range is still needed (or the clip ends)
        assert hi == T or hi in later
