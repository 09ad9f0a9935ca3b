"""GPU parity at the BASELINE.json configurations, through the NODE METHODS (the drop-in boundary).

Fixtures (tests/golden/cfg*_node.npz) were minted by running the reference's own node methods
(propainter_nodes.py:93-154 / :231-310) on CPU fp32 with the seeded synthetic weights and the seeded synthetic clip
(tests/golden/make_golden.py: run_node_case), capturing stage tensors on the way:
  cfg1_node      BASELINE configs[0]: 16 frames 320x180 -> 320x176 (PIL bicubic resize path), raft_iter 5
  cfg2_24f_node  configs[1] geometry: 640x360, nl 10, rs 10, raft_iter 20 (30x54 tokens, 6x6 windows), 24-frame truncation
  cfg3_12f_node  configs[2] geometry: outpaint 640x360 -> 768x360, 12-frame truncation
  cfg2_80f_node  configs[1] IN FULL (r03): the 80-frame clip bench.py times
  cfg4_100f_node configs[3]'s mode at its size (r03): 100 frames > subvideo_length 80 -> local reference frames (ref_num 8),
                 flow completion in two sub-videos with 5-frame halos, image propagation with 10-frame halos
  mov_20f_node   per-frame (moving) MASK input (r03): mask.shape[0] == T, one dilation per mask frame
  cfg3_80f_node  configs[2] IN FULL (r04): 80-frame outpaint 640x360 -> 768x360 (reference run 1267 s)
  cfg5_90f_node  configs[4]'s size AND mode (r04): 90 frames of 1280x720 > subvideo_length 80, neighbor_length 20, raft_iter 20:
                 local reference frames, RAFT in the reference's short clips of 4, two flow-completion / image-propagation
                 sub-videos, 60x107 -> 60x108 token grid, 21-frame windows
  cfg4_170f_node an INTERIOR sub-video at real size (r04): 170 frames of 640x360 -> sub-videos [0,80) [80,160) [160,170): the
                 middle one has halos on both sides
Tolerances (north_star: PSNR >= 40 dB on the pixels, masks / schedules bit-exact):
  RAFT flows 2e-3 px; updated masks <= 0.5 % differing pixels; final uint8 frames: exactly the input outside the dilated
  mask, PSNR >= 40 dB and >= 99 % within 2 LSB inside it; node mask outputs bit-exact.
  Completed flows: with fp16 "disable" the flow-completion network keeps fp32 tensors like the fixture's reference run;
  its input (our RAFT flows) differs from the reference's by ~1.3e-4 px and the synthetic (untrained, non-contractive)
  recurrence amplifies input perturbations ~100x over 24 frames, so the stage agrees to 2e-2 px max / 2e-3 mean there
  (identical with PP_F32_GEMM=exact; the network alone on the reference's own input agrees to 1e-3: tests/test_rfc.py)
  -- asserted: max < 5e-2, mean < 5e-3 (1e-3 of the mean is the fixture's f16 storage).  With fp16 "enable" it is an f16
  network (the reference's `.half()` mode): its second-order deformable recurrence amplifies f16 rounding with the clip
  length under these synthetic (untrained, non-contractive) weights -- measured max / p99.9 / mean: 1.2e-2 / 4.9e-3 /
  5e-4 px at 16 frames of 320x176, 6e-2 / 2e-2 / 2e-3 at 12 frames of 768x360, 1.4 / 0.55 / 1.6e-2 at 24 frames of
  640x360 -- so that mode only asserts mean < 5e-2 px and max < 3 px on the stage and relies on the fp32 mode for the
  tight stage check; the final-frame bound is the same in both modes.
  Clips of more than 40 frames (r03: the full 80-frame configs[1], the 100-frame long-clip case): INSIDE the hole the
  recurrence is chaotic with these weights -- the reference's own fp32 arithmetic (the CPU oracle) fed with our RAFT flows
  (1.4e-4 px from the reference's) lands 3.9 px max / 5e-2 mean from the reference's result, exactly where our fp32 and f16
  runs land (3.8 / 4.1 px), while the stage agrees to 1e-5 px with the oracle on identical input where the recurrence is
  stable (tests/test_rfc.py::test_rfc_80_frames_teacher_forced; profiles/r03_flow_completion_sensitivity.md).  So there
  the test asserts the completed flows tightly OUTSIDE the flow mask (2e-3 px beyond the fixture's f16 storage
  rounding: they are the RAFT flows) and only
  mean < 0.25 / max < 10 px inside; masks, schedules and the final frames keep their bounds.
  cfg2_80f_contractive_node (r04): configs[1] in full with the CONTRACTIVE synthetic-weight variant -- the recurrence damps
  input perturbations (8e-4 px response to 1.4e-4 px at 80 frames on the MI355X, tools/diag_recurrence_sensitivity.py), so the
  completed flows are asserted pointwise inside the hole at the full length: max < 2.5e-2 / mean < 3e-3 px (fp32 storage; measured
  1.56e-2 / 1.9e-3 = the fixture's own f16 storage of flows of up to 36 px), max < 0.15 / mean < 5e-3 px (f16 storage: measured
  4.7e-2 / 2.2e-3); final frames 58.1 / 77.4 dB, max 1 LSB.
  r05, the stated LENGTHS (VERDICT r04 missing #3): cfg4_640f_node = configs[3] in full, 640 frames of 640x360 = 8 sub-videos (the
  plan the 8-GPU run shards); cfg5_160f_node = configs[4] in full, 160 frames of 1280x720, nl 20; cfg5_90f_contractive_node =
  configs[4]'s size and mode with the contractive weights (completed flows pointwise inside the hole at 1280x720).  Stored for a
  subset of the frames (every k-th + both sides of every sub-video seam; the other frames by the sum of their masked pixels:
  tests/golden/make_golden.py keep_every), minted without the oracle pin (the fixture IS the reference's output).  Measured on the
  MI355X (profiles/r05_pytest_gpu.log, r05_pytest_cfg5_160f.log, r05_pytest_cfg5_contractive.log): 640 f 57.5 / 61.4 dB max 2 LSB;
  160 f 1280x720 57.8 / 63.8 dB max 3 / 2 LSB; contractive 1280x720: completed flows 3.1e-2 / 1.56e-2 px max inside the hole,
  58.1 / 77.1 dB max 1 LSB.
A live-oracle case covers configs[4]'s geometry (1280x720, nl 20: 60x107 -> 60x108 token grid, 405 pooled keys)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from comfyui_propainter_nodes_amd import pipeline, synth
from node_case import check_node_case, psnr


@pytest.fixture()
def synthetic_models(monkeypatch):
    monkeypatch.setenv("PP_ALLOW_SYNTHETIC_WEIGHTS", "1")
    pipeline.drop_model_cache()
    yield
    pipeline.drop_model_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("fp16", ["enable", "disable"])
@pytest.mark.parametrize("case", ["cfg1_node", "cfg2_24f_node", "cfg3_12f_node", "cfg2_80f_node", "cfg4_100f_node", "mov_20f_node",
                                  "cfg3_80f_node", "cfg5_90f_node", "cfg4_170f_node", "cfg2_80f_contractive_node",
                                  # r05: the configurations at their STATED length (skipped until minted: hours of reference time)
                                  "cfg4_640f_node", "cfg5_160f_node", "cfg5_90f_contractive_node"])
def test_node_matches_reference_fixture(hip_lib, synthetic_models, case, fp16):
    check_node_case(case, fp16)


@pytest.mark.gpu
def test_cfg5_geometry_against_live_oracle(hip_lib):
    """BASELINE configs[4] geometry on a short clip: 1280x720, neighbor_length 20, ref_stride 10 -> token grid 60x107
    padded to 60x108, 12x12 windows, 405 pooled keys per frame; the CPU oracle runs beside it (raft_iter 2, 5 frames)."""
    from comfyui_propainter_nodes_amd import image_utils, weights
    from oracle import pipeline as OP

    T, H, W = 5, 720, 1280
    kw = dict(raft_iter=2, neighbor_length=20, ref_stride=10, subvideo_length=80)
    image, mask = synth.synthetic_clip(T, H, W)
    frames_u8 = image_utils.image_to_uint8_frames(image)
    frames_u8, fm, md = image_utils.prepare_frames_and_masks(frames_u8, mask, image_utils.ImageConfig(W, H, 5, 8, (W, H), T))
    sds = weights.synth_state_dicts(0)
    dev = torch.device("cuda:0")
    models = pipeline.models_from_state_dicts(sds, dev)
    cfg = pipeline.ProPainterConfig(kw["ref_stride"], kw["neighbor_length"], kw["subvideo_length"], kw["raft_iter"], "enable",
                                    T, dev, (W, H))
    tr = {}
    got = pipeline.run_inpainting(models, frames_u8, fm, md, cfg, trace=tr).numpy()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    frames = (torch.from_numpy(frames_u8).float().div(255) * 2 - 1).permute(0, 3, 1, 2)[None]
    ref, otr = OP.run(sds, frames, torch.from_numpy(fm).float()[None, :, None], torch.from_numpy(md).float()[None, :, None],
                      [f for f in frames_u8], return_trace=True, **kw)
    ref = np.stack(ref, 0)
    e_gt = max(float((tr["gt_flows"][i].cpu() - otr["gt_flows"][i][0].permute(0, 2, 3, 1)).abs().max()) for i in (0, 1))
    d_pf = torch.cat([(tr["pred_flows"][i].cpu() - otr["pred_flows"][i][0].permute(0, 2, 3, 1)).abs().flatten() for i in (0, 1)])
    e_pf, q_pf = float(d_pf.max()), float(torch.quantile(d_pf[::7], 0.999))
    sel = md.astype(bool)
    p = psnr(got[sel], ref[sel])
    print(f"cfg5 geometry: gt_flow {e_gt:.2e} px, pred_flow max {e_pf:.2e} p99.9 {q_pf:.2e} px, masked-pixel PSNR {p:.1f} dB, "
          f"max {int(np.abs(got.astype(int) - ref.astype(int)).max())} LSB")
    assert np.array_equal(got[~sel], ref[~sel])
    assert e_gt < 2e-3 and q_pf < 3e-2 and e_pf < 2.0 and p >= 40.0
