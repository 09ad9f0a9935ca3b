"""The WHOLE product path on CPU: host pipeline (pipeline.py / raft.py / rfc.py / imgprop.py / generator.py) driving
every kernel source through the x86 emulator (tests/emu), against the oracle, on the smallest clip the reference accepts
(2 frames, 128x128: RAFT's lower size limit).  About two minutes; the GPU suite repeats it at real sizes through the
gfx950 library against the reference-minted fixtures."""
import numpy as np
import pytest
import torch

from comfyui_propainter_nodes_amd import image_utils, pipeline, synth, weights
from oracle import pipeline as OP


@pytest.mark.slow
def test_whole_pipeline_under_emulation_matches_oracle(emu_lib, pp_knobs):
    # (the 1024-fiber split-K work-groups are slow to emulate and have their own tests: tests/test_conv.py)
    pp_knobs(PP_CONV_KSPLIT="0")
    T, H, W = 2, 128, 128
    image, mask = synth.synthetic_clip(T, H, W, 3)
    u8 = image_utils.image_to_uint8_frames(image)
    fr, fm, md = image_utils.prepare_frames_and_masks(u8, mask, image_utils.ImageConfig(W, H, 2, 3, (W, H), T))
    assert 0.05 < md.mean() < 0.5  # a real hole
    sds = weights.synth_state_dicts(0)
    dev = torch.device("cpu")
    models = pipeline.models_from_state_dicts(sds, dev)
    cfg = pipeline.ProPainterConfig(2, 2, 80, 1, "enable", T, dev, (W, H))
    class Sink:  # the node's streaming hook: every reported range must already hold its final pixels
        def __init__(self):
            self.parts = []

        def frames_final(self, comp, lo, hi):
            self.parts.append((lo, hi, comp[lo:hi].clone()))

    sink = Sink()
    got = pipeline.run_inpainting(models, torch.from_numpy(fr), torch.from_numpy(fm), torch.from_numpy(md), cfg,
                                  to_host=False, sink=sink).numpy()
    assert [p[:2] for p in sink.parts][0][0] == 0 and sink.parts[-1][1] == T
    assert all(np.array_equal(part.numpy(), got[lo:hi]) for lo, hi, part in sink.parts)
    frames = (torch.from_numpy(fr).float().div(255) * 2 - 1).permute(0, 3, 1, 2)[None]
    torch.set_num_threads(8)
    ref = np.stack(OP.run(sds, frames, torch.from_numpy(fm).float()[None, :, None], torch.from_numpy(md).float()[None, :, None],
                          [f for f in fr], raft_iter=1, neighbor_length=2, ref_stride=2, subvideo_length=80), 0)
    d = got.astype(np.float32) - ref.astype(np.float32)
    psnr = 10 * np.log10(255.0 ** 2 / max(float((d ** 2).mean()), 1e-12))
    assert psnr >= 40.0, psnr                       # BASELINE.json north_star tolerance
    assert np.abs(d).max() <= 2.55, np.abs(d).max()  # max abs diff < 1e-2 of full scale
    outside = md == 0
    assert np.array_equal(got[outside], fr[outside])  # untouched pixels are bit exact
