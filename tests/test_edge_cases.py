"""The corners of the node's parameter ranges, through the NODE METHODS, against fixtures minted from the reference's own
node methods (tests/golden/make_golden.py: EDGE_CASES; every one of them runs to completion in the reference, and the
oracle reproduces it bit for bit on the stage tensors):

  2-frame clip with neighbor_length 2 / ref_stride 1 / raft_iter 1 / no dilation; odd neighbor_length with ref_stride
  beyond the clip; neighbor_length 300 (one window spans the clip); subvideo_length 1 and 2 (sub-videos that are all
  halo); T = subvideo_length and T = subvideo_length + 1 (global / local reference mode boundary); a 200x136 clip
  (34x50 tokens: partial windows on both axes); every frame a reference frame; an all-zero MASK (nothing to inpaint), an
  all-one MASK and a dilation that swallows the frame (nothing known); outpainting by height only, by both axes, and
  with both scales 1 (empty border).

`-m gpu`: every case with fp16 "enable" (and five of them with "disable") on the MI355X.  On CPU the same check can run with the kernel sources under the x86
emulator -- 4 to 10 minutes per case, so it is opt-in: PP_EDGE_EMU=all (or a comma-separated list of case names); the CPU
suite's own emulated end-to-end run is tests/test_e2e_emulation.py."""
import os

import pytest
import torch

from comfyui_propainter_nodes_amd import nodes, pipeline
from node_case import EDGE, check_node_case

_EMU = os.environ.get("PP_EDGE_EMU", "")
EMULATED = EDGE if _EMU == "all" else [c for c in _EMU.split(",") if c]


@pytest.fixture(scope="module")
def synthetic_models():
    """Seeded synthetic weights, prepared once for the whole module (the model cache is keyed on device, fp16 and arithmetic)."""
    saved = os.environ.get("PP_ALLOW_SYNTHETIC_WEIGHTS")
    os.environ["PP_ALLOW_SYNTHETIC_WEIGHTS"] = "1"
    pipeline.drop_model_cache()
    yield
    pipeline.drop_model_cache()
    if saved is None:
        os.environ.pop("PP_ALLOW_SYNTHETIC_WEIGHTS", None)
    else:
        os.environ["PP_ALLOW_SYNTHETIC_WEIGHTS"] = saved


# fp16 "disable" (fp32 storage, other kernels under the same host logic) for the cases whose host path differs most
GPU_CASES = [(c, "enable") for c in EDGE] + [(c, "disable") for c in ("edge_T2_min", "edge_T6_sv1", "edge_T4_no_mask",
                                                                      "edge_T4_full_mask", "edge_T4_outpaint_both")]


@pytest.mark.gpu
@pytest.mark.parametrize("case,fp16", GPU_CASES)
def test_edge_case_matches_reference_fixture(hip_lib, synthetic_models, case, fp16):
    check_node_case(case, fp16)


@pytest.mark.slow
@pytest.mark.skipif(not EMULATED, reason="opt-in: PP_EDGE_EMU=all")
@pytest.mark.parametrize("case", EMULATED or ["-"])
def test_edge_case_under_emulation(emu_lib, pp_knobs, synthetic_models, monkeypatch, case):
    """The same check with every kernel source executed by the x86 emulator (tests/emu): the node runs on CPU tensors."""
    pp_knobs(PP_CONV_KSPLIT="0")   # (the 1024-fiber split-K work-groups are slow to emulate; own tests in tests/test_conv.py)
    monkeypatch.setattr(nodes, "get_torch_device", lambda: torch.device("cpu"))
    check_node_case(case, "enable")
