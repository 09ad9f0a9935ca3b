"""Whole stages executed on CPU under the kernel emulator (tests/emu) against the oracle -- the same host
schedulers and the same kernel sources that run on the MI355X, at sizes the emulator finishes in a minute.
(The GPU suite repeats these at real sizes through the gfx950 library.)"""
import torch
import torch.nn.functional as F

from comfyui_propainter_nodes_amd import rfc, weights
from oracle import rfc as OC


import pytest


@pytest.mark.parametrize("dtype,gemm,tol", [(torch.float16, "split", 2e-2), (torch.float32, "split", 1e-3),
                                            (torch.float32, "exact", 1e-3)])
def test_flow_completion_stage_under_emulation(emu_lib, monkeypatch, dtype, gemm, tol):
    """f16 storage (fp16 "enable") and f32 storage (fp16 "disable": PP_F32X2 split products, or the f32 MFMA
    instructions with PP_F32_GEMM=exact) against the fp32 oracle."""
    monkeypatch.setenv("PP_F32_GEMM", gemm)
    monkeypatch.setenv("PP_DEFORM_FUSED", "1")
    sds = weights.synth_state_dicts(0)
    T, H, W = 3, 32, 40
    g = torch.Generator().manual_seed(5)
    flows = torch.randn(2, T, H, W, 2, generator=g) * 2
    masks = torch.zeros(T + 1, H, W, dtype=torch.uint8)
    masks[:, H // 3:2 * H // 3, W // 4:3 * W // 4] = 1
    out = rfc.FlowCompleter(sds["rfc"], "cpu", dtype)(flows, masks)
    of, ob = flows[0].permute(0, 3, 1, 2)[None], flows[1].permute(0, 3, 1, 2)[None]
    m = masks.float()[None, :, None]
    with torch.no_grad():
        ref = OC.combine_flow((of, ob), OC.forward_bidirect_flow(sds["rfc"], (of, ob), m), m)
    for d in (0, 1):
        err = (out[d].permute(0, 3, 1, 2) - ref[d][0]).abs().max().item()
        assert err < tol, err  # f16 activations: 2e-2 px on flows of a few px; f32: fp32 rounding noise
    if dtype == torch.float16:
        # the stage ran the one-launch deformable convolution (pp_deform_conv); the two-launch form (pp_deform_cols + 1x1
        # pp_conv2d) it replaces gives the same flows bit for bit: same sampled f16 values, same summation order
        monkeypatch.setenv("PP_DEFORM_FUSED", "0")
        assert torch.equal(rfc.FlowCompleter(sds["rfc"], "cpu", dtype)(flows, masks), out)
