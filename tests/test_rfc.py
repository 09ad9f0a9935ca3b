"""Recurrent flow completion stage (HIP path) against the oracle and the reference-minted fixture.

The stage runs f16 activations with fp32 accumulation (the reference runs it `.half()`); against
the fp32 oracle the completed flow must agree to 2e-2 px (flows of a few px; observed ~3e-3)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from comfyui_propainter_nodes_amd import rfc, weights
from oracle import rfc as OC

GOLD = Path(__file__).parent / "golden"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-2), (torch.float32, 1e-3)])
def test_rfc_matches_oracle_and_golden(hip_lib, dtype, tol):
    """f16 storage (fp16 "enable") and f32 storage (fp16 "disable") against the fixture and the live fp32 oracle."""
    g = np.load(GOLD / "e2e_small.npz")
    sds = weights.synth_state_dicts(int(g["params"][9]))
    gt = torch.stack([torch.from_numpy(g["gt_flow_f"]), torch.from_numpy(g["gt_flow_b"])], 0).permute(0, 1, 3, 4, 2).contiguous()
    masks = torch.from_numpy(g["flow_masks"])
    C = rfc.FlowCompleter(sds["rfc"], "cuda:0", dtype)
    out = C(gt.cuda(), masks.cuda()).cpu()
    gold = torch.stack([torch.from_numpy(g["pred_flow_f"]), torch.from_numpy(g["pred_flow_b"])], 0).float().permute(0, 1, 3, 4, 2)
    assert (out - gold).abs().max().item() < 2e-2, (out - gold).abs().max().item()
    # live oracle on the same inputs (fp32, not the f16-rounded fixture)
    m = masks.float()[None, :, None]
    fl = (gt[0].permute(0, 3, 1, 2)[None], gt[1].permute(0, 3, 1, 2)[None])
    with torch.no_grad():
        ref = OC.combine_flow(fl, OC.forward_bidirect_flow(sds["rfc"], fl, m), m)
    err = max((out[d].permute(0, 3, 1, 2) - ref[d][0]).abs().max().item() for d in (0, 1))
    assert err < tol, err
