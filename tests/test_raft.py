"""RAFT stage (product path, HIP kernels) against the oracle and the reference-minted fixture.

Tolerance: RAFT is fp32 on both sides (exact-f32 MFMA, different summation order); the flow of
the synthetic model amplifies rounding over the iterations, so the bound is 2e-3 px absolute on
flows of a few px (the oracle itself reproduces the reference bit-exactly on CPU)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from comfyui_propainter_nodes_amd import raft, synth, weights
from oracle import raft as OR

GOLD = Path(__file__).parent / "golden"


@pytest.mark.gpu
def test_raft_matches_oracle_and_golden(hip_lib):
    g = np.load(GOLD / "e2e_small.npz")
    T, H, W, iters = [int(v) for v in g["params"][:4]]
    sds = weights.synth_state_dicts(int(g["params"][9]))
    frames = torch.from_numpy(g["frames_u8"]).float().div(255) * 2 - 1  # [T,H,W,3]
    R = raft.RaftFlow(sds["raft"], "cuda:0")
    ff, fb = R(frames.cuda(), iters)
    ff, fb = ff.cpu(), fb.cpu()
    gf = torch.from_numpy(g["gt_flow_f"]).permute(0, 2, 3, 1)
    gb = torch.from_numpy(g["gt_flow_b"]).permute(0, 2, 3, 1)
    assert (ff - gf).abs().max().item() < 2e-3, (ff - gf).abs().max().item()
    assert (fb - gb).abs().max().item() < 2e-3
    # independent oracle run on one pair (not only the stored fixture)
    fr = frames.permute(0, 3, 1, 2)
    with torch.no_grad():
        o = OR.raft_forward(sds["raft"], fr[1:2], fr[2:3], iters)
    assert (ff[1] - o[0].permute(1, 2, 0)).abs().max().item() < 2e-3


@pytest.mark.gpu
def test_raft_pair_batching_is_chunk_invariant(hip_lib):
    """compute_flow's chunking must not change per-pair results (propainter_inference.py:65-90)."""
    sds = weights.synth_state_dicts(0)
    image, _ = synth.synthetic_clip(5, 128, 128)
    frames = (image * 2 - 1).cuda()
    a = raft.RaftFlow(sds["raft"], "cuda:0")(frames, 2)
    b = raft.RaftFlow(sds["raft"], "cuda:0", max_volume_bytes=1, enc_chunk=2)(frames, 2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.gpu
def test_raft_split_gemm_matches_exact_f32_gemm(hip_lib, monkeypatch):
    """PP_F32X2 (f32 convolutions as three f16 MFMA products per multiply-add, the default) against the exact f32
    MFMA kernels (PP_F32_GEMM=exact) on the same clip: both must sit inside the fixture tolerance of each other."""
    sds = weights.synth_state_dicts(0)
    image, _ = synth.synthetic_clip(4, 128, 144)
    frames = (image * 2 - 1).cuda()
    monkeypatch.setenv("PP_F32_GEMM", "exact")
    exact = raft.RaftFlow(sds["raft"], "cuda:0")
    assert not exact.convc1.split
    fe, be = exact(frames, 6)
    monkeypatch.setenv("PP_F32_GEMM", "split")
    split = raft.RaftFlow(sds["raft"], "cuda:0")
    assert split.convc1.split and split.gru["zr1"].split
    fs, bs = split(frames, 6)
    assert torch.isfinite(fs).all() and torch.isfinite(bs).all()
    assert (fs - fe).abs().max().item() < 1e-3, (fs - fe).abs().max().item()
    assert (bs - be).abs().max().item() < 1e-3


@pytest.mark.gpu
def test_raft_stress_undamped_weights_large_motion(hip_lib, monkeypatch):
    """The synthetic weights keep RAFT's flow head damped (weights.py: flow_head.conv2 gain 0.1, sub-pixel updates).  Here
    the head is UN-damped (gain 1.0: updates of several pixels per iteration, the correlation windows leave the image, the
    1/8-res flow reaches tens of px) on a clip with 20-px motion, 12 iterations: PP_F32X2 against the exact-f32 MFMA kernels
    and against the fp32 oracle.  The update loop amplifies rounding differences when the head is un-damped, so the bounds
    are relative to the flow magnitude (which the test also checks is large)."""
    sds = weights.synth_state_dicts(0)
    sd = dict(sds["raft"])
    for k in sd:
        if "flow_head.conv2.weight" in k:
            sd[k] = sd[k] * 10.0
    g = torch.Generator().manual_seed(3)
    base = torch.rand(1, 3, 40, 40, generator=g)
    base = torch.nn.functional.interpolate(base, size=(200, 200), mode="bicubic", align_corners=False).clamp(0, 1)[0]
    frames = torch.stack([base[:, 20 + 20 * i:148 + 20 * i, 10 + 12 * i:138 + 12 * i].permute(1, 2, 0) for i in range(3)], 0)
    frames = (frames * 2 - 1).contiguous()
    iters = 12
    monkeypatch.setenv("PP_F32_GEMM", "exact")
    fe, be = raft.RaftFlow(sd, "cuda:0")(frames.cuda(), iters)
    monkeypatch.setenv("PP_F32_GEMM", "split")
    fs, bs = raft.RaftFlow(sd, "cuda:0")(frames.cuda(), iters)
    fr = frames.permute(0, 3, 1, 2)
    with torch.no_grad():
        o = OR.raft_forward(sd, fr[0:1], fr[1:2], iters)[0].permute(1, 2, 0)
    mag = float(o.abs().max())
    e_se = float((fs - fe).abs().max())
    e_so = float((fs[0].cpu() - o).abs().max())
    e_eo = float((fe[0].cpu() - o).abs().max())
    print(f"stress: |flow| max {mag:.1f} px; split-exact {e_se:.2e}, split-oracle {e_so:.2e}, exact-oracle {e_eo:.2e} px")
    assert torch.isfinite(fs).all() and torch.isfinite(bs).all()
    assert mag > 5.0
    assert e_so <= max(2e-3, 5.0 * e_eo) and e_se <= max(2e-3, 5.0 * e_eo)   # the split is as close to the oracle as exact f32 is
