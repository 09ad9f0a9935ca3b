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
    assert split.convc1.split and split.gru["z1"].split
    fs, bs = split(frames, 6)
    assert torch.isfinite(fs).all() and torch.isfinite(bs).all()
    assert (fs - fe).abs().max().item() < 1e-3, (fs - fe).abs().max().item()
    assert (bs - be).abs().max().item() < 1e-3
