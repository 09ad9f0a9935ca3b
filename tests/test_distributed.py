"""Multi-GPU path: the sub-video sharding plan and the seam-exchange orchestration.

CPU: the real driver (comfyui_propainter_nodes_amd/distributed.py) over torch.distributed/gloo with world_size 2 and 3,
stage functions replaced by a toy backend; the sharded result must equal the single-rank result exactly.
GPU: the same driver with the real MI355X backend (in-process virtual ranks, and two real processes over gloo) must
reproduce the single-GPU pipeline on the chunked fixture bit for bit (r03: every kernel choice is a function of the layer
shape, never of how many frames / windows a rank batches); two ranks over RCCL ("nccl") when the box has two GPUs."""
import os
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from comfyui_propainter_nodes_amd import distributed as D
from comfyui_propainter_nodes_amd import pipeline

GOLD = Path(__file__).parent / "golden"
# the halo rows of a posted exchange are NaN until they land: a window that reads one too early cannot pass
os.environ.setdefault("PP_POISON_HALOS", "1")


def test_shard_plan_is_the_reference_chunking():
    # cfg 4: 640 frames, 8 ranks -> the reference's own 8 sub-videos (propainter_inference.py:115-144,172-212)
    for r in range(8):
        p = D.ShardPlan(640, 80, 8, r)
        assert p.frames == (80 * r, 80 * r + 80)
        assert p.flow_chunks() == [(80 * r, min(639, 80 * r + 80), max(0, 80 * r - 5), min(639, 80 * r + 85))]
        assert p.frame_chunks() == [(80 * r, 80 * r + 80, max(0, 80 * r - 10), min(640, 80 * r + 90))]
    # uneven: 9 frames, sub-videos of 4, 2 ranks -> chunks {0,1} and {2}
    p0, p1 = D.ShardPlan(9, 4, 2, 0), D.ShardPlan(9, 4, 2, 1)
    assert p0.frames == (0, 8) and p1.frames == (8, 9)
    assert p0.flow_chunks() == [(0, 4, 0, 8), (4, 8, 0, 8)] and p1.flow_chunks() == []
    assert p1.frame_chunks() == [(8, 9, 0, 9)] and p1.raft_frames() == (0, 0)
    covered = sorted(i for r in range(3) for i in range(*D.ShardPlan(37, 10, 3, r).frames))
    assert covered == list(range(37))


def _toy_inputs(T, H=6, W=8):
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (T, H, W, 3), generator=g, dtype=torch.uint8)
    md = (torch.rand(T, H, W, generator=g) > 0.5).to(torch.uint8)
    fm = (torch.rand(T, H, W, generator=g) > 0.4).to(torch.uint8)
    return frames, fm, md


def _cfg(T, nl, rs, sv):
    return pipeline.ProPainterConfig(rs, nl, sv, 2, "enable", T, torch.device("cpu"), (8, 6))


def _worker(rank, world, port, T, nl, rs, sv, out_dir, gather_root=None):
    import sys

    sys.path.insert(0, str(Path(__file__).parent))
    from toy_backend import ToyBackend
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, fm, md = _toy_inputs(T)
    tl = []     # (r05) the per-rank segment clock bench.py --gpus N prints: same results with it, every exchange of the protocol listed
    res = D.run_distributed(ToyBackend(), _cfg(T, nl, rs, sv), frames, fm, md, gather_root=gather_root, timeline=tl)
    kinds = [k for k, _, _ in tl]
    assert kinds[0] == "compute" and kinds[-1] == "compute" and kinds.count("p2p_start") == kinds.count("p2p_wait") == 2
    assert all(ms >= 0 for _, ms, _ in tl) and any(b > 0 for k, _, b in tl if k != "compute")
    torch.save(res, Path(out_dir) / f"r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,T,nl,rs,sv", [(2, 40, 6, 3, 10), (3, 37, 4, 2, 10), (2, 9, 4, 2, 4), (2, 60, 30, 10, 15)])
def test_sharded_equals_single_rank_over_gloo(tmp_path, world, T, nl, rs, sv):
    import sys

    sys.path.insert(0, str(Path(__file__).parent))
    from toy_backend import ToyBackend

    frames, fm, md = _toy_inputs(T)
    ref = D.run_simulated(lambda r: ToyBackend(), 1, _cfg(T, nl, rs, sv), frames, fm, md)[0]
    port = 29500 + (os.getpid() + world * 7 + T) % 2000
    mp.spawn(_worker, args=(world, port, T, nl, rs, sv, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(tmp_path / f"r{r}.pt")
        assert got.shape == ref.shape and torch.equal(got, ref), f"rank {r} differs from the single-rank result"
    # the in-process simulator (used for the single-GPU functional test) follows the same code path
    sim = D.run_simulated(lambda r: ToyBackend(), world, _cfg(T, nl, rs, sv), frames, fm, md)
    assert all(torch.equal(s, ref) for s in sim)
    assert ref.float().std() > 10  # the toy clip is not degenerate
    # ... and not saturated: zeroing the completed flows of the seam region must change the single-rank result
    class ZeroFlows(ToyBackend):
        def make_state(self, enc, flows, md, upd):
            flows = flows.clone()
            flows[:, flows.shape[1] // 2:] = 0
            return super().make_state(enc, flows, md, upd)
    broken = D.run_simulated(lambda r: ZeroFlows(), 1, _cfg(T, nl, rs, sv), frames, fm, md)[0]
    assert not torch.equal(broken, ref), "the toy backend hides wrong flows"



def test_gather_to_root_over_gloo(tmp_path):
    """r04: `gather_root` = 0 (bench.py --gpus N, the node's multi-device mode): the composed frames travel to rank 0 only --
    point-to-point sends instead of an all_gather that hands every rank the whole clip; rank 0 returns the single-rank result,
    the other ranks None."""
    import sys

    sys.path.insert(0, str(Path(__file__).parent))
    from toy_backend import ToyBackend

    world, T, nl, rs, sv = 3, 37, 4, 2, 10
    frames, fm, md = _toy_inputs(T)
    ref = D.run_simulated(lambda r: ToyBackend(), 1, _cfg(T, nl, rs, sv), frames, fm, md)[0]
    port = 29500 + (os.getpid() + 977) % 2000
    mp.spawn(_worker, args=(world, port, T, nl, rs, sv, str(tmp_path), 0), nprocs=world, join=True)
    assert torch.equal(torch.load(tmp_path / "r0.pt"), ref)
    assert torch.load(tmp_path / "r1.pt") is None and torch.load(tmp_path / "r2.pt") is None

@pytest.mark.parametrize("world,T,nl,rs,sv", [
    (4, 13, 4, 2, 3),    # 3-flow chunks: a 5-flow completion halo spans two neighbouring ranks (seam exchange x0)
    (5, 11, 4, 2, 2),    # 2-flow chunks, more ranks than a halo is long
    (8, 17, 6, 3, 100),  # more ranks than chunks: 7 idle ranks take part in every exchange
    (3, 2, 2, 1, 1),     # a single flow
    (2, 23, 10, 5, 7),   # ragged last chunk
    (2, 60, 30, 10, 15), # neighbor_length // 2 = 15 > 10: seam windows read flows beyond the image-propagation halo
    (3, 50, 24, 6, 10),  # the same with rank boundaries that are not window centres
])
def test_sharded_equals_single_rank_edge_cases(world, T, nl, rs, sv):
    """In-process virtual ranks (the same generator the RCCL runner drives) on shard plans with short chunks, idle
    ranks and ragged tails: every rank must return the single-rank result bit for bit."""
    import sys

    sys.path.insert(0, str(Path(__file__).parent))
    from toy_backend import ToyBackend

    frames, fm, md = _toy_inputs(T)
    ref = D.run_simulated(lambda r: ToyBackend(), 1, _cfg(T, nl, rs, sv), frames, fm, md)[0]
    sim = D.run_simulated(lambda r: ToyBackend(), world, _cfg(T, nl, rs, sv), frames, fm, md)
    for r, got in enumerate(sim):
        assert got.shape == ref.shape and torch.equal(got, ref), f"virtual rank {r} differs"



@pytest.mark.parametrize("world,T,nl,rs,sv", [(2, 40, 6, 3, 10), (3, 37, 4, 2, 10), (4, 13, 4, 2, 3), (2, 60, 30, 10, 15), (8, 17, 6, 3, 100)])
def test_multi_device_runner_equals_single_rank(world, T, nl, rs, sv):
    """r04: the in-process multi-device runner behind the node's PP_GPUS=N (one THREAD per device, a mailbox + events + peer
    copies instead of torch.distributed) drives the same generator: over CPU "devices" with the toy backend every plan must
    reproduce the single-rank result bit for bit -- with the gather to rank 0 (the node's mode) and with every rank returning
    the clip; each rank is handed only the frame slab it asked for."""
    import sys

    sys.path.insert(0, str(Path(__file__).parent))
    from toy_backend import ToyBackend

    frames, fm, md = _toy_inputs(T)
    cfg = _cfg(T, nl, rs, sv)
    ref = D.run_simulated(lambda r: ToyBackend(), 1, cfg, frames, fm, md)[0]
    asked = []

    def load_slab(rank, lo, hi, dev):
        asked.append((rank, lo, hi))
        return frames[lo:hi].clone()

    devs = [torch.device("cpu")] * world
    got = D.run_multi_device([ToyBackend() for _ in range(world)], cfg, load_slab, fm, md, devs, gather_root=0)
    assert got.shape == ref.shape and torch.equal(got, ref)
    for rank, lo, hi in asked:
        assert (lo, hi) == D.frames_needed(D.ShardPlan(T, sv, world, rank))
    everyone = D.run_multi_device([ToyBackend() for _ in range(world)], cfg, load_slab, fm, md, devs, gather_root=None)
    assert len(everyone) == world and all(torch.equal(e, ref) for e in everyone)
    # the lock-step simulator follows the gather-to-root protocol too: only the root returns the clip
    sim = D.run_simulated(lambda r: ToyBackend(), world, cfg, frames, fm, md, gather_root=0)
    assert torch.equal(sim[0], ref) and all(s is None for s in sim[1:])


def test_multi_device_runner_surfaces_a_failing_rank():
    """A rank that raises must not leave its peers waiting on the mailbox: the error reaches the caller."""
    import sys

    sys.path.insert(0, str(Path(__file__).parent))
    from toy_backend import ToyBackend

    class Broken(ToyBackend):
        def complete(self, flows, masks):
            raise RuntimeError("rank 1 lost its GPU")

    T, nl, rs, sv = 40, 6, 3, 10
    frames, fm, md = _toy_inputs(T)
    with pytest.raises(RuntimeError):
        D.run_multi_device([ToyBackend(), Broken()], _cfg(T, nl, rs, sv), lambda r, lo, hi, d: frames[lo:hi].clone(), fm, md,
                           [torch.device("cpu")] * 2)


def _assert_same_frames(got, single, exact, what):
    if exact:
        assert torch.equal(got, single), what
    else:
        d = (got.int() - single.int()).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-2, (what, int(d.max()), float((d > 0).float().mean()))


@pytest.mark.gpu
def test_sharded_gpu_pipeline_is_bit_identical_to_single_gpu(hip_lib, monkeypatch):
    from comfyui_propainter_nodes_amd import weights

    pinned = True  # r03: kernel selection depends on the layer, never on a rank's batch -> bit-identical by default
    # (virtual ranks share one model object: its captured recurrence graphs hand out static buffers, which two ranks
    #  advancing in lock-step would overwrite for each other -- a real rank has its own process and models)
    monkeypatch.setenv("PP_GRAPHS", "0")

    g = np.load(GOLD / "e2e_chunked.npz")
    T, H, W, iters, nl, rs, sv, _, _, seed = [int(v) for v in g["params"]]
    dev = torch.device("cuda:0")
    models = pipeline.models_from_state_dicts(weights.synth_state_dicts(seed), dev)
    cfg = pipeline.ProPainterConfig(rs, nl, sv, iters, "enable", T, dev, (W, H))
    fr, fm, md = (torch.from_numpy(g[k]).to(dev) for k in ("frames_u8", "flow_masks", "masks_dilated"))
    single = pipeline.run_inpainting(models, fr, fm, md, cfg)
    for world in (2, 3):
        res = D.run_simulated(lambda r: D.GpuBackend(models, cfg), world, cfg, fr, fm, md)
        for r in range(world):
            _assert_same_frames(res[r].cpu(), single, pinned, f"world {world} rank {r}")


def _gpu_worker(rank, world, port, out_dir, backend="gloo"):
    """One REAL process per rank, the real MI355X backend, a real process group (gloo: both ranks share the one GPU
    of the test box, tensors are staged through the host for the collectives; nccl = RCCL: one GPU per rank)."""
    import torch.distributed as dist

    from comfyui_propainter_nodes_amd import lib, weights

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lib.load()
    g = np.load(GOLD / "e2e_chunked.npz")
    T, H, W, iters, nl, rs, sv, _, _, seed = [int(v) for v in g["params"]]
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    models = pipeline.models_from_state_dicts(weights.synth_state_dicts(seed), dev)
    cfg = pipeline.ProPainterConfig(rs, nl, sv, iters, "enable", T, dev, (W, H))
    fr, fm, md = (torch.from_numpy(g[k]).to(dev) for k in ("frames_u8", "flow_masks", "masks_dilated"))
    res = D.run_distributed(D.GpuBackend(models, cfg), cfg, fr, fm, md)
    torch.save(res.cpu(), Path(out_dir) / f"r{rank}.pt")
    if rank == 0:
        torch.save(pipeline.run_inpainting(models, fr, fm, md, cfg), Path(out_dir) / "single.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_processes_with_the_gpu_backend_match_single_process(hip_lib, tmp_path):
    pinned = True  # bit-identical by default (see above)
    port = 29500 + (os.getpid() * 3 + 11) % 2000
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = torch.load(tmp_path / "single.pt")
    for r in range(2):
        _assert_same_frames(torch.load(tmp_path / f"r{r}.pt"), single, pinned, f"rank {r}")


@pytest.mark.gpu
def test_two_ranks_over_rccl_match_single_process(hip_lib, tmp_path):
    """The RCCL branch of the runner (backend "nccl", device tensors on the wire, grouped isend / irecv + all_gather over
    xGMI): two ranks, one GPU each, against the single-process result.  Skips on a 1-GPU box (gpurun); runs on the driver's
    multi-GPU node so that the branch the SCALE bench uses has been executed before it is timed."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL, one rank per GPU)")
    port = 29500 + (os.getpid() * 5 + 17) % 2000
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path), "nccl"), nprocs=2, join=True)
    single = torch.load(tmp_path / "single.pt")
    for r in range(2):
        _assert_same_frames(torch.load(tmp_path / f"r{r}.pt"), single, True, f"rank {r} (RCCL)")


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_the_sharded_driver(hip_lib, tmp_path):
    """The RCCL code path on the ONE GPU of the test box: a 1-rank "nccl" process group (communicator creation with
    device_id, device tensors on the wire, all_gather / barrier through RCCL; no peer so no point-to-point traffic) driving
    the sharded runner -- the branch `bench.py --gpus N` takes, executed before the driver times it on N GPUs."""
    port = 29500 + (os.getpid() * 7 + 23) % 2000
    mp.spawn(_gpu_worker, args=(1, port, str(tmp_path), "nccl"), nprocs=1, join=True)
    assert torch.equal(torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "single.pt"))



def _multi_device_case(devices):
    from comfyui_propainter_nodes_amd import weights

    g = np.load(GOLD / "e2e_chunked.npz")
    T, H, W, iters, nl, rs, sv, _, _, seed = [int(v) for v in g["params"]]
    dev0 = devices[0]
    sds = weights.synth_state_dicts(seed)
    cfg = pipeline.ProPainterConfig(rs, nl, sv, iters, "enable", T, dev0, (W, H))
    fr_h = torch.from_numpy(g["frames_u8"])
    fm, md = (torch.from_numpy(g[k]).to(dev0) for k in ("flow_masks", "masks_dilated"))
    single = pipeline.run_inpainting(pipeline.models_from_state_dicts(sds, dev0), fr_h.to(dev0), fm, md, cfg)
    # one model object per rank (ranks that share a device must not share captured graphs), graphs ON: the runner gives every
    # rank its own compute stream and captures with thread-local error mode
    backends = []
    for d in devices:
        with torch.cuda.device(d):
            backends.append(D.GpuBackend(pipeline.models_from_state_dicts(sds, d), cfg))
    got = D.run_multi_device(backends, cfg, lambda r, lo, hi, d: fr_h[lo:hi].to(d), fm, md, devices, gather_root=0)
    assert got.device == dev0
    return got.cpu(), single


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_multi_device_runner_virtual_ranks_on_one_gpu(hip_lib, world):
    """The in-process multi-device runner (PP_GPUS=N behind the node) with the real MI355X backend: N rank threads sharing the
    one GPU of the test box, each on its own stream, peer copies degenerate to same-device copies ordered by events; the
    result must equal the single-process pipeline bit for bit."""
    got, single = _multi_device_case([torch.device("cuda:0")] * world)
    _assert_same_frames(got, single, True, f"{world} rank threads on one GPU")


@pytest.mark.gpu
def test_multi_device_runner_on_two_gpus(hip_lib):
    """... and with one GPU per rank thread (hipMemcpyPeer over xGMI) wherever two GPUs are visible (skips on the 1-GPU gpurun box;
    runs on the driver's multi-GPU node)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    got, single = _multi_device_case([torch.device("cuda:0"), torch.device("cuda:1")])
    _assert_same_frames(got, single, True, "one rank thread per GPU")


@pytest.mark.gpu
def test_node_shards_a_long_clip_with_pp_gpus(hip_lib, monkeypatch):
    """The drop-in itself shards (r04): the same node call with PP_GPUS=2 (virtual ranks on the one GPU of the test box) returns
    the IMAGE / masks of the single-GPU call bit for bit; a clip of one sub-video stays on one GPU."""
    from comfyui_propainter_nodes_amd import nodes, synth

    monkeypatch.setenv("PP_ALLOW_SYNTHETIC_WEIGHTS", "1")
    pipeline.drop_model_cache()
    T, H, W = 9, 128, 128
    image, mask = synth.synthetic_clip(T, H, W)
    kw = dict(mask_dilates=2, flow_mask_dilates=3, ref_stride=2, neighbor_length=4, subvideo_length=4, raft_iter=2, fp16="enable")
    node = nodes.ProPainterInpaint()
    want = node.propainter_inpainting(image, mask, W, H, **kw)
    monkeypatch.setenv("PP_GPUS", "2")
    monkeypatch.setenv("PP_GPUS_VIRTUAL", "1")
    cfg = pipeline.ProPainterConfig(2, 4, 4, 2, "enable", T, torch.device("cuda:0"), (W, H))
    assert len(nodes._shard_devices(cfg, torch.device("cuda:0"))) == 2
    got = node.propainter_inpainting(image, mask, W, H, **kw)
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.equal(a.cpu(), b.cpu())
    out = nodes.ProPainterOutpaint().propainter_outpainting(image, W, H, 1.25, 1.0, **kw)
    monkeypatch.delenv("PP_GPUS")
    ref = nodes.ProPainterOutpaint().propainter_outpainting(image, W, H, 1.25, 1.0, **kw)
    assert out[2:] == ref[2:] and torch.equal(out[1].cpu(), ref[1].cpu())
    if not torch.equal(out[0], ref[0]):     # diagnostic: which of the two paths is unstable?
        d = (out[0] - ref[0]).abs()
        ref2 = nodes.ProPainterOutpaint().propainter_outpainting(image, W, H, 1.25, 1.0, **kw)
        monkeypatch.setenv("PP_GPUS", "2")
        out2 = nodes.ProPainterOutpaint().propainter_outpainting(image, W, H, 1.25, 1.0, **kw)
        monkeypatch.setenv("PP_OUTPUT", "device")
        monkeypatch.delenv("PP_GPUS")
        ref3 = nodes.ProPainterOutpaint().propainter_outpainting(image, W, H, 1.25, 1.0, **kw)
        raise AssertionError(("outpaint IMAGE differs", float(d.max()) * 255, float((d > 0).float().mean()),
                              [round(float(v) * 255, 1) for v in d.flatten(1).max(1).values.tolist()],
                              "single again == single", bool(torch.equal(ref2[0], ref[0])), "sharded again == sharded",
                              bool(torch.equal(out2[0], out[0])), "single(PP_OUTPUT=device) == single", bool(torch.equal(ref3[0], ref[0])),
                              "single(device) == sharded", bool(torch.equal(ref3[0], out[0]))))
    monkeypatch.setenv("PP_GPUS", "8")
    short = pipeline.ProPainterConfig(2, 4, 80, 2, "enable", T, torch.device("cuda:0"), (W, H))
    assert len(nodes._shard_devices(short, torch.device("cuda:0"))) == 1     # one sub-video: nothing to shard
    pipeline.drop_model_cache()

@pytest.mark.gpu
def test_bench_spawns_its_own_ranks(hip_lib):
    """`python bench.py --gpus 2` run bare launches two ranks under torch.distributed.run (gloo here: one GPU on the
    test box; the driver's SCALE run uses RCCL, one rank per GPU) and prints one JSON line with n_gpus = 2."""
    import json
    import subprocess
    import sys

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PP_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "40",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["frames_per_gpu"] == 40 and line["value"] > 0
    # r05: the line carries every rank's segment clock of one instrumented step (compute / exchange / wait, bytes sent)
    ranks = line["per_rank"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["compute_total_ms"] > 0 and r["exchanges"] for r in ranks)
    assert any(e["kind"] == "p2p_wait" for e in ranks[0]["exchanges"]) and any(e["MB_sent"] > 0 for e in ranks[1]["exchanges"])
    assert line["host_enqueue_ms"] > 0
