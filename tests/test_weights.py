"""Checkpoint handling: the three files of the reference (utils/model_utils.py:20-43) round-trip
through the strict loader, missing checkpoints fail loudly in the node path, and the per-process model cache is keyed on
device, precision mode and checkpoint identity."""
import os
import time

import pytest
import torch

from comfyui_propainter_nodes_amd import pipeline, weights


def _write_checkpoints(dirpath, sds):
    for net, fname in weights.FILES.items():
        torch.save(sds[net], dirpath / fname)


def test_checkpoint_files_round_trip(tmp_path):
    sds = weights.synth_state_dicts(3)
    # raft-things.pth is a DataParallel state dict: every key carries the `module.` prefix (flow_comp_raft.py:17-19)
    assert all(k.startswith("module.") for k in sds["raft"])
    assert (len(sds["raft"]), len(sds["rfc"]), len(sds["gen"])) == (179, 74, 216)
    _write_checkpoints(tmp_path, sds)
    assert weights.weights_available(tmp_path)
    back = weights.load_state_dicts(tmp_path)
    for net in sds:
        assert list(back[net]) == list(sds[net])
        for k, v in sds[net].items():
            assert torch.equal(back[net][k], v.float() if v.is_floating_point() else v), (net, k)


def test_loader_is_strict(tmp_path):
    sds = weights.synth_state_dicts(0)
    bad = dict(sds["gen"])
    bad.pop("decoder.6.bias")
    with pytest.raises(RuntimeError, match="missing"):
        weights.check_state_dict("gen", bad)
    bad = dict(sds["raft"], **{"module.extra": torch.zeros(1)})
    with pytest.raises(RuntimeError, match="unexpected"):
        weights.check_state_dict("raft", bad)
    bad = dict(sds["rfc"])
    bad["fusion.weight" if "fusion.weight" in bad else next(iter(bad))] = torch.zeros(3)
    with pytest.raises(RuntimeError):
        weights.check_state_dict("rfc", bad)
    with pytest.raises(FileNotFoundError, match="raft-things.pth"):
        weights.load_state_dicts(tmp_path)  # nothing there: no silent fallback


class _Dummy:
    built = 0

    def __init__(self, sd, device, dtype=None):
        _Dummy.built += 1
        self.n = len(sd)


@pytest.fixture()
def dummy_networks(monkeypatch, tmp_path):
    for name in ("RaftFlow", "FlowCompleter", "InpaintGeneratorMI355"):
        monkeypatch.setattr(pipeline, name, _Dummy)
    monkeypatch.setattr(weights, "WEIGHT_DIR", tmp_path)
    monkeypatch.delenv("PP_ALLOW_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.delenv("PP_F32_GEMM", raising=False)
    pipeline.drop_model_cache()
    _Dummy.built = 0
    yield tmp_path
    pipeline.drop_model_cache()


def test_missing_checkpoints_fail_in_the_node_path(dummy_networks, monkeypatch):
    with pytest.raises(FileNotFoundError, match="raft-things.pth"):
        pipeline.initialize_models(torch.device("cpu"))
    monkeypatch.setenv("PP_ALLOW_SYNTHETIC_WEIGHTS", "1")       # explicit opt-in (bench / tests)
    m = pipeline.initialize_models(torch.device("cpu"))
    assert m.provenance.startswith("synthetic")


def test_model_cache_key(dummy_networks, monkeypatch):
    monkeypatch.setenv("PP_ALLOW_SYNTHETIC_WEIGHTS", "1")
    dev = torch.device("cpu")
    a = pipeline.initialize_models(dev)
    assert pipeline.initialize_models(dev) is a and _Dummy.built == 3          # second execution: cached
    assert pipeline.initialize_models(dev, seed=1) is not a                    # another seed is another model
    # checkpoints dropped into weights/ after a synthetic run are picked up ...
    _write_checkpoints(dummy_networks, weights.synth_state_dicts(5))
    b = pipeline.initialize_models(dev)
    assert b is not a and b.provenance == "pretrained"
    assert pipeline.initialize_models(dev) is b
    # ... and so is a replaced file (mtime / size are part of the key)
    f = dummy_networks / weights.FILES["gen"]
    os.utime(f, ns=(time.time_ns(), f.stat().st_mtime_ns + 10_000_000))
    assert pipeline.initialize_models(dev) is not b
    # RAFT arithmetic mode is part of the key
    c = pipeline.initialize_models(dev)
    monkeypatch.setenv("PP_F32_GEMM", "exact")
    assert pipeline.initialize_models(dev) is not c
