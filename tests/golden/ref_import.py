"""Import the upstream reference (/root/reference) on CPU -- BUILD-CONTAINER ONLY.

Used exclusively by tests/golden/make_golden.py to mint the committed fixtures and by the
optional `test_oracle_vs_reference_live` checks; nothing in the GPU suite, smoke() or
bench.py touches /root/reference.  The reference does not import here as-is;
four shims are injected first:
  1. `comfy.model_management.get_torch_device` -> cpu
  2. `cv2` -> MagicMock (only touched at import time by RAFT/utils/frame_utils.py)
  3. `torchvision` -> stub exposing ops.deform_conv2d (our restatement of its documented
     contract, oracle/ops.py), transforms.Compose, transforms.functional.to_pil_image
  4. `torch.__version__` temporarily "2.10.0" (model/misc.py parses it with a regex)
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path
from unittest import mock

import torch

REF_ROOT = Path("/root/reference")


def available() -> bool:
    return (REF_ROOT / "propainter_nodes.py").exists()


def load_reference():
    """Return the imported `reference` package (cached in sys.modules)."""
    if "reference" in sys.modules and hasattr(sys.modules["reference"], "propainter_nodes"):
        return sys.modules["reference"]
    if not available():
        raise RuntimeError("/root/reference is not present on this machine")
    root = Path(__file__).resolve().parents[2]
    if str(root) not in sys.path:
        sys.path.insert(0, str(root))
    from oracle import ops as oracle_ops

    comfy = types.ModuleType("comfy")
    mm = types.ModuleType("comfy.model_management")
    mm.get_torch_device = lambda: torch.device("cpu")
    comfy.model_management = mm
    sys.modules["comfy"] = comfy
    sys.modules["comfy.model_management"] = mm
    sys.modules.setdefault("cv2", mock.MagicMock())

    tv = types.ModuleType("torchvision")
    tv_ops = types.ModuleType("torchvision.ops")
    tv_ops.deform_conv2d = oracle_ops.deform_conv2d
    tv_tr = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, fns):
            self.fns = fns

        def __call__(self, x):
            for f in self.fns:
                x = f(x)
            return x

    tv_tr.Compose = Compose
    tv_fn = types.ModuleType("torchvision.transforms.functional")

    def to_pil_image(t):
        from PIL import Image

        a = t.numpy()
        if a.ndim == 3:
            a = a[0]
        return Image.fromarray(a)

    tv_fn.to_pil_image = to_pil_image
    tv_tr.functional = tv_fn
    tv.ops, tv.transforms = tv_ops, tv_tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tv_ops
    sys.modules["torchvision.transforms"] = tv_tr
    sys.modules["torchvision.transforms.functional"] = tv_fn

    if "/root" not in sys.path:
        sys.path.insert(0, "/root")
    real_version = torch.__version__
    torch.__version__ = "2.10.0"
    try:
        ref = importlib.import_module("reference")
        importlib.import_module("reference.propainter_nodes")
    finally:
        torch.__version__ = real_version
    return ref
