"""pytest configuration.

Markers:
  gpu  -- needs a real MI355X (run by the driver with `-m gpu` on the GPU box); these are the
          parity tests proper and call the gfx950 library through the C ABI.
Everything else runs on CPU: oracle-vs-golden checks, host logic, ABI/export checks and the
kernel-logic checks that execute the kernel sources under the x86 emulator (tests/emu).
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests" / "emu") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests" / "emu"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture()
def pp_knobs():
    """Set PP_CONV_* knobs for ONE test inside this process: the library caches them at first use (csrc/pp_options.h), so
    the environment is changed, the library told to read it again, and both are undone afterwards."""
    from comfyui_propainter_nodes_amd import lib

    saved = {}

    def set_knobs(**kw):
        for k, v in kw.items():
            saved.setdefault(k, os.environ.get(k))
            os.environ[k] = v
        lib.reload_options()

    yield set_knobs
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    lib.reload_options()


@pytest.fixture()
def emu_lib():
    """Build (if needed) and load the x86 emulation of the kernel sources. TEST ONLY."""
    import emu_loader
    from comfyui_propainter_nodes_amd import lib

    L = emu_loader.load_emulator()
    yield L
    lib.unload()


@pytest.fixture(scope="session")
def hip_lib():
    """Load the gfx950 library; building it first when hipcc is present."""
    import torch

    from comfyui_propainter_nodes_amd import build, lib

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    if not lib.HIP_LIB.exists():
        build.build_hip()
    lib.unload()
    return lib.load()


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """Run a kernel test on the emulator (CPU suite) and on the MI355X (gpu suite)."""
    import torch

    from comfyui_propainter_nodes_amd import build, lib

    if request.param == "emu":
        import emu_loader

        emu_loader.load_emulator()
        yield torch.device("cpu")
        lib.unload()
    else:
        if not torch.cuda.is_available():
            pytest.skip("no GPU visible")
        if not lib.HIP_LIB.exists():
            build.build_hip()
        lib.unload()
        lib.load()
        yield torch.device("cuda:0")
        torch.cuda.synchronize()
