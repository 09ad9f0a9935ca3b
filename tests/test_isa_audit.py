"""The PP_F32X2 convolution hides its pixel loads from hipcc's s_waitcnt bookkeeping (inline asm, counted by hand):
the emitted gfx950 ISA must not touch a load's destination registers before the counted wait on ANY path, must not
use scratch memory and must not contain calls.  Cross-compiles on CPU (no GPU needed)."""
import re
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="no hipcc")
@pytest.mark.parametrize("source,nkernels", [("conv_split.hip", 9), ("conv_halo.hip", 9)])
def test_hidden_loads_are_never_touched_in_flight(source, nkernels):
    """Audits the ISA of the PRODUCT build: build.build_hip() keeps the gfx950 assembly of these translation units next
    to their objects (a no-op when the library is up to date, a cross-compile otherwise)."""
    import audit_hidden_loads as A

    from comfyui_propainter_nodes_amd import build

    build.build_hip()
    asm = build.isa_path(Path(source).stem)
    assert asm.exists(), asm
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN2pp\w*conv_(?:halo_)?split(?:_ct|_tall)?_kernel\w+):", text, flags=re.M)
    assert len(kernels) == nkernels  # conv_split: 7 flat tiles + the 8-wave and 16-pixel tiles; conv_halo: 128 / 96 / 64 channels x (3x3, 1x5, 5x1) (PP_F32X2 form; the f16 form in conv_halo_f16.hip has no hidden loads); conv_halo_tall: 128 / 96 channels x (3x3, 1x5, 5x1)
    assert "global_load_lds_dwordx4" in text and ";;#ASMSTART" in text
    assert A.main(str(asm)) == 0
    assert "s_swappc" not in text  # no real calls: helper lambdas are always inlined
    scratch = re.findall(r"\.private_segment_fixed_size: (\d+)", text)
    assert scratch and all(int(v) == 0 for v in scratch), scratch
