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


def test_every_barrier_of_an_lds_dma_kernel_retires_its_lds_reads_first():
    """The r04 race (csrc/pp_device.h: pp_barrier): in a kernel that restages LDS buffers by global_load_lds, an
    `s_waitcnt lgkmcnt(0)` must stand between a wave's last ds_reads and the barrier that lets the other waves overwrite
    the buffer -- hipcc sinks that wait below the barrier when left to itself.  Audits the disassembly of the code objects
    INSIDE the shipped library (tools/shipped_isa.py; no recompilation; the pre-fix objects fail it: 52 of 56 barriers of
    conv_halo_f16.hip)."""
    import shipped_isa as S

    from comfyui_propainter_nodes_amd import build

    if not S.tools_available():
        pytest.skip("no llvm-objcopy / llvm-objdump")
    build.build_hip()
    r = S.audit_barriers()
    assert r["dma_kernels"] >= 80 and r["barriers"] >= 250, r   # every LDS-DMA kernel family is inside the library
    assert not r["violations"], r["violations"][:10]


def test_no_packed_fp32_instruction_swaps_operand_halves():
    """r06 (profiles/r06_pk_f32_op_sel_erratum.md): on the MI355X `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` -- hipcc's SLP
    vectoriser emits it for crossed scalar pairs -- returns a wrong low half for lanes 48..63 of a wave while MFMA instructions of
    another wave share the SIMD: pp_deform_cols computed wrong sample positions whenever a convolution ran next to it on a second
    stream (tools/diag_deform_debug.py; minimal reproducer without memory traffic: tools/probes/pk_f32_next_to_mfma.hip).  The
    library is built with -fno-slp-vectorize; this audits the code objects INSIDE the shipped library: no packed fp32 instruction
    with a set op_sel bit."""
    import shipped_isa as S

    from comfyui_propainter_nodes_amd import build

    if not S.tools_available():
        pytest.skip("no llvm-objcopy / llvm-objdump")
    build.build_hip()
    r = S.audit_packed_f32_op_sel()
    assert r["kernels"] >= 150, r["kernels"]
    assert not r["violations"], r["violations"][:10]
