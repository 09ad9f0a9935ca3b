"""TEST INFRASTRUCTURE ONLY: build and load the x86 emulation of the kernel sources (csrc/*.hip compiled with -DPP_EMU
against tests/emu/pp_emu.h).  The product package has no reference to this library; tests install it as the active
`Library` object so that the same host code (ops.py, raft.py, ...) drives the emulated kernels with host pointers."""
from __future__ import annotations

import os
from pathlib import Path

from comfyui_propainter_nodes_amd import build as B
from comfyui_propainter_nodes_amd import lib

EMU_DIR = Path(__file__).resolve().parent
# PP_EMU_DEFINES="-D<hook>": the emulation of an experiment build (csrc hooks compiled in by tools/build_variant.sh for
# the GPU) in its own object directory and library, so that a kernel variant is checked on CPU before it is timed on the MI355X
VARIANT_DEFINES = os.environ.get("PP_EMU_DEFINES", "").split()
_TAG = "".join(c if c.isalnum() else "_" for c in "_".join(VARIANT_DEFINES))
EMU_LIB = EMU_DIR / ("libpropainter_emu.so" if not VARIANT_DEFINES else f"libpropainter_emu{_TAG}.so")
HOST_CLANG = os.environ.get("PP_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


def build_emu(force: bool = False) -> Path:
    flags = ["-O3", "-mavx2", "-mf16c", "-std=c++17", "-fPIC", "-DPP_EMU", "-x", "c++", "-Wno-unused-value", "-ffp-contract=off",
             "-I", str(B.CSRC), "-I", str(B.ROOT / "include"), "-I", str(EMU_DIR), *VARIANT_DEFINES]

    def compile_one(src: Path, obj: Path) -> None:
        B._run([HOST_CLANG, *flags, "-c", str(src), "-o", str(obj)])

    def link(objs: list[Path], out: Path) -> None:
        rt = objs[0].parent / "pp_emu_rt.o"
        B._run([HOST_CLANG, "-O2", "-std=c++17", "-fPIC", "-I", str(EMU_DIR), "-c",
                str(EMU_DIR / "pp_emu.cpp"), "-o", str(rt)])
        B._run([HOST_CLANG, "-shared", "-fPIC", *map(str, objs), str(rt), "-lpthread", "-o", str(out)])

    extra = "emu" + " ".join(flags) + (EMU_DIR / "pp_emu.h").read_text() + (EMU_DIR / "pp_emu.cpp").read_text()
    return B._compile_all(EMU_DIR / "build" / _TAG if VARIANT_DEFINES else EMU_DIR / "build", compile_one, link, EMU_LIB, extra, force, extra_dirs=(EMU_DIR,))


def load_emulator() -> lib.Library:
    """Build (if needed) and install the emulator as the active library."""
    build_emu()
    lib._lib = lib.Library(EMU_LIB, is_emulator=True)
    return lib._lib
