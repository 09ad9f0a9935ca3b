"""ctypes binding of libpropainter_mi355.so (the C ABI in include/propainter_mi355.h).

The ctypes mirrors of the parameter structs are generated from the header itself, and their
sizes are cross-checked against `pp_struct_size()` at load time, so the Python side can
never silently drift from the C side.

The product path loads ONLY the gfx950 library and raises loudly when it is missing.
(`Library.is_emulator` marks a library that takes host pointers: the CPU test-suite installs an x86
emulation build of the same kernel sources from tests/emu/loader.py; nothing here loads it.)
"""
from __future__ import annotations

import ctypes
import re
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
HEADER = ROOT / "include" / "propainter_mi355.h"
HIP_LIB = PKG / "libpropainter_mi355.so"

_CT = {
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "const void*": ctypes.c_void_p,
    "void*": ctypes.c_void_p,
}


class ABIError(RuntimeError):
    pass


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def parse_header(path: Path = HEADER):
    """Return (constants, structs, functions) parsed from the C header."""
    text = _strip_comments(path.read_text())
    consts: dict[str, int] = {}
    for m in re.finditer(r"#define\s+(PP_\w+)\s+(-?\d+)", text):
        consts[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"enum\s+\w+\s*\{(.*?)\}", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = item.split("=")
                nxt = int(v.strip(), 0)
                consts[k.strip()] = nxt
            else:
                consts[item] = nxt
            nxt += 1
    structs: dict[str, type] = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            dm = re.match(r"^(const void\*|void\*|int32_t|int64_t|float)\s+(.*)$", decl)
            if not dm:
                raise ABIError(f"unsupported member declaration in {m.group(2)}: '{decl}'")
            ctype = _CT[dm.group(1)]
            for name in dm.group(2).split(","):
                name = name.strip()
                am = re.match(r"^(\w+)\[(\w+)\]$", name)
                if am:
                    n = am.group(2)
                    n = consts[n] if n in consts else int(n)
                    fields.append((am.group(1), ctype * n))
                else:
                    fields.append((name, ctype))
        structs[m.group(2)] = type(m.group(2), (ctypes.Structure,), {"_fields_": fields})
    funcs: list[str] = re.findall(r"\b(pp_\w+)\s*\(", re.sub(r"typedef\s+struct.*?;\s*\n", "", text, flags=re.S))
    funcs = sorted({f for f in funcs if f not in structs})
    return consts, structs, funcs


CONSTS, STRUCTS, FUNCS = parse_header()
globals().update(CONSTS)


class Library:
    """A loaded libpropainter build (gfx950 product build, or the x86 emulator in tests)."""

    def __init__(self, path: Path, is_emulator: bool):
        if not path.exists():
            raise ABIError(
                f"{path} is missing. The MI355X hot path has no CPU fallback: build it with "
                f"`python -m comfyui_propainter_nodes_amd.build` (needs hipcc, gfx950)."
            )
        self.path = path
        self.is_emulator = is_emulator
        self.cdll = ctypes.CDLL(str(path))
        self.cdll.pp_last_error.restype = ctypes.c_char_p
        self.cdll.pp_struct_size.restype = ctypes.c_int64
        self.cdll.pp_struct_size.argtypes = [ctypes.c_char_p]
        self.cdll.pp_version.restype = ctypes.c_int32
        self.cdll.pp_reload_options.restype = None
        if self.cdll.pp_version() != CONSTS["PP_ABI_VERSION"]:
            raise ABIError("libpropainter ABI version mismatch with the header")
        for name, st in STRUCTS.items():
            got = self.cdll.pp_struct_size(name.encode())
            if got != ctypes.sizeof(st):
                raise ABIError(f"struct {name}: library says {got} bytes, ctypes mirror has {ctypes.sizeof(st)}")
        for f in FUNCS:
            if not hasattr(self.cdll, f):
                raise ABIError(f"{path.name} does not export {f}")

    def call(self, fname: str, stream, params) -> None:
        rc = getattr(self.cdll, fname)(ctypes.c_void_p(stream), ctypes.byref(params))
        if rc != 0:
            raise RuntimeError(f"{fname} failed ({rc}): {self.cdll.pp_last_error().decode()}")


    def call2(self, fname: str, stream, p1, p2) -> None:
        """Entry points that take two parameter blocks (pp_deform_conv: the sampling block and the convolution block)."""
        rc = getattr(self.cdll, fname)(ctypes.c_void_p(stream), ctypes.byref(p1), ctypes.byref(p2))
        if rc != 0:
            raise RuntimeError(f"{fname} failed ({rc}): {self.cdll.pp_last_error().decode()}")


_lib: Library | None = None


def load() -> Library:
    """Load the gfx950 library (product path). Raises if it has not been built."""
    global _lib
    if _lib is None:
        _lib = Library(HIP_LIB, is_emulator=False)
    return _lib


def reload_options() -> None:
    """Re-read the PP_CONV_* knobs from the environment (the library caches them at first use)."""
    if _lib is not None:
        _lib.cdll.pp_reload_options()


def unload() -> None:
    global _lib
    _lib = None


def current() -> Library:
    return load() if _lib is None else _lib
