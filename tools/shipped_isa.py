"""Disassembles the gfx950 code objects of the SHIPPED libpropainter_mi355.so (no recompilation): the .hip_fatbin section is a
sequence of clang offload bundles, one per translation unit; each holds one hipv4-amdgcn-amd-amdhsa--gfx950 entry.

    python tools/shipped_isa.py [--barriers]        # kernel count per code object / the barrier audit below
    python tools/shipped_isa.py --resources         # registers, LDS, scratch and spills of every kernel (markdown table)

audit_barriers(): the r04 race.  A kernel that restages an LDS buffer by LDS-DMA (global_load_lds) right after a barrier must
have retired its own ds_reads of that buffer BEFORE the barrier: hipcc is free to sink the `s_waitcnt lgkmcnt(0)` of the last
fragment reads below `s_barrier` (their values are only needed by later MFMAs), after which a faster wave's DMA overwrites rows a
slower wave is still reading -- 0.2-1 % of the launches of conv_halo_f16_ct_kernel gave different bits next to a busy stream
(tools/diag_kernels_under_load.py).  pp_barrier() (csrc/pp_device.h) therefore carries its own wait; the audit checks the emitted
code: walking back from every s_barrier of a kernel that uses LDS-DMA, an `s_waitcnt ... lgkmcnt(0)` must come before any ds_read
or block boundary."""
from __future__ import annotations

import re
import shutil
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "comfyui_propainter_nodes_amd" / "libpropainter_mi355.so"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _tool(name: str) -> str | None:
    for cand in (Path("/opt/rocm/lib/llvm/bin") / name, Path("/opt/rocm/llvm/bin") / name):
        if cand.exists():
            return str(cand)
    return shutil.which(name)


def tools_available() -> bool:
    return _tool("llvm-objcopy") is not None and _tool("llvm-objdump") is not None


def code_objects(lib: Path = LIB) -> list[bytes]:
    """The gfx950 ELF images inside the library, in link order."""
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run([_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
        blob = fat.read_bytes()
    out = []
    at = blob.find(MAGIC)
    while at >= 0:
        p = at + len(MAGIC)
        (entries,) = struct.unpack_from("<Q", blob, p)
        p += 8
        for _ in range(entries):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tlen].decode()
            p += tlen
            if "gfx950" in triple and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(MAGIC, at + 1)
    return out


def disassemble(image: bytes) -> dict[str, list[str]]:
    """symbol -> instruction lines (mnemonic first) of one code object."""
    with tempfile.TemporaryDirectory() as td:
        obj = Path(td) / "co.o"
        obj.write_bytes(image)
        text = subprocess.run([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", str(obj)],
                              check=True, capture_output=True, text=True).stdout
    kernels: dict[str, list[str]] = {}
    cur = None
    for ln in text.split("\n"):
        m = re.match(r"^(?:[0-9a-f]+ )?<([^>]+)>:", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None and ln.strip():
            kernels[cur].append(ln.strip())
    return kernels


def audit_packed_f32_op_sel(lib: Path = LIB) -> dict:
    """r06: the MI355X returns a wrong LOW half for lanes 48..63 of `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (a packed fp32
    instruction whose low lane reads the HIGH dword of an operand) when MFMA instructions of another wave share the SIMD
    (tools/probes/pk_f32_next_to_mfma.hip; profiles/r06_pk_f32_op_sel_erratum.md).  No kernel of the library may contain a packed
    fp32 instruction with a set `op_sel` bit (the op_sel_hi broadcast forms are fine: the low lane reads the low dword).
    -> {"kernels": n, "packed_fp32": n, "violations": [(kernel, instruction)]}"""
    res = {"kernels": 0, "packed_fp32": 0, "violations": []}
    for image in code_objects(lib):
        for name, ins in disassemble(image).items():
            res["kernels"] += 1
            for x in ins:
                if not re.match(r"v_pk_(add|mul|fma|min|max)_f32\b", x):
                    continue
                res["packed_fp32"] += 1
                m = re.search(r"op_sel:\[([01,]+)\]", x)
                if m and "1" in m.group(1):
                    res["violations"].append((name, x.strip()))
    return res


def audit_barriers(lib: Path = LIB) -> dict:
    """-> {"dma_kernels": n, "barriers": n, "violations": [(kernel, instruction index, what was found first)]}"""
    res = {"dma_kernels": 0, "barriers": 0, "violations": []}
    for image in code_objects(lib):
        for name, ins in disassemble(image).items():
            if not any(x.startswith("global_load_lds") for x in ins):
                continue
            res["dma_kernels"] += 1
            for i, x in enumerate(ins):
                if not x.startswith("s_barrier"):
                    continue
                res["barriers"] += 1
                found = "start of the kernel"
                for j in range(i - 1, -1, -1):
                    y = ins[j]
                    if y.startswith("s_waitcnt") and "lgkmcnt(0)" in y:
                        found = None
                        break
                    if y.startswith(("ds_read", "ds_load")):
                        found = "a pending " + y.split()[0]
                        break
                    if y.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
                        found = "a block boundary (" + y.split()[0] + ")"
                        break
                if found is not None:
                    res["violations"].append((name, i, found))
    return res


def kernel_resources(lib: Path = LIB) -> list[dict]:
    """One record per kernel from the code objects' metadata notes (what the loader uses to size a wave)."""
    rows = []
    keys = {"name": ".name", "vgpr": ".vgpr_count", "agpr": ".agpr_count", "sgpr": ".sgpr_count",
            "lds_static": ".group_segment_fixed_size", "scratch": ".private_segment_fixed_size",
            "vgpr_spills": ".vgpr_spill_count", "sgpr_spills": ".sgpr_spill_count", "max_wg": ".max_flat_workgroup_size"}
    for image in code_objects(lib):
        with tempfile.TemporaryDirectory() as td:
            obj = Path(td) / "co.o"
            obj.write_bytes(image)
            notes = subprocess.run([_tool("llvm-readelf"), "--notes", str(obj)], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n  - (?=\.agpr_count:)", notes)[1:]:
            rec = {}
            for k, field in keys.items():
                m = re.search(r"(?:^|\n)\s*" + re.escape(field) + r":\s+(\S+)", blk)
                v = m.group(1) if m else "0"
                rec[k] = v if k == "name" else int(v)
            rows.append(rec)
    return rows


def _demangle(names: list[str]) -> list[str]:
    tool = _tool("llvm-cxxfilt") or shutil.which("c++filt")
    if tool is None:
        return names
    out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return [re.sub(r"^void ", "", x) for x in out[:len(names)]]


def main(argv: list[str]) -> int:
    if "--resources" in argv:
        rows = sorted(kernel_resources(), key=lambda r: (-(r["vgpr"] + r["agpr"]), r["name"]))
        names = _demangle([r["name"] for r in rows])
        print("| kernel | VGPR | AGPR | waves / SIMD | SGPR | static LDS | scratch B | VGPR spills | SGPR spills |")
        print("|---|---|---|---|---|---|---|---|---|")
        for r, n in zip(rows, names):
            regs = max(r["vgpr"] + r["agpr"], 1)
            alloc = (regs + 7) // 8 * 8
            waves = min(8, 512 // alloc)
            print(f"| `{n[:110]}` | {r['vgpr']} | {r['agpr']} | {waves} | {r['sgpr']} | {r['lds_static']} | {r['scratch']} | "
                  f"{r['vgpr_spills']} | {r['sgpr_spills']} |")
        return 0
    if "--barriers" in argv:
        r = audit_barriers()
        print(f"{r['dma_kernels']} kernels with LDS-DMA, {r['barriers']} barriers, {len(r['violations'])} without a retired-reads wait")
        for v in r["violations"]:
            print("  ", *v)
        return 1 if r["violations"] else 0
    for k, image in enumerate(code_objects()):
        ks = disassemble(image)
        print(f"code object {k}: {len(image)} bytes, {len(ks)} symbols")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
