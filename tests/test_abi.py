"""The C-ABI shared library loads and exports every symbol include/propainter_mi355.h declares (no compute)."""
import ctypes

import pytest

from comfyui_propainter_nodes_amd import build, lib


def test_header_parses_to_structs_and_functions():
    consts, structs, funcs = lib.parse_header()
    assert consts["PP_ABI_VERSION"] >= 1 and consts["PP_MAX_SEG"] == 4
    assert "pp_conv2d_params" in structs and "pp_window_attention_params" in structs
    for f in ("pp_conv2d", "pp_corr_lookup", "pp_deform_cols", "pp_img_prop_step", "pp_window_attention", "pp_compose_u8"):
        assert f in funcs
    # every pp_<op> entry point has a matching pp_<op>_params struct
    for f in funcs:
        if f not in ("pp_version", "pp_last_error", "pp_struct_size", "pp_reload_options"):
            assert f + "_params" in structs, f


def test_gfx950_library_builds_loads_and_exports_everything():
    """hipcc cross-compiles for gfx950 without a GPU; loading needs no device either."""
    build.build_hip()
    L = lib.Library(lib.HIP_LIB, is_emulator=False)  # checks ABI version, struct sizes, all exports
    for f in lib.FUNCS:
        assert hasattr(L.cdll, f)
    assert L.cdll.pp_struct_size(b"no_such_struct") == -1


def test_product_path_fails_loudly_without_the_extension(tmp_path, monkeypatch):
    monkeypatch.setattr(lib, "HIP_LIB", tmp_path / "libpropainter_mi355.so")
    lib.unload()
    with pytest.raises(lib.ABIError, match="no CPU fallback"):
        lib.load()
    lib.unload()


def test_gfx950_library_rejects_host_tensors():
    """Product library + CPU tensor must raise, never fall back."""
    import torch

    from comfyui_propainter_nodes_amd import ops

    build.build_hip()
    lib.unload()
    lib.load()
    try:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ops.avgpool2x2(torch.zeros(1, 4, 4), torch.zeros(1, 2, 2))
    finally:
        lib.unload()


def test_bad_arguments_return_error_codes(emu_lib):
    P = lib.STRUCTS["pp_conv2d_params"]()
    rc = emu_lib.cdll.pp_conv2d(ctypes.c_void_p(0), ctypes.byref(P))
    assert rc == lib.CONSTS["PP_ERR_BAD_ARG"]
    assert b"nseg" in emu_lib.cdll.pp_last_error()
