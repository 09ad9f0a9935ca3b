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
    # every pp_<op> entry point has a matching pp_<op>_params struct (pp_deform_conv, the fused form of pp_deform_cols +
    # pp_conv2d, takes the parameter blocks of those two; r06 pp_corr_lookup_conv those of pp_corr_lookup + pp_conv2d)
    assert "pp_corr_lookup_conv" in funcs
    for f in funcs:
        if f not in ("pp_version", "pp_last_error", "pp_struct_size", "pp_reload_options", "pp_deform_conv", "pp_corr_lookup_conv"):
            assert f + "_params" in structs, f
    assert "pp_deform_conv" in funcs


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


def test_knobs_are_cached_until_reload(emu_lib, capfd, monkeypatch):
    """csrc/pp_options.h: the PP_CONV_* knobs are read from the environment once; a change is seen only after
    pp_reload_options() (lib.reload_options(), what the `pp_knobs` fixture of the tests calls)."""
    import torch

    from comfyui_propainter_nodes_amd import lib, ops

    def run():
        g = torch.Generator().manual_seed(1)
        x = torch.randn(1, 12, 20, 32, generator=g)
        spec = ops.make_conv_spec(torch.randn(64, 32, 3, 3, generator=g) * 0.1, None, torch.float32, padding=1, split=True)
        out = torch.empty(1, 12, 20, 64)
        ops.conv2d(spec, [x], out)
        return capfd.readouterr().err

    monkeypatch.setenv("PP_CONV_HALO", "force")
    lib.reload_options()
    assert "halo-tile kernel" not in run()               # PP_CONV_TRACE is not set: nothing is printed
    monkeypatch.setenv("PP_CONV_TRACE", "1")
    assert "halo-tile kernel" not in run()               # ... and setting it is not seen: the knobs are cached
    lib.reload_options()
    assert "halo-tile kernel" in run()                   # ... until the library is told to read them again
    monkeypatch.delenv("PP_CONV_TRACE")
    monkeypatch.delenv("PP_CONV_HALO")
    lib.reload_options()
    assert "halo-tile kernel" not in run()
