"""CPU-side checks of the boundary: the library loads without a GPU and exports exactly the
symbols include/nerf_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "nerf_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nrf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from nerf_amd import _lib
    names = _declared()
    assert sorted(_lib.SYMBOLS) == names
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n


def test_version_and_strerror_without_gpu():
    from nerf_amd import _lib
    L = _lib.lib()
    assert L.nrf_version() >= 100
    assert L.nrf_strerror(0) == b"ok"
    assert b"NULL" in L.nrf_strerror(-1)
    assert L.nrf_packed_floats() == (2320 + 8) * 256 + 9 * 256 + 128 + 256 + 16 + 384 + 16
    assert L.nrf_render_workspace_bytes(1024, 64, 128) > 1024 * 192 * 16


def test_null_and_shape_validation_without_gpu():
    """Argument validation happens on the host before any launch, so it is testable on CPU."""
    from nerf_amd import _lib
    L = _lib.lib()
    assert L.nrf_mlp_forward_rays(None, None, None, None, 0, 4, 64, None, None) == -1
    assert L.nrf_mlp_forward_rays(None, None, None, None, 0, 4, 48, None, None) == -2     # 48 % 32 != 0
    assert L.nrf_mlp_forward_rays(None, None, None, None, 0, 0, 64, None, None) == 0      # empty batch
    assert L.nrf_composite(None, None, 0, 1, 64, 1, None, None, None, None, None) == -1
    assert L.nrf_render(None, None) == -1
    a = _lib.RenderArgs()
    assert L.nrf_render(ctypes.byref(a), None) == -1
