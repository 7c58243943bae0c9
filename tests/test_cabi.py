"""C-ABI contract: the hipcc-built library loads and exports every symbol include/pcdm.h declares
(no compute calls here -- there is no GPU on the CPU test box); argument validation paths return -1;
the emulator build exports the same set."""
from __future__ import annotations

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "pcdm.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pcdm_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from pcdms_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_product_library_exports_every_symbol():
    from pcdms_amd.build import build_lib
    lib = ctypes.CDLL(str(build_lib()))
    for sym in _declared():
        assert hasattr(lib, sym), sym
    assert lib.pcdm_is_emulator() == 0 and lib.pcdm_version() >= 1
    # argument validation happens before any launch: safe without a GPU
    assert lib.pcdm_gemm(None, None) == -1
    lib.pcdm_groupnorm_ws_floats.restype = ctypes.c_int64
    assert lib.pcdm_groupnorm_ws_floats(8, 320) > 0


def test_emulator_exports_every_symbol():
    from tests.emu import build_emu
    lib = build_emu.load()
    for sym in _declared():
        assert hasattr(lib, sym), sym
    assert lib.pcdm_is_emulator() == 1


def test_missing_library_fails_loudly(tmp_path):
    from pcdms_amd import _lib
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load(tmp_path / "libpcdm.so")
