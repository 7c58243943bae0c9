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


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors in pcdms_amd/_lib.py against the C structs of include/pcdm.h as gcc lays them out: size and the offset of every
    field (pcdm_gemm_params grew without a version bump -- an ABI drift between the header and a binding must fail a test, and
    ``pcdm_version()`` must say 5 for the struct that starts with ``struct_size`` and ends with ``a3``, ``lda3``)."""
    import shutil
    import subprocess

    from pcdms_amd import _lib
    from pcdms_amd.build import build_lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    structs = {"pcdm_gemm_params": _lib.GemmParams, "pcdm_gn_splitk_src": _lib.GnSplitKSrc, "pcdm_unet_config": _lib.UNetConfig}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT / "include" / "pcdm.h"}"', "int main(void) {"]
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call([gcc, "-std=c11", "-Wall", "-Werror", str(src), "-o", str(exe)])   # (also: the header is valid C11 on its own)
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == ctypes.sizeof(ct), (cname, got[cname], ctypes.sizeof(ct))
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, (cname, fname)
    # every field of the header's struct is mirrored (a field added to the header only would shift nothing above but be left unset)
    hdr = (ROOT / "include" / "pcdm.h").read_text()
    body = re.sub(r"/\*.*?\*/", "", hdr[hdr.index("typedef struct pcdm_gemm_params {"):hdr.index("} pcdm_gemm_params;")], flags=re.S)
    names = re.findall(r"(\w+)\s*(?:,|;)", body.split("{", 1)[1])
    assert names == [f for f, _ in _lib.GemmParams._fields_], (names, [f for f, _ in _lib.GemmParams._fields_])
    assert ctypes.CDLL(str(build_lib())).pcdm_version() == 5


def test_gemm_params_of_another_size_are_refused(monkeypatch):
    """ABI 4: ``pcdm_gemm_params.struct_size`` is the first field; a host compiled against an older (shorter) or newer header is refused with -1
    before any other field is read (version agreement used to be advisory).  On the emulator build: the SAME call succeeds with
    the right size and is refused with any other; on the product library (no GPU here) the refusal precedes every launch."""
    import torch

    from pcdms_amd import _lib, ops
    from pcdms_amd.build import build_lib
    from tests.emu import build_emu
    assert _lib.GemmParams.struct_size.offset == 0 and _lib.GemmParams().struct_size == ctypes.sizeof(_lib.GemmParams)
    _lib.use_library(build_emu.load())
    a = torch.randn(64, 64).to(torch.bfloat16)
    pw = ops.pack_linear(torch.randn(64, 64), None, torch.device("cpu"))
    out = torch.empty(64, 64, dtype=torch.bfloat16)
    ops.gemm(a, pw, out, tile=2)                                             # well-formed: runs
    real_init = _lib.GemmParams.__init__
    for bad in (0, ctypes.sizeof(_lib.GemmParams) - 16, ctypes.sizeof(_lib.GemmParams) + 8, 0x7f000000):
        def init(self, *args, _bad=bad, **kw):
            real_init(self, *args, **kw)
            self.struct_size = _bad
        monkeypatch.setattr(_lib.GemmParams, "__init__", init)
        with pytest.raises(RuntimeError, match="code -1"):
            ops.gemm(a, pw, out, tile=2)
    monkeypatch.setattr(_lib.GemmParams, "__init__", real_init)
    ops.gemm(a, pw, out, tile=2)
    lib = ctypes.CDLL(str(build_lib()))
    lib.pcdm_gemm.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    p = _lib.GemmParams()
    p.struct_size = ctypes.sizeof(_lib.GemmParams) - 16                      # (the version-3 struct was 16 bytes shorter)
    assert lib.pcdm_gemm(ctypes.byref(p), None) == -1
