"""Build the lane-emulator variant of libpcdm (TEST INFRASTRUCTURE; see tests/emu/hip_emu.h).

Compiles the *same* pcdms_amd/csrc/*.hip sources for the host with -DPCDM_EMU into
tests/emu/_build/libpcdm_emu.so.  Only tests load it (through pcdms_amd._lib.use_library).
"""
from __future__ import annotations

import ctypes
import shutil
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "pcdms_amd" / "csrc"
OUT = HERE / "_build"
LIB = OUT / "libpcdm_emu.so"
SOURCES = ["norm.hip", "gemm.hip", "gemm_ext.hip", "rowgemm.hip", "attn.hip", "misc.hip", "unet_ctx.hip"]


def _cxx() -> str:
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++"), shutil.which("amdclang++")):
        if c and Path(c).exists():
            return c
    raise RuntimeError("clang++ (ext_vector_type support) not found for the emulator build")


def build(force: bool = False) -> Path:
    OUT.mkdir(exist_ok=True)
    deps = [CSRC / s for s in SOURCES] + [CSRC / "pcdm_device.h", CSRC / "gemm_args.h", CSRC / "gemm_kernel.inc", ROOT / "include" / "pcdm.h", HERE / "hip_emu.h",
                                          HERE / "hip_emu.cpp", ROOT / "pcdms_amd" / "tuning" / "gfx950.json"]
    if not force and LIB.exists() and all(d.stat().st_mtime <= LIB.stat().st_mtime for d in deps):
        return LIB
    from pcdms_amd.build import write_tuning_include
    write_tuning_include()        # (csrc/tuning_table.inc: included by unet_ctx.hip)
    cxx = _cxx()
    objs = []
    common = ["-O2", "-std=c++17", "-fPIC", "-DPCDM_EMU", "-Wno-unknown-attributes", "-Wno-unused-value"]
    for s in SOURCES:
        o = OUT / (s + ".o")
        subprocess.check_call([cxx, *common, "-x", "c++", "-include", str(HERE / "hip_emu.h"), "-c", str(CSRC / s),
                               "-o", str(o)])
        objs.append(str(o))
    o = OUT / "hip_emu.o"
    subprocess.check_call([cxx, *common, "-c", str(HERE / "hip_emu.cpp"), "-o", str(o)])
    objs.append(str(o))
    subprocess.check_call([cxx, "-shared", "-fPIC", "-o", str(LIB), *objs])
    return LIB


def load() -> ctypes.CDLL:
    return ctypes.CDLL(str(build()))


if __name__ == "__main__":
    print(build(force=True))
