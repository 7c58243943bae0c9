"""Build libpcdm.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

``python -m pcdms_amd.build`` or ``pcdms_amd.build.build_lib()``.  hipcc cross-compiles for
gfx950 without a GPU; the resulting ``pcdms_amd/lib/libpcdm.so`` is git-ignored but travels to the
GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
LIB = LIBDIR / "libpcdm.so"
SOURCES = ["norm.hip", "gemm.hip", "rowgemm.hip", "attn.hip", "misc.hip", "unet_ctx.hip"]
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]   # MFMA accumulators in arch VGPRs (no v_accvgpr moves)
NO_VGPR_FORM: set = set()
EXTRA_DEPS: dict = {}
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libpcdm.so)")


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    LIBDIR.mkdir(exist_ok=True)
    # the -D defines of the last build are part of the staleness check: an A/B build (PCDM_BUILD_DEFINES=...) must never silently
    # reuse objects compiled without them, nor the default build objects compiled with them
    defines = " ".join(sorted(os.environ.get("PCDM_BUILD_DEFINES", "").split()))
    stamp = LIBDIR / "build_defines.txt"
    if not stamp.exists() or stamp.read_text() != defines:
        force = True
    headers = [CSRC / "pcdm_device.h", CSRC / "gemm_args.h", ROOT.parent / "include" / "pcdm.h"]
    objs, jobs = [], []
    for src in SOURCES:
        s = CSRC / src
        o = LIBDIR / (src + ".o")
        if force or _stale(o, [s, *headers, *[CSRC / d for d in EXTRA_DEPS.get(src, [])]]):
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
                         *([] if src in NO_VGPR_FORM else VGPR_FORM), "-Wno-unused-result",
                         *(["-D" + d for d in os.environ.get("PCDM_BUILD_DEFINES", "").split()]), "-c", str(s), "-o", str(o)])
        objs.append(str(o))
    if jobs:   # the translation units are independent: compile them side by side (gemm.hip alone is ~1 min)
        from concurrent.futures import ThreadPoolExecutor
        if verbose:
            for cmd in jobs:
                print(" ".join(cmd), flush=True)
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(subprocess.check_call, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    stamp.write_text(defines)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
