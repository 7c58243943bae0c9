#!/usr/bin/env python3
"""Build a VARIANT of libpcdm.so with extra -D defines into pcdms_amd/lib_alt/<name>/libpcdm.so, next to (not instead of) the product library:
same-box A/B runs through PCDM_LIB=<path> (tools/with_lib.py, bench.py).  No GPU needed.

    python tools/build_alt_lib.py aux17 PCDM_STORE_AUX=17
"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pcdms_amd import build as B  # noqa: E402


def main():
    name, defines = sys.argv[1], sys.argv[2:]
    out = B.ROOT / "lib_alt" / name
    out.mkdir(parents=True, exist_ok=True)
    hipcc = B._hipcc()
    B.write_tuning_include()
    jobs, objs = [], []
    for src in B.SOURCES:
        o = out / (src + ".o")
        jobs.append([hipcc, f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", *B.VGPR_FORM, *B.EXTRA_FLAGS.get(src, []),
                     "-Wno-unused-result", *["-D" + d for d in defines], "-c", str(B.CSRC / src), "-o", str(o)])
        objs.append(str(o))
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        list(ex.map(subprocess.check_call, jobs))
    lib = out / "libpcdm.so"
    subprocess.check_call([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", str(lib), *objs])
    for o in objs:
        Path(o).unlink()
    print(lib)


if __name__ == "__main__":
    main()
