"""Register / scratch / LDS usage of every kernel in a hipcc ``-save-temps`` assembly file (``*-gfx950.s``).

    cd /tmp/dev && hipcc --offload-arch=gfx950 -O3 ... -save-temps -c x.hip -o x.o && python tools/kernel_resources.py /tmp/dev/*gfx950*.s
Rules of thumb (MI355X_MICROARCH.md "Register files"): waves per SIMD = min(8, 512 // alloc) with alloc = ceil(next_free_vgpr / 8) * 8
(arch VGPRs + AGPRs); any non-zero scratch is a spill."""
from __future__ import annotations

import glob
import re
import sys


def main():
    files = sys.argv[1:] or glob.glob("*gfx950*.s")
    for f in files:
        t = open(f).read()
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", t, re.S):
            name, b = m.group(1), m.group(2)

            def g(k):
                r = re.search(k + r"\s+(\d+)", b)
                return int(r.group(1)) if r else -1
            name = re.sub(r"_ZN\d+_GLOBAL__N_1\d+", "", name)
            v = g(r"\.amdhsa_next_free_vgpr")
            alloc = (v + 7) // 8 * 8
            print(f"{name[:64]:64s} vgpr+agpr {v:4d} (accum_offset {g(r'.amdhsa_accum_offset'):3d}) waves/SIMD {min(8, 512 // max(alloc, 1))} "
                  f"sgpr {g(r'.amdhsa_next_free_sgpr'):3d} scratch {g(r'.amdhsa_private_segment_fixed_size'):4d} B "
                  f"lds {g(r'.amdhsa_group_segment_fixed_size')} B")


if __name__ == "__main__":
    main()
