#!/usr/bin/env python3
"""Static instruction mix of every kernel in a hipcc ``-save-temps`` assembly file (no GPU needed): MFMA / VALU / SALU / branches / waits /
barriers / LDS / vector-memory instruction counts per kernel, and optionally one kernel's text.  How the run-time ``switch`` behind
rowgemm.hip's counted wait was found (a tree of ~35 scalar compares and branches per ring stage), and the check that the tiled kernels'
K loops carry no such code.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -save-temps -c pcdms_amd/csrc/gemm.hip -o /tmp/gemm.o
    python tools/kernel_isa_mix.py gemm-hip-amdgcn-amd-amdhsa-gfx950.s [substring of a mangled kernel name to dump -> <name>.s]
"""
import collections
import re
import sys


def main():
    src = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else None
    starts = [i for i, ln in enumerate(src) if re.match(r"^_Z\w+: +; @", ln)]
    for i in starts:
        name = src[i].split(":")[0]
        j = i
        while j < len(src) and not src[j].startswith(".Lfunc_end"):
            j += 1
        body = src[i:j]
        cnt = collections.Counter()
        for ln in body:
            t = ln.strip().split()[0] if ln.strip() else ""
            if t.startswith("v_mfma"):
                cnt["mfma"] += 1
            elif t.startswith("v_"):
                cnt["valu"] += 1
            elif t.startswith(("s_cbranch", "s_branch", "s_setpc")):
                cnt["branch"] += 1
            elif t.startswith("s_waitcnt"):
                cnt["wait"] += 1
            elif t.startswith("s_barrier"):
                cnt["barrier"] += 1
            elif t.startswith("s_"):
                cnt["salu"] += 1
            elif t.startswith("ds_"):
                cnt["lds"] += 1
            elif t.startswith(("buffer_", "global_", "scratch_", "flat_")):
                cnt["vmem"] += 1
        print(f"{name[:110]:110s} lines {len(body):6d}  " + "  ".join(f"{k} {cnt[k]}" for k in ("mfma", "valu", "salu", "branch", "wait", "barrier", "lds", "vmem")))
        if want and want in name:
            out = re.sub(r"\W+", "_", name)[:80] + ".s"
            open(out, "w").write("\n".join(body))
            print(f"  -> {out}")


if __name__ == "__main__":
    main()
