# Round 6, GPU session 4: the shader clock the MFMA-dense launches actually run at (in-kernel: shader cycles / 100 MHz wall counter)
set -u
OUT=gpurun_out/r6_s4
mkdir -p $OUT
(PCDM_ANATOMY=clock timeout 600 python tools/gemm_anatomy.py 2>&1 | grep -v amdgpu.ids | cut -c1-400) > $OUT/anatomy_clock.txt
cat $OUT/anatomy_clock.txt
