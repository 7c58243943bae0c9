#!/bin/bash
# round 4, GPU session 1: new-kernel tests, then same-box A/B of (a) the one-exp GELU against the Abramowitz-Stegun build
# (pcdms_amd/lib/libpcdm_prev.so = PCDM_BUILD_DEFINES=PCDM_GELU_AS7126) and (b) the split-K reduce folded into the GroupNorm against
# PCDM_DEFER_SPLITK=0.   usage: bash tools/gpu_r4_s1.sh
set -u
OUT=gpurun_out/r4_s1
mkdir -p $OUT
PREV=pcdms_amd/lib/libpcdm_prev.so
B="--no-cpu-baseline --no-vae --no-roofline"
(timeout 600 python -m pytest tests/test_kernels.py tests/test_unet_ctx.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15) > $OUT/tests.txt
tail -3 $OUT/tests.txt
(timeout 150 python tools/bench_rowgemm.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_rowgemm_new.txt
(timeout 150 python tools/with_lib.py $PREV tools/bench_rowgemm.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_rowgemm_prevgelu.txt
for i in 1 2; do
(timeout 300 python bench.py $B) > $OUT/bench_new_$i.json 2>$OUT/bench_new_$i.err
(PCDM_DEFER_SPLITK=0 timeout 300 python bench.py $B) > $OUT/bench_nodefer_$i.json 2>/dev/null
(timeout 300 python tools/with_lib.py $PREV bench.py $B) > $OUT/bench_prevgelu_$i.json 2>/dev/null
done
for f in new_1 nodefer_1 prevgelu_1 new_2 nodefer_2 prevgelu_2; do echo $f; cut -c1-130 $OUT/bench_$f.json; done
grep -i "ff1\|tiled\|tile 3" $OUT/bench_rowgemm_new.txt | head -20
echo ---; grep -i "ff1\|tiled\|tile 3" $OUT/bench_rowgemm_prevgelu.txt | head -20
(timeout 200 python tools/profile_step.py 2>&1 | grep -v amdgpu.ids) > $OUT/step_breakdown.txt
head -3 $OUT/step_breakdown.txt
