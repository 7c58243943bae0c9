#!/bin/bash
# A/B of the attention output epilogue (rows through LDS vs direct 8-byte pieces): GPU attention tests, the attention micro-benchmark and
# interleaved bench lines, previous library (pcdms_amd/lib_alt/prev) against the current one.   usage: bash tools/gpu_ab_attn_out.sh [name]
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-r5_ab_attn_out}
mkdir -p $OUT
(timeout 200 python -m pytest tests/test_kernels.py -m gpu -q -k "attn or attention" 2>&1 | tail -3) > $OUT/tests_attn.txt; cat $OUT/tests_attn.txt
(PCDM_LIB=$REPO/pcdms_amd/lib_alt/prev/libpcdm.so timeout 100 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_attn_prev.txt
(timeout 100 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_attn_new.txt
for i in 1 2; do
  (PCDM_LIB=$REPO/pcdms_amd/lib_alt/prev/libpcdm.so timeout 200 python bench.py --no-cpu-baseline --no-vae --no-roofline 2>/dev/null) > $OUT/bench_prev_$i.json
  (timeout 200 python bench.py --no-cpu-baseline --no-vae --no-roofline 2>/dev/null) > $OUT/bench_new_$i.json
done
head -20 $OUT/bench_attn_prev.txt; head -20 $OUT/bench_attn_new.txt
for f in $OUT/bench_prev_1.json $OUT/bench_new_1.json $OUT/bench_prev_2.json $OUT/bench_new_2.json; do cut -c1-110 $f; done
