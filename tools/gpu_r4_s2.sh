#!/bin/bash
# round 4, GPU session 2: the whole -m gpu suite (records gpurun_out/parity_values.json), then same-box A/B of the row-wise GroupNorm
# (historical: PCDM_GN_ROWS selected the row-wise GroupNorm kernel, removed after this session -- profiles/r4_bench_gn_rows_ab.txt)
# (PCDM_GN_ROWS=0) and of the CFG-shared prefix (PCDM_SHARE_CFG_PREFIX=0), GroupNorm micro-benchmark.   usage: bash tools/gpu_r4_s2.sh
set -u
OUT=gpurun_out/r4_s2
mkdir -p $OUT
B="--no-cpu-baseline --no-vae --no-roofline"
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40) > $OUT/tests.txt
tail -5 $OUT/tests.txt
(timeout 120 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn.txt
(PCDM_GN_ROWS=0 timeout 120 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn_norows.txt
for i in 1 2; do
(timeout 300 python bench.py $B) > $OUT/bench_new_$i.json 2>$OUT/bench_new_$i.err
(PCDM_GN_ROWS=0 timeout 300 python bench.py $B) > $OUT/bench_norows_$i.json 2>/dev/null
(PCDM_SHARE_CFG_PREFIX=0 timeout 300 python bench.py $B) > $OUT/bench_noshare_$i.json 2>/dev/null
done
for f in new_1 norows_1 noshare_1 new_2 norows_2 noshare_2; do echo $f; cut -c1-130 $OUT/bench_$f.json; done
paste $OUT/bench_gn.txt $OUT/bench_gn_norows.txt | cut -c1-200 | head -30
