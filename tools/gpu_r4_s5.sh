#!/bin/bash
# round 4, GPU session 5: re-tune of the bench configuration's shapes on the round-4 kernels (incl. the dup_rows launches) + same-box A/B
# (historical: PCDM_ATTN_LOWREG selected the four-workgroups-per-CU attention instance, removed after this session -- profiles/r4_bench_attn_lowreg_ab.txt)
# against the committed table; the four-workgroups-per-CU attention instance (PCDM_ATTN_LOWREG=1) A/B.
set -u
OUT=gpurun_out/r4_s5
mkdir -p $OUT
B="--no-cpu-baseline --no-vae --no-roofline"
(timeout 900 python tools/tune_gemm_shapes.py --only-main --merge --out $OUT/gfx950_retuned.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune.log
tail -60 $OUT/tune.log
for v in 1 0; do
(PCDM_ATTN_LOWREG=$v timeout 150 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | cut -c1-150) > $OUT/bench_attn_lowreg$v.txt
done
for f in lowreg1 lowreg0; do echo $f; cat $OUT/bench_attn_$f.txt; done
for i in 1 2; do
(timeout 300 python bench.py $B) > $OUT/bench_committed_$i.json 2>$OUT/bench_committed_$i.err
(PCDM_TUNING_TABLE=$OUT/gfx950_retuned.json timeout 300 python bench.py $B) > $OUT/bench_retuned_$i.json 2>/dev/null
(PCDM_ATTN_LOWREG=1 timeout 300 python bench.py $B) > $OUT/bench_lowreg_$i.json 2>/dev/null
done
for f in committed_1 retuned_1 lowreg_1 committed_2 retuned_2 lowreg_2; do echo $f; cut -c1-130 $OUT/bench_$f.json; done
