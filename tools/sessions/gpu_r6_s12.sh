# Round 6, GPU session 12: in-step tuning of the LayerNorm -> Linear pairs with a localized objective (batch 4), A/B; then the whole in-step pass at
# batch 8 per GPU (configs[2]'s share), A/B with --batch 8
set -u
OUT=gpurun_out/r6_s12
mkdir -p $OUT
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v1.json
(timeout 1500 python tools/tune_in_step.py --write --skip-plain --out $OUT/tune_ln.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_ln.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v3.json
for i in 1 2 3; do
(PCDM_TUNING_TABLE=$OUT/gfx950_v1.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_v1_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v3.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_v3_$i.json 2>/dev/null
done
(timeout 2400 python tools/tune_in_step.py --write --batch 8 --out $OUT/tune_b8.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_b8.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v4.json
for i in 1 2; do
(PCDM_TUNING_TABLE=$OUT/gfx950_v3.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_b8_v3_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v4.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_b8_v4_$i.json 2>/dev/null
done
cut -c1-220 $OUT/tune_ln.txt
for f in v1_1 v3_1 v1_2 v3_2 v1_3 v3_3; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
grep "CHANGED\|in-step total\|baseline" $OUT/tune_b8.txt | cut -c1-220
for f in b8_v3_1 b8_v4_1 b8_v3_2 b8_v4_2; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
