# Round 6, GPU session 13: in-step pass at batch 16 per GPU (configs[4]'s share), A/B with --batch 16 (bf16 and fp8 attention)
set -u
OUT=gpurun_out/r6_s13
mkdir -p $OUT
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v4.json
(timeout 2400 python tools/tune_in_step.py --write --batch 16 --out $OUT/tune_b16.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_b16.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v5.json
for i in 1 2; do
(PCDM_TUNING_TABLE=$OUT/gfx950_v4.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 16 --steps 2) > $OUT/bench_b16_v4_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v5.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 16 --steps 2) > $OUT/bench_b16_v5_$i.json 2>/dev/null
done
(PCDM_TUNING_TABLE=$OUT/gfx950_v5.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 16 --steps 2 --attn fp8) > $OUT/bench_b16_v5_fp8.json 2>/dev/null
grep "CHANGED\|in-step total\|baseline" $OUT/tune_b16.txt | cut -c1-220
for f in b16_v4_1 b16_v5_1 b16_v4_2 b16_v5_2 b16_v5_fp8; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
