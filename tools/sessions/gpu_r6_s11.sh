# Round 6, GPU session 11: second pass of the in-step tuner (gain judged against the key's own time; the LayerNorm -> Linear pairs too), then the A/B of
# three tables: 176-row tiles only (back-to-back tuned) / in-step pass 1 / in-step pass 2
set -u
OUT=gpurun_out/r6_s11
mkdir -p $OUT
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v1.json
(timeout 2400 python tools/tune_in_step.py --write --out $OUT/tune_in_step.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_in_step.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v2.json
for i in 1 2 3; do
(PCDM_TUNING_TABLE=tools/ab/gfx950_r6_tiles22.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_tiles22_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v1.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_v1_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v2.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_v2_$i.json 2>/dev/null
done
grep "CHANGED\|in-step total\|baseline" $OUT/tune_in_step.txt | cut -c1-230
for f in tiles22_1 v1_1 v2_1 tiles22_2 v1_2 v2_2 tiles22_3 v1_3 v2_3; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
