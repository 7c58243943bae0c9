# Round 6, GPU session 9: the same launches IN THE STEP (eager step, HIP events per launch, by shape) under two tables: where do the per-launch gains go?
set -u
OUT=gpurun_out/r6_s9
mkdir -p $OUT
for t in r6_tiles22 r6_twins_noln; do
for i in 1 2; do
(PCDM_TUNING_TABLE=tools/ab/gfx950_$t.json timeout 200 python tools/profile_step.py 2>&1 | grep -v amdgpu.ids) > $OUT/step_${t}_$i.txt
done
done
head -3 $OUT/step_r6_tiles22_1.txt; head -3 $OUT/step_r6_twins_noln_1.txt
