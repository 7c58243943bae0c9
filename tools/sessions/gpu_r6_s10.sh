# Round 6, GPU session 10: the tuning table re-tuned IN THE STEP (tools/tune_in_step.py), then the end-to-end A/B against the committed table
set -u
OUT=gpurun_out/r6_s10
mkdir -p $OUT
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_before.json
(timeout 1500 python tools/tune_in_step.py --write --out $OUT/tune_in_step.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_in_step.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_in_step.json
for i in 1 2 3; do
(PCDM_TUNING_TABLE=$OUT/gfx950_before.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_before_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_in_step.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_in_step_$i.json 2>/dev/null
done
cut -c1-230 $OUT/tune_in_step.txt | tail -90
for f in before_1 in_step_1 before_2 in_step_2 before_3 in_step_3; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
