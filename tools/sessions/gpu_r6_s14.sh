# Round 6, GPU session 14: in-step passes for the driver's default geometry (512 x 512 images: latent 64 x 128, N = 4) and for the stage-3 UNet (N = 4, 352 x 512)
set -u
OUT=gpurun_out/r6_s14
mkdir -p $OUT
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v5.json
(timeout 2400 python tools/tune_in_step.py --write --width 512 --out $OUT/tune_w512.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_w512.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v6.json
for i in 1 2; do
(PCDM_TUNING_TABLE=$OUT/gfx950_v5.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline --width 512) > $OUT/bench_w512_v5_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v6.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline --width 512) > $OUT/bench_w512_v6_$i.json 2>/dev/null
done
(timeout 2400 python tools/tune_in_step.py --write --stage3 --batch 4 --out $OUT/tune_stage3.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_stage3.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v7.json
grep "CHANGED\|in-step total\|baseline" $OUT/tune_w512.txt | cut -c1-220
for f in w512_v5_1 w512_v6_1 w512_v5_2 w512_v6_2; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
grep "CHANGED\|in-step total\|baseline\|Error\|error" $OUT/tune_stage3.txt | cut -c1-220; tail -3 $OUT/tune_stage3.txt | cut -c1-200
