# Round 6, GPU session 19: the table keys of the composed launches (ff.net.2 -> proj_out, conv2 + conv_shortcut) for the other configurations:
# in-step passes at batch 16 per GPU (configs[4]'s share), the 512-wide driver default and the stage-3 UNet; then the lines that use them.
set -u
OUT=gpurun_out/r6_s19
mkdir -p $OUT
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v11.json
(timeout 1500 python tools/tune_in_step.py --write --batch 16 --out $OUT/tune_b16.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_b16.txt
(timeout 1500 python tools/tune_in_step.py --write --width 512 --out $OUT/tune_w512.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_w512.txt
(timeout 1500 python tools/tune_in_step.py --write --stage3 --batch 4 --out $OUT/tune_stage3.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_stage3.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v12.json
for i in 1 2; do
(PCDM_TUNING_TABLE=$OUT/gfx950_v11.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_v11_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v12.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_v12_$i.json 2>/dev/null
done
(PCDM_TUNING_TABLE=$OUT/gfx950_v12.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline --width 512) > $OUT/bench_w512_v12.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v12.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 16 --steps 2) > $OUT/bench_b16_v12.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v12.json timeout 400 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 16 --steps 2 --attn fp8) > $OUT/bench_b16_fp8_v12.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v12.json timeout 400 python tools/bench_three_stage.py 2>&1 | grep -v amdgpu.ids | tail -2) > $OUT/three_stage_v12.json
for t in b16 w512 stage3; do grep "CHANGED\|in-step total\|new to the table" $OUT/tune_$t.txt | cut -c1-200; done
for f in v11_1 v12_1 v11_2 v12_2 w512_v12 b16_v12 b16_fp8_v12; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
cut -c1-420 $OUT/three_stage_v12.json
