# Round 6, GPU session 30: table keys of the phase-decomposed upsample launches (and whatever else the last changes made new) for batch 16 per GPU, the
# 512-wide driver default and the stage-3 UNet -- ADD ONLY (tools/tune_in_step.py --add-only: no existing entry can move), so the batch-4 line cannot change.
set -u
OUT=gpurun_out/r6_s30
mkdir -p $OUT
(timeout 900 python tools/tune_in_step.py --write --add-only --batch 16 --out $OUT/tune_b16.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_b16.txt
(timeout 900 python tools/tune_in_step.py --write --add-only --width 512 --out $OUT/tune_w512.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_w512.txt
(timeout 900 python tools/tune_in_step.py --write --add-only --stage3 --batch 4 --out $OUT/tune_stage3.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_stage3.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v15.json
for t in b16 w512 stage3; do grep "CHANGED\|new to the table\|in-step total" $OUT/tune_$t.txt | cut -c1-300; done
