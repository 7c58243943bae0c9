# Round 6, GPU session 7: kernel tests on the sources with the 16x16x32 twins as LayerNorm producers / consumers; the folded-LayerNorm keys re-tuned with
# them; same-box A/B of three tables (round 5 / 176-row tiles only / + twins)
set -u
OUT=gpurun_out/r6_s7
mkdir -p $OUT
python -m pytest tests/test_kernels.py tests/test_unet_ctx.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/tests.txt
(timeout 900 python tools/retune_keys.py ln --write 2>&1 | grep -v amdgpu.ids | tail -25) > $OUT/retune_ln.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_after_retune.json
for i in 1 2; do
(PCDM_TUNING_TABLE=tools/ab/gfx950_r5.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_r5table_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=tools/ab/gfx950_r6_tiles22.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_tiles22_$i.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_committed_$i.json 2>/dev/null
done
tail -3 $OUT/tests.txt; cat $OUT/retune_ln.txt
for f in r5table_1 tiles22_1 committed_1 r5table_2 tiles22_2 committed_2; do echo $f; cut -c1-140 $OUT/bench_$f.json; done
