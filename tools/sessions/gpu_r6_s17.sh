# Round 6, GPU session 17: ff.net.2 (+ residual) -> proj_out (+ residual) as one two-source GEMM against [Wp W2 | Wp] (pcdms_amd/unet.py FUSE_FF_OUT).
# Bound written down first: the two launches take 49 + 28.7 us per level-0 block back to back (59 + 27-31 in the step); the fused K = 1600 launch
# streams the same operands minus the M x C state in between.  Kill criterion: adopted only if three interleaved pairs gain >= 0.5 % end to end
# and the full-size parity tests stay inside their tolerances.
set -u
OUT=gpurun_out/r6_s17
mkdir -p $OUT
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v7.json
(timeout 1500 python tools/tune_in_step.py --write --out $OUT/tune_ffo.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_ffo.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v8.json
for i in 1 2 3; do
(PCDM_FUSE_FF_OUT=0 PCDM_TUNING_TABLE=$OUT/gfx950_v7.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_unfused_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v8.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_fused_$i.json 2>/dev/null
done
(PCDM_TUNING_TABLE=$OUT/gfx950_v8.json timeout 150 python tools/profile_step.py 2>&1 | grep -v amdgpu.ids) > $OUT/step_breakdown_fused.txt
(timeout 1500 python tools/tune_in_step.py --write --batch 8 --out $OUT/tune_ffo_b8.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_ffo_b8.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v9.json
for i in 1 2; do
(PCDM_FUSE_FF_OUT=0 PCDM_TUNING_TABLE=$OUT/gfx950_v7.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_b8_unfused_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v9.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_b8_fused_$i.json 2>/dev/null
done
(PCDM_TUNING_TABLE=$OUT/gfx950_v9.json timeout 1200 python -m pytest tests/test_unet.py tests/test_unet_ctx.py tests/test_fullsize_parity.py -q -m gpu 2>&1 | tail -4) > $OUT/tests.txt
(timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "gemm_linear or preference" 2>&1 | tail -3) > $OUT/tests_gemm.txt
cat $OUT/tests_gemm.txt
grep "CHANGED\|in-step total\|baseline\|1600\|3200\|6400" $OUT/tune_ffo.txt | cut -c1-220
for f in unfused_1 fused_1 unfused_2 fused_2 unfused_3 fused_3 b8_unfused_1 b8_fused_1 b8_unfused_2 b8_fused_2; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
grep "1600\|3200\|6400\|family" $OUT/step_breakdown_fused.txt | cut -c1-160
grep "CHANGED\|in-step total" $OUT/tune_ffo_b8.txt | cut -c1-220
cat $OUT/tests.txt
