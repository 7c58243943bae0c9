# round-5 GPU session 2: LayerNorm partials from the producer (EXT 2 / 3), tuning of the "ln" keys with mode 2, micro-benchmark, same-box A/B,
# the new full-size tests (driver default UniPC 64x128, three-stage chain, stress)
set -u
OUT=gpurun_out/r5_s2
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "row_stats or folded_layernorm or rowgemm" 2>&1 | tail -5) > $OUT/tests_kernels.txt
(timeout 500 python tools/tune_missing_keys.py --out $OUT/gfx950_merged.json 2>&1 | grep -v amdgpu.ids | tail -40) > $OUT/tune_missing.txt
(timeout 400 python tools/bench_ln_gemm.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_ln_gemm.txt
for i in 1 2; do
(PCDM_TUNING_TABLE=$OUT/gfx950_merged.json PCDM_LN_TILED=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_lnoff_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_merged.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_lnon_$i.json 2>/dev/null
done
(PCDM_TUNING_TABLE=$OUT/gfx950_merged.json timeout 600 python -m pytest tests/test_unet_ctx.py -m gpu -x -q 2>&1 | tail -6) > $OUT/tests_ctx.txt
(PCDM_TUNING_TABLE=$OUT/gfx950_merged.json timeout 900 python -m pytest tests/test_fullsize_parity.py tests/test_three_stage_flow.py -m gpu -x -q -s -k "driver_default or three_stage_full or stress" 2>&1 | grep -v amdgpu.ids | tail -30) > $OUT/tests_new_fullsize.txt
cat $OUT/tests_kernels.txt $OUT/tune_missing.txt $OUT/bench_ln_gemm.txt; for f in lnoff_1 lnon_1 lnoff_2 lnon_2; do cut -c1-120 $OUT/bench_$f.json; done; cat $OUT/tests_ctx.txt; tail -20 $OUT/tests_new_fullsize.txt
