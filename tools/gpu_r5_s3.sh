# round-5 GPU session 3: cross-attention with the query projection inside (pcdm_flash_attn_qproj) + the partials prologue fix of the folded-LayerNorm consumers
set -u
OUT=gpurun_out/r5_s3
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "row_stats or folded_layernorm or flash_attn_qproj" 2>&1 | tail -5) > $OUT/tests_kernels.txt
(timeout 500 python tools/tune_missing_keys.py --out $OUT/gfx950_merged.json 2>&1 | grep -v amdgpu.ids | tail -40) > $OUT/tune_missing.txt
(timeout 400 python tools/bench_ln_gemm.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_ln_gemm.txt
(PCDM_TUNING_TABLE=$OUT/gfx950_merged.json timeout 300 python tools/bench_xattn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_xattn.txt
export PCDM_TUNING_TABLE=$OUT/gfx950_merged.json
for i in 1 2; do
(PCDM_LN_TILED=0 PCDM_XATTN_QPROJ=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_off_$i.json 2>/dev/null
(PCDM_XATTN_QPROJ=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_ln_$i.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_ln_xq_$i.json 2>/dev/null
done
(timeout 600 python -m pytest tests/test_unet_ctx.py tests/test_unet.py -m gpu -x -q 2>&1 | tail -6) > $OUT/tests_ctx.txt
(timeout 600 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -s -k "single_forward or 50_step or stress" 2>&1 | grep -v amdgpu.ids | tail -12) > $OUT/tests_fullsize.txt
cat $OUT/tests_kernels.txt $OUT/tune_missing.txt $OUT/bench_ln_gemm.txt $OUT/bench_xattn.txt; for f in off_1 ln_1 ln_xq_1 off_2 ln_2 ln_xq_2; do cut -c1-100 $OUT/bench_$f.json; done; cat $OUT/tests_ctx.txt; tail -12 $OUT/tests_fullsize.txt
