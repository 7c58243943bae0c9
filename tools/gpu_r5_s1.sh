# round-5 GPU session 1: GroupNorm statistics rewrite + LayerNorm fold on the tiled GEMM (kernel tests, tuning of the new keys, micro-benchmark, same-box A/B)
set -u
OUT=gpurun_out/r5_s1
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "groupnorm or layernorm or rowgemm or gelu or splitk" 2>&1 | tail -5) > $OUT/tests_kernels.txt
(timeout 400 python tools/tune_missing_keys.py --out $OUT/gfx950_merged.json 2>&1 | grep -v amdgpu.ids | tail -40) > $OUT/tune_missing.txt
(timeout 300 python tools/bench_ln_gemm.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_ln_gemm.txt
(timeout 200 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids | head -20) > $OUT/bench_gn.txt
for i in 1 2; do
(PCDM_TUNING_TABLE=$OUT/gfx950_merged.json PCDM_LN_TILED=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_lnoff_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_merged.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_lnon_$i.json 2>/dev/null
done
(PCDM_TUNING_TABLE=$OUT/gfx950_merged.json timeout 400 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -k "stress or single_forward" 2>&1 | tail -15) > $OUT/tests_fullsize.txt
cat $OUT/tests_kernels.txt $OUT/tune_missing.txt $OUT/bench_ln_gemm.txt; for f in lnoff_1 lnon_1 lnoff_2 lnon_2; do cut -c1-150 $OUT/bench_$f.json; done; tail -8 $OUT/tests_fullsize.txt
