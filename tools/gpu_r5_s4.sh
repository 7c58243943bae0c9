# round-5 GPU session 4: bound of a normalise-only GroupNorm pass (apply with known statistics), shortcut convolutions on a side stream (A/B), failed tests of session 3 re-run
set -u
OUT=gpurun_out/r5_s4
mkdir -p $OUT
(timeout 300 python tools/bench_gn_apply.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn_apply.txt
(timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "flash_attn_qproj" 2>&1 | tail -5) > $OUT/tests_kernels.txt
for i in 1 2; do
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_base_$i.json 2>/dev/null
(PCDM_SHORTCUT_STREAM=352 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_side352_$i.json 2>/dev/null
(PCDM_SHORTCUT_STREAM=1408 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_side1408_$i.json 2>/dev/null
done
(timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -k "configs4 or stage3 or config0" 2>&1 | tail -6) > $OUT/tests_fullsize.txt
cat $OUT/bench_gn_apply.txt $OUT/tests_kernels.txt; for f in base_1 side352_1 side1408_1 base_2 side352_2 side1408_2; do cut -c1-100 $OUT/bench_$f.json; done; cat $OUT/tests_fullsize.txt
