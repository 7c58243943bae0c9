# round-5 GPU session 6 (the EXT 4 instances without their scratch frame, apply kernel without per-element divisions): GroupNorm statistics from the producing convolution (EXT 4) + normalise-only GroupNorm: tests, micro-benchmark, same-box A/B
set -u
OUT=gpurun_out/r5_s6
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "groupnorm_producer or groupnorm_large or row_stats" 2>&1 | tail -5) > $OUT/tests_kernels.txt
(timeout 300 python tools/bench_gn_apply.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn_apply.txt
for i in 1 2; do
(PCDM_GN_PRODUCER_STATS=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_gnoff_$i.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_gnon_$i.json 2>/dev/null
done
(timeout 600 python -m pytest tests/test_unet_ctx.py tests/test_unet.py -m gpu -x -q 2>&1 | tail -6) > $OUT/tests_ctx.txt
(timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -s -k "single_forward or 50_step or stress or configs4" 2>&1 | grep -v amdgpu.ids | tail -12) > $OUT/tests_fullsize.txt
cat $OUT/tests_kernels.txt $OUT/bench_gn_apply.txt; for f in gnoff_1 gnon_1 gnoff_2 gnon_2; do cut -c1-100 $OUT/bench_$f.json; done; cat $OUT/tests_ctx.txt; tail -12 $OUT/tests_fullsize.txt
