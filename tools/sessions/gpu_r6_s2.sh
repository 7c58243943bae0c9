# Round 6, GPU session 2: kernel tests on the sources with the 176-row tiles + the partials consumer on tile 23; the folded-LayerNorm GEGLU key of
# level 1 re-tuned; same-box A/B of three tuning tables (round 5 / committed round 6 / every 21 -> 22); which operand's staging is exposed.
set -u
OUT=gpurun_out/r6_s2
mkdir -p $OUT
python -m pytest tests/test_kernels.py -m gpu -x -q -k "uneven_176 or full_row_tiles or row_stats_producer or folded_layernorm_tiled or gemm_bias" 2>&1 | grep -v amdgpu.ids | tail -5 > $OUT/tests.txt
(timeout 600 python tools/retune_keys.py ln,11264,5120 ln,11264,1920 2>&1 | grep -v amdgpu.ids | tail -4) > $OUT/retune_ln.txt
for i in 1 2; do
(PCDM_TUNING_TABLE=tools/ab/gfx950_r5.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_r5table_$i.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_committed_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=tools/ab/gfx950_all22.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_all22_$i.json 2>/dev/null
done
(timeout 600 python tools/ablate_gemm.py 21,22 2>&1 | grep -v amdgpu.ids) > $OUT/ablate_21_22.txt
tail -3 $OUT/tests.txt; cat $OUT/retune_ln.txt
for f in r5table_1 committed_1 all22_1 r5table_2 committed_2 all22_2; do echo $f; cut -c1-140 $OUT/bench_$f.json; done
cat $OUT/ablate_21_22.txt
