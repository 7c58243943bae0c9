# Round 6, GPU session 18: ResnetBlock2D conv2 + conv_shortcut as ONE contraction (extra K behind the nine taps: pcdm_gemm_params.a3, ABI 5;
# pcdms_amd/unet.py FUSE_SHORTCUT).  Bound written down first: the 14 shortcut launches of a step take ~0.38 ms (profiles/r6_step_breakdown_fused.txt);
# as extra K-tiles of conv2 they cost about half of that, and conv2 no longer reads a residual.  Kill criterion: adopted only if three
# interleaved pairs gain >= 0.5 % end to end, the plain convolutions do not lose against a build without the extra-K branch
# (pcdms_amd/lib_alt/noxk: PCDM_NO_CONV_XK=1), and the full-size parity tests stay inside their tolerances.
set -u
OUT=gpurun_out/r6_s18
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_kernels.py -q -m gpu -k "shortcut_k or conv3x3 or gemm_linear or preference" 2>&1 | tail -4) > $OUT/tests_kernels.txt
cat $OUT/tests_kernels.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v9.json
# (a) what carrying the branch costs the plain convolutions: both builds with the fusion OFF, same table
for i in 1 2 3; do
(PCDM_FUSE_SHORTCUT=0 PCDM_LIB=$PWD/pcdms_amd/lib_alt/noxk/libpcdm.so timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_noxk_$i.json 2>/dev/null
(PCDM_FUSE_SHORTCUT=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_unfused_$i.json 2>/dev/null
done
# (b) the fusion: in-step pass for the new keys, then interleaved pairs
(timeout 1500 python tools/tune_in_step.py --write --out $OUT/tune_sc.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_sc.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v10.json
for i in 1 2 3; do
(PCDM_FUSE_SHORTCUT=0 PCDM_TUNING_TABLE=$OUT/gfx950_v9.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_unfused2_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v10.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_fused_$i.json 2>/dev/null
done
(PCDM_TUNING_TABLE=$OUT/gfx950_v10.json timeout 150 python tools/profile_step.py 2>&1 | grep -v amdgpu.ids) > $OUT/step_breakdown_fused.txt
(timeout 1500 python tools/tune_in_step.py --write --batch 8 --out $OUT/tune_sc_b8.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_sc_b8.txt
cp pcdms_amd/tuning/gfx950.json $OUT/gfx950_v11.json
for i in 1 2; do
(PCDM_FUSE_SHORTCUT=0 PCDM_TUNING_TABLE=$OUT/gfx950_v9.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_b8_unfused_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$OUT/gfx950_v11.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_b8_fused_$i.json 2>/dev/null
done
(PCDM_TUNING_TABLE=$OUT/gfx950_v11.json timeout 1200 python -m pytest tests/test_unet.py tests/test_unet_ctx.py tests/test_fullsize_parity.py -q -m gpu 2>&1 | tail -4) > $OUT/tests.txt
grep "CHANGED\|in-step total\|baseline\|new to the table" $OUT/tune_sc.txt | cut -c1-260
for f in noxk_1 unfused_1 noxk_2 unfused_2 noxk_3 unfused_3 unfused2_1 fused_1 unfused2_2 fused_2 unfused2_3 fused_3 b8_unfused_1 b8_fused_1 b8_unfused_2 b8_fused_2; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
grep "True, \|family" $OUT/step_breakdown_fused.txt | cut -c1-160 | head -70
grep "CHANGED\|in-step total\|new to the table" $OUT/tune_sc_b8.txt | cut -c1-260
cat $OUT/tests.txt
