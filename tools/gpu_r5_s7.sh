# round-5 GPU session 7: cache policy of the GEMM epilogues' output stores (PCDM_STORE_AUX: 17 = sc0 sc1 system-scope write-through, 16 = sc1, 1 = sc0)
# against the default write-back stores -- is the end-of-kernel L2 write-back of a 28.8 MB output part of the per-launch floor?  Same-box A/B.
set -u
OUT=gpurun_out/r5_s7
mkdir -p $OUT
for i in 1 2; do
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_base_$i.json 2>/dev/null
for v in aux17 aux16 aux1; do
(PCDM_LIB=$PWD/pcdms_amd/lib_alt/$v/libpcdm.so timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_${v}_$i.json 2>/dev/null
done
done
for v in base aux17 aux16 aux1; do
L=""; [ $v != base ] && L=$PWD/pcdms_amd/lib_alt/$v/libpcdm.so
(PCDM_LIB=$L timeout 200 python tools/profile_step.py 2>&1 | grep -v amdgpu.ids | head -45) > $OUT/step_$v.txt
done
for f in base_1 aux17_1 aux16_1 aux1_1 base_2 aux17_2 aux16_2 aux1_2; do echo $f $(cut -c1-95 $OUT/bench_$f.json); done
for v in base aux17; do head -1 $OUT/step_$v.txt; grep "(45056, 320, 320, False, 5\|(11264, 640, 640, False, 18\|(45056, 320, 2880, True, 21\|groupnorm          (8, 5632, 320)\|flash_attn_kernel  (8, 5, 5632" $OUT/step_$v.txt; done
