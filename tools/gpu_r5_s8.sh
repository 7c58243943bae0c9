# round-5 GPU session 8: agent-scope (sc1) output stores in EVERY kernel (GEMM / rowgemm epilogues, norms, attention, split-K slabs) = the new product
# build, against the write-back stores of rounds 1-4 (lib_alt/aux0) and against sc1 on the GEMM / rowgemm epilogue stores only (lib_alt/aux16).  Same box.
set -u
OUT=gpurun_out/r5_s8
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "groupnorm or layernorm or flash_attn or splitk or gemm_linear" 2>&1 | tail -4) > $OUT/tests_kernels.txt
for i in 1 2; do
(PCDM_LIB=$PWD/pcdms_amd/lib_alt/aux0/libpcdm.so timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_aux0_$i.json 2>/dev/null
(PCDM_LIB=$PWD/pcdms_amd/lib_alt/aux16/libpcdm.so timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_aux16gemm_$i.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_all_$i.json 2>/dev/null
done
(timeout 200 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids | head -6) > $OUT/bench_gn.txt
(PCDM_LIB=$PWD/pcdms_amd/lib_alt/aux0/libpcdm.so timeout 200 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids | head -6) > $OUT/bench_gn_aux0.txt
(timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | head -8) > $OUT/bench_attn.txt
cat $OUT/tests_kernels.txt; for f in aux0_1 aux16gemm_1 all_1 aux0_2 aux16gemm_2 all_2; do echo $f $(cut -c1-95 $OUT/bench_$f.json); done; paste -d'|' $OUT/bench_gn.txt $OUT/bench_gn_aux0.txt | cut -c1-200; head -4 $OUT/bench_attn.txt
