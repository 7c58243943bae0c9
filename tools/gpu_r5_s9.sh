# round-5 GPU session 9: agent-scope (sc1) stores for the norms' outputs only / the attention outputs only, on top of the product build (sc1 on the GEMM epilogue stores). Same box.
set -u
OUT=gpurun_out/r5_s9
mkdir -p $OUT
for i in 1 2; do
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_product_$i.json 2>/dev/null
(PCDM_LIB=$PWD/pcdms_amd/lib_alt/sc1norm/libpcdm.so timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_sc1norm_$i.json 2>/dev/null
(PCDM_LIB=$PWD/pcdms_amd/lib_alt/sc1attn/libpcdm.so timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_sc1attn_$i.json 2>/dev/null
done
for f in product_1 sc1norm_1 sc1attn_1 product_2 sc1norm_2 sc1attn_2; do echo $f $(cut -c1-95 $OUT/bench_$f.json); done
