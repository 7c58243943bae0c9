# round-5 GPU session 10: streaming (nt) loads for data read exactly once -- the epilogues' residual rows / the GroupNorm inputs -- against the product build. Same box.
set -u
OUT=gpurun_out/r5_s10
mkdir -p $OUT
for i in 1 2; do
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_product_$i.json 2>/dev/null
(PCDM_LIB=$PWD/pcdms_amd/lib_alt/resnt/libpcdm.so timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_resnt_$i.json 2>/dev/null
(PCDM_LIB=$PWD/pcdms_amd/lib_alt/gnnt/libpcdm.so timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_gnnt_$i.json 2>/dev/null
done
for f in product_1 resnt_1 gnnt_1 product_2 resnt_2 gnnt_2; do echo $f $(cut -c1-95 $OUT/bench_$f.json); done
