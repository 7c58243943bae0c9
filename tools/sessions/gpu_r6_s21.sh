# Round 6, GPU session 21: the attention row sums on the VALU (PCDM_ATTN_ROWSUM=valu) against the MFMA-against-ones default, IN the step
# (earlier rounds ranked the two per launch, back to back).  Adopted only if three interleaved pairs gain >= 0.3 %.
set -u
OUT=gpurun_out/r6_s21
mkdir -p $OUT
for i in 1 2 3; do
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_mfma_$i.json 2>/dev/null
(PCDM_ATTN_ROWSUM=valu timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_valu_$i.json 2>/dev/null
done
for f in mfma_1 valu_1 mfma_2 valu_2 mfma_3 valu_3; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
