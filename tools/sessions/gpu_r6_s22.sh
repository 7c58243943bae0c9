# Round 6, GPU session 22: XCD-aware placement of the rowgemm grid (PCDM_ROWGEMM_XCD=0 = the plain grid): the A-in-registers kernel was the one
# producer / consumer of level-0 rows whose row blocks were dealt round robin over the XCDs.  Adopted only if three interleaved pairs gain >= 0.3 %.
set -u
OUT=gpurun_out/r6_s22
mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "rowgemm or preference or gemm_geglu or layernorm" 2>&1 | tail -3) > $OUT/tests.txt
for i in 1 2 3; do
(PCDM_ROWGEMM_XCD=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_plain_$i.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_xcd_$i.json 2>/dev/null
done
for i in 1 2; do
(PCDM_ROWGEMM_XCD=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_b8_plain_$i.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_b8_xcd_$i.json 2>/dev/null
done
cat $OUT/tests.txt
for f in plain_1 xcd_1 plain_2 xcd_2 plain_3 xcd_3 b8_plain_1 b8_xcd_1 b8_plain_2 b8_xcd_2; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
