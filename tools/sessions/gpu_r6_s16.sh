# Round 6, GPU session 16: XCD-aware placement of the attention grid (attn_block_coords; PCDM_ATTN_XCD=0 = the plain grid).
# Kill criterion written down first: adopted only if the level-0 self-attention launch gets >= 3 % faster back to back AND the end-to-end
# line does not lose (three interleaved pairs).
set -u
OUT=gpurun_out/r6_s16
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2; do
for x in 0 1; do
(PCDM_ATTN_XCD=$x timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_attn_xcd${x}_$i.txt
done
done
REPO=$PWD
cd /tmp
for x in 0 1; do
PCDM_ATTN_XCD=$x timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$x -- python $REPO/tools/pmc_attn.py > /dev/null 2>&1
done
cd $REPO
for x in 0 1; do python tools/pmc_attn_fetch.py $OUT/pmc_fetch_$x > $OUT/attn_fetch_xcd$x.json 2> $OUT/attn_fetch_xcd$x.err; rm -rf $OUT/pmc_fetch_$x; done
for i in 1 2 3; do
for x in 0 1; do
(PCDM_ATTN_XCD=$x timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_xcd${x}_$i.json 2>/dev/null
done
done
(PCDM_ATTN_XCD=0 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --attn fp8) > $OUT/bench_fp8_xcd0.json 2>/dev/null
(PCDM_ATTN_XCD=1 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --attn fp8) > $OUT/bench_fp8_xcd1.json 2>/dev/null
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "flash_attn" 2>&1 | tail -3 > $OUT/attn_tests.txt
for x in 0 1; do for i in 1 2; do echo xcd$x $i; cat $OUT/bench_attn_xcd${x}_$i.txt; done; done
cat $OUT/attn_fetch_xcd0.json $OUT/attn_fetch_xcd1.json
for x in 0 1; do for i in 1 2 3; do echo xcd$x $i; cut -c1-120 $OUT/bench_xcd${x}_$i.json; done; done
cut -c1-120 $OUT/bench_fp8_xcd0.json $OUT/bench_fp8_xcd1.json
cat $OUT/attn_tests.txt
