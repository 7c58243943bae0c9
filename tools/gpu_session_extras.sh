set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3_s19
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_vae.py tests/test_bench_contract.py tests/test_pipeline.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 > $OUT/tests.log
(timeout 500 python tools/bench_three_stage.py 2>$OUT/three_stage.err | tail -1) > $OUT/three_stage.json
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_bf16.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --attn fp8) > $OUT/bench_fp8.json 2>/dev/null
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline --batch 8) > $OUT/bench_batch8.json 2>/dev/null
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/tools/profile_step.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/tools/profile_step.py > /dev/null 2>&1
cd $REPO
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/gemm_traffic.json > $OUT/kernel_traffic.json 2> $OUT/kernel_traffic.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
(timeout 300 python bench.py --no-cpu-baseline --no-vae) > $OUT/bench_roofline.json 2>/dev/null
cat $OUT/tests.log; cut -c1-400 $OUT/three_stage.json; for f in bf16 fp8 batch8; do cut -c1-200 $OUT/bench_$f.json; done
