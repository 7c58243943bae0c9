# HBM traffic passes on the current kernel sources + one roofline bench line that quotes them: bash tools/gpu_traffic_pass.sh <outdir under gpurun_out>
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-traffic}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/tools/profile_step.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/tools/profile_step.py > /dev/null 2>&1
cd $REPO
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/gemm_traffic.json > $OUT/kernel_traffic.json 2> $OUT/kernel_traffic.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/gemm_traffic.json profiles/gemm_traffic.json
(timeout 400 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json; grep -o '"traffic": [^,]*' $OUT/bench.json
