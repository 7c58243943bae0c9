# Round 6, GPU session 31 (last): the tuning table gained seven add-only keys (csrc/tuning_table.inc is part of the kernel-source hash): the HBM traffic
# passes once more on exactly the committed sources, and the default bench line.
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r6_s31
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/tools/profile_step.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/tools/profile_step.py > /dev/null 2>&1
cd $REPO
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/gemm_traffic.json > $OUT/kernel_traffic.json 2> $OUT/kernel_traffic.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/gemm_traffic.json profiles/gemm_traffic.json
(timeout 500 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json; grep -o '"traffic": [0-9a-z]*' $OUT/bench.json
