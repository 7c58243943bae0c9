#!/bin/bash
# The round's committed measurements in one GPU session: bench line, rocprofv3 kernel stats of the same command, per-launch step
# breakdown, HBM traffic passes, GEMM anatomy, attention / GroupNorm micro-benchmarks.   usage: bash tools/gpu_final_session.sh <name>
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-final}
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 400 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline --no-roofline --no-vae > $OUT/bench_prof.json 2> $OUT/bench_prof.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/tools/profile_step.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/tools/profile_step.py > /dev/null 2>&1
cd $REPO
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/gemm_traffic.json > $OUT/kernel_traffic.json 2> $OUT/kernel_traffic.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python tools/kernel_trace_summary.py $OUT/stats > $OUT/kernel_step_summary.txt 2> $OUT/kernel_step_summary.err
rm -rf $OUT/stats
(timeout 120 python tools/profile_step.py 2>&1 | grep -v amdgpu.ids) > $OUT/step_breakdown.txt
(timeout 120 python tools/gemm_anatomy.py 2>&1 | grep -v amdgpu.ids) > $OUT/gemm_anatomy.txt
(timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_attn.txt
(timeout 120 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn.txt
cut -c1-300 $OUT/bench.json; head -12 $OUT/kernel_stats.csv | cut -c1-160; head -c 600 $OUT/kernel_traffic.json
