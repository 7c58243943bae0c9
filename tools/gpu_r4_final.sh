#!/bin/bash
# round 4, final measurement session: the bench line (roofline + pinned CPU baseline), rocprofv3 kernel stats of the same command and the
# per-step summary, HBM traffic passes (FETCH_SIZE / WRITE_SIZE) + the bench line that quotes them, per-launch step breakdown, GEMM
# anatomy, attention / GroupNorm micro-benchmarks, the fp8 / batch-8 lines and the three-stage run.   usage: bash tools/gpu_r4_final.sh
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4_final
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-vae --no-roofline"
(timeout 600 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py $B > $OUT/bench_prof.json 2> $OUT/bench_prof.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/tools/profile_step.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/tools/profile_step.py > /dev/null 2>&1
cd $REPO
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/gemm_traffic.json > $OUT/kernel_traffic.json 2> $OUT/kernel_traffic.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/gemm_traffic.json profiles/gemm_traffic.json
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python tools/kernel_trace_summary.py $OUT/stats > $OUT/kernel_step_summary.txt 2> $OUT/kernel_step_summary.err
rm -rf $OUT/stats
(timeout 400 python bench.py --no-cpu-baseline) > $OUT/bench_with_traffic.json 2> $OUT/bench_with_traffic.err
grep -o '"traffic": [^,]*' $OUT/bench_with_traffic.json
(timeout 120 python tools/profile_step.py 2>&1 | grep -v amdgpu.ids) > $OUT/step_breakdown.txt
(timeout 120 python tools/gemm_anatomy.py 2>&1 | grep -v amdgpu.ids) > $OUT/gemm_anatomy.txt
(timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_attn.txt
(timeout 120 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn.txt
(timeout 300 python bench.py $B --attn fp8) > $OUT/bench_fp8.json 2>/dev/null
(timeout 300 python bench.py $B --batch 8) > $OUT/bench_batch8.json 2>/dev/null
(timeout 300 python bench.py $B) > $OUT/bench_bf16_same_box.json 2>/dev/null
(timeout 500 python tools/bench_three_stage.py 2>$OUT/three_stage.err | tail -1) > $OUT/three_stage.json
for f in fp8 batch8 bf16_same_box; do cut -c1-140 $OUT/bench_$f.json; done; cut -c1-400 $OUT/three_stage.json
head -14 $OUT/kernel_step_summary.txt | cut -c1-140
