#!/bin/bash
# Round 6, the committed measurements in one GPU session: HBM traffic passes on the final kernel sources, the bench line that quotes them, rocprofv3
# kernel stats + the per-step summary with FLOP-weighted family lines, the per-launch step breakdown, SQ counters, anatomy, micro-benchmarks, the
# three-stage chain.   usage: bash tools/gpu_r6_final.sh [name]
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-r6_final}
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
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --no-cpu-baseline --no-roofline --no-vae > $OUT/bench_prof.json 2> $OUT/bench_prof.err
cd $REPO
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
(timeout 150 python tools/profile_step.py --flops-json $OUT/step_flops.json 2>&1 | grep -v amdgpu.ids) > $OUT/step_breakdown.txt
python tools/kernel_trace_summary.py $OUT/stats --flops $OUT/step_flops.json > $OUT/kernel_step_summary.txt 2> $OUT/kernel_step_summary.err
rm -rf $OUT/stats
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
timeout 200 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/step_a -- python $REPO/tools/profile_step.py > $OUT/step_a.log 2>&1
cd $REPO
python tools/pmc_summary.py $OUT/step_a $OUT/step_a > $OUT/pmc_step.json 2>$OUT/pmc_step.err
rm -rf $OUT/step_a
cd /tmp
for x in 0 1; do
PCDM_ATTN_XCD=$x timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_attn_fetch_$x -- python $REPO/tools/pmc_attn.py > /dev/null 2>&1
done
cd $REPO
for x in 0 1; do python tools/pmc_attn_fetch.py $OUT/pmc_attn_fetch_$x > $OUT/attn_fetch_xcd$x.json 2> $OUT/attn_fetch_xcd$x.err; rm -rf $OUT/pmc_attn_fetch_$x; done
(timeout 120 python tools/gemm_anatomy.py 2>&1 | grep -v amdgpu.ids) > $OUT/gemm_anatomy.txt
(timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_attn.txt
(timeout 120 python tools/bench_gn.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn.txt
(timeout 200 python tools/bench_ln_gemm.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_ln_gemm.txt
(timeout 400 python tools/bench_three_stage.py 2>&1 | grep -v amdgpu.ids | tail -3) > $OUT/three_stage.json
(timeout 300 python bench.py --no-cpu-baseline --no-vae --attn fp8 2>/dev/null) > $OUT/bench_attn_fp8.json
(timeout 300 python bench.py --no-cpu-baseline --no-vae --batch 8 2>/dev/null) > $OUT/bench_batch8.json
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2) > $OUT/smoke.txt
cat $OUT/attn_fetch_xcd0.json $OUT/attn_fetch_xcd1.json; cat $OUT/gpu_tests.txt $OUT/smoke.txt
cut -c1-400 $OUT/bench.json; tail -12 $OUT/kernel_step_summary.txt; tail -4 $OUT/step_breakdown.txt; cat $OUT/three_stage.json | cut -c1-300
