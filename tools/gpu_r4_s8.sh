#!/bin/bash
# (historical: the feature under test -- GroupNorm statistics from the producing GEMM -- was removed after these sessions: DESIGN.md §7, profiles/r4_gn_producer_stats_ab.txt)
# round 4, GPU session 8: what the cross-lane instructions of the statistics reduction do on the hardware (probe), one kernel test with
# its full report, the norms alone, and the per-kernel times of a step with the feature on (rocprofv3 kernel trace).
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4_s8
mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 tools/probe_permlane.hip -o /tmp/probe > /dev/null 2>&1 && /tmp/probe > $OUT/probe.txt 2>&1
cut -c1-400 $OUT/probe.txt
(timeout 300 python -m pytest tests/test_kernels.py -x -q -m gpu -k "statistics and conv_temb_t21" --tb=short 2>&1 | tail -40) > $OUT/test_one.txt
cat $OUT/test_one.txt
(timeout 300 python tools/bench_gn_stats.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn_stats.txt
cat $OUT/bench_gn_stats.txt
B="--no-cpu-baseline --no-vae --no-roofline"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py $B > $OUT/bench_prof.json 2> $OUT/bench_prof.err
cd $REPO
python tools/kernel_trace_summary.py $OUT/stats > $OUT/kernel_step_summary.txt 2> $OUT/kernel_step_summary.err
rm -rf $OUT/stats
head -45 $OUT/kernel_step_summary.txt | cut -c1-150
