#!/bin/bash
# (historical: the feature under test -- GroupNorm statistics from the producing GEMM -- was removed after these sessions: DESIGN.md §7, profiles/r4_gn_producer_stats_ab.txt)
# round 4, GPU session 9: the statistics reduction with the inline-asm lane swaps (tools/debug_stats.py, the kernel and schedule tests),
# producer -> norm pair timings and the norms alone, end-to-end A/B (PCDM_GN_PRODUCER_STATS=0/1 interleaved).
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4_s9
mkdir -p $OUT
export TMPDIR=/tmp
(timeout 120 python tools/debug_stats.py 2>&1 | grep -v amdgpu.ids | grep "bad entries") > $OUT/debug_stats.txt
cat $OUT/debug_stats.txt
(timeout 600 python -m pytest tests/test_kernels.py tests/test_unet_ctx.py -q -m gpu -k "statistics" --tb=short 2>&1 | tail -25) > $OUT/tests_stats.txt
cat $OUT/tests_stats.txt
(timeout 300 python tools/bench_gn_stats.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_gn_stats.txt
cat $OUT/bench_gn_stats.txt
B="--no-cpu-baseline --no-vae --no-roofline"
for i in 1 2 3; do
  for v in 0 1; do
    (PCDM_GN_PRODUCER_STATS=$v timeout 300 python bench.py $B) > $OUT/bench_stats${v}_$i.json 2> $OUT/bench_stats${v}_$i.err
    echo "stats=$v run $i: $(grep -o '"value": [0-9.]*' $OUT/bench_stats${v}_$i.json | head -1) $(grep -o '"ms_per_denoise_step": [0-9.]*' $OUT/bench_stats${v}_$i.json)"
  done
done
