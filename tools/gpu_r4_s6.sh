#!/bin/bash
# round 4, GPU session 6: same-box A/B of the per-call time-embedding table (PCDM_TIME_TABLE=0), then the whole -m gpu suite.
set -u
OUT=gpurun_out/r4_s6
mkdir -p $OUT
B="--no-cpu-baseline --no-vae --no-roofline"
for i in 1 2; do
(timeout 300 python bench.py $B) > $OUT/bench_new_$i.json 2>$OUT/bench_new_$i.err
(PCDM_TIME_TABLE=0 timeout 300 python bench.py $B) > $OUT/bench_notable_$i.json 2>/dev/null
done
for f in new_1 notable_1 new_2 notable_2; do echo $f; cut -c1-130 $OUT/bench_$f.json; done
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40) > $OUT/tests.txt
tail -6 $OUT/tests.txt
