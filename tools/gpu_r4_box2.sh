#!/bin/bash
# round 4, after the final session: the default bench line once more on another box (box-to-box spread; the CPU-baseline sweep now
# includes 16 threads and the container's cgroup CPU quota), and the GPU test suite at HEAD.   usage: bash tools/gpu_r4_box2.sh
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4_box2
mkdir -p $OUT
export TMPDIR=/tmp
cat /sys/fs/cgroup/cpu.max > $OUT/cgroup_cpu_max.txt 2>&1
cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us >> $OUT/cgroup_cpu_max.txt 2>&1
nproc >> $OUT/cgroup_cpu_max.txt
(time timeout 900 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json
grep -o '"per_thread_count": {.*"config1"' $OUT/bench.json | cut -c1-900
tail -4 $OUT/bench.err
cat $OUT/cgroup_cpu_max.txt
(timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -6) > $OUT/gpu_tests.txt
cat $OUT/gpu_tests.txt
