#!/bin/bash
# round 4, GPU session 10: does the second (dup_rows) epilogue repetition that every convolution instance carries cost the
# convolutions that never use it?  (The statistics instances of the removed GroupNorm experiment had no such repetition and their
# convolutions measured 3-9 % faster.)  A/B of two builds on one box: libpcdm.so against libpcdm_nodup.so (-DPCDM_NO_DUP_ROWS around
# the repetition), both with PCDM_SHARE_CFG_PREFIX=0; convolution micro-timings (tools/bench_conv.py) and the bench line.
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r4_s10
mkdir -p $OUT
export TMPDIR=/tmp
export PCDM_SHARE_CFG_PREFIX=0
ALT=pcdms_amd/lib/libpcdm_nodup.so
B="--no-cpu-baseline --no-vae --no-roofline"
for i in 1 2; do
(timeout 120 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids) > $OUT/conv_base_$i.txt
(timeout 120 python tools/with_lib.py $ALT tools/bench_conv.py 2>&1 | grep -v amdgpu.ids) > $OUT/conv_nodup_$i.txt
paste -d'|' $OUT/conv_base_$i.txt $OUT/conv_nodup_$i.txt | cut -c1-90,100-140
done
for i in 1 2 3; do
  (timeout 300 python bench.py $B) > $OUT/bench_base_$i.json 2>/dev/null
  (timeout 300 python tools/with_lib.py $ALT bench.py $B) > $OUT/bench_nodup_$i.json 2>/dev/null
  echo "run $i: base $(grep -o '"value": [0-9.]*' $OUT/bench_base_$i.json | head -1) nodup $(grep -o '"value": [0-9.]*' $OUT/bench_nodup_$i.json | head -1)"
done
