# Round 6, GPU session 15: which table ships.  v4 = in-step passes at batch 4 + batch 8 (A/B'd in s12), v5 = + batch 16 (s13), v7 = + 512-wide + stage 3 (s14).
# Interleaved A/B at batch 4 (the metric's configuration: later passes changed keys it shares), then the three-stage chain under v5 and v7.
set -u
OUT=gpurun_out/r6_s15
mkdir -p $OUT
S12=gpurun_out/r6_s12; S14=gpurun_out/r6_s14
cp tools/ab/gfx950_v4.json tools/ab/gfx950_v5.json tools/ab/gfx950_v7.json $OUT/
for i in 1 2 3; do
for v in v4 v5 v7; do
(PCDM_TUNING_TABLE=$OUT/gfx950_$v.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_${v}_$i.json 2>/dev/null
done
done
for v in v5 v7; do
(PCDM_TUNING_TABLE=$OUT/gfx950_$v.json timeout 400 python tools/bench_three_stage.py 2>&1 | grep -v amdgpu.ids | tail -2) > $OUT/three_stage_$v.json
done
for v in v4 v5 v7; do for i in 1 2 3; do echo ${v}_$i; cut -c1-120 $OUT/bench_${v}_$i.json; done; done
for v in v5 v7; do echo $v; cut -c1-400 $OUT/three_stage_$v.json; done
