# Round 6, GPU session 8: which part of the twins table loses end to end?  (176-row tiles only / + twins / + twins on convs and GEGLU only / + re-tuned ln keys)
set -u
OUT=gpurun_out/r6_s8
mkdir -p $OUT
for i in 1 2 3; do
for t in r6_tiles22 r6_twins_noln r6_twins_convs r6_twins_ln; do
(PCDM_TUNING_TABLE=tools/ab/gfx950_$t.json timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_${t}_$i.json 2>/dev/null
done
done
for f in $OUT/bench_*.json; do echo $f; cut -c1-120 $f; done
