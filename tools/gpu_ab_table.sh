# same-box A/B of two tuning tables: bash tools/gpu_ab_table.sh <other table> <outdir under gpurun_out> [extra bench.py flags]
set -u
OUT=gpurun_out/${2:-ab}
EXTRA="${3:-}"
mkdir -p $OUT
for i in 1 2; do
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline $EXTRA) > $OUT/bench_committed_$i.json 2>/dev/null
(PCDM_TUNING_TABLE=$1 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline $EXTRA) > $OUT/bench_other_$i.json 2>/dev/null
done
(timeout 200 python tools/profile_step.py 2>&1 | grep -v amdgpu.ids) > $OUT/step_committed.txt
for f in committed_1 other_1 committed_2 other_2; do cut -c1-140 $OUT/bench_$f.json; done; head -1 $OUT/step_committed.txt; grep -c layernorm $OUT/step_committed.txt
