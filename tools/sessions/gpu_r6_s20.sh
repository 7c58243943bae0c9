# Round 6, GPU session 20: the committed sources once more after the "composed weights are packed once" cleanup: full GPU suite, smoke, the default bench line.
set -u
OUT=gpurun_out/r6_s20
mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4) > $OUT/gpu_tests.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2) > $OUT/smoke.txt
(timeout 500 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_2.json 2>/dev/null
cat $OUT/gpu_tests.txt $OUT/smoke.txt; cut -c1-300 $OUT/bench.json; cut -c1-120 $OUT/bench_2.json
