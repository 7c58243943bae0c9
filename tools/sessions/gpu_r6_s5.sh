# Round 6, GPU session 5: the whole -m gpu suite on the ABI-4 sources (parity values -> gpurun_out/parity_values.json), then a bench line
set -u
OUT=gpurun_out/r6_s5
mkdir -p $OUT
(timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -25) > $OUT/gpu_tests.txt
(timeout 600 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
tail -25 $OUT/gpu_tests.txt; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err
