# Round 6, GPU session 6: the 16x16x32-fragment twins (24 / 25 / 29 / 30) of the 32x32x16 tiles, per launch against the committed choice; kernel tests
set -u
OUT=gpurun_out/r6_s6
mkdir -p $OUT
python -m pytest tests/test_kernels.py -m gpu -x -q -k "twins or uneven_176 or rowvec_step" 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/tests.txt
(timeout 1500 python tools/bench_tile176.py --tiles 24,25,29,30 2>&1 | grep -v amdgpu.ids) > $OUT/tile_f16.txt
tail -3 $OUT/tests.txt; cut -c1-330 $OUT/tile_f16.txt
