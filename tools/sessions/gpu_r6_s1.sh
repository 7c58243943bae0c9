# Round 6, GPU session 1: the practical MFMA roof probe, the 176-row tiles (parity on the GPU, per-shape A/B against the committed
# tiles, no-load ablation), baseline bench line of the committed table.
set -u
OUT=gpurun_out/r6_s1
mkdir -p $OUT
cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o mfma_power mfma_power.hip 2>&1 | grep -i error; cd ../..
(timeout 300 tools/ubench/mfma_power) > $OUT/ubench_mfma_power.txt 2>&1
python -m pytest tests/test_kernels.py -m gpu -x -q -k "uneven_176 or full_row_tiles or gemm_bias" 2>&1 | grep -v amdgpu.ids | tail -5 > $OUT/tests.txt
(timeout 900 python tools/bench_tile176.py 2>&1 | grep -v amdgpu.ids) > $OUT/tile176.txt
(timeout 300 python tools/ablate_gemm.py 21,22 2>&1 | grep -v amdgpu.ids) > $OUT/ablate_21_22.txt
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_base_1.json 2>/dev/null
tail -3 $OUT/tests.txt; cat $OUT/ubench_mfma_power.txt; cat $OUT/tile176.txt | cut -c1-260; cat $OUT/ablate_21_22.txt; cut -c1-200 $OUT/bench_base_1.json
