set -u
OUT=gpurun_out/r3_s35
mkdir -p $OUT
PREV=pcdms_amd/lib/libpcdm_prev.so
python -m pytest tests/test_kernels.py tests/test_unet_ctx.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
(timeout 120 python tools/gemm_anatomy.py 2>&1 | grep -v amdgpu.ids | grep "res=True" | cut -c1-230) > $OUT/anatomy_new.txt
(timeout 120 python tools/with_lib.py $PREV tools/gemm_anatomy.py 2>&1 | grep -v amdgpu.ids | grep "res=True" | cut -c1-230) > $OUT/anatomy_prev.txt
for i in 1 2; do
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_new_$i.json 2>/dev/null
(timeout 300 python tools/with_lib.py $PREV bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_prev_$i.json 2>/dev/null
done
paste -d'\n' $OUT/anatomy_prev.txt $OUT/anatomy_new.txt | grep "tile 5 \|tile 18 \|tile 4 \|tile 6 " | cut -c1-215
for f in new_1 prev_1 new_2 prev_2; do cut -c1-120 $OUT/bench_$f.json; done
