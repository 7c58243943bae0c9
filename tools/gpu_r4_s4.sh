#!/bin/bash
# round 4, GPU session 4: the software-pipelined self-attention kernel (flash_attn_pipe_kernel): parity tests, micro-benchmark A/B against
# (historical: PCDM_ATTN_PIPE selected flash_attn_pipe_kernel, removed after this session -- profiles/r4_bench_attn_pipe_ab.txt)
# the round-2 kernel (PCDM_ATTN_PIPE=0), end-to-end A/B.
set -u
OUT=gpurun_out/r4_s4
mkdir -p $OUT
B="--no-cpu-baseline --no-vae --no-roofline"
(timeout 600 python -m pytest tests/test_kernels.py tests/test_ref_funcs.py tests/test_pipeline.py -k "flash_attn or attn or unipc_and_identities" -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -12) > $OUT/tests.txt
tail -4 $OUT/tests.txt
for v in 1 0; do
(PCDM_ATTN_PIPE=$v timeout 150 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | cut -c1-150) > $OUT/bench_attn_pipe$v.txt
done
(PCDM_ATTN_PIPE=1 PCDM_ATTN_ROWSUM=valu timeout 150 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | cut -c1-150) > $OUT/bench_attn_pipe1_rowsum_valu.txt
for f in pipe1 pipe0 pipe1_rowsum_valu; do echo $f; cat $OUT/bench_attn_$f.txt; done
for i in 1 2; do
(timeout 300 python bench.py $B) > $OUT/bench_new_$i.json 2>$OUT/bench_new_$i.err
(PCDM_ATTN_PIPE=0 timeout 300 python bench.py $B) > $OUT/bench_nopipe_$i.json 2>/dev/null
done
for f in new_1 nopipe_1 new_2 nopipe_2; do echo $f; cut -c1-130 $OUT/bench_$f.json; done
