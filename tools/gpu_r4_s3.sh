#!/bin/bash
# round 4, GPU session 3: re-run of the tests fixed after session 2 + the new rowgemm tiles; rowgemm micro-benchmark with tiles 35 / 36;
# re-tune of the K = 320 residual linears with them; SQ counter passes of a whole step and of the attention kernel.
set -u
OUT=gpurun_out/r4_s3
mkdir -p $OUT
(timeout 900 python -m pytest tests/test_pipeline.py tests/test_fullsize_parity.py::test_stage3_full_size tests/test_kernels.py -k "pipeline or stage3 or rowgemm or groupnorm or dup_rows" -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -30) > $OUT/tests.txt
tail -4 $OUT/tests.txt
(timeout 200 python tools/bench_rowgemm.py 2>&1 | grep -v amdgpu.ids) > $OUT/bench_rowgemm.txt
cat $OUT/bench_rowgemm.txt
(timeout 300 python tools/retune_keys.py 45056,320,320 22528,320,320 2>&1 | grep -v amdgpu.ids) > $OUT/retune.txt
cat $OUT/retune.txt
bash tools/gpu_profile_session.sh r4_s3/pmc > $OUT/pmc.log 2>&1
ls -la $OUT/pmc; head -c 1500 $OUT/pmc/pmc_step.json
