# Round 6, GPU session 29: the VAE decoder's three Upsample2D convolutions as their phase decomposition too (pcdms_amd/vae.py; Python only: the
# kernels are the ones of session 28).  Full GPU suite, the default bench line (its VAE encode / decode timing), the three-stage chain.
set -u
OUT=gpurun_out/r6_s29
mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2) > $OUT/smoke.txt
(timeout 500 python bench.py) > $OUT/bench.json 2> $OUT/bench.err
(PCDM_PHASE_UPSAMPLE=0 timeout 500 python bench.py --no-cpu-baseline --no-roofline) > $OUT/bench_gather.json 2>/dev/null
(timeout 400 python tools/bench_three_stage.py 2>&1 | grep -v amdgpu.ids | tail -2) > $OUT/three_stage.json
cat $OUT/gpu_tests.txt $OUT/smoke.txt
for f in bench bench_gather; do python - $OUT/$f.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j["config"]
print(sys.argv[1].split("/")[-1], j["value"], c["ms_per_denoise_step"], "vae enc/dec ms", c.get("vae_encode_ms"), c.get("vae_decode_uint8_ms"), "incl vae", c.get("images_per_s_incl_vae"))
PY
done
cut -c1-420 $OUT/three_stage.json
