# Round 6, GPU session 24: weights one launch ahead on a side stream (ops.WeightPrefetch / pcdm_prefetch; PCDM_WEIGHT_PREFETCH=1).
# Bound measured first (tools/probe_prefetch.py, tools/probe_cold_launch.py): a level-3 convolution 54.9 us cold -> 47.5 us prefetched, the prefetch
# kernel 13.5 us per 29.5 MB and free for a concurrent dense convolution; summed over the weight-heavy launches of levels 1-3 <= 0.25 ms (2 %).
# Kill criterion: adopted (default on) only if three interleaved pairs gain >= 0.5 % and the outputs are bit-identical.
set -u
OUT=gpurun_out/r6_s24
mkdir -p $OUT
python - > $OUT/equal.txt 2>&1 <<'PY'
import os, torch, sys
sys.path.insert(0, '.')
from oracle.pipeline import synth_inputs
from oracle.unet import UNetConfig, synth_state_dict
from pcdms_amd import ops, pipeline as PL
from pcdms_amd.schedulers import DDIMScheduler
from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
from tests.test_unet import _kwargs
dev = torch.device("cuda:0")
cfg = UNetConfig()
m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg)); m.load_state_dict(synth_state_dict(cfg, seed=0)); m.to(dev)
sched = lambda: DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1)
inp = {k: v.to(dev) for k, v in synth_inputs(cfg, 64, 88, 4).items()}
outs = {}
for on in (False, True):
    PL.WEIGHT_PREFETCH = on
    ops.PREFETCH = None
    pipe = PL.Stage2_InpaintDiffusionPipeline(m, sched())
    pipe(height=512, width=704, num_images_per_prompt=4, guidance_scale=2.0, num_inference_steps=6, output_type="latent", **inp)
    torch.cuda.synchronize()
    outs[on] = pipe._st["lat"].clone()
    if on:
        print("plan entries", len(ops.PREFETCH.plan), "prefetches per step", sum(1 for j in range(1, len(ops.PREFETCH.plan)) if ops.PREFETCH.plan[j][1] >= ops.PREFETCH.min_bytes))
print("bit-identical:", torch.equal(outs[False], outs[True]))
PY
cat $OUT/equal.txt | grep -v amdgpu.ids
for i in 1 2 3; do
(timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_off_$i.json 2>/dev/null
(PCDM_WEIGHT_PREFETCH=1 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_on_$i.json 2>$OUT/bench_on_$i.err
done
(PCDM_WEIGHT_PREFETCH=1 PCDM_PREFETCH_AHEAD=2 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_ahead2.json 2>/dev/null
(PCDM_WEIGHT_PREFETCH=1 PCDM_PREFETCH_WAVES=1024 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_w1024.json 2>/dev/null
(PCDM_WEIGHT_PREFETCH=1 PCDM_PREFETCH_MIN_MB=8 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_min8.json 2>/dev/null
(PCDM_WEIGHT_PREFETCH=1 PCDM_PREFETCH_MIN_MB=1 timeout 300 python bench.py --no-cpu-baseline --no-vae --no-roofline) > $OUT/bench_min1.json 2>/dev/null
for f in off_1 on_1 off_2 on_2 off_3 on_3 ahead2 w1024 min8 min1; do echo $f; cut -c1-120 $OUT/bench_$f.json; done
tail -3 $OUT/bench_on_1.err
