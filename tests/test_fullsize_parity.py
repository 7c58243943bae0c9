"""Full-size, full-length parity for BASELINE.json configs[1] (and the per-GPU share of configs[2]) against the committed
fp32-oracle fixture ``tests/golden/fullsize_config2.npz`` (made by ``tests/golden/make_fullsize_config2_fixture.py`` in the
build container: 868.9 M-parameter seeded UNet with random norm affines, latent 64x88, N = 4 => UNet batch 8, 50 DDIM
steps, guidance 2.0, then the full-size VAE decode to uint8 canvases).

What is compared (reference: /root/reference/src/pipelines/stage2_inpaint_pipeline.py:494-532):

* ONE forward at the oracle's own state -- eps at steps 0 / 10 / 25 / 49 with the fixture's latents as input (M = 45 056
  rows through every GEMM / conv tile the bench uses): rel-L2 <= FWD_TOL;
* the 50-step hipGraph trajectory from the same initial latents -- latents before steps 10 / 25 / 49 and the final latents:
  rel-L2 <= TRAJ_TOL (bf16 activations between kernels vs the fp32 oracle, error carried through 50 steps).  The latents are
  ~85 % a deterministic fp32 rescale of the initial noise (cosine(lat_final, lat_0) = 0.988 in the fixture), so the check that
  bites is the EPS-DRIVEN PART ``lat_i - c_x(i) * lat_0`` (c_x(i) = the product of the DDIM x-coefficients up to step i, obtained
  by running the oracle scheduler with eps = 0): everything the UNet contributed, rel-L2 <= EPS_PART_TOL;
* uint8 canvases after VAE decode (HIP VAE on the HIP latents vs oracle VAE on the oracle latents): mean |diff| <= PIX_TOL
  levels of 255, and per-canvas mean within PIX_MEAN_TOL levels;
* configs[2]'s per-GPU share (N = 8, UNet batch 16): forwards at TWO oracle states (step 0: ``b16_eps``; step 25:
  ``fullsize_b16_mid.npz``) and a 50-step N = 8 hipGraph run whose first four samples reproduce the oracle's N = 4 trajectory
  (samples are independent: same pair, per-sample noise) and whose two halves are mutually consistent;
* configs[4]'s per-GPU share (fp8 attention, N = 16, UNet batch 32): fp8 forward vs the bf16 path, and run-to-run determinism;
* configs[0] -- the reference's own CPU-runnable case (one 256x256 pair: latent 32x64, N = 1, 20 DDIM steps): the COMPLETE call
  against ``fullsize_config0.npz`` (same weights; M = 4096 rows, UNet batch 2 -- other tiles, other split-K than configs[1]).

Tolerances are stated here and were set from the measured values on MI355X (recorded in DESIGN.md §5)."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest
import torch

from oracle.pipeline import build_conditioning, synth_inputs
from oracle.schedulers import DDIMOracle
from oracle.unet import UNetConfig, synth_state_dict
from pcdms_amd.pipeline import Stage2_InpaintDiffusionPipeline
from pcdms_amd.schedulers import DDIMScheduler
from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
from tests.parity_record import check as record_check
from tests.test_schedulers import SD21
from tests.test_unet import _kwargs

FIXTURE = Path(__file__).resolve().parent / "golden" / "fullsize_config2.npz"
FWD_TOL = 2.5e-2      # one forward, rel-L2 of the guided eps: the suite's forward tolerance (tests/test_unet.py); measured 0.95e-2 .. 1.67e-2
FP8_FWD_TOL = 6e-2    # one forward with fp8 (e4m3) attention operands -- configs[4]'s own, looser tolerance
TRAJ_TOL = 1.5e-3     # latents along / at the end of the 50-step trajectory (measured 0.7e-3 .. 0.8e-3: the DDIM update is dominated by
                      # its deterministic rescale of the latents, which both sides compute in fp32)
EPS_PART_TOL = 1.2e-2  # the eps-driven part of the same latents, lat_i - c_x(i) lat_0: the accumulated UNet contribution (measured 0.52e-2 .. 0.67e-2;
                       # a single forward is at 0.95e-2 .. 1.67e-2: the per-step errors partly average out along the trajectory)
FP8_VS_BF16_TOL = 2e-2  # guided eps with fp8 attention vs the bf16-attention path, same weights and inputs (UNet batch 32)
PIX_TOL = 1.0         # mean absolute difference in uint8 levels over a canvas (measured 0.34; bf16 VAE alone 0.34)
PIX_MEAN_TOL = 0.25   # |mean(canvas) - mean(oracle canvas)| in uint8 levels (measured <= 0.022)


def _rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(np.asarray(b, dtype=np.float32))
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def full(request):
    if not FIXTURE.exists():
        pytest.fail(f"{FIXTURE} missing: run tests/golden/make_fullsize_config2_fixture.py")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pcdms_amd import _lib
    _lib.load()
    fx = np.load(FIXTURE)
    assert str(fx["torch_version"]) == torch.__version__, "seeded CPU generators: the fixture was made with another torch build"
    cfg = UNetConfig()
    sd = synth_state_dict(cfg, seed=0, random_affine=True)
    dev = torch.device("cuda:0")
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(sd)
    m.to(dev)
    return fx, cfg, m, dev


def _guided_eps(m, cfg, inp, lat, t, N, dev):
    c = build_conditioning(inp["masked_latents"], inp["s_img_proj_f"], inp["st_pose_f"], inp["pred_t_img_embed"], N, True)
    x = torch.cat([lat] * 2)
    eps = m(torch.cat([x, c["mask"], c["masked_latents"]], 1).to(dev), t, encoder_hidden_states=c["feature_f"].to(dev),
            class_labels=c["prior_embed"].to(dev), my_pose_cond=c["pose_cond"].to(dev)).sample.float().cpu()
    u, cn = eps.chunk(2)
    return u + 2.0 * (cn - u)


def _ddim_x_coefficients(steps: int):
    """c_x(i): latents before step i = c_x(i) * lat_0 + (eps-driven part); from the ORACLE scheduler run with eps = 0."""
    sch = DDIMOracle()
    sch.set_timesteps(steps)
    x, cx = torch.ones(1, dtype=torch.float64), {0: 1.0}
    for i, t in enumerate(sch.timesteps):
        x = sch.step(torch.zeros_like(x), t, x)
        cx[i + 1] = float(x)
    return cx


@pytest.mark.gpu
def test_single_forward_at_oracle_states(full):
    fx, cfg, m, dev = full
    N, h, w = 4, 64, 88
    inp = synth_inputs(cfg, h, w, N)
    sch = DDIMOracle()
    sch.set_timesteps(int(fx["steps"]))
    rels = {}
    for i in [int(v) for v in fx["check"]]:
        eps = _guided_eps(m, cfg, inp, torch.from_numpy(fx[f"lat_{i}"]), sch.timesteps[i], N, dev)
        rels[i] = _rel(eps, fx[f"eps_{i}"])
    print("full-size single-forward rel-L2 (guided eps) per step:", {k: round(v, 5) for k, v in rels.items()})
    for k, v in rels.items():
        record_check(f"configs1.forward.step{k}", v, FWD_TOL)


@pytest.mark.gpu
def test_50_step_trajectory_and_pixels(full):
    fx, cfg, m, dev = full
    N, h, w = 4, 64, 88
    steps = int(fx["steps"])
    inp = synth_inputs(cfg, h, w, N)
    assert np.array_equal(inp["latents"].numpy(), fx["lat_0"])     # same seeded inputs as the fixture run
    pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler.from_config(SD21))
    seen = {}
    out = pipe(height=h * 8, width=w * 8, masked_latents=inp["masked_latents"].to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
               st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev), latents=inp["latents"].to(dev),
               num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=steps, output_type="latent",
               callback=lambda i, t, lat: seen.__setitem__(i + 1, lat)).latents          # hipGraph replay + per-step snapshot
    assert pipe._graph is not None and len(seen) == steps
    rels = {i: _rel(seen[i], fx[f"lat_{i}"]) for i in [int(v) for v in fx["check"]] if i > 0}
    rels["final"] = _rel(out, fx["lat_final"])
    print("full-size 50-step trajectory rel-L2 (latents before step i / final):", {k: round(v, 5) for k, v in rels.items()})
    for k, v in rels.items():
        record_check(f"configs1.trajectory.{k}", v, TRAJ_TOL)
    # the eps-driven part: subtract the deterministic image of the initial noise, c_x(i) * lat_0
    cx = _ddim_x_coefficients(steps)
    lat0 = torch.from_numpy(fx["lat_0"])
    parts = {}
    for i in [int(v) for v in fx["check"]] + [steps]:
        if i == 0:
            continue
        hip = (seen[i] if i < steps else out).float().cpu() - cx[i] * lat0
        ref = torch.from_numpy(fx[f"lat_{i}"] if i < steps else fx["lat_final"]) - cx[i] * lat0
        parts[i] = ((hip - ref).norm() / ref.norm()).item()
        assert ref.norm() > 0.02 * lat0.norm()
    print("full-size 50-step trajectory, eps-driven part rel-L2 (before step i / after the last):", {k: round(v, 5) for k, v in parts.items()})
    for k, v in parts.items():
        record_check(f"configs1.eps_part.{k}", v, EPS_PART_TOL)
    # a second call with the graph already captured reproduces the first bit for bit
    out2 = pipe(height=h * 8, width=w * 8, masked_latents=inp["masked_latents"].to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
                st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev), latents=inp["latents"].to(dev),
                num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=steps, output_type="latent").latents
    assert torch.equal(out, out2)
    # ---- pixels: VAE decode + VaeImageProcessor.postprocess (ref :528-532) on the HIP latents
    from oracle import vae as OV
    from pcdms_amd.vae import AutoencoderKL
    vcfg = OV.VAEConfig()
    vae = AutoencoderKL()
    vae.load_state_dict(OV.synth_state_dict(vcfg, 0))
    vae.to(dev)
    u8 = vae.decode_to_uint8(out / vcfg.scaling_factor).cpu().numpy()
    assert u8.shape == (N, h * 8, w * 8, 3)
    pix = {k: float(np.abs(u8[k].astype(np.int32) - fx[f"img_{k}"].astype(np.int32)).mean()) for k in (0, 3)}
    means = np.abs(u8.reshape(N, -1).astype(np.float64).mean(1) - fx["img_mean"])
    # the VAE alone, on the ORACLE's final latents (separates decoder error from trajectory error)
    u8o = vae.decode_to_uint8(torch.from_numpy(fx["lat_final"]).to(dev) / vcfg.scaling_factor).cpu().numpy()
    pix_vae = {k: float(np.abs(u8o[k].astype(np.int32) - fx[f"img_{k}"].astype(np.int32)).mean()) for k in (0, 3)}
    print("full-size uint8 canvases: mean|diff| levels", {k: round(v, 3) for k, v in pix.items()}, "VAE alone", {k: round(v, 3) for k, v in pix_vae.items()},
          "canvas-mean diff", np.round(means, 3).tolist())
    assert max(pix.values()) <= PIX_TOL and max(pix_vae.values()) <= PIX_TOL and means.max() <= PIX_MEAN_TOL, (pix, pix_vae, means)
    record_check("configs1.pixels.mean_abs_levels", max(pix.values()), PIX_TOL)


@pytest.mark.gpu
def test_fp8_attention_forward_at_oracle_states(full):
    """BASELINE.json configs[4]: the same full-size forward with every attention on e4m3 operands.  Its own stated
    tolerance: rel-L2 <= FP8_FWD_TOL against the fp32 oracle (the bf16-attention path: FWD_TOL)."""
    fx, cfg, m, dev = full
    N, h, w = 4, 64, 88
    inp = synth_inputs(cfg, h, w, N)
    sch = DDIMOracle()
    sch.set_timesteps(int(fx["steps"]))
    m.set_attention_precision("fp8")
    try:
        rels = {i: _rel(_guided_eps(m, cfg, inp, torch.from_numpy(fx[f"lat_{i}"]), sch.timesteps[i], N, dev), fx[f"eps_{i}"]) for i in (0, 25, 49)}
    finally:
        m.set_attention_precision("bf16")
    print("full-size single-forward rel-L2 with fp8 attention:", {k: round(v, 5) for k, v in rels.items()})
    for k, v in rels.items():
        record_check(f"configs4.fp8_forward.step{k}", v, FP8_FWD_TOL)


@pytest.mark.gpu
def test_batch16_forward_configs2_share(full):
    fx, cfg, m, dev = full
    if "b16_eps" not in fx:
        pytest.skip("fixture made with --no-b16")
    N, h, w = 8, 64, 88
    inp = synth_inputs(cfg, h, w, N)
    sch = DDIMOracle()
    sch.set_timesteps(int(fx["steps"]))
    eps = _guided_eps(m, cfg, inp, inp["latents"], sch.timesteps[0], N, dev)
    r = _rel(eps, fx["b16_eps"])
    print("full-size UNet-batch-16 forward rel-L2:", round(r, 5))
    record_check("configs2.forward_b16.step0", r, FWD_TOL)
    # second oracle state: the middle of the schedule (tests/golden/make_fullsize_b16_fixture.py)
    mid_path = FIXTURE.parent / "fullsize_b16_mid.npz"
    if not mid_path.exists():
        pytest.fail(f"{mid_path} missing: run tests/golden/make_fullsize_b16_fixture.py")
    from tests.golden.make_fullsize_b16_fixture import mid_state_latents
    mid = np.load(mid_path)
    t = int(mid["t"])
    assert t == int(sch.timesteps[int(mid["step"])])
    lat = mid_state_latents(float(sch.alphas_cumprod[t]), N, h, w)
    assert abs(float(lat.double().abs().sum()) - float(mid["lat_checksum"])) <= 1e-9 * float(mid["lat_checksum"])
    r2 = _rel(_guided_eps(m, cfg, inp, lat, torch.tensor(t), N, dev), mid["eps"])
    print("full-size UNet-batch-16 forward at step 25 rel-L2:", round(r2, 5))
    record_check("configs2.forward_b16.step25", r2, FWD_TOL)


@pytest.mark.gpu
def test_configs2_share_50_step_graph_run(full):
    """BASELINE.json configs[2], one GPU's share: N = 8 samples of one pair (UNet batch 16), all 50 DDIM steps under the hipGraph.
    Samples are independent given the pair, so with per-sample noise = [the N = 4 fixture's noise | fresh noise] the first four
    must reproduce the ORACLE's N = 4 trajectory (fixture), and a second call must reproduce the first bit for bit."""
    fx, cfg, m, dev = full
    N, h, w = 8, 64, 88
    steps = int(fx["steps"])
    inp = synth_inputs(cfg, h, w, 4)
    lat8 = torch.cat([inp["latents"], torch.randn(4, 4, h, w, generator=torch.Generator().manual_seed(77))])
    pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler.from_config(SD21))
    kw = dict(height=h * 8, width=w * 8, masked_latents=inp["masked_latents"].to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
              st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev), num_images_per_prompt=N,
              guidance_scale=2.0, num_inference_steps=steps, output_type="latent")
    out = pipe(latents=lat8.to(dev), **kw).latents
    assert pipe._graph is not None and out.shape == (N, 4, h, w) and bool(torch.isfinite(out).all())
    r = _rel(out[:4], fx["lat_final"])
    cx = _ddim_x_coefficients(steps)[steps]
    lat0 = inp["latents"]
    ref = torch.from_numpy(fx["lat_final"]) - cx * lat0
    rp = ((out[:4].float().cpu() - cx * lat0 - ref).norm() / ref.norm()).item()
    print("configs[2] share, N = 8 50-step run: first four samples vs the oracle rel-L2", round(r, 5), "eps-driven part", round(rp, 5))
    record_check("configs2.trajectory_n8.final", r, TRAJ_TOL)
    record_check("configs2.eps_part_n8.final", rp, EPS_PART_TOL)
    # the other four samples carry the same pair: their statistics match the first four's (a property, not a pin)
    s0, s1 = out[:4].float().std().item(), out[4:].float().std().item()
    assert abs(s0 - s1) <= 0.05 * s0, (s0, s1)
    out2 = pipe(latents=lat8.to(dev), **kw).latents
    assert torch.equal(out, out2)


@pytest.mark.gpu
def test_configs4_share_fp8_batch32(full):
    """BASELINE.json configs[4], one GPU's share (batch 128 over 8 GPUs = N 16, UNet batch 32): the fp8-attention forward against the
    bf16-attention forward of the same weights and inputs (no batch-32 oracle state is committed: 38 TFLOP of fp32 CPU work),
    and bit-exact run-to-run determinism of the fp8 path."""
    fx, cfg, m, dev = full
    N, h, w = 16, 64, 88
    inp = synth_inputs(cfg, h, w, N)
    sch = DDIMOracle()
    sch.set_timesteps(int(fx["steps"]))
    t = sch.timesteps[25]
    a = float(sch.alphas_cumprod[int(t)])
    lat = a ** 0.5 * 0.9 * torch.randn(N, 4, h, w, generator=torch.Generator().manual_seed(31)) + (1 - a) ** 0.5 * inp["latents"]
    ref = _guided_eps(m, cfg, inp, lat, t, N, dev)
    m.set_attention_precision("fp8")
    try:
        e1 = _guided_eps(m, cfg, inp, lat, t, N, dev)
        e2 = _guided_eps(m, cfg, inp, lat, t, N, dev)
    finally:
        m.set_attention_precision("bf16")
    r = ((e1 - ref).norm() / ref.norm()).item()
    print("configs[4] share, UNet batch 32: fp8-attention vs bf16-attention guided eps rel-L2", round(r, 5))
    assert torch.equal(e1, e2) and bool(torch.isfinite(e1).all())
    record_check("configs4.fp8_vs_bf16_b32", r, FP8_VS_BF16_TOL)


@pytest.mark.gpu
def test_config0_complete_run(full):
    """BASELINE.json configs[0]: 1 pair at 256x256 (latent 32x64), num_images_per_prompt = 1, 20 DDIM steps, guidance 2.0 -- the whole
    call on the full-size weights against the fp32 oracle's run of the same call (tests/golden/make_fullsize_config0_fixture.py)."""
    _, cfg, m, dev = full
    fx = np.load(Path(__file__).resolve().parent / "golden" / "fullsize_config0.npz")
    assert str(fx["torch_version"]) == torch.__version__
    N, h, w, steps = 1, 32, 64, int(fx["steps"])
    inp = synth_inputs(cfg, h, w, N)
    assert np.array_equal(inp["latents"].numpy(), fx["lat_0"])
    # single forwards at the oracle's own states
    fw = {}
    sch = DDIMOracle()
    sch.set_timesteps(steps)
    for i in (0, 10, 19):
        eps = _guided_eps(m, cfg, inp, torch.from_numpy(fx[f"lat_{i}"]), int(sch.timesteps[i]), N, dev)
        fw[i] = _rel(eps, fx[f"eps_{i}"].astype(np.float32))
    print("configs[0] guided eps rel-L2 at oracle states:", {k: round(v, 5) for k, v in fw.items()})
    for k, v in fw.items():
        record_check(f"configs0.forward.step{k}", v, FWD_TOL)
    pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler.from_config(SD21))
    seen = {}
    out = pipe(height=h * 8, width=w * 8, masked_latents=inp["masked_latents"].to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
               st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev), latents=inp["latents"].to(dev),
               num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=steps, output_type="latent",
               callback=lambda i, t, lat: seen.__setitem__(i + 1, lat)).latents
    cx = _ddim_x_coefficients(steps)
    lat0 = torch.from_numpy(fx["lat_0"])
    rels, parts = {}, {}
    for i in (5, 10, 15, steps):
        hip = (seen[i] if i < steps else out).float().cpu()
        ref = torch.from_numpy(fx[f"lat_{i}"] if i < steps else fx["lat_final"])
        rels[i] = ((hip - ref).norm() / ref.norm()).item()
        parts[i] = (((hip - cx[i] * lat0) - (ref - cx[i] * lat0)).norm() / (ref - cx[i] * lat0).norm()).item()
    print("configs[0] 20-step run rel-L2:", {k: round(v, 5) for k, v in rels.items()}, "eps-driven part:", {k: round(v, 5) for k, v in parts.items()})
    for k in rels:   # (20 coarse steps: larger eps weight per step)
        record_check(f"configs0.trajectory.{k}", rels[k], 2 * TRAJ_TOL)
        record_check(f"configs0.eps_part.{k}", parts[k], 2 * EPS_PART_TOL)


@pytest.mark.gpu
def test_bf16_hip_path_against_the_reference_precision_budget(full):
    """BASELINE.json asks for "a stated fp16 tolerance"; the reference itself runs hard-cast to fp16
    (/root/reference/stage2_batchtest_inpaint_model.py:123-128, src/pipelines/stage2_inpaint_pipeline.py:431-519).  tests/golden/fp16_budget.npz
    (made by tests/golden/make_fp16_budget.py in the build container: the fp32 oracle with every op output rounded to fp16 and the loop's
    arithmetic in fp16 tensors -- the reference's own numerics) holds how far THAT sits from the fp32 oracle.  Here the bf16 HIP path is put
    next to it on the same weights and inputs:
      * per forward (three oracle states of configs[1], sample 0): bf16 carries 3 mantissa bits less than fp16, so a forward is allowed 12 x the
        fp16 reference's distance (measured 6-9 x) and must not exceed 1.25 x what hard-casting the reference to bf16 would give (measured 0.8 x:
        fp32 residual adds in the epilogues, fp32 split-K slabs);
      * per trajectory (configs[0] in full: 20 DDIM steps): the HIP path keeps latents and scheduler in fp32, the reference rounds them to fp16
        every step, so final latents / eps-driven part / decoded pixels must be within 1.25 x the fp16 reference's own distance (measured 0.6-0.8 x)."""
    _, cfg, m, dev = full
    bpath = Path(__file__).resolve().parent / "golden" / "fp16_budget.npz"
    bx = np.load(bpath)
    assert str(bx["torch_version"]) == torch.__version__
    import json
    budget = json.loads(str(bx["json"]))
    fx = np.load(FIXTURE)
    # ---- per forward
    N, h, w = 1, 64, 88
    inp = synth_inputs(cfg, h, w, N)
    sch = DDIMOracle()
    sch.set_timesteps(50)
    ratios = {}
    for i in (0, 10, 25):
        eps = _guided_eps(m, cfg, inp, torch.from_numpy(fx[f"lat_{i}"][:1]), sch.timesteps[i], N, dev)
        hip = _rel(eps, fx[f"eps_{i}"][:1].astype(np.float32))
        b = budget["forward_configs1"][str(i)]
        ratios[i] = (hip, hip / b["fp16ref_vs_fp32"], hip / b["bf16ref_vs_fp32"])
        record_check(f"budget.forward.step{i}.hip_vs_fp32", hip, FWD_TOL)
        record_check(f"budget.forward.step{i}.x_fp16ref", hip / b["fp16ref_vs_fp32"], 12.0)
        record_check(f"budget.forward.step{i}.x_bf16cast", hip / b["bf16ref_vs_fp32"], 1.25)
    print("forward: (hip vs fp32, x fp16-reference, x bf16-hard-cast):", {k: tuple(round(x, 4) for x in v) for k, v in ratios.items()})
    # ---- per trajectory: configs[0] in full
    N, h, w, steps = 1, 32, 64, 20
    inp = synth_inputs(cfg, h, w, N)
    pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler.from_config(SD21))
    out = pipe(height=h * 8, width=w * 8, masked_latents=inp["masked_latents"].to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
               st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev), latents=inp["latents"].to(dev),
               num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=steps, output_type="latent").latents.float().cpu()
    ref = torch.from_numpy(bx["config0_final_fp32"])
    cx = _ddim_x_coefficients(steps)[steps]
    lat0 = inp["latents"]
    hip_lat = ((out - ref).norm() / ref.norm()).item()
    hip_part = (((out - cx * lat0) - (ref - cx * lat0)).norm() / (ref - cx * lat0).norm()).item()
    b16 = budget["config0"]["fp16"]
    # pixels: the fp32 oracle VAE on both final latents (the UNet / loop precision alone, as in the budget)
    from oracle import vae as ovae
    vcfg = ovae.VAEConfig()
    vsd = ovae.synth_state_dict(vcfg, 0)
    with torch.no_grad():
        img_h = ovae.postprocess_uint8(ovae.decode(vsd, vcfg, out / vcfg.scaling_factor))[0].float()
        img_r = ovae.postprocess_uint8(ovae.decode(vsd, vcfg, ref / vcfg.scaling_factor))[0].float()
    hip_px = (img_h - img_r).abs().mean().item()
    print(f"configs[0] 20 steps: hip vs fp32 latents {hip_lat:.3e} ({hip_lat / b16['final_latents']:.2f} x the fp16 reference's {b16['final_latents']:.3e}), "
          f"eps-driven part {hip_part:.3e} ({hip_part / b16['final_eps_part']:.2f} x {b16['final_eps_part']:.3e}), "
          f"pixels {hip_px:.3f} / 255 ({hip_px / max(b16['pixels_mean_abs_diff_of_255'], 1e-9):.2f} x {b16['pixels_mean_abs_diff_of_255']:.3f})")
    record_check("budget.config0.latents.x_fp16ref", hip_lat / b16["final_latents"], 1.25)
    record_check("budget.config0.eps_part.x_fp16ref", hip_part / b16["final_eps_part"], 1.25)
    record_check("budget.config0.pixels.hip_levels", hip_px, 0.25)
    record_check("budget.config0.pixels.x_fp16ref", hip_px / b16["pixels_mean_abs_diff_of_255"], 2.0)


@pytest.mark.gpu
def test_real_image_inputs(full):
    """Real pictures instead of Gaussian tensors (tests/golden/make_real_image_fixture.py: the reference's own sample image and pose maps through
    the driver's canvas preparation, /root/reference/stage2_batchtest_inpaint_model.py:150-174; only arrays travel): the HIP chain VAE encode (injected
    posterior noise: the target half of the masked latents is the VAE's code of BLACK, not the constant 0 of the synthetic fixtures) -> pose net ->
    3 DDIM steps of the full-size UNet -> VAE decode -> uint8, each stage consuming the HIP output of the one before, against the fp32 oracle chain."""
    from oracle import cond as OC
    from oracle import vae as OV
    from pcdms_amd.cond import ControlNetConditioningEmbedding
    from pcdms_amd.vae import AutoencoderKL
    from tests.golden.make_real_image_fixture import H, N, SEED_POSE_NET, STEPS, W, seeded, to_model_input
    _, cfg, m, dev = full
    fx = np.load(Path(__file__).resolve().parent / "golden" / "real_image.npz")
    assert str(fx["torch_version"]) == torch.__version__
    sdd = seeded()
    vcfg = OV.VAEConfig()
    vae = AutoencoderKL()
    vae.load_state_dict(OV.synth_state_dict(vcfg, 0))
    vae.to(dev)
    pose_net = ControlNetConditioningEmbedding()
    pose_net.load_state_dict(OC.synth(OC.pose_param_shapes(), seed=SEED_POSE_NET))
    pose_net.to(dev)
    canvas = to_model_input(fx["canvas_u8"])[None].to(dev)
    assert float(canvas[..., W:].max()) == -1.0                                    # the black target half
    # VAE encode of the real canvas
    ml = vae.encode(canvas).latent_dist.sample(noise=sdd["post_noise"].to(dev)) * vcfg.scaling_factor
    r_ml = _rel(ml, fx["masked_latents"])
    # the target (right) half on its own: a constant-colour input, i.e. what the synthetic fixtures replace by 0
    r_black = _rel(ml[..., 2 * W // 16:], fx["masked_latents"][..., 2 * W // 16:])
    # pose net on the real stick-figure canvas
    st_pose_f = pose_net(to_model_input(fx["pose_u8"])[None].to(dev)).float()
    r_pose = _rel(st_pose_f[:, :, ::4, ::4], fx["st_pose_f_sub"].astype(np.float32))
    assert abs(float(st_pose_f.norm()) / float(fx["st_pose_f_norm"]) - 1.0) < 5e-3
    syn = synth_inputs(cfg, H // 8, 2 * W // 8, N)
    pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler.from_config(SD21), vae=vae)
    seen = {}
    out = pipe(height=H, width=2 * W, masked_latents=ml, s_img_proj_f=syn["s_img_proj_f"].to(dev), st_pose_f=st_pose_f,
               pred_t_img_embed=syn["pred_t_img_embed"].to(dev), latents=sdd["latents"].to(dev), num_images_per_prompt=N, guidance_scale=2.0,
               num_inference_steps=STEPS, output_type="uint8", callback=lambda i, t, lat: seen.__setitem__(i + 1, lat))
    lat = out.latents.float().cpu()
    ref = torch.from_numpy(fx["lat_final"])
    cx = _ddim_x_coefficients(STEPS)[STEPS]
    lat0 = sdd["latents"]
    r_lat = ((lat - ref).norm() / ref.norm()).item()
    r_part = (((lat - cx * lat0) - (ref - cx * lat0)).norm() / (ref - cx * lat0).norm()).item()
    # first-step eps: from the latents after step 1 (x1 = cx x0 + ce eps  =>  the eps-driven part of step 1 is ce eps)
    l1, r1 = seen[1].float().cpu(), torch.from_numpy(fx["lat_before"][1])
    c1 = _ddim_x_coefficients(STEPS)[1]
    r_eps0 = (((l1 - c1 * lat0) - (r1 - c1 * lat0)).norm() / (r1 - c1 * lat0).norm()).item()
    d = (out.images[0].cpu().int() - torch.from_numpy(fx["img"]).int()).abs().float()
    print(f"real-image chain: masked latents {r_ml:.3e} (black half {r_black:.3e}), pose feature {r_pose:.3e}, first-step eps part {r_eps0:.3e}, "
          f"3-step latents {r_lat:.3e}, eps-driven part {r_part:.3e}, pixels mean |diff| {d.mean():.3f} / 255 (max {d.max():.0f})")
    record_check("real_image.masked_latents", r_ml, 2e-2)
    record_check("real_image.masked_latents_black_half", r_black, 2e-2)
    record_check("real_image.pose_feature", r_pose, 2e-2)
    record_check("real_image.eps_part.step1", r_eps0, 3e-2)
    record_check("real_image.trajectory.final", r_lat, 1e-2)          # (3 coarse steps: each carries a third of the schedule)
    record_check("real_image.eps_part.final", r_part, 3e-2)
    record_check("real_image.pixels.mean_abs_levels", d.mean().item(), 1.5)


FP8_TRAJ_TOL = 1.5e-3     # 50-step latents with fp8 attention operands against the fp32 oracle: measured 0.82e-3 -- the same as with bf16
FP8_EPS_PART_TOL = 1.2e-2  # attention (0.81e-3), so the same tolerances; eps-driven part measured 0.53e-2 (bf16: 0.52e-2)


@pytest.mark.gpu
def test_configs4_share_fp8_50_step_graph_run(full):
    """BASELINE.json configs[4], one GPU's share, as a SAMPLING RUN (VERDICT r3 #3b): N = 16 samples of one pair (UNet batch 32), fp8
    (e4m3) attention operands, all 50 DDIM steps under the hipGraph.  The first four samples get the N = 4 fixture's noise, so they must
    reproduce the fp32 ORACLE's N = 4 trajectory (``fullsize_config2.npz``) within the stated fp8 trajectory tolerance -- on the
    latents and on their eps-driven part -- and a second call must reproduce the first bit for bit."""
    fx, cfg, m, dev = full
    N, h, w = 16, 64, 88
    steps = int(fx["steps"])
    inp = synth_inputs(cfg, h, w, 4)
    lat16 = torch.cat([inp["latents"], torch.randn(12, 4, h, w, generator=torch.Generator().manual_seed(78))])
    m.set_attention_precision("fp8")
    try:
        pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler.from_config(SD21))
        kw = dict(height=h * 8, width=w * 8, masked_latents=inp["masked_latents"].to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
                  st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev), num_images_per_prompt=N,
                  guidance_scale=2.0, num_inference_steps=steps, output_type="latent")
        out = pipe(latents=lat16.to(dev), **kw).latents
        assert pipe._graph is not None and out.shape == (N, 4, h, w) and bool(torch.isfinite(out).all())
        out2 = pipe(latents=lat16.to(dev), **kw).latents
        assert torch.equal(out, out2)
    finally:
        m.set_attention_precision("bf16")
    r = _rel(out[:4], fx["lat_final"])
    cx = _ddim_x_coefficients(steps)[steps]
    lat0 = inp["latents"]
    ref = torch.from_numpy(fx["lat_final"]) - cx * lat0
    rp = ((out[:4].float().cpu() - cx * lat0 - ref).norm() / ref.norm()).item()
    print("configs[4] share, fp8 attention, N = 16 50-step run: first four samples vs the oracle rel-L2", round(r, 5), "eps-driven part", round(rp, 5))
    record_check("configs4.fp8_trajectory_n16.final", r, FP8_TRAJ_TOL)
    record_check("configs4.fp8_eps_part_n16.final", rp, FP8_EPS_PART_TOL)
    s0, s1 = out[:4].float().std().item(), out[4:].float().std().item()
    assert abs(s0 - s1) <= 0.05 * s0, (s0, s1)


@pytest.mark.gpu
def test_stage3_full_size():
    """BASELINE.json configs[3]'s third stage at FULL size (VERDICT r3 #3a): the stock 865.9 M-parameter topology with in_channels = 8
    (``pcdms_amd.UNet2DConditionModel``), latent 64 x 44 (levels 64x44 / 32x22 / 16x11 / 8x6: the odd-size stride-2 and
    upsample-to-skip-size gathers), N = 8 under CFG (UNet batch 16), against ``tests/golden/fullsize_stage3.npz``
    (tests/golden/make_fullsize_stage3_fixture.py; the fp32 oracle restating /root/reference/src/pipelines/stage3_refined_pipeline.py:483-557):
    * single forwards at two oracle states (step 0 and step 10 of a 20-step DDIM schedule): guided eps rel-L2 <= FWD_TOL;
    * the 20-step hipGraph run through ``Stage3_RefinedDiffusionPipeline`` with N = 8: its first two samples reproduce the oracle's
      N = 2 trajectory (latents and eps-driven part), all samples finite with matching statistics, a second call bit-identical."""
    from pcdms_amd import _lib
    from pcdms_amd.pipeline import Stage3_RefinedDiffusionPipeline
    from pcdms_amd.unet import UNet2DConditionModel
    from tests.golden.make_fullsize_stage3_fixture import H, MID, N, STEPS, W, mid_latents, stage3_config, synth_stage3_inputs
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = FIXTURE.parent / "fullsize_stage3.npz"
    if not path.exists():
        pytest.fail(f"{path} missing: run tests/golden/make_fullsize_stage3_fixture.py")
    _lib.load()
    fx = np.load(path)
    assert str(fx["torch_version"]) == torch.__version__
    dev = torch.device("cuda:0")
    cfg = stage3_config()
    m = UNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(synth_state_dict(cfg, seed=0, random_affine=True))
    m.to(dev)
    assert sum(int(np.prod(s)) for s in m.expected_shapes().values()) == 865_922_244
    inp = synth_stage3_inputs()
    assert abs(float(inp["latents"].double().abs().sum()) - float(fx["lat_checksum"])) <= 1e-9 * float(fx["lat_checksum"])
    sch = DDIMOracle()
    sch.set_timesteps(STEPS)

    def guided(lat, t):
        feat = inp["s_img_proj_f"].repeat(N, 1, 1)
        gl = inp["gen_t_img_latents"].repeat(N, 1, 1, 1)
        feat = torch.cat([torch.zeros_like(feat), feat])
        gl = torch.cat([torch.zeros_like(gl), gl])
        eps = m(torch.cat([torch.cat([lat] * 2), gl], 1).to(dev), t, encoder_hidden_states=feat.to(dev)).sample.float().cpu()
        u, c = eps.chunk(2)
        return u + 2.0 * (c - u)
    r0 = _rel(guided(inp["latents"], sch.timesteps[0]), fx["eps_0"])
    tm = int(fx["t_mid"])
    assert tm == int(sch.timesteps[MID])
    r1 = _rel(guided(mid_latents(float(sch.alphas_cumprod[tm])), torch.tensor(tm)), fx["eps_mid"])
    print("stage-3 full-size forwards (UNet batch 16, latent 64x44) rel-L2 at step 0 / step 10:", round(r0, 5), round(r1, 5))
    record_check("stage3.forward_b16.step0", r0, FWD_TOL)
    record_check("stage3.forward_b16.step10", r1, FWD_TOL)
    # ---- the sampler: 20 DDIM steps, N = 8, hipGraph
    pipe = Stage3_RefinedDiffusionPipeline(m, DDIMScheduler.from_config(SD21))
    kw = dict(height=H * 8, width=W * 8, num_inference_steps=STEPS, guidance_scale=2.0, num_images_per_prompt=N, output_type="latent",
              s_img_proj_f=inp["s_img_proj_f"].to(dev), gen_t_img_latents=inp["gen_t_img_latents"].to(dev))
    out = pipe(latents=inp["latents"].to(dev), **kw).latents
    assert pipe._graph is not None and out.shape == (N, 4, H, W) and bool(torch.isfinite(out).all())
    ref = torch.from_numpy(fx["lat_final_n2"])
    r = _rel(out[:2], ref)
    cx = _ddim_x_coefficients(STEPS)[STEPS]
    lat0 = inp["latents"][:2]
    rp = (((out[:2].float().cpu() - cx * lat0) - (ref - cx * lat0)).norm() / (ref - cx * lat0).norm()).item()
    print("stage-3 full-size 20-step run, first two samples vs the oracle rel-L2", round(r, 5), "eps-driven part", round(rp, 5))
    record_check("stage3.trajectory_n8.final", r, 2 * TRAJ_TOL)       # (20 coarse steps, as configs[0])
    record_check("stage3.eps_part_n8.final", rp, 2 * EPS_PART_TOL)
    s0, s1 = out[:2].float().std().item(), out[2:].float().std().item()
    assert abs(s0 - s1) <= 0.05 * s0, (s0, s1)
    assert torch.equal(pipe(latents=inp["latents"].to(dev), **kw).latents, out)


# ---------------------------------------------------------------------------------------------------------------------------------
# Checkpoint-like statistics: outlier channels (x30) in conv_in / every proj_in / every ff.net.2 and every
# GroupNorm / LayerNorm beta ~ N(0, 3^2) -- tests/golden/make_fullsize_stress_fixture.py, oracle.unet.stress_state_dict.
STRESS_FIXTURE = Path(__file__).resolve().parent / "golden" / "fullsize_stress.npz"
STRESS_FWD_TOL = 2 * FWD_TOL     # "within 2x of configs1.forward.*"
STRESS_TAP_TOL = 2 * FWD_TOL     # the residual stream at every block boundary the schedule keeps (conv_in, down skips, cross-attention up blocks)


@pytest.mark.gpu
def test_stress_checkpoint_statistics_forward():
    """One full-size forward (UNet batch 2) on weights with the activation statistics the seeded U(+-1/sqrt(fan_in)) weights lack:
    residual-stream outlier channels of |x| ~ 70 against a unit bulk, norm betas of std 3.  Compared with the fp32 oracle: the guided
    eps at two timesteps AND -- because with such betas the output is dominated by an input-independent part -- the residual stream
    itself at 16 block boundaries (16 seeded token rows each, all channels).  Exercises the folded LayerNorm of rowgemm.hip, the
    centred / Chan-merged GroupNorm statistics of all three paths and the fp8-free attention on rows with dominant channels."""
    if not STRESS_FIXTURE.exists():
        pytest.fail(f"{STRESS_FIXTURE} missing: run tests/golden/make_fullsize_stress_fixture.py")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.unet import stress_state_dict
    from pcdms_amd import _lib
    from tests.golden.make_fullsize_stress_fixture import tap_rows
    _lib.load()
    fx = np.load(STRESS_FIXTURE)
    assert str(fx["torch_version"]) == torch.__version__
    cfg = UNetConfig()
    dev = torch.device("cuda:0")
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(stress_state_dict(cfg, seed=0))
    m.to(dev)
    N, h, w = 1, 64, 88
    inp = synth_inputs(cfg, h, w, N)
    sch = DDIMOracle()
    sch.set_timesteps(50)
    steps_at = [int(v) for v in fx["steps_at"]]
    for k, i in enumerate(steps_at):
        m._taps = {} if k == 0 else None
        eps = _guided_eps(m, cfg, inp, inp["latents"], sch.timesteps[i], N, dev)
        assert torch.isfinite(eps).all()
        record_check(f"stress.forward.step{i}", _rel(eps, fx[f"eps_{i}"]), STRESS_FWD_TOL)
        if k == 0:
            taps, m._taps = m._taps, None
            rels = {}
            for name, v in taps.items():
                if f"tap_{name}" not in fx.files:
                    continue            # (ds<i>: the oracle does not tap the downsampling convs)
                ref = torch.from_numpy(fx[f"tap_{name}"].astype(np.float32))          # [B, 16, C]
                B, C = ref.shape[0], ref.shape[2]
                rows = v.view(B, -1, C)
                got = rows[:, tap_rows(rows.shape[1]).to(rows.device)].float().cpu()
                rels[name] = ((got - ref).norm() / ref.norm()).item()
            print("stress: residual-stream rel-L2 per tap:", {k_: round(v_, 5) for k_, v_ in rels.items()})
            print("stress: |row mean| / std (max, median), max |x| per tap:", {n[5:]: [round(float(z), 2) for z in fx[n]] for n in fx.files if n.startswith("stat_")})
            assert len(rels) >= 14, sorted(rels)
            for name, r in rels.items():
                record_check(f"stress.tap.{name}", r, STRESS_TAP_TOL)


# ---------------------------------------------------------------------------------------------------------------------------------
# The configuration the reference's stage-2 driver SHIPS WITH (VERDICT r4 missing #6 / next 4a): UniPC, 20 steps, 512x512 images =>
# canvas 1024x512 => latent 64x128, num_images_per_prompt 4, guidance 2.0 (/root/reference/stage2_batchtest_inpaint_model.py:132,196,
# 256-260).  M = 65 536 rows at level 0, self-attention over 8192 tokens -- shapes no other full-size test has.
DRIVER_FIXTURE = Path(__file__).resolve().parent / "golden" / "fullsize_driver_default.npz"
UNIPC_TRAJ_TOL = 3e-3      # latents along the 20-step UniPC trajectory (multistep: the error of two model outputs enters every update)
UNIPC_EPS_PART_TOL = 2e-2  # their eps-driven part, lat_i - c_x(i) lat_0


def _unipc_x_coefficients(steps: int):
    """c_x(i) of the UniPC update (linear in (sample, model outputs)): the oracle scheduler run with eps = 0."""
    from oracle.schedulers import UniPCOracle
    sch = UniPCOracle()
    sch.set_timesteps(steps)
    x, cx = torch.ones(1, dtype=torch.float64), {0: 1.0}
    for i, t in enumerate(sch.timesteps):
        x = sch.step(torch.zeros_like(x), t, x)
        cx[i + 1] = float(x)
    return cx


@pytest.mark.gpu
def test_driver_default_unipc_512(full):
    """stage2_batchtest_inpaint_model.py's own defaults at full size against the fp32 oracle: forwards at three oracle states of the
    UniPC trajectory, then the complete 20-step call through the captured fused step (pcdm_unipc_step on the static history slots)."""
    from oracle.schedulers import UniPCOracle
    from pcdms_amd.schedulers import UniPCMultistepScheduler
    _, cfg, m, dev = full
    if not DRIVER_FIXTURE.exists():
        pytest.fail(f"{DRIVER_FIXTURE} missing: run tests/golden/make_fullsize_driver_default_fixture.py")
    fx = np.load(DRIVER_FIXTURE)
    assert str(fx["torch_version"]) == torch.__version__
    N, h, w, steps = 4, 64, 128, int(fx["steps"])
    inp = synth_inputs(cfg, h, w, N)
    assert np.array_equal(inp["latents"].numpy(), fx["lat_0"])
    sch = UniPCOracle()
    sch.set_timesteps(steps)
    rels = {}
    for i in [int(v) for v in fx["eps_at"]]:
        eps = _guided_eps(m, cfg, inp, torch.from_numpy(fx[f"lat_{i}"]), sch.timesteps[i], N, dev)
        rels[i] = _rel(eps, fx[f"eps_{i}"])
    print("driver default (UniPC, 64x128): single-forward rel-L2 per step:", {k: round(v, 5) for k, v in rels.items()})
    for k, v in rels.items():
        record_check(f"driver512.forward.step{k}", v, FWD_TOL)
    pipe = Stage2_InpaintDiffusionPipeline(m, UniPCMultistepScheduler.from_config(SD21))
    seen = {}
    out = pipe(height=h * 8, width=w * 8, masked_latents=inp["masked_latents"].to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
               st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev), latents=inp["latents"].to(dev),
               num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=steps, output_type="latent",
               callback=lambda i, t, lat: seen.__setitem__(i + 1, lat)).latents
    assert pipe._graph is not None and len(seen) == steps and torch.isfinite(out).all()
    cx = _unipc_x_coefficients(steps)
    lat0 = torch.from_numpy(fx["lat_0"])
    tr, parts = {}, {}
    for i in [int(v) for v in fx["check"]] + [steps]:
        if i == 0:
            continue
        ref = torch.from_numpy(fx["lat_final"] if i == steps else fx[f"lat_{i}"])
        got = (out if i == steps else seen[i]).float().cpu()
        key = "final" if i == steps else i
        tr[key] = ((got - ref).norm() / ref.norm()).item()
        parts[key] = (((got - cx[i] * lat0) - (ref - cx[i] * lat0)).norm() / (ref - cx[i] * lat0).norm()).item()
    print("driver default: 20-step UniPC trajectory rel-L2:", {k: round(v, 5) for k, v in tr.items()}, "eps-driven part:",
          {k: round(v, 5) for k, v in parts.items()})
    for k, v in tr.items():
        record_check(f"driver512.trajectory.{k}", v, UNIPC_TRAJ_TOL)
    for k, v in parts.items():
        record_check(f"driver512.eps_part.{k}", v, UNIPC_EPS_PART_TOL)
