#!/usr/bin/env python3
"""How far does the REFERENCE'S OWN precision sit from the fp32 oracle?  (BASELINE.json: "to within a stated fp16
tolerance".)

The reference runs the stage-2 path hard-cast to fp16: ``torch_dtype=torch.float16`` for the UNet
(/root/reference/stage2_batchtest_inpaint_model.py:123-128), every loop input ``.to(dtype=torch.float16)``
(src/pipelines/stage2_inpaint_pipeline.py:431,440,449,452,487,501), fp16 latents through ``scheduler.step`` (:519) and an fp16 CFG
combine (:510-512).  In PyTorch eager that means: every tensor an op materialises is rounded to fp16, the op itself (cuDNN / cuBLAS /
xformers / ATen norms) accumulates in fp32.  This script reproduces exactly that on the CPU oracle -- weights rounded once, ``oracle.unet.
ROUND_DTYPE`` rounding every op output, the loop's elementwise arithmetic in real ``torch.float16`` tensors -- and measures its distance
from the fp32 oracle on the same seeded weights / inputs as the full-size parity fixtures:

  * three single forwards of configs[1] (latent 64x88; sample 0 of the oracle states before steps 0 / 10 / 25 of fullsize_config2.npz,
    UNet batch 2): rel-L2 of the guided eps;
  * BASELINE configs[0] IN FULL (one 256x256 pair, latent 32x64, N = 1, 20 DDIM steps): rel-L2 of the latents after every step, of
    their eps-driven part (lat_i - c_x(i) lat_0, tests/test_fullsize_parity.py), and mean |diff| of the decoded uint8 canvases (fp32
    oracle VAE on both trajectories: the UNet / loop precision alone).

The same is done with bf16 in place of fp16 ("the reference hard-cast to bf16"): the bf16 HIP path rounds LESS often than that (fp32
residual adds inside epilogues, fp32 latents and scheduler, fp32 split-K slabs), so its distance from the fp32 oracle should lie between
the two.  tests/test_fullsize_parity.py::test_bf16_hip_path_against_the_reference_precision_budget compares (GPU).

    python tests/golden/make_fp16_budget.py          (~15 min on the 8 build-container cores)
writes tests/golden/fp16_budget.npz (numbers + the two emulated final latents of configs[0]) and profiles/r6_fp16_budget.txt.
"""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def main():
    import oracle.unet as OU
    from oracle import vae as ovae
    from oracle.pipeline import build_conditioning, synth_inputs
    from oracle.schedulers import DDIMOracle
    from oracle.unet import UNetConfig, synth_state_dict, unet_forward
    t0 = time.time()
    torch.manual_seed(0)
    cfg = UNetConfig()
    sd32 = synth_state_dict(cfg, seed=0, random_affine=True)
    sds = {"fp32": sd32, "fp16": {k: v.half().float() for k, v in sd32.items()}, "bf16": {k: v.bfloat16().float() for k, v in sd32.items()}}
    dts = {"fp32": None, "fp16": torch.float16, "bf16": torch.bfloat16}
    log = []

    def say(s):
        print(s, flush=True)
        log.append(s)

    def forward(mode, inp, t, c):
        """guided eps of one CFG-doubled forward in precision ``mode`` (returns fp32 values that are exactly representable in the mode's dtype)"""
        dt = dts[mode]
        OU.ROUND_DTYPE = dt
        try:
            q = (lambda x: x) if dt is None else (lambda x: x.to(dt).float())
            eps = unet_forward(sds[mode], cfg, q(inp), t, q(c["feature_f"]), q(c["prior_embed"]), q(c["pose_cond"]))
        finally:
            OU.ROUND_DTYPE = None
        return eps

    def guided(mode, eps, g=2.0):
        dt = dts[mode]
        if dt is None:
            u, cn = eps.chunk(2)
            return u + g * (cn - u)
        e = eps.to(dt)
        u, cn = e.chunk(2)
        return u + g * (cn - u)          # (real fp16 / bf16 tensor arithmetic: one rounding per op, as torch eager does)

    out = {"torch_version": np.array(torch.__version__)}
    res = {"forward_configs1": {}, "config0": {}}
    with torch.no_grad():
        # ---------------------------------------------------------------- configs[1]: three forwards at latent 64 x 88
        fx = np.load(ROOT / "tests" / "golden" / "fullsize_config2.npz")
        h, w = 64, 88
        inp1 = synth_inputs(cfg, h, w, 1)
        c1 = build_conditioning(inp1["masked_latents"], inp1["s_img_proj_f"], inp1["st_pose_f"], inp1["pred_t_img_embed"], 1, True)
        sch = DDIMOracle()
        sch.set_timesteps(50)
        for i in (0, 10, 25):
            lat = torch.from_numpy(fx[f"lat_{i}"][:1])
            t = sch.timesteps[i]
            x = torch.cat([torch.cat([lat] * 2), c1["mask"], c1["masked_latents"]], 1)
            e = {m: guided(m, forward(m, x, t, c1)).float() for m in ("fp32", "fp16", "bf16")}
            r16, rbf = rel(e["fp16"], e["fp32"]), rel(e["bf16"], e["fp32"])
            # the stored fp32 oracle eps of the fixture (fp16-stored) as a cross-check of the recomputation
            chk = rel(e["fp32"], torch.from_numpy(fx[f"eps_{i}"][:1].astype(np.float32)))
            res["forward_configs1"][str(i)] = {"fp16ref_vs_fp32": r16, "bf16ref_vs_fp32": rbf, "fp32_vs_stored_fixture": chk}
            say(f"configs[1] forward at the oracle state before step {i:2d} (t = {int(t):3d}): guided eps rel-L2  fp16-reference {r16:.3e}   bf16-hard-cast {rbf:.3e}"
                f"   (fp32 recomputed vs stored fixture {chk:.1e})   [{time.time() - t0:.0f} s]")
        # ---------------------------------------------------------------- configs[0] in full
        H0, W0, STEPS = 32, 64, 20
        inp0 = synth_inputs(cfg, H0, W0, 1)
        c0 = build_conditioning(inp0["masked_latents"], inp0["s_img_proj_f"], inp0["st_pose_f"], inp0["pred_t_img_embed"], 1, True)
        traj = {}
        for mode in ("fp32", "fp16", "bf16"):
            dt = dts[mode]
            sch = DDIMOracle()
            sch.set_timesteps(STEPS)
            lat = inp0["latents"].clone() if dt is None else inp0["latents"].to(dt)
            lats = []
            for i, t in enumerate(sch.timesteps):
                x = torch.cat([torch.cat([lat] * 2).float(), c0["mask"], c0["masked_latents"]], 1)
                e = guided(mode, forward(mode, x, t, c0))
                lat = sch.step(e, t, lat)          # fp16 / bf16 tensors x python floats: every op of the step rounds (ref :519 on fp16 latents)
                assert lat.dtype == (torch.float32 if dt is None else dt)
                lats.append(lat.float().clone())
            traj[mode] = lats
            say(f"configs[0] {mode} trajectory done [{time.time() - t0:.0f} s]")
        # c_x(i): the deterministic rescaling of the initial noise (DDIM: product of sqrt(a_prev / a) over the steps so far)
        sch = DDIMOracle()
        sch.set_timesteps(STEPS)
        cx, acc = [], 1.0
        for t in sch.timesteps:
            a, ap = sch.coefficients(int(t))
            acc *= (ap / a) ** 0.5
            cx.append(acc)
        lat0 = inp0["latents"]
        per_step = {"fp16": [], "bf16": []}
        for m in ("fp16", "bf16"):
            for i in range(STEPS):
                ref = traj["fp32"][i]
                per_step[m].append({"latents": rel(traj[m][i], ref), "eps_part": rel(traj[m][i] - cx[i] * lat0, ref - cx[i] * lat0)})
        # pixels: fp32 oracle VAE on the fp32 / fp16 / bf16 final latents
        vcfg = ovae.VAEConfig()
        vsd = ovae.synth_state_dict(vcfg, 0)
        img = {m: ovae.postprocess_uint8(ovae.decode(vsd, vcfg, traj[m][-1] / vcfg.scaling_factor))[0].float() for m in ("fp32", "fp16", "bf16")}
        for m in ("fp16", "bf16"):
            fin = per_step[m][-1]
            px = float((img[m] - img["fp32"]).abs().mean())
            res["config0"][m] = {"final_latents": fin["latents"], "final_eps_part": fin["eps_part"], "pixels_mean_abs_diff_of_255": px,
                                 "per_step_latents": [p["latents"] for p in per_step[m]], "per_step_eps_part": [p["eps_part"] for p in per_step[m]]}
            say(f"configs[0] complete ({STEPS} DDIM steps, latent {H0}x{W0}), {m} vs fp32 oracle: final latents {fin['latents']:.3e}   eps-driven part {fin['eps_part']:.3e}"
                f"   decoded uint8 canvas mean |diff| {px:.3f} / 255   (latents after steps 1 / 5 / 10: "
                f"{per_step[m][0]['latents']:.2e} / {per_step[m][4]['latents']:.2e} / {per_step[m][9]['latents']:.2e})")
        out["config0_final_fp16"] = traj["fp16"][-1].numpy()
        out["config0_final_bf16"] = traj["bf16"][-1].numpy()
        out["config0_final_fp32"] = traj["fp32"][-1].numpy()
    out["json"] = np.array(json.dumps(res))
    path = ROOT / "tests" / "golden" / "fp16_budget.npz"
    np.savez_compressed(path, **out)
    say(f"wrote {path} ({path.stat().st_size / 1e3:.0f} kB) in {time.time() - t0:.0f} s; torch {torch.__version__}")
    (ROOT / "profiles" / "r6_fp16_budget.txt").write_text("\n".join(log) + "\n")
    (ROOT / "profiles" / "r6_fp16_budget.json").write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
