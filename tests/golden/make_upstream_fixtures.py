#!/usr/bin/env python3
"""Upstream pin: golden vectors from the REAL third-party code the reference's arithmetic lives in.

The stage-2 path's block internals and schedulers are `diffusers==0.24.0` (/root/reference/README.md:37; imports at
/root/reference/src/models/stage2_inpaint_unet_2d_condition.py:21,35-44), which is neither vendored in the reference nor
installable in the offline build container -- the in-repo oracle restates it ("parity unpinned").  Anyone with

    pip install diffusers==0.24.0            and a checkout of tencent-ailab/PCDMs

turns that into a pin with ONE command (CPU only, a few seconds):

    python tests/golden/make_upstream_fixtures.py --reference-root /path/to/PCDMs

It writes tests/golden/upstream_{unet,ddim,unipc,ddpm,vae}.npz: seeded inputs + the outputs of the reference's own
`Stage2_InapintUNet2DConditionModel` / diffusers' schedulers / `AutoencoderKL` on the tiny configs the test-suite uses (weights:
`oracle.unet.synth_state_dict(UNetConfig.tiny(), seed)` loaded into the upstream modules -- diffusers key names, so only the seed is
stored).  `tests/test_upstream_pin.py` then checks the ORACLE against them on CPU (pinning every tolerance in the suite to upstream)
and the HIP path on the GPU; the tests skip while the files are absent.

`--backend oracle` writes the same files from the in-repo restatement instead: a FORMAT self-test (tests/test_upstream_pin.py runs
it into a temp dir), never to be committed as a pin -- the files record their backend and the tests refuse to call that a pin.

STATUS: the diffusers backend is written against the 0.24.0 API from memory and has NOT been executed (no diffusers here)."""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import vae as OV                                   # noqa: E402
from oracle.schedulers import DDIMOracle, DDPMOracle, UniPCOracle   # noqa: E402
from oracle.unet import UNetConfig, synth_state_dict, unet_forward   # noqa: E402

SD21_SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
SEED = 11


def unet_inputs(cfg: UNetConfig, B=2, h=16, w=16, L=10):
    g = torch.Generator().manual_seed(SEED)
    x = torch.randn(B, cfg.in_channels, h, w, generator=g)
    ehs = torch.randn(B, L, cfg.cross_attention_dim, generator=g)
    ehs[: B // 2] = 0
    cl = torch.randn(B, 1, cfg.projection_class_embeddings_input_dim, generator=g) * 0.4
    pose = torch.randn(B, cfg.block_out_channels[0], h, w, generator=g) * 0.1
    return x, ehs, cl, pose


def make_unet(out: Path, backend: str, reference_root: str | None):
    cfg = UNetConfig.tiny()
    sd = synth_state_dict(cfg, seed=SEED, random_affine=True)
    x, ehs, cl, pose = unet_inputs(cfg)
    ts = [981, 500, 1]
    if backend == "diffusers":
        sys.path.insert(0, reference_root)
        from src.models.stage2_inpaint_unet_2d_condition import Stage2_InapintUNet2DConditionModel as RefUNet
        m = RefUNet(sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=4, block_out_channels=cfg.block_out_channels,
                    layers_per_block=2, attention_head_dim=cfg.attention_head_dim, cross_attention_dim=cfg.cross_attention_dim,
                    use_linear_projection=True, class_embed_type="projection",
                    projection_class_embeddings_input_dim=cfg.projection_class_embeddings_input_dim).float().eval()
        m.load_state_dict(sd)
        with torch.no_grad():
            eps = [m(x, torch.tensor(t), encoder_hidden_states=ehs, class_labels=cl, my_pose_cond=pose, return_dict=False)[0] for t in ts]
    else:
        with torch.no_grad():
            eps = [unet_forward(sd, cfg, x, torch.tensor(t), ehs, cl, pose) for t in ts]
    np.savez_compressed(out / "upstream_unet.npz", backend=backend, seed=SEED, timesteps=np.array(ts), x=x.numpy(), ehs=ehs.numpy(),
                        cl=cl.numpy(), pose=pose.numpy(), eps=torch.stack(eps).numpy())


def make_schedulers(out: Path, backend: str):
    g = torch.Generator().manual_seed(SEED + 1)
    for name, n in (("ddim", 50), ("unipc", 20), ("ddpm", 25)):
        if backend == "diffusers":
            import diffusers
            sch = {"ddim": lambda: diffusers.DDIMScheduler(**SD21_SCHED, clip_sample=False, set_alpha_to_one=False, steps_offset=1),
                   "unipc": lambda: diffusers.UniPCMultistepScheduler(**SD21_SCHED),
                   "ddpm": lambda: diffusers.DDPMScheduler(**SD21_SCHED, clip_sample=False)}[name]()
        else:
            sch = {"ddim": DDIMOracle, "unipc": UniPCOracle,
                   "ddpm": lambda: DDPMOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")}[name]()
        sch.set_timesteps(n)
        ts = [int(t) for t in sch.timesteps]
        x = torch.randn(2, 4, 8, 8, generator=g)
        xs, es, zs = [x.numpy().copy()], [], []
        for t in ts:                      # a TRAJECTORY (UniPC is multistep: its history must be the real one)
            e = torch.randn(2, 4, 8, 8, generator=g)
            z = torch.randn(2, 4, 8, 8, generator=g)
            if backend == "diffusers":
                if name == "ddpm":        # diffusers draws the variance noise from `generator`: hand it a generator that yields z
                    gz = torch.Generator().manual_seed(1000 + t)
                    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(1000 + t))
                    x = sch.step(e, t, x, generator=gz).prev_sample
                else:
                    x = sch.step(e, t, x).prev_sample
            else:
                x = sch.step(e, t, x, variance_noise=z) if name == "ddpm" else sch.step(e, t, x)
            es.append(e.numpy())
            zs.append(z.numpy())
            xs.append(x.numpy().copy())
        np.savez_compressed(out / f"upstream_{name}.npz", backend=backend, n=n, timesteps=np.array(ts), eps=np.stack(es), noise=np.stack(zs),
                            x=np.stack(xs))


def make_vae(out: Path, backend: str):
    cfg = OV.VAEConfig.tiny()
    sd = OV.synth_state_dict(cfg, SEED)
    g = torch.Generator().manual_seed(SEED + 2)
    img = torch.rand(1, 3, 64, 96, generator=g) * 2 - 1
    z = torch.randn(1, 4, 8, 12, generator=g)
    if backend == "diffusers":
        import diffusers
        m = diffusers.AutoencoderKL(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=cfg.block_out_channels, layers_per_block=2,
                                    norm_num_groups=32, down_block_types=("DownEncoderBlock2D",) * 4,
                                    up_block_types=("UpDecoderBlock2D",) * 4).float().eval()
        m.load_state_dict(sd)
        with torch.no_grad():
            mom, dec = m.encode(img).latent_dist.parameters, m.decode(z, return_dict=False)[0]
    else:
        with torch.no_grad():
            mom, dec = OV.encode_moments(sd, cfg, img), OV.decode(sd, cfg, z)
    np.savez_compressed(out / "upstream_vae.npz", backend=backend, seed=SEED, img=img.numpy(), z=z.numpy(), moments=mom.numpy(), decoded=dec.numpy())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=("diffusers", "oracle"), default="diffusers")
    ap.add_argument("--reference-root", default=None, help="checkout of tencent-ailab/PCDMs (diffusers backend)")
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden"))
    a = ap.parse_args()
    if a.backend == "diffusers":
        import diffusers
        if not a.reference_root:
            ap.error("--reference-root is required with the diffusers backend")
        print("diffusers", diffusers.__version__, "(the reference pins 0.24.0)")
    out = Path(a.out)
    out.mkdir(parents=True, exist_ok=True)
    make_unet(out, a.backend, a.reference_root)
    make_schedulers(out, a.backend)
    make_vae(out, a.backend)
    print("wrote", sorted(p.name for p in out.glob("upstream_*.npz")))


if __name__ == "__main__":
    main()
