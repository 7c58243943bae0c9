#!/usr/bin/env python3
"""Generate tests/golden/ref_wiring_prior.npz by running the REFERENCE's own stage-1 code in this container.

    python tests/golden/make_reference_prior_fixture.py        (needs /root/reference; dev container only)

Executed from /root/reference (imported, never copied):
  * ``Stage1_PriorTransformer.__init__`` / ``.forward`` / ``.post_process_latents``
    (src/models/stage1_prior_transformer.py:66-134, :197-301), including its own ``MLP`` pose encoders (:18-36)
  * ``Stage1_PriorPipeline.__call__`` (src/pipelines/stage1_prior_pipeline.py:355-504)
on top of tests/golden/diffusers_stub.py (transformer block internals = oracle.prior.transformer_block, scheduler =
oracle.schedulers.UnCLIPOracle with injected variance noise, time embedding = oracle.unet).  The fixture pins the
reference's own wiring -- pose MLPs, token order, positional add, last-token read-out, loop / prev_timestep handling,
post_process_latents -- not the diffusers block internals.  Inputs + expected outputs only; weights are regenerated
from ``oracle.prior.synth_state_dict(cfg, seed)`` and guarded by a checksum.
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")

from oracle.prior import PriorConfig, synth_state_dict  # noqa: E402
from tests.golden.make_reference_wiring_fixtures import weights_checksum  # noqa: E402

FIX_CFG = dict(num_attention_heads=2, num_layers=2)
SEED = 9


class _FakeEncoder:
    """stands in for CLIPVisionModelWithProjection in get_zero_embed (:281-288); its output is not part of the fixture"""
    dtype = torch.float32
    config = types.SimpleNamespace(image_size=8)

    def __call__(self, x):
        return {"image_embeds": torch.zeros(x.shape[0], 1024)}


def main():
    from tests.golden import diffusers_stub
    diffusers_stub.install()
    sys.path.insert(0, str(REF))
    from src.models.stage1_prior_transformer import Stage1_PriorTransformer as RefPrior
    from src.pipelines.stage1_prior_pipeline import Stage1_PriorPipeline as RefPipe
    cfg = PriorConfig(**FIX_CFG)
    sd = synth_state_dict(cfg, SEED)
    prior = RefPrior(num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim,
                     num_layers=cfg.num_layers, embedding_dim=cfg.embedding_dim, num_embeddings=cfg.num_embeddings,
                     additional_embeddings=cfg.additional_embeddings).eval()
    missing, unexpected = prior.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    meta = dict(seed=SEED, weights_checksum=weights_checksum(sd), torch_version=torch.__version__,
                **{k: np.array(v) for k, v in FIX_CFG.items()})

    g = torch.Generator().manual_seed(21)
    B = 2
    x = torch.randn(B, 1, 1024, generator=g)
    emb = torch.randn(B, 1, 1024, generator=g) * 0.4
    sp, tp = torch.rand(B, 1, 36, generator=g), torch.rand(B, 1, 36, generator=g)
    with torch.no_grad():
        pred = prior(x, timestep=torch.tensor(473), proj_embedding=emb, encoder_hidden_states=sp,
                     encoder_hidden_states1=tp, attention_mask=None).predicted_image_embedding
        pred_t = prior(x, 473, emb, sp, tp, return_dict=False)[0]
    assert torch.equal(pred, pred_t)

    steps = 4
    s_embed = torch.randn(1, 1, 1024, generator=g) * 0.4
    s_pose, t_pose = torch.rand(1, 1, 36, generator=g), torch.rand(1, 1, 36, generator=g)
    latents = torch.randn(1, 1024, generator=g)
    noises = [torch.randn(1, 1024, generator=g) for _ in range(steps)]
    pipe = RefPipe(prior=prior, image_encoder=_FakeEncoder(), scheduler=diffusers_stub.UnCLIPSchedulerStub(noises),
                   image_processor=None)
    with torch.no_grad():
        out = pipe(s_embed=s_embed, s_pose=s_pose, t_pose=t_pose, num_images_per_prompt=1, num_inference_steps=steps,
                   generator=None, latents=latents.clone(), guidance_scale=0)   # driver default guidance (:153)
    np.savez_compressed(HERE / "ref_wiring_prior.npz", x=x.numpy(), timestep=473, proj_embedding=emb.numpy(), s_pose_b=sp.numpy(),
                        t_pose_b=tp.numpy(), pred=pred.numpy(), steps=steps, s_embed=s_embed.numpy(), s_pose=s_pose.numpy(),
                        t_pose=t_pose.numpy(), latents=latents.numpy(), noises=torch.stack(noises).numpy(),
                        image_embeds=out[0].numpy(), **meta)
    print("pred std", float(pred.std()), "image_embeds std", float(out[0].std()))


if __name__ == "__main__":
    main()
