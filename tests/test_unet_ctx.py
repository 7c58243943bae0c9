"""pcdm_unet_forward(ctx, ...): the C-side schedule of the UNet (include/pcdm.h) against the Python schedule of
pcdms_amd/unet.py -- the same kernels on the same packed weights with the same tile choices, so the two must agree BIT FOR BIT -- and,
through it, against the oracle (tests/test_unet.py's tolerance)."""
from __future__ import annotations

import pytest
import torch

from oracle.unet import UNetConfig, synth_state_dict, unet_forward
from pcdms_amd import ops
from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel, UNet2DConditionModel
from pcdms_amd.unet_ctx import UNetContext
from tests.test_unet import _kwargs


def _run_both(backend, cfg, B, h, w, L, n0, cls=Stage2_InapintUNet2DConditionModel, pose=True, fp8=False):
    dev = backend.device
    sd = synth_state_dict(cfg, seed=3, random_affine=True)
    m = cls(**_kwargs(cfg))
    m.load_state_dict(sd)
    m.to(dev)
    if fp8:
        m.set_attention_precision("fp8")
    g = torch.Generator().manual_seed(0)
    s = torch.randn(B, cfg.in_channels, h, w, generator=g)
    e = torch.randn(B, L, cfg.cross_attention_dim, generator=g)
    e[:n0] = 0
    c = None if cfg.projection_class_embeddings_input_dim is None else torch.randn(B, 1, cfg.projection_class_embeddings_input_dim, generator=g) * 0.4
    p = torch.randn(1, cfg.block_out_channels[0], h, w, generator=g) * 0.1 if pose else None
    t = torch.tensor([417], dtype=torch.int64, device=dev)
    # Python schedule (tunes unseen shapes on the GPU on its first pass)
    cond = m.prepare_conditioning(B, h, w, e.to(dev), None if c is None else c.to(dev), None if p is None else p.to(dev), zero_ctx_batches=n0)
    x_in = ops.nchw_to_nhwc_bf16(s.to(dev), cpad=m._w["conv_in"].cin)
    ref = m._forward_nhwc(x_in, B, h, w, t, cond).clone()
    ref2 = m._forward_nhwc(x_in, B, h, w, t, cond).clone()
    backend.sync()
    assert torch.equal(ref, ref2)
    # C schedule
    ctx = UNetContext(m)
    pose_b = ctx.prepare_conditioning(B, h, w, e, c, p, zero_ctx_batches=n0)
    out = ctx.forward(x_in, t, None, B, h, w, pose_b)
    backend.sync()
    return sd, (s, e, c, p), ref, out


def test_c_schedule_equals_python_schedule_stage2(backend):
    cfg = UNetConfig.tiny()
    B, h, w, L, n0 = (2, 8, 8, 5, 1) if backend.is_emu else (4, 16, 24, 9, 2)
    sd, (s, e, c, p), ref, out = _run_both(backend, cfg, B, h, w, L, n0)
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert torch.equal(out, ref), (out - ref).abs().max()
    if not backend.is_emu:   # and the oracle, at the suite's forward tolerance
        o = unet_forward(sd, cfg, s, torch.tensor(417), e, c, p)
        rel = ((out.cpu() - o).norm() / o.norm()).item()
        assert rel <= 2.5e-2, rel


def test_c_schedule_without_the_composed_weights(backend, monkeypatch):
    """A host that registers the plain layers only (no ``<resnet>conv2s``, no ``<transformer>ffo``: pcdm.h lists both as optional) gets the
    two-launch forms of ``conv2 + conv_shortcut`` and ``ff.net.2 -> proj_out`` from ``pcdm_unet_forward`` -- still bit for bit the Python
    schedule of a model packed the same way (pcdms_amd/unet.py FUSE_SHORTCUT / FUSE_FF_OUT off)."""
    from pcdms_amd import unet as U
    monkeypatch.setattr(U, "FUSE_SHORTCUT", False)
    monkeypatch.setattr(U, "FUSE_FF_OUT", False)
    cfg = UNetConfig.tiny()
    B, h, w, L, n0 = (2, 8, 8, 5, 1) if backend.is_emu else (4, 16, 24, 9, 2)
    _, _, ref, out = _run_both(backend, cfg, B, h, w, L, n0)
    assert torch.equal(out, ref), (out - ref).abs().max()


def test_c_schedule_fp8_attention(backend):
    """``pcdm_unet_set_attention_fp8``: the C schedule with every attention on e4m3 operands (BASELINE.json configs[4]) -- K / V^T quantised
    per attention and, for the context, once per conditioning -- equals the Python schedule's fp8 path bit for bit."""
    cfg = UNetConfig.tiny()
    B, h, w, L, n0 = (2, 8, 8, 5, 1) if backend.is_emu else (4, 16, 24, 9, 2)
    _, _, ref, out = _run_both(backend, cfg, B, h, w, L, n0, fp8=True)
    assert torch.equal(out, ref), (out - ref).abs().max()
    _, _, ref_bf, _ = _run_both(backend, cfg, B, h, w, L, n0)
    assert not torch.equal(ref, ref_bf) and ((ref - ref_bf).norm() / ref_bf.norm()).item() < 6e-2   # (it IS another precision)


@pytest.mark.gpu
def test_c_schedule_full_size_and_graph_capture(gpu_backend):
    """The 868.9 M-parameter configuration at latent 64x88, UNet batch 8 (BASELINE.json configs[1]): C schedule == Python schedule, also
    when the single C call is captured into a hipGraph and replayed with the timestep taken from a device table."""
    cfg = UNetConfig()
    dev = gpu_backend.device
    B, h, w, L, n0 = 8, 64, 88, 258, 4
    sd = synth_state_dict(cfg, seed=3, random_affine=True)
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(sd)
    m.to(dev)
    g = torch.Generator().manual_seed(0)
    s = torch.randn(B, cfg.in_channels, h, w, generator=g)
    e = torch.randn(B, L, cfg.cross_attention_dim, generator=g)
    e[:n0] = 0
    c = torch.randn(B, 1, cfg.projection_class_embeddings_input_dim, generator=g) * 0.4
    p = torch.randn(1, cfg.block_out_channels[0], h, w, generator=g) * 0.1
    # a device timestep TABLE and a device step counter: what the sampler's captured step reads (nothing host-side changes between replays)
    ts = torch.tensor([417, 801, 33], dtype=torch.int64, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    cond = m.prepare_conditioning(B, h, w, e.to(dev), c.to(dev), p.to(dev), zero_ctx_batches=n0)
    x_in = ops.nchw_to_nhwc_bf16(s.to(dev), cpad=m._w["conv_in"].cin)
    refs = []
    for i in range(3):       # Python schedule, eager, at the three timesteps (the first pass tunes unseen shapes)
        step.fill_(i)
        refs.append(m._forward_nhwc(x_in, B, h, w, ts, cond, step).clone())
    assert not torch.equal(refs[0], refs[1])
    ctx = UNetContext(m)
    pose_b = ctx.prepare_conditioning(B, h, w, e, c, p, zero_ctx_batches=n0)
    out = torch.empty_like(refs[0])
    step.fill_(0)
    ctx.forward(x_in, ts, step, B, h, w, pose_b, out=out)            # eager C schedule
    torch.cuda.synchronize()
    assert torch.equal(out, refs[0]), (out - refs[0]).abs().max()
    # ---- the single C call captured into a hipGraph, replayed with the step counter advanced on the device
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        ctx.forward(x_in, ts, step, B, h, w, pose_b, out=out)
    for i in (1, 2, 0, 1):
        step.fill_(i)
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, refs[i]), (i, (out - refs[i]).abs().max())


@pytest.mark.gpu
def test_c_schedule_stage3_topology(gpu_backend):
    """No class embedding, no pose (the stage-3 refinement UNet, in_channels 8), a latent that is not divisible by 8."""
    cfg = UNetConfig.tiny(in_channels=8, class_embed_type=None, projection_class_embeddings_input_dim=None)
    sd, _, ref, out = _run_both(gpu_backend, cfg, 2, 16, 12, 7, 1, cls=UNet2DConditionModel, pose=False)
    assert torch.equal(out, ref), (out - ref).abs().max()


def test_c_packers_equal_python_packers():
    """pcdm_pack_linear / _conv3x3 / _geglu (what a C host uses) produce the very bytes pcdms_amd.ops.pack_* upload."""
    import ctypes as C

    from pcdms_amd import _lib
    try:
        lib = _lib.lib()
    except RuntimeError:
        from tests.emu import build_emu
        _lib.use_library(build_emu.load())
        lib = _lib.lib()
    g = torch.Generator().manual_seed(1)
    cpu = torch.device("cpu")
    # linear (N not a multiple of 64)
    N, K = 100, 128
    w, b = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    pw = ops.pack_linear(w, b, cpu)
    ow, ob = torch.empty(pw.Npad, K, dtype=torch.int16), torch.empty(pw.Npad)
    assert lib.pcdm_pack_linear(w.data_ptr(), b.data_ptr(), N, K, 64, ow.data_ptr(), ob.data_ptr()) == pw.Npad
    assert torch.equal(ow, pw.w.view(torch.int16)) and torch.equal(ob, pw.bias)
    # conv3x3 (Cin padded 9 -> 64)
    N, Cin = 64, 9
    w, b = torch.randn(N, Cin, 3, 3, generator=g), torch.randn(N, generator=g)
    pw = ops.pack_conv3x3(w, b, cpu)
    ow, ob = torch.empty(pw.Npad, pw.K, dtype=torch.int16), torch.empty(pw.Npad)
    kk, cc = C.c_int(0), C.c_int(0)
    assert lib.pcdm_pack_conv3x3(w.contiguous().data_ptr(), b.data_ptr(), N, Cin, 64, ow.data_ptr(), ob.data_ptr(), C.addressof(kk), C.addressof(cc)) == pw.Npad
    assert (kk.value, cc.value) == (pw.K, pw.cin) and torch.equal(ow, pw.w.view(torch.int16)) and torch.equal(ob, pw.bias)
    # GEGLU (D not a multiple of 64)
    D, K = 96, 64
    w, b = torch.randn(2 * D, K, generator=g), torch.randn(2 * D, generator=g)
    pw = ops.pack_geglu(w, b, cpu)
    ow, ob = torch.empty(pw.Npad, K, dtype=torch.int16), torch.empty(pw.Npad)
    assert lib.pcdm_pack_geglu(w.data_ptr(), b.data_ptr(), D, K, ow.data_ptr(), ob.data_ptr()) == pw.Npad and pw.N == D
    assert torch.equal(ow, pw.w.view(torch.int16)) and torch.equal(ob, pw.bias)


def test_pipeline_with_c_schedule_equals_python_schedule(backend):
    """The fused + hipGraph sampler with the UNet of every step run by ONE captured ``pcdm_unet_forward`` call: bit-identical final latents
    to the default (Python-scheduled) capture, DDIM and UniPC."""
    from oracle.pipeline import synth_inputs
    from pcdms_amd.pipeline import Stage2_InpaintDiffusionPipeline
    from pcdms_amd.schedulers import DDIMScheduler, UniPCMultistepScheduler
    from tests.test_schedulers import SD21
    cfg = UNetConfig.tiny()
    dev = backend.device
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(synth_state_dict(cfg, seed=2, random_affine=True))
    m.to(dev)
    N, h, w, L, steps = (1, 8, 8, 4, 1) if backend.is_emu else (2, 16, 24, 9, 6)
    inp = synth_inputs(cfg, h, w, N, L_img=L)
    kw = dict(height=h * 8, width=w * 8, masked_latents=inp["masked_latents"].to(dev), s_img_proj_f=inp["s_img_proj_f"].to(dev),
              st_pose_f=inp["st_pose_f"].to(dev), pred_t_img_embed=inp["pred_t_img_embed"].to(dev), latents=inp["latents"].to(dev),
              num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=steps, output_type="latent")
    for sched in (DDIMScheduler,) if backend.is_emu else (DDIMScheduler, UniPCMultistepScheduler):
        a = Stage2_InpaintDiffusionPipeline(m, sched.from_config(SD21))(**kw).latents
        pc = Stage2_InpaintDiffusionPipeline(m, sched.from_config(SD21), c_schedule=True)
        b = pc(**kw).latents
        backend.sync()
        assert pc._st["ctx"] is not None and (backend.is_emu or pc._graph is not None)
        assert torch.equal(a, b), (sched.__name__, (a - b).abs().max())
        if not backend.is_emu:
            assert torch.equal(pc(**kw).latents, b)      # replay


def test_splitk_deferral_through_both_schedules(backend, monkeypatch):
    """The split-K reduce folded into the consuming GroupNorm (``ops.DeferredGemm`` / the C schedule's pending reduce), at the level of a
    whole forward: every convolution of the tiny UNet is forced onto a split-K configuration, then the forward with the fold must equal
    the forward without it (``PCDM_DEFER_SPLITK=0``) BIT FOR BIT, in the Python schedule and in ``pcdm_unet_forward``."""
    cfg = UNetConfig.tiny()
    B, h, w, L, n0 = (2, 8, 8, 5, 1) if backend.is_emu else (4, 16, 24, 9, 2)

    class Rec(dict):
        seen: list = []

        def get(self, k, d=None):
            self.seen.append(k)
            return super().get(k, d)

    rec = Rec(ops._TUNED)
    monkeypatch.setattr(ops, "_TUNED", rec)
    _run_both(backend, cfg, B, h, w, L, n0)           # records the problem keys of one forward (and tunes them on the GPU)
    forced = 0
    for k in set(rec.seen):
        if isinstance(k[0], int) and k[3] and k[6] == ops.EPI_STORE and k[2] // 64 >= 4 and k[1] % 64 == 0:   # conv3x3, plain store
            dict.__setitem__(rec, k, (2, 2))          # 64x64 tiles, K split in two
            forced += 1
    assert forced >= 6
    monkeypatch.setattr(ops, "DEFER_SPLITK", False)
    monkeypatch.setenv("PCDM_DEFER_SPLITK", "0")
    _, _, ref_py, ref_c = _run_both(backend, cfg, B, h, w, L, n0)
    monkeypatch.setattr(ops, "DEFER_SPLITK", True)
    monkeypatch.setenv("PCDM_DEFER_SPLITK", "1")
    made = []
    orig = ops.DeferredGemm.__init__
    monkeypatch.setattr(ops.DeferredGemm, "__init__", lambda self, **kw: (made.append(kw["store"]), orig(self, **kw))[1])
    _, _, out_py, out_c = _run_both(backend, cfg, B, h, w, L, n0)
    assert len(made) >= 2 * forced - 4 and any(made) and not all(made)   # conv1s (not stored) and conv2s / resampling convs (stored)
    assert torch.equal(ref_py, ref_c)
    assert torch.equal(out_py, ref_py), (out_py - ref_py).abs().max()
    assert torch.equal(out_c, ref_py), (out_c - ref_py).abs().max()


def test_layernorm_partials_through_both_schedules(backend, monkeypatch):
    """Round 5 at the level of a whole forward: the LayerNorm -> Linear pairs of the tiny UNet forced onto the folded instances with producer
    partials (table entries (tile, mode 2) for the ``ln`` keys, the linears that write the LayerNorm inputs on a tile that has a statistics
    instance).  The Python schedule and ``pcdm_unet_forward`` must agree BIT FOR BIT -- including where the folded form is refused (16 tokens
    per image: the V^T pass wants multiples of 32) and both fall back to LayerNorm + GEMM --, the result must stay within the forward
    tolerance of the unforced schedule, and LayerNorm launches must actually disappear."""
    cfg = UNetConfig.tiny()
    B, h, w, L, n0 = (2, 8, 8, 5, 1) if backend.is_emu else (4, 16, 24, 9, 2)

    class Rec(dict):
        seen: list = []

        def get(self, k, d=None):
            self.seen.append(k)
            return super().get(k, d)

    rec = Rec(ops._TUNED)
    monkeypatch.setattr(ops, "_TUNED", rec)
    calls = []
    orig_ln = ops.layernorm
    monkeypatch.setattr(ops, "layernorm", lambda *a, **k: (calls.append(1), orig_ln(*a, **k))[1])
    # reference: every LayerNorm as its own launch (on the GPU the online tuner would otherwise pick folded forms already); the lookups of
    # this run are recorded -- the ``ln`` keys too
    monkeypatch.setattr(ops, "LN_TILED", False)
    monkeypatch.setenv("PCDM_LN_TILED", "0")
    _, _, ref_py, ref_c = _run_both(backend, cfg, B, h, w, L, n0)
    assert torch.equal(ref_py, ref_c)
    n_plain = len(calls)
    monkeypatch.setattr(ops, "LN_TILED", True)
    monkeypatch.setenv("PCDM_LN_TILED", "1")
    widths = set(cfg.block_out_channels)
    forced_ln = forced_prod = 0
    for k in set(rec.seen):
        if k[0] == "ln":
            _, M, Npad, K, epi = k
            if K % 64 == 0 and K != 320:
                tile = 4 if (epi == ops.EPI_GEGLU and Npad % 128 == 0) else 2
                if epi != ops.EPI_GEGLU or Npad % 128 == 0:
                    dict.__setitem__(rec, k, (tile, 2))
                    forced_ln += 1
        elif isinstance(k[0], int) and not k[3] and k[6] == ops.EPI_STORE and k[2] in widths and k[1] == k[2] and not k[7] and len(k) == 9:
            dict.__setitem__(rec, k, (2, 1))      # proj_in / to_out (+ residual): 64 x 64 tiles, a row-statistics producer instance
            forced_prod += 1
        elif isinstance(k[0], int) and not k[3] and k[6] == ops.EPI_STORE and k[2] in widths and k[1] == k[2] and len(k) == 10 and k[9] is True:
            dict.__setitem__(rec, k, (2, 1))      # attn2.to_out with the zero-context rows
            forced_prod += 1
    assert forced_ln >= 6 and forced_prod >= 4, (forced_ln, forced_prod)
    del calls[:]
    _, _, out_py, out_c = _run_both(backend, cfg, B, h, w, L, n0)
    assert torch.equal(out_py, out_c), (out_py - out_c).abs().max()
    assert len(calls) < n_plain, (len(calls), n_plain)
    rel = ((out_py - ref_py).norm() / ref_py.norm()).item()
    assert rel <= 1e-2, rel


def test_cfg_shared_prefix_is_exact(backend, monkeypatch):
    """The CFG-shared prefix (``prepare_conditioning(shared_cfg_input=True)`` / ``pcdm_unet_set_shared_cfg_input``): with both halves of
    the batch carrying the same sample and pose, conv_in, the first norm1 and the first conv1's contraction run once for B/2 entries and
    are written for both halves.  With the half-batch problems held to the tile configuration of the full-batch ones the forward must
    equal the unshared forward BIT FOR BIT, in the Python schedule and in ``pcdm_unet_forward``; a sample that differs between the halves
    must NOT be shared (the flag is the caller's promise) -- checked by the pipeline-level test in tests/test_pipeline.py."""
    cfg = UNetConfig.tiny()
    dev = backend.device
    B, h, w, L, n0 = (2, 8, 8, 5, 1) if backend.is_emu else (8, 16, 24, 9, 4)
    sd = synth_state_dict(cfg, seed=3, random_affine=True)
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(sd)
    m.to(dev)
    g = torch.Generator().manual_seed(0)
    s_half = torch.randn(B // 2, cfg.in_channels, h, w, generator=g)
    s = torch.cat([s_half, s_half])                                       # both CFG halves: the same sample
    e = torch.randn(B, L, cfg.cross_attention_dim, generator=g)
    e[:n0] = 0
    c = torch.randn(B, 1, cfg.projection_class_embeddings_input_dim, generator=g) * 0.4     # class labels DIFFER between the halves
    p = torch.randn(1, cfg.block_out_channels[0], h, w, generator=g) * 0.1
    t = torch.tensor([417], dtype=torch.int64, device=dev)
    x_in = ops.nchw_to_nhwc_bf16(s.to(dev), cpad=m._w["conv_in"].cin) if m._w else None
    if x_in is None:
        m._pack()
        x_in = ops.nchw_to_nhwc_bf16(s.to(dev), cpad=m._w["conv_in"].cin)

    class Rec(dict):
        seen: list = []

        def get(self, k, d=None):
            self.seen.append(k)
            return super().get(k, d)
    rec = Rec(ops._TUNED)
    monkeypatch.setattr(ops, "_TUNED", rec)

    def run(shared):
        cond = m.prepare_conditioning(B, h, w, e.to(dev), c.to(dev), p.to(dev), zero_ctx_batches=n0, shared_cfg_input=shared)
        assert cond.shared_halves == shared
        a = m._forward_nhwc(x_in, B, h, w, t, cond).clone()
        ctx = UNetContext(m)
        pose_b = ctx.prepare_conditioning(B, h, w, e, c, p, zero_ctx_batches=n0, shared_cfg_input=shared)
        b = ctx.forward(x_in, t, None, B, h, w, pose_b)
        backend.sync()
        return a, b
    run(False); run(True)                                                  # record (and, on the GPU, tune) every problem key of both forms
    dup_keys = [k for k in set(rec.seen) if isinstance(k[0], int) and len(k) > 9 and k[9] == 2]
    assert len(dup_keys) == 2, dup_keys                                    # conv_in and the first conv1
    for k in dup_keys:                                                     # the half-batch problems on the tile of their full-batch twins
        full = (2 * k[0],) + k[1:9]
        dict.__setitem__(rec, full, (5, 1))
        dict.__setitem__(rec, k, (5, 1))
    ref_py, ref_c = run(False)
    out_py, out_c = run(True)
    assert torch.equal(ref_py, ref_c) and torch.equal(out_py, out_c)
    assert torch.equal(out_py, ref_py), (out_py - ref_py).abs().max()


def test_c_context_starts_with_the_committed_tuning_table():
    """``pcdm_unet_create`` compiles the committed tuning table in (csrc/tuning_table.inc, generated from pcdms_amd/tuning/gfx950.json): a host
    without the Python tuner runs on the measured tile set.  The committed include must be what the JSON generates, and
    a fresh context must answer with the table's entries -- tiled, LayerNorm-folded, zero_rows and split-K ones."""
    import ctypes as C
    import json

    from pcdms_amd import _lib
    from pcdms_amd.build import CSRC, ROOT, write_tuning_include
    before = (CSRC / "tuning_table.inc").read_text()
    assert write_tuning_include().read_text() == before, "csrc/tuning_table.inc is stale: run python -m pcdms_amd.build and commit it"
    if not torch.cuda.is_available():
        # (no device: the HIP library loads but finds no gfx950 and -- by design -- starts on the heuristic; the emulator build stands in for
        # the architecture.  Decided here, not by whichever library an earlier test left loaded: the test must pass when run alone.)
        from tests.emu import build_emu
        _lib.use_library(build_emu.load())
    lib = _lib.lib()
    cfg = _lib.UNetConfig()
    cfg.out_channels, cfg.n_levels, cfg.layers_per_block, cfg.cross_attention_dim, cfg.norm_groups, cfg.norm_eps = 4, 1, 1, 64, 32, 1e-5
    cfg.block_out_channels[0], cfg.heads[0], cfg.cross_attn[0] = 64, 1, 1
    h = lib.pcdm_unet_create(C.byref(cfg))
    assert h
    tab = json.loads((ROOT / "tuning" / "gfx950.json").read_text())["gemm"]
    seen = {"ln": 0, "zero_rows": 0, "split": 0, "plain": 0}
    for k, (tile, split) in tab.items():
        f = k.split(",")
        if f[0] == "ln":
            args, kind = (1, int(f[1]), int(f[2]), int(f[3]), 0, 0, 0, int(f[4]), 0, 0, 0), "ln"
        else:
            flag = 0 if len(f) < 10 else (1 if f[9] == "True" else int(f[9]))
            args = (0, int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7] == "True"), int(f[8] == "True"), flag)
            kind = "zero_rows" if flag == 1 else ("split" if split > 1 else "plain")
        t, sp = C.c_int(-1), C.c_int(-1)
        assert lib.pcdm_unet_get_tile(h, *args, C.byref(t), C.byref(sp)) == 0, k
        assert (t.value, sp.value) == (tile, split), (k, t.value, sp.value)
        seen[kind] += 1
    assert all(v > 0 for v in seen.values()), seen
    assert lib.pcdm_unet_get_tile(h, 0, 12345, 64, 64, 0, 0, 0, 0, 0, 0, 0, None, None) == -1
    lib.pcdm_unet_destroy(h)


def test_step_counter_beyond_the_time_table_is_clamped_and_reported(backend):
    """ADVICE r4 #1 / VERDICT r5 next #7: ``pcdm_unet_forward`` picks the time-embedding block of a step by a DEVICE counter; a counter beyond the
    table of ``pcdm_unet_prepare_timesteps`` used to read out of bounds (documented, not enforced).  ABI 4: every consumer clamps the counter
    into the table on the device and raises a flag -- ``pcdm_unet_step_overflow`` (C schedule) / ``unet.step_overflow()`` (Python schedule).
    In range: flag 0 and the output of the step; out of range: the LAST step's output (bit for bit), flag 1, no fault."""
    dev = backend.device
    cfg = UNetConfig.tiny()
    B, h, w, L, n0 = (2, 8, 8, 5, 1) if backend.is_emu else (4, 16, 24, 9, 2)
    sd = synth_state_dict(cfg, seed=3, random_affine=True)
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(sd)
    m.to(dev)
    g = torch.Generator().manual_seed(0)
    s = torch.randn(B, cfg.in_channels, h, w, generator=g)
    e = torch.randn(B, L, cfg.cross_attention_dim, generator=g)
    e[:n0] = 0
    c = torch.randn(B, 1, cfg.projection_class_embeddings_input_dim, generator=g) * 0.4
    p = torch.randn(1, cfg.block_out_channels[0], h, w, generator=g) * 0.1
    ts = torch.tensor([801, 401, 1], dtype=torch.int64, device=dev)
    # ---- Python schedule
    cond = m.prepare_conditioning(B, h, w, e.to(dev), c.to(dev), p.to(dev), zero_ctx_batches=n0, timesteps=ts)
    x_in = ops.nchw_to_nhwc_bf16(s.to(dev), cpad=m._w["conv_in"].cin)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    outs = {}
    for sv in (0, 2, 7):
        step.fill_(sv)
        outs[sv] = m._forward_nhwc(x_in, B, h, w, ts, cond, step_dev=step).clone()
        backend.sync()
        assert m.step_overflow() == (sv >= 3), sv
    assert not torch.equal(outs[0], outs[2]) and torch.equal(outs[7], outs[2])
    m.prepare_conditioning(B, h, w, e.to(dev), c.to(dev), p.to(dev), zero_ctx_batches=n0, timesteps=ts)   # a new table clears the flag
    assert not m.step_overflow()
    # ---- C schedule
    ctx = UNetContext(m)
    pose_b = ctx.prepare_conditioning(B, h, w, e, c, p, zero_ctx_batches=n0)
    ctx.prepare_timesteps(ts, B, h, w)
    couts = {}
    for sv in (2, 5):
        step.fill_(sv)
        couts[sv] = ctx.forward(x_in, ts, step, B, h, w, pose_b).clone()
        backend.sync()
        assert ctx.step_overflow(B, h, w) == (sv >= 3), sv
    assert torch.equal(couts[2], outs[2]) and torch.equal(couts[5], outs[2])
    ctx.prepare_timesteps(ts, B, h, w)
    assert not ctx.step_overflow(B, h, w)
