"""N > 1 path on CPU: world_size-2 gloo processes, pairs sharded like the reference, one all-gather."""
from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pcdms_amd.parallel import chunk_index_ranges, run_sharded, split_list_into_chunks


def test_split_matches_reference_semantics():
    # remainder folded into the last chunk (ref stage2_batchtest_inpaint_model.py:25-31)
    assert split_list_into_chunks(list(range(10)), 3) == [[0, 1, 2], [3, 4, 5], [6, 7, 8, 9]]
    assert split_list_into_chunks(list(range(8)), 8) == [[i] for i in range(8)]
    assert split_list_into_chunks(list(range(9)), 2) == [[0, 1, 2, 3], [4, 5, 6, 7, 8]]
    assert split_list_into_chunks([1, 2], 4) == [[1], [2], [], []]   # the reference raises here; we stay total
    assert [list(r) for r in chunk_index_ranges(5, 2)] == [[0, 1], [2, 3, 4]]
    with pytest.raises(ValueError):
        split_list_into_chunks([1], 0)


def _sample(pair):
    g = torch.Generator().manual_seed(int(pair))
    return torch.randn(2, 4, 3, 5, generator=g) + pair


_sample.example_output = torch.zeros(2, 4, 3, 5)


def _worker(rank, world, port, npairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}
    orig = dist.all_gather

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    dist.all_gather = counting
    try:
        res = run_sharded(list(range(npairs)), _sample)
    finally:
        dist.all_gather = orig
    ok = len(res) == npairs and all(torch.equal(r, _sample(i)) for i, r in enumerate(res)) and calls["n"] == (1 if npairs else 0)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("npairs", [5, 1])
def test_run_sharded_gloo_world2(npairs):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, npairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_run_sharded_single_process():
    res = run_sharded([3, 4], _sample)
    assert torch.equal(res[1], _sample(4))


def _nccl_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))

    def sample(pair):
        return (_sample(pair)).to(f"cuda:{rank}")
    sample.example_output = torch.zeros(2, 4, 3, 5, device=f"cuda:{rank}")
    res = run_sharded(list(range(5)), sample)
    ok = len(res) == 5 and all(torch.equal(r.cpu(), _sample(i)) for i, r in enumerate(res))
    q.put((rank, ok, dist.get_world_size()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_run_sharded_rccl_world2():
    """The same harness over RCCL (backend "nccl"), one process per GPU; needs two devices (the 1-GPU box skips)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True, 2), (1, True, 2)]


def _pipe_worker(rank, world, port, q):
    """run_sharded over the REAL sampler: the stage-2 pipeline on the lane-emulator build of the kernel sources (tiny UNet, one DDIM step)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pipeline import synth_inputs
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import _lib
    from pcdms_amd.pipeline import Stage2_InpaintDiffusionPipeline
    from pcdms_amd.schedulers import DDIMScheduler
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel
    from tests.emu import build_emu
    from tests.test_schedulers import SD21
    from tests.test_unet import _kwargs
    _lib.use_library(build_emu.load())
    cfg = UNetConfig.tiny()
    m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
    m.load_state_dict(synth_state_dict(cfg, seed=0, random_affine=True))      # replicated weights: the same seed on every rank
    pipe = Stage2_InpaintDiffusionPipeline(m, DDIMScheduler.from_config(SD21))
    N, h, w = 1, 8, 8
    base = synth_inputs(cfg, h, w, N, L_img=4)
    calls = {"n": 0}

    def sample(pair):   # one (source, target) pair = its own seeded latents and conditioning
        calls["n"] += 1
        g = torch.Generator().manual_seed(100 + int(pair))
        inp = dict(base, latents=torch.randn(base["latents"].shape, generator=g), s_img_proj_f=torch.randn(base["s_img_proj_f"].shape, generator=g))
        return pipe(height=h * 8, width=w * 8, num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=1, output_type="latent",
                    use_graph=False, **inp).latents
    sample.example_output = torch.zeros(N, 4, h, w)
    pairs = [0, 1, 2]                                     # chunks [0] and [1, 2] (the reference's split: remainder into the last chunk)
    res = run_sharded(pairs, sample)
    mine = calls["n"]
    # every rank holds every pair's result, in pair order; check one pair of the OTHER rank's chunk by recomputing it here
    other = 2 if rank == 0 else 0
    again = sample(other)
    ok = len(res) == 3 and mine == (1 if rank == 0 else 2) and torch.equal(res[other], again) and not torch.equal(res[0], res[1]) and \
        all(torch.isfinite(r).all() for r in res)
    gathered = [torch.zeros(3, N, 4, h, w) for _ in range(world)]
    dist.all_gather(gathered, torch.stack(res))           # (test only: both ranks must hold the SAME list)
    ok = ok and torch.equal(gathered[0], gathered[1])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_run_sharded_gloo_world2_with_the_real_pipeline_on_the_emulator():
    """The data-parallel harness around the product's own sampler: two gloo ranks, three pairs split like the reference
    (/root/reference/stage2_batchtest_inpaint_model.py:25-31,266-285), each pair sampled by ``Stage2_InpaintDiffusionPipeline`` on the lane emulator,
    one all-gather; every rank ends with all three results in pair order, and a pair computed on the other rank equals a local recomputation."""
    from tests.emu import build_emu
    build_emu.build()                                     # (before the ranks start: they must not race on the emulator build)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
