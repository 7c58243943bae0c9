#!/usr/bin/env python3
"""bench.py -- stage-2 inpainting sampler throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one complete sampling call of the hot path: 50 DDIM denoise steps (guidance 2.0, CFG
=> UNet batch 2*batch) for ``--batch`` generated 352x512 images (canvas 704x512, latent 64x88) of one
(source, target) pair, inputs resident in HBM, final latents returned (VAE decode is outside the hot
path).  Workload at N=1 = BASELINE.json configs[1] ("stage2 inpaint, 352x512,
batch=4, 50 DDIM steps, bf16, 1xMI355X").  N>1: every rank samples its own pair with the same
per-GPU batch (weak scaling), then ONE RCCL all-gather collects the final latents.  Launched bare
(``python bench.py --gpus 8``, no WORLD_SIZE in the environment) it spawns the N ranks itself through
``torch.distributed.run`` on 127.0.0.1; under an external launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.
At N=8 it also reports BASELINE.json configs[2]'s own per-GPU batch (8 => 64 images per step) as ``config.configs2``.

Synthetic data: seeded random-init weights of the full 868.9 M-parameter stage-2
UNet and seeded synthetic conditioning -- there are no checkpoints / DeepFashion pairs offline.
Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FLOP_PER_IMAGE = 118.84e12        # 50 steps x 2 CFG rows x 1188.4 GFLOP (latent 64x88, L=258)
FLOP_PER_ROW_FWD = 1188.4e9
PEAK_HBM_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA peak
# The PRACTICAL matrix-pipe roof (round 6, tools/ubench/mfma_power.hip -> profiles/r6_ubench_mfma_power.txt): a register-resident
# v_mfma_f32_16x16x32_bf16-only loop (no LDS, no memory traffic) on N(0,1) operands, two waves per SIMD on all 256 CUs, sustains 1952 TF/s at an
# in-kernel shader clock of 1.96 GHz (all-zero operands: 2182-2260 TF/s at 2.2-2.3 GHz; 32x32x16 fragments on N(0,1): 1720-1790 at 1.75 GHz) --
# the chip clocks to its power budget.  No bf16 kernel on real data can beat this number on this part; fractions are quoted against both.
PRACTICAL_BF16_TFLOPS = 1952.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="generated images per GPU per step (num_images_per_prompt)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=352, help="single image width; the canvas is 2x this")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the separate VAE encode/decode timing (SURVEY.md §8f N1)")
    ap.add_argument("--no-configs2", action="store_true", help="at --gpus 8: skip the extra batch-8-per-GPU (configs[2]) measurement")
    ap.add_argument("--attn", choices=("bf16", "fp8"), default="bf16",
                    help="fp8 = BASELINE.json configs[4]: e4m3 attention operands on the MX-scaled fp8 MFMA (looser parity; NOT the headline)")
    # ---- crash-proofing of the world > 1 code path where no multi-GPU node exists (tests/test_bench_contract.py; never a measurement):
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="process-group backend; gloo only together with --emu")
    ap.add_argument("--emu", action="store_true", help="TEST HOOK: CPU tensors through the lane emulator build of the same kernel sources "
                                                       "(tests/emu), no GPU, no hipGraph -- executes this file's multi-rank plumbing, measures nothing")
    ap.add_argument("--tiny", action="store_true", help="the tiny UNet configuration of the CPU suite (same topology) instead of the 868.9 M one")
    ap.add_argument("--device-weights", action="store_true", help="draw the weights as the world > 1 ranks do (device_state_dict) also at world 1")
    ap.add_argument("--pair-seed", type=int, default=None, help="seed of this process's pair (default 1000 + rank)")
    ap.add_argument("--configs2-world", type=int, default=8, help="world size at which BASELINE configs[2]'s own per-GPU batch is also timed")
    ap.add_argument("--configs2-batch", type=int, default=8)
    ap.add_argument("--dump", default=None, help="directory: every rank writes rank<r>.pt (its final latents, the gathered tensor, a hash of its weights)")
    args = ap.parse_args()
    if args.emu != (args.backend == "gloo"):
        raise SystemExit("--emu and --backend gloo go together (the product path is RCCL on GPUs)")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher (one process per GPU, rendezvous on 127.0.0.1); rank 0's
        # JSON line is the only thing the children print to stdout
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("PCDM_BENCH_FORCE_DIST") == "1"   # (1-rank RCCL group: test hook)
    if world > 1:
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if args.emu:
        from pcdms_amd import _lib
        from tests.emu import build_emu
        _lib.use_library(build_emu.load())
        dev = torch.device("cpu")
        args.no_graph = True
    else:
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    from oracle.pipeline import synth_inputs           # seeded synthetic inputs (data only)
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import ops
    from pcdms_amd.pipeline import Stage2_InpaintDiffusionPipeline
    from pcdms_amd.schedulers import DDIMScheduler
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel

    cfg = UNetConfig.tiny() if args.tiny else UNetConfig()
    h, w = args.height // 8, 2 * args.width // 8
    N = args.batch
    t0 = time.time()
    # N = 1: the CPU-seeded weights the parity tests and the cpu_baseline leg use.  N > 1: every rank would spend ~15 s x N of shared
    # host cores on 869 M host random numbers; the ranks draw the same distribution on their GPU instead (same seeds => the same
    # replicated weights on every rank)
    sd = synth_state_dict(cfg, seed=0) if world == 1 and not args.device_weights else device_state_dict(cfg, 0, dev)
    unet = Stage2_InapintUNet2DConditionModel(
        in_channels=9, block_out_channels=cfg.block_out_channels, attention_head_dim=cfg.attention_head_dim,
        cross_attention_dim=cfg.cross_attention_dim, use_linear_projection=True, class_embed_type="projection",
        projection_class_embeddings_input_dim=cfg.projection_class_embeddings_input_dim, sample_size=64)
    unet.load_state_dict(sd)
    unet.to(dev)
    unet.set_attention_precision(args.attn)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)  # notebook cell 15
    pipe = Stage2_InpaintDiffusionPipeline(unet, sched)
    synth_kw = dict(L_img=4) if args.tiny else {}
    inp = synth_inputs(cfg, h, w, N, **synth_kw)
    # per-rank pair: different seeded latents / conditioning per rank (data parallel over pairs)
    g = torch.Generator().manual_seed(args.pair_seed if args.pair_seed is not None else 1000 + rank)
    inp["latents"] = torch.randn(inp["latents"].shape, generator=g)
    dinp = {k: v.to(dev) for k, v in inp.items()}
    setup_s = time.time() - t0

    def timed(dinp, n_img, steps, warmup):
        """W warm-up + K timed sampling calls of n_img images per GPU, barrier + synchronize on both sides, max over ranks."""
        gathered = torch.empty(world * n_img, 4, h, w, dtype=torch.float32, device=dev) if use_dist else None

        def step():
            lat = pipe(height=args.height, width=2 * args.width, masked_latents=dinp["masked_latents"],
                       s_img_proj_f=dinp["s_img_proj_f"], st_pose_f=dinp["st_pose_f"],
                       pred_t_img_embed=dinp["pred_t_img_embed"], latents=dinp["latents"], num_images_per_prompt=n_img,
                       guidance_scale=2.0, num_inference_steps=args.ddim_steps, output_type="latent",
                       use_graph=not args.no_graph).latents
            if use_dist and args.backend == "nccl":
                dist.all_gather_into_tensor(gathered, lat)   # the single collective of the path (RCCL over xGMI)
            elif use_dist:
                dist.all_gather(list(gathered.view(world, n_img, 4, h, w).unbind(0)), lat.contiguous())   # (gloo: the CPU test hook)
            return lat

        for _ in range(warmup):
            step()
        if use_dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        if use_dist:
            dist.barrier()
        sync()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        assert torch.isfinite(out).all()
        last.update(lat=out.detach().clone(), gathered=None if gathered is None else gathered.clone())
        return elapsed

    last = {}

    elapsed = timed(dinp, N, args.steps, args.warmup)
    if args.dump:
        import hashlib
        hsh = hashlib.sha256()
        for k in sorted(sd):
            hsh.update(k.encode()); hsh.update(sd[k].detach().float().cpu().numpy().tobytes())
        Path(args.dump).mkdir(parents=True, exist_ok=True)
        torch.save({"lat": last["lat"].cpu(), "gathered": None if last["gathered"] is None else last["gathered"].cpu(), "sd_hash": hsh.hexdigest(),
                    "rank": rank, "world": world}, Path(args.dump) / f"rank{rank}.pt")
    images = world * N * args.steps
    value = images / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    e2e_tflops = value * FLOP_PER_IMAGE * (h * w) / (64 * 88) / world / 1e12

    result = {
        "metric": "images/sec (50-step DDIM, 352x512 stage2)", "value": round(value, 4), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.attn == "bf16" else "bf16 (attention: fp8 e4m3)",
        "data": "synthetic" if not args.emu else "synthetic (LANE EMULATOR on CPU: a plumbing test of the multi-rank path, not a measurement)",
        "config": {"workload": f"stage2 inpaint, {args.width}x{args.height} (canvas {2 * args.width}x{args.height}, "
                               f"latent {h}x{w}), batch={N} per GPU, {args.ddim_steps} DDIM steps, guidance 2.0 (UNet batch {2 * N}), "
                               "868.9M-param UNet, 258 context tokens, bf16 MFMA / fp32 accumulate",
                   "global_batch": world * N, "ms_per_denoise_step": round(ms_per_step / args.ddim_steps, 3),
                   "parallelism": f"dp{world}", "hipgraph": not args.no_graph, "setup_s": round(setup_s, 1),
                   "rccl_world_size": dist.get_world_size() if use_dist else 1, "process_group": dist.get_backend() if use_dist else None,
                   "attention": args.attn,
                   "e2e_tflops_per_gpu": round(e2e_tflops, 1),
                   "flops_note": "e2e_tflops_per_gpu and roofline.e2e_frac credit the UN-HOISTED algorithmic FLOPs of SURVEY.md §8d "
                                 "(118.84 TFLOP per image); executed FLOPs are ~3.5% lower: the cross-attention K/V projections run once "
                                 "per call instead of once per step, and the all-zero-context CFG half of every cross-attention "
                                 "(LN2, to_q, QK^T/PV, to_out contraction) is skipped (output == to_out.bias exactly); round 6: the three "
                                 "Upsample2D convolutions run as their phase decomposition (2x2 convolutions of the low-res tensor with summed taps: "
                                 "4/9 of their 2*M*N*9C FLOPs, another ~4.5% of the step) -- roofline.executed_gflop_per_denoise_step and "
                                 "e2e_frac_executed count what is issued"},
    }
    N8 = args.configs2_batch
    if world == args.configs2_world and world > 1 and N != N8 and not args.no_configs2:
        # BASELINE.json configs[2] as written: batch 64 over 8 GPUs = 8 images (UNet batch 16) per GPU; same protocol
        inp8 = synth_inputs(cfg, h, w, N8, **synth_kw)
        inp8["latents"] = torch.randn(inp8["latents"].shape, generator=torch.Generator().manual_seed(2000 + rank))
        el8 = timed({k: v.to(dev) for k, v in inp8.items()}, N8, args.steps, args.warmup if args.no_graph else max(1, args.warmup))
        result["config"]["configs2"] = {"workload": f"stage2 inpaint, 352x512, batch={world * N8} ({N8} per GPU, UNet batch {2 * N8}), {args.ddim_steps} DDIM steps, dp{world}",
                                        "value": round(world * N8 * args.steps / el8, 4), "unit": "images/s",
                                        "ms_per_step": round(el8 / args.steps * 1e3, 3)}
        if args.dump:
            torch.save({"lat": last["lat"].cpu(), "gathered": last["gathered"].cpu()}, Path(args.dump) / f"rank{rank}_configs2.pt")
        timed(dinp, N, 1, 0)   # every rank: back to the headline workload's captured state (kernel_roofline re-runs single steps of it)

    if rank == 0 and not args.no_vae:
        result["config"].update(vae_timing(dev, N, args.height, 2 * args.width, ms_per_step))
    if rank == 0 and not args.no_roofline:
        result["roofline"] = kernel_roofline(pipe, ops, dinp, N, h, w)
        # whole-path fraction of the MFMA roof beside the dominant family's (un-hoisted algorithmic FLOPs, see config.flops_note)
        result["roofline"]["e2e_frac"] = round(e2e_tflops / PEAK_BF16_TFLOPS, 4)
        result["roofline"]["e2e_frac_of_practical"] = round(e2e_tflops / PRACTICAL_BF16_TFLOPS, 4)
        # launch boundaries of the replayed graph: wall time of a denoise step minus the kernels' own durations (eager per-launch events of the same
        # step, each corrected by the measured cost of an empty event bracket)
        busy_ms = result["roofline"].pop("_busy_ms")
        result["config"]["graph_wall_minus_busy_ms"] = round(ms_per_step / args.ddim_steps - busy_ms, 3)
        result["config"]["kernel_busy_ms_per_denoise_step"] = round(busy_ms, 3)
        # the same with the FLOPs the kernels actually execute (K/V hoisting and the CFG cross-attention skip taken out)
        ex_tflops = result["roofline"]["executed_gflop_per_denoise_step"] * 1e9 * args.ddim_steps * args.steps / elapsed / 1e12
        result["roofline"]["e2e_frac_executed"] = round(ex_tflops / PEAK_BF16_TFLOPS, 4)
        result["config"]["e2e_tflops_per_gpu_executed"] = round(ex_tflops, 1)
    if rank == 0:
        result["parity"] = parity_summary()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N=1 only (other ranks would idle)
        result["cpu_baseline"] = cpu_baseline(N, args.ddim_steps)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def parity_summary():
    """Where the bf16 HIP path sits relative to the fp32 oracle AND relative to the reference's own fp16 numerics (committed measurements:
    profiles/r6_parity_values.json from the -m gpu suite on MI355X, profiles/r6_fp16_budget.json from tests/golden/make_fp16_budget.py) -- not
    re-measured by this run.  rel-L2 throughout; "parity vs the in-repo oracle; upstream diffusers 0.24 unverified" (DESIGN.md section 5)."""
    out = {"oracle": "in-repo fp32 restatement (oracle/); parity UNPINNED upstream for the diffusers 0.24 block internals"}
    try:
        pv = json.loads((ROOT / "profiles" / "r6_parity_values.json").read_text())["values"]
        fb = json.loads((ROOT / "profiles" / "r6_fp16_budget.json").read_text())
        fwd = [pv[k] for k in pv if k.startswith("configs1.forward.step")]
        out["hip_vs_fp32"] = {"forward_configs1": [round(min(fwd), 5), round(max(fwd), 5)], "trajectory_50_steps_configs1": round(pv["configs1.trajectory.final"], 6),
                              "trajectory_20_steps_configs0": round(pv["configs0.trajectory.20"], 6)}
        f1 = [v["fp16ref_vs_fp32"] for v in fb["forward_configs1"].values()]
        out["fp16ref_vs_fp32"] = {"forward_configs1": [round(min(f1), 5), round(max(f1), 5)], "trajectory_20_steps_configs0": round(fb["config0"]["fp16"]["final_latents"], 6)}
        out["hip_over_fp16ref"] = {k: round(v, 3) for k, v in pv.items() if k.startswith("budget.") and k.endswith(("x_fp16ref", "x_bf16cast"))}
    except (OSError, KeyError, ValueError) as e:
        out["note"] = f"committed parity records not readable: {e!r}"
    return out


def device_state_dict(cfg, seed, dev):
    """The distribution of oracle.unet.synth_state_dict drawn with per-tensor seeded DEVICE generators: identical
    on every rank of one job (same seeds, same GPU type), no host RNG time.  Multi-GPU bench only -- parity tests and the CPU
    baseline use the CPU-seeded weights."""
    import math
    import zlib

    from oracle import unet as OU
    shapes = dict(OU.param_shapes(cfg))
    sd = {}
    for key, shape in shapes.items():
        wshape = shapes[key[: key.rfind(".") + 1] + "weight"]
        if len(wshape) == 1:
            sd[key] = torch.ones(shape) if key.endswith("weight") else torch.zeros(shape)
            continue
        g = torch.Generator(device=dev).manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
        t = (torch.rand(shape, generator=g, device=dev) * 2 - 1) / math.sqrt(math.prod(wshape[1:]))
        if key.endswith("weight") and any(key.endswith(sfx) for sfx in OU._HALF_SCALED):
            t = t * 0.5
        sd[key] = t
    return sd


def vae_timing(dev, N, height, width, ms_per_call):
    """Outside the headline metric (VAE is a 'next' row): one encode of the masked canvas + decode/uint8 of the N
    samples with the full SD-2.1 AutoencoderKL topology (83.65 M params, synthetic weights), HIP path."""
    from oracle.vae import VAEConfig, synth_state_dict as vae_sd
    from pcdms_amd.vae import AutoencoderKL
    vae = AutoencoderKL()
    vae.load_state_dict(vae_sd(VAEConfig(), 0))
    vae.to(dev)
    x = torch.rand(1, 3, height, width, device=dev) * 2 - 1
    z = torch.randn(N, 4, height // 8, width // 8, device=dev)

    def t(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    enc = t(lambda: vae.encode(x).latent_dist.sample())
    dec = t(lambda: vae.decode_to_uint8(z))
    return {"vae_encode_ms": round(enc, 2), "vae_decode_uint8_ms": round(dec, 2),
            "images_per_s_incl_vae": round(N / ((ms_per_call + enc + dec) * 1e-3), 4)}


def kernel_roofline(pipe, ops, dinp, N, h, w):
    """Per-launch HIP-event timing (on the launch stream) of every MFMA GEMM / implicit-conv launch of
    ONE eager denoise step; `achieved` = algorithmic FLOPs per launch / average launch duration for the
    dominant kernel family ``gemm_kernel`` (conv3x3 + linear: 79% of the step's FLOPs)."""
    st = pipe._st
    st["step"].zero_()
    st["lat"].copy_(dinp["latents"])
    t_host = 0.0
    for _ in range(2):   # warm
        t0 = time.perf_counter()
        pipe._step_eager(st)
        t_host = time.perf_counter() - t0   # host time to ENQUEUE one eager step (no sync inside)
        torch.cuda.synchronize()
    # The events bracket each launch on its stream, so a host that enqueues slower than the GPU executes would add idle
    # time between an event and its kernel.  Keep the GPU busy with a spin kernel while the whole step (launches +
    # events) is enqueued behind it, and take the per-launch minimum over 3 repetitions of the (deterministic) step.
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); e1.synchronize()
    cyc_per_s = 20_000_000 / (e0.elapsed_time(e1) * 1e-3)
    reps = []
    for _ in range(3):
        torch.cuda._sleep(int(cyc_per_s * min(3.0 * t_host + 0.01, 1.0)))
        ops.LAUNCH_LOG = []
        pipe._step_eager(st)
        torch.cuda.synchronize()
        reps.append(ops.LAUNCH_LOG)
        ops.LAUNCH_LOG = None
    assert len({len(r) for r in reps}) == 1
    log = []
    for entries in zip(*reps):
        name, flops, info = entries[0][0], entries[0][1], entries[0][4]
        log.append((name, flops, min(e[2].elapsed_time(e[3]) for e in entries), info))
    # cost of an empty event bracket on this stream (what every per-launch duration above includes)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    torch.cuda._sleep(int(cyc_per_s * 0.002))
    for a_, b_ in ev:
        a_.record(); b_.record()
    torch.cuda.synchronize()
    bracket_ms = sorted(a_.elapsed_time(b_) for a_, b_ in ev)[len(ev) // 2]
    busy_ms = sum(max(ms - bracket_ms, 0.0) for _, _, ms, _ in log)
    fam = {}
    tiles = {}
    executed = sum(ex for _, _, _, _, _, ex in (e if len(e) > 5 else (*e, e[1]) for e in reps[0]))   # FLOPs actually issued in one step
    for name, flops, ms, info in log:
        if name == "gemm_kernel":
            tiles[info[-2]] = tiles.get(info[-2], 0) + 1
        f = fam.setdefault(name, [0, 0.0, 0.0])
        f[0] += 1
        f[1] += flops
        f[2] += ms * 1e-3
    gk = fam["gemm_kernel"]
    achieved = gk[1] / gk[2] / 1e12
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes are separate runs (tools/pmc_traffic.py): the figure is quoted only while the
    # kernel sources it was measured on are the ones in this tree (hash recorded by the tool), never silently stale
    traffic, traffic_note = None, "no PMC traffic file"
    tj = ROOT / "profiles" / "gemm_traffic.json"
    if tj.exists():
        tdoc = json.loads(tj.read_text())
        if tdoc.get("kernel_src_sha256") == kernel_source_hash():
            traffic, traffic_note = tdoc.get("hbm_bytes_per_launch"), f"profiles/gemm_traffic.json ({tdoc.get('measured', 'rocprofv3 --pmc')})"
        else:
            traffic_note = "profiles/gemm_traffic.json was measured on other kernel sources (hash mismatch): not quoted"
    out = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
           "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "practical_peak": PRACTICAL_BF16_TFLOPS,
           "frac_of_practical": round(achieved / PRACTICAL_BF16_TFLOPS, 4),
           "practical_peak_source": "profiles/r6_ubench_mfma_power.txt: register-resident 16x16x32 bf16 MFMA-only loop on N(0,1) operands, 256 CUs x 2 waves per SIMD (1.96 GHz in-kernel clock)",
           "_busy_ms": busy_ms, "event_bracket_us": round(bracket_ms * 1e3, 2),
           "traffic": traffic, "traffic_source": traffic_note,
           "kernel": "gemm_kernel<BM,BN,CONV> (implicit-GEMM conv3x3 + linear, all instances)",
           "launches_per_denoise_step": gk[0], "avg_launch_us": round(gk[2] / gk[0] * 1e6, 2),
           "alg_gflop_per_launch": round(gk[1] / gk[0] / 1e9, 2),
           "tile_configs_used": {str(k): v for k, v in sorted(tiles.items())},
           "executed_gflop_per_denoise_step": round(executed / 1e9, 1),
           "note": "HIP events around each launch of one eager denoise step enqueued behind a spin kernel, per-launch min of 3 (UNet batch %d, latent %dx%d)" % (2 * N, h, w)}
    if "flash_attn_kernel" in fam:
        fa = fam["flash_attn_kernel"]
        out["flash_attn_kernel"] = {"achieved": round(fa[1] / fa[2] / 1e12, 1), "launches": fa[0],
                                    "avg_launch_us": round(fa[2] / fa[0] * 1e6, 2)}
    # the HBM-bound kernel families of the step against the 8 TB/s roof: algorithmic bytes = tensor read
    # once + written once, same per-launch event timing
    hbm = {}
    for name in ("groupnorm", "layernorm"):
        if name in fam:
            f = fam[name]
            hbm[name] = {"bound": "hbm", "achieved": round(f[1] / f[2] / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(f[1] / f[2] / 1e9 / PEAK_HBM_GBS, 4), "launches": f[0], "total_us": round(f[2] * 1e6, 1)}
    if hbm:
        out["hbm_bound_kernels"] = hbm
    return out


def kernel_source_hash() -> str:
    """sha256 over the HIP kernel sources + headers libpcdm.so is built from (what a PMC measurement is valid for)."""
    import hashlib
    hsh = hashlib.sha256()
    for f in sorted((ROOT / "pcdms_amd" / "csrc").glob("*")) + [ROOT / "include" / "pcdm.h"]:
        if f.is_file():
            hsh.update(f.name.encode()); hsh.update(f.read_bytes())
    return hsh.hexdigest()


def cpu_baseline(N, ddim_steps):
    """The fp32 PyTorch CPU restatement (oracle/) timed on this box's host cores -- substitute for the reference's CPU diffusers path,
    which cannot run (diffusers is not installed / vendored).  Timed in a CHILD process so that the OpenMP runtime starts
    pinned: ``OMP_PROC_BIND=close`` with explicit ``OMP_PLACES`` = one place per physical core, the cores of NUMA node 0
    first (then the following nodes) -- unpinned threads migrating across NUMA domains were why 64 threads measured slower than 32 in round 3.
      * ``value``: a BOUNDED sample of configs[1] -- one denoise step (UNet forward at the full latent size + CFG + DDIM update) for ONE
        generated image (UNet batch 2), 1 warm-up + 2 timed steps per thread count (8 / 16 / 32 / 64 where the host has them, plus the container's cgroup CPU quota when it has one), the best count
        extrapolated to the 50-step call; images/s per process;
      * ``config1``: BASELINE.json configs[0] in FULL -- one 256x256 pair (latent 32x64), N = 1, 20 DDIM steps, guidance 2.0, fp32."""
    import subprocess
    topo = host_topology()
    counts = baseline_thread_counts(topo)
    # explicit places, one per physical core, NUMA node 0's cores first: thread i of the child's OpenMP team is bound to cpus[i], so a
    # run with n threads uses exactly the first n cores of that order (computed HERE: once OMP_PROC_BIND is in the environment the
    # child's own main thread is already bound to one CPU when it could look)
    cpus = pinned_cpu_order(topo)[: max(counts)]
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES=",".join("{%d}" % c for c in cpus), OMP_NUM_THREADS=str(max(counts)),
               MKL_NUM_THREADS=str(max(counts)), PCDM_CPU_BASELINE_COUNTS=",".join(map(str, counts)), PCDM_CPU_BASELINE_STEPS=str(ddim_steps),
               PCDM_CPU_BASELINE_CPUS=",".join(map(str, cpus)), PCDM_CPU_BASELINE_TOPO=json.dumps(topo))
    env.pop("HIP_VISIBLE_DEVICES", None)
    try:
        r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--cpu-baseline-worker"], env=env, capture_output=True, text=True, timeout=900)
        doc = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:   # the GPU line must not die with the host leg
        return {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": f"CPU baseline worker failed: {e!r}", "host": topo}
    per = {int(k): v for k, v in doc["per_thread_count"].items()}
    best = min(per, key=lambda n: per[n]["s_per_step"])
    per_step = per[best]["s_per_step"]
    return {"value": round(1.0 / (per_step * ddim_steps), 5), "unit": "images/s", "cores": best, "kind": "port",
            "sample": f"one denoise step of configs[1] for ONE generated image (UNet batch 2, latent 64x88, fp32, torch {torch.__version__} CPU ops, attention through F.scaled_dot_product_attention as diffusers 0.24 does): "
                      f"1 warm-up + 2 timed steps per thread count, threads pinned one per physical core (OMP_PROC_BIND=close, explicit OMP_PLACES, NUMA node 0's cores first); {per_step:.2f} s/step at {best} threads, extrapolated x{ddim_steps}",
            "host": doc["host"], "per_thread_count": doc["per_thread_count"], "config1": doc["config1"],
            "note": "a reported baseline, not a target: one host process; the GPU line is whole-job throughput of N images per call"}


def baseline_thread_counts(topo):
    """Thread counts of the CPU-baseline sweep: 8 / 16 / 32 / 64 where the host has the cores; with a cgroup CPU-bandwidth quota below the
    visible core count (threads beyond it only take turns: 64 threads measured slower than 32 in rounds 3 / 4 on a 16-core quota) the
    quota's core count is added and counts beyond twice the quota are dropped."""
    counts = {c for c in (8, 16, 32, 64) if c <= topo["physical_cores"]} | {min(8, topo["physical_cores"])}
    quota = topo.get("cgroup_cpu_quota_cores")
    if quota:
        counts = {c for c in counts if c <= 2 * quota} | {max(1, min(int(quota), topo["physical_cores"]))}
    return sorted(counts)


def cpu_baseline_worker():
    """Child of ``cpu_baseline`` (``bench.py --cpu-baseline-worker``): its OpenMP team is placed by the environment the parent built."""
    topo = json.loads(os.environ["PCDM_CPU_BASELINE_TOPO"]) if "PCDM_CPU_BASELINE_TOPO" in os.environ else host_topology()
    counts = [int(c) for c in os.environ.get("PCDM_CPU_BASELINE_COUNTS", "8").split(",")]
    ddim_steps = int(os.environ.get("PCDM_CPU_BASELINE_STEPS", "50"))
    cpus = [int(c) for c in os.environ.get("PCDM_CPU_BASELINE_CPUS", "0").split(",")]   # (the parent's placement: OMP_PLACES)
    from oracle.pipeline import build_conditioning, stage2_sample, synth_inputs
    from oracle.schedulers import DDIMOracle
    import oracle.unet as OU
    from oracle.unet import UNetConfig, synth_state_dict, unet_forward
    OU.ATTENTION_IMPL = "sdpa"   # what diffusers 0.24 runs on torch >= 2 (AttnProcessor2_0); the explicit-softmax form is the parity checker's
    cfg = UNetConfig()
    torch.set_num_threads(max(counts))
    sd = synth_state_dict(cfg, seed=0)
    N, h, w = 1, 64, 88
    inp = synth_inputs(cfg, h, w, N)
    c = build_conditioning(inp["masked_latents"], inp["s_img_proj_f"], inp["st_pose_f"], inp["pred_t_img_embed"], N, True)
    sch = DDIMOracle()
    sch.set_timesteps(ddim_steps)
    lat = inp["latents"].clone()

    def one_step():
        t = sch.timesteps[0]
        t0 = time.perf_counter()
        x = torch.cat([lat] * 2)
        eps = unet_forward(sd, cfg, torch.cat([x, c["mask"], c["masked_latents"]], 1), t, c["feature_f"], c["prior_embed"], c["pose_cond"])
        u, cn = eps.chunk(2)
        out = sch.step(u + 2.0 * (cn - u), t, lat)
        assert torch.isfinite(out).all()
        return time.perf_counter() - t0
    per = {}
    with torch.no_grad():
        for n in counts:
            torch.set_num_threads(n)                     # the first n places
            one_step()                                   # warm-up (thread pool, primitive caches)
            ts = [one_step(), one_step()]
            v = min(ts)
            per[str(n)] = {"s_per_step": round(v, 3), "s_per_step_both": [round(x, 3) for x in ts], "images_per_s": round(1.0 / (v * ddim_steps), 5),
                           "tflops_fp32": round(2 * N * FLOP_PER_ROW_FWD / v / 1e12, 3)}
        best = min(per, key=lambda k: per[k]["s_per_step"])
        torch.set_num_threads(int(best))
        inp1 = synth_inputs(cfg, 32, 64, 1)
        t0 = time.perf_counter()
        out1 = stage2_sample(sd, cfg, DDIMOracle(), num_images_per_prompt=1, guidance_scale=2.0, num_inference_steps=20, **inp1)
        c1_s = time.perf_counter() - t0
    assert torch.isfinite(out1).all()
    topo["numa_nodes"] = numa_nodes()
    topo["pinned_cpus_first16"] = cpus[:16]
    print(json.dumps({"host": topo, "per_thread_count": per,
                      "config1": {"workload": "configs[0]: 1 pair 256x256 (latent 32x64), N=1, 20 DDIM steps, guidance 2.0, fp32 CPU, full run",
                                  "threads": int(best), "seconds": round(c1_s, 2), "images_per_s": round(1.0 / c1_s, 5)}}), flush=True)


def cgroup_cpu_quota(root: Path = Path("/sys/fs/cgroup")):
    """CPU-bandwidth quota of this container in cores (cgroup v2 ``cpu.max`` / v1 ``cpu.cfs_quota_us``), or None when unlimited / unreadable."""
    try:
        txt = (root / "cpu.max").read_text().split()
        if txt and txt[0] != "max":
            return round(int(txt[0]) / int(txt[1]), 2)
        if txt:
            return None
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = int((root / "cpu" / "cpu.cfs_quota_us").read_text())
        per = int((root / "cpu" / "cpu.cfs_period_us").read_text())
        return round(q / per, 2) if q > 0 and per > 0 else None
    except (OSError, ValueError):
        return None


def numa_nodes():
    """{node: cpulist string} from sysfs (what ``numactl -H`` prints)."""
    out = {}
    base = Path("/sys/devices/system/node")
    if base.exists():
        for d in sorted(base.glob("node[0-9]*")):
            try:
                out[d.name] = (d / "cpulist").read_text().strip()
            except OSError:
                pass
    return out


def _parse_cpulist(txt):
    cpus = []
    for part in txt.split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def pinned_cpu_order(topo):
    """Logical CPUs in the order the baseline's threads take them: ONE hyper-thread per physical core, NUMA node 0's cores first, then
    the other nodes in order (node numbering follows the sockets), restricted to this process's allowed set."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    order, seen_cores = [], set()
    nodes = numa_nodes() or {"node0": ",".join(map(str, sorted(allowed)))}
    for _, lst in sorted(nodes.items(), key=lambda kv: int(kv[0][4:])):
        for cpu in _parse_cpulist(lst):
            if cpu not in allowed:
                continue
            try:
                sib = Path(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read_text().strip()
                core = min(_parse_cpulist(sib))
            except OSError:
                core = cpu
            if core in seen_cores:
                continue
            seen_cores.add(core)
            order.append(cpu)
    return order or sorted(allowed)


def host_topology():
    """Sockets / physical cores / logical CPUs of this host from /proc/cpuinfo (what ``cores`` of the CPU baseline refers to)."""
    logical = os.cpu_count() or 1
    cores, sockets = set(), set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core)); sockets.add(phys)
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core)); sockets.add(phys)
    except OSError:
        pass
    n_phys = len(cores) or logical
    n_sock = len(sockets) or 1
    try:   # a container may be limited to fewer CPUs than the host shows
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = logical
    n_phys = max(1, min(n_phys, avail))
    return {"sockets": n_sock, "physical_cores": n_phys, "cores_per_socket": max(1, n_phys // n_sock), "logical_cpus": logical,
            "cgroup_cpu_quota_cores": cgroup_cpu_quota()}


if __name__ == "__main__":
    if "--cpu-baseline-worker" in sys.argv:
        cpu_baseline_worker()
    else:
        main()
