#!/usr/bin/env python3
"""IN-STEP tuning of the GEMM tuning table (round 6).

Three rounds of evidence say that back-to-back launches of ONE problem do not rank tile configurations the way the denoise step does (round 3 / 4: a
full re-tune was 0.2-1.0 % slower end to end; round 6: sixteen-by-sixteen-fragment twins 4-15 % faster per launch and 0.3 % slower end to end --
profiles/r6_ab_tiles.json): in the step every launch finds its weights cold (1.74 GB of them pass through the caches per step), its input freshly
written by another kernel, and the chip in whatever clock state the previous launches left.  So measure THERE: for every plain table key of the
stage-2 step (UNet batch 8, latent 64x88), for every valid (tile, split-K) candidate, run the whole eager denoise step with that one entry changed
and take the HIP-event time of the launches of that key (+ all GroupNorm launches when split-K is in play: a deferred reduce moves work into the norm
that follows) -- per launch the minimum over REPS steps -- keep the fastest, move on to the next key (greedy coordinate descent; the entries are
independent to first order).  Producers of LayerNorm partials keep a tile that can write them.

    python tools/tune_in_step.py [--write] [--reps 3] [--only 45056,320]      (MI355X; ~3 min)
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from collections import defaultdict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="", help="key prefix filter, e.g. 11264,640")
    ap.add_argument("--min-gain", type=float, default=0.02, help="relative in-step gain a candidate needs over the current entry")
    ap.add_argument("--out", default="gpurun_out/r6_tune_in_step.json")
    ap.add_argument("--batch", type=int, default=4, help="generated images per call (UNet batch = 2 x this): 4 = configs[1], 8 = the per-GPU share of configs[2]")
    ap.add_argument("--skip-plain", action="store_true", help="only the LayerNorm -> Linear pairs")
    ap.add_argument("--add-only", action="store_true", help="tune nothing: run the step (the online autotuner picks tiles for unseen shapes at first use) and, with "
                                                            "--write, add the keys the table does not know yet -- no existing entry can move")
    ap.add_argument("--width", type=int, default=352, help="single image width (the stage-2 canvas is twice this): 352 = the metric's, 512 = the driver's default")
    ap.add_argument("--stage3", action="store_true", help="the stage-3 refine UNet (in_channels 8, no class / pose embedding) on one image of --width: "
                                                          "N = --batch samples, as stage3_batchtest_refined_model.py runs it")
    args = ap.parse_args()
    from oracle.pipeline import synth_inputs
    from oracle.unet import UNetConfig, synth_state_dict
    from pcdms_amd import ops
    from pcdms_amd.pipeline import Stage2_InpaintDiffusionPipeline
    from pcdms_amd.schedulers import DDIMScheduler
    from pcdms_amd.pipeline import Stage3_RefinedDiffusionPipeline
    from pcdms_amd.unet import Stage2_InapintUNet2DConditionModel, UNet2DConditionModel
    from tests.test_unet import _kwargs
    dev = torch.device("cuda:0")
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                          set_alpha_to_one=False, steps_offset=1)
    N = args.batch
    if args.stage3:
        cfg = UNetConfig(in_channels=8, class_embed_type=None, projection_class_embeddings_input_dim=None)
        m = UNet2DConditionModel(**_kwargs(cfg))
        m.load_state_dict(synth_state_dict(cfg, seed=0))
        m.to(dev)
        pipe = Stage3_RefinedDiffusionPipeline(m, sched)
        h, w = 64, args.width // 8
        g = torch.Generator().manual_seed(3)
        lat0 = torch.randn(N, 4, h, w, generator=g).to(dev)
        pipe(height=h * 8, width=w * 8, num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=20, output_type="latent", use_graph=False,
             gen_t_img_latents=(torch.randn(1, 4, h, w, generator=g) * 0.9).to(dev), s_img_proj_f=torch.randn(1, 257, 1024, generator=g).to(dev), latents=lat0)
    else:
        cfg = UNetConfig()
        m = Stage2_InapintUNet2DConditionModel(**_kwargs(cfg))
        m.load_state_dict(synth_state_dict(cfg, seed=0))
        m.to(dev)
        pipe = Stage2_InpaintDiffusionPipeline(m, sched)
        h, w = 64, 2 * args.width // 8
        inp = {k: v.to(dev) for k, v in synth_inputs(cfg, h, w, N).items()}
        pipe(height=h * 8, width=w * 8, num_images_per_prompt=N, guidance_scale=2.0, num_inference_steps=50, output_type="latent", use_graph=False, **inp)
        lat0 = inp["latents"].clone()
    st = pipe._st
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); e1.synchronize()
    cyc_per_s = 20_000_000 / (e0.elapsed_time(e1) * 1e-3)

    def one_step():
        st["lat"].copy_(lat0)
        st["step"].zero_()
        torch.cuda.synchronize()
        torch.cuda._sleep(int(cyc_per_s * 0.03))          # the host enqueues the whole step behind a spin kernel: no idle gaps inside the brackets
        ops.LAUNCH_LOG, ops.LAUNCH_KEYS, ops.LAUNCH_SPANS = [], [], []
        try:
            pipe._step_eager(st)
            torch.cuda.synchronize()
            log, keys, spans = ops.LAUNCH_LOG, ops.LAUNCH_KEYS, ops.LAUNCH_SPANS
        finally:
            ops.LAUNCH_LOG, ops.LAUNCH_KEYS, ops.LAUNCH_SPANS = None, None, None
        one_step.spans = spans
        return log, keys, st["lat"].clone()

    def measure(reps):
        """per-launch minimum over `reps` steps -> (ms per launch list, names, keys map, output latents)"""
        runs = [one_step() for _ in range(reps)]
        n = len(runs[0][0])
        assert all(len(r[0]) == n for r in runs)
        ms = [min(r[0][i][2].elapsed_time(r[0][i][3]) for r in runs) for i in range(n)]
        names = [runs[0][0][i][0] for i in range(n)]
        return ms, names, runs[0][1], runs[0][2]

    ms0, names0, keys0, lat_ref = measure(args.reps)
    by_key = defaultdict(list)
    for idx, key in keys0:
        by_key[key].append(idx)
    ln_keys = sorted({sp[0] for sp in one_step.spans}, key=str)
    plain = [k for k in by_key if k[0] != "ln"]
    if args.only:
        pref = tuple(int(x) for x in args.only.split(","))
        plain = [k for k in plain if k[: len(pref)] == pref]
    total0 = sum(ms0)
    print(f"baseline: {len(ms0)} timed launches, {total0:.3f} ms; {len(plain)} plain keys to tune (reps {args.reps})", flush=True)
    # keys whose launches are asked for LayerNorm partials keep a producer tile: find them by a dry run that records ops.gemm's row_stats argument
    producer_keys = set()
    real_gemm = ops.gemm

    def spy2(*a, **kw):   # (ops.gemm computes the key itself: the key logged by this very call)
        want = kw.get("row_stats") is not None and kw.get("ln") is None
        before = len(ops.LAUNCH_KEYS) if ops.LAUNCH_KEYS is not None else 0
        out = real_gemm(*a, **kw)
        if want and ops.LAUNCH_KEYS is not None and len(ops.LAUNCH_KEYS) > before:
            producer_keys.add(ops.LAUNCH_KEYS[before][1])
        return out
    ops.gemm = spy2
    try:
        one_step()
    finally:
        ops.gemm = real_gemm
    print(f"{len(producer_keys)} keys are producers of LayerNorm partials (tiles restricted to {ops.STATS_TILES})", flush=True)

    def gn_total(ms, names):
        return sum(t for t, nm in zip(ms, names) if nm == "groupnorm")

    report, changed = [], {}
    t_start = time.time()
    for key in ([] if (args.skip_plain or args.add_only) else sorted(plain, key=lambda k: -sum(ms0[i] for i in by_key[k]))):
        M, Npad, K, conv, stride, ups, epi = key[:7]
        cur = ops._TUNED.get(key)
        if cur is None:
            continue
        nkt = K // 64
        cands = []
        for tile, (bm, bn) in ops.TILE_SHAPES.items():
            if Npad % bn or (tile in ops.ROWGEMM_TILES and (K != 320 or conv)):
                continue
            if key in producer_keys and tile not in ops.STATS_TILES:
                continue
            ntiles = -(-M // bm) * (Npad // bn)
            splits = [1]
            if epi == ops.EPI_STORE and tile not in ops.ROWGEMM_TILES and len(key) == 9:      # (flagged variants -- zero_rows / dup_rows -- never split)
                splits += [s for s in (2, 3, 4, 6, 8, 12, 16) if ntiles * s <= 1024 and nkt // s >= 4 and ntiles < 512]
            cands += [(tile, s) for s in splits]
        idxs = by_key[key]
        any_split = cur[1] > 1 or any(s > 1 for _, s in cands)

        def objective(ms, names):
            return sum(ms[i] for i in idxs) + (gn_total(ms, names) if any_split else 0.0)
        ms_c, names_c, _, _ = measure(args.reps)          # the current entry, re-measured now (drift)
        best, best_t = tuple(cur), objective(ms_c, names_c)
        t_cur = best_t
        own = sum(ms_c[i] for i in idxs)                  # the key's own launches: what a gain is judged against (the GroupNorm total rides along as a constant)
        need = max(args.min_gain * own, 0.002)            # ms: >= min-gain of the key's time and >= 2 us (the noise of a sum of event brackets)
        tried = 0
        for cand in cands:
            if cand == tuple(cur):
                continue
            ops._TUNED[key] = cand
            try:
                ms_, names_, keys_, lat = measure(1)
            except RuntimeError:
                continue                                  # (the library refuses this tile for this problem)
            finally:
                ops._TUNED[key] = tuple(cur)
            if len(ms_) != len(ms0) and not any_split:
                continue
            t1 = objective(ms_, names_) if len(ms_) == len(ms0) else float("inf")
            tried += 1
            if t1 < best_t - 0.5 * need:                       # promising on one step: confirm with the full repetitions
                ops._TUNED[key] = cand
                try:
                    ms_, names_, _, lat = measure(args.reps)
                finally:
                    ops._TUNED[key] = tuple(cur)
                ok = bool(torch.isfinite(lat).all()) and float((lat - lat_ref).norm() / lat_ref.norm()) < 2e-3
                t2 = objective(ms_, names_) if len(ms_) == len(ms0) else float("inf")
                if ok and t2 < best_t:
                    best, best_t = cand, t2
        gain = t_cur - best_t                             # ms per step
        line = (f"{','.join(str(x) for x in key):48s} n={len(idxs):2d}  current {cur[0]}/{cur[1]} own {own * 1e3:8.1f} us -> best {best[0]}/{best[1]} "
                f"{-gain * 1e3:+7.1f} us per step ({-100 * gain / own:+5.1f} % of its own time)  [{tried} candidates]" + ("  (+GroupNorm total in the objective)" if any_split else ""))
        if best != tuple(cur) and gain >= need:
            ops._TUNED[key] = best
            changed[",".join(str(x) for x in key)] = list(best)
            line += "  CHANGED"
        print(line, flush=True)
        report.append(line)
    # ---- the LayerNorm -> Linear pairs ("ln" keys: (tile, mode); (0, 1) = LayerNorm launch + plain GEMM).  A pair is one launch or two depending on
    # the candidate: its launches are the LAUNCH_LOG span ops._gemm_ln recorded for it.  Mode 2 makes the PRODUCER of the rows write partials, so the
    # producers' launches ride along in the objective.  (A first version judged these keys by a step-wide total: its "gains" were drift, and the
    # table it produced lost 0.4 % end to end -- profiles/r6_tune_in_step_v2_ln_noise.txt.)
    def ln_objective(key, reps):
        vals = []
        for _ in range(reps):
            log, keys, lat = one_step()
            t = sum(log[i][2].elapsed_time(log[i][3]) for sp in one_step.spans if sp[0] == key for i in range(sp[1], sp[2]))
            t += sum(log[i][2].elapsed_time(log[i][3]) for i, k in keys if k in producer_keys)
            vals.append(t)
        return min(vals), lat
    for key in ([] if (args.only or args.add_only) else ln_keys):
        _, M, Npad, K, epi = key
        cur = ops._TUNED.get(key)
        if cur is None:
            continue
        cur = tuple(cur) if len(cur) > 1 else (cur[0], 1)
        cands = [(0, 1)]
        for t in (ops.ROWGEMM_TILES if K == 320 else ()) + ops.LN_TILED_TILES + ops.LN_PARTIALS_TILES:
            if Npad % ops.TILE_SHAPES[t][1]:
                continue
            for md in (1, 2):
                if (md == 2 and (t in ops.ROWGEMM_TILES or K == 320)) or (md == 1 and t in ops.LN_PARTIALS_TILES):
                    continue
                cands.append((t, md))
        t_cur, _ = ln_objective(key, args.reps)
        own = t_cur - sum(ms0[i] for i, k in keys0 if k in producer_keys)
        best, best_t = cur, t_cur
        tried = 0
        for cand in cands:
            if cand == cur:
                continue
            ops._TUNED[key] = cand
            try:
                t1, lat = ln_objective(key, args.reps)
            except RuntimeError:
                continue
            finally:
                ops._TUNED[key] = cur
            tried += 1
            ok = bool(torch.isfinite(lat).all()) and float((lat - lat_ref).norm() / lat_ref.norm()) < 2e-3
            if ok and t1 < best_t:
                best, best_t = cand, t1
        gain = t_cur - best_t
        line = (f"{','.join(str(x) for x in key):48s} current {cur[0]}/mode {cur[1]} (its launches ~{own * 1e3:7.1f} us per step) -> best {best[0]}/mode {best[1]}  "
                f"{-gain * 1e3:+7.1f} us per step  [{tried} candidates]")
        if best != cur and gain >= max(args.min_gain * max(own, 0.0), 0.003):
            # confirm once more against the current entry, back to back
            ops._TUNED[key] = best
            try:
                t2, _ = ln_objective(key, args.reps)
            finally:
                ops._TUNED[key] = cur
            t3, _ = ln_objective(key, args.reps)
            if t3 - t2 >= max(args.min_gain * max(own, 0.0), 0.003):
                ops._TUNED[key] = best
                changed[",".join(str(x) for x in key)] = list(best)
                line += f"  CHANGED (confirmed {-(t3 - t2) * 1e3:+.1f} us)"
            else:
                line += f"  (not confirmed: {-(t3 - t2) * 1e3:+.1f} us)"
        print(line, flush=True)
        report.append(line)
    ms1, names1, _, lat1 = measure(args.reps)
    print(f"in-step total of the timed launches: {total0:.3f} -> {sum(ms1):.3f} ms ({len(changed)} entries changed) in {time.time() - t_start:.0f} s; "
          f"output rel diff {float((lat1 - lat_ref).norm() / lat_ref.norm()):.2e}", flush=True)
    Path(args.out).parent.mkdir(exist_ok=True)
    Path(args.out).write_text(json.dumps({"changed": changed, "total_ms_before": total0, "total_ms_after": sum(ms1), "report": report}, indent=1))
    # keys of this step the committed table does not know yet (the online autotuner chose them at first use; the passes above kept or changed them)
    tab_now = json.loads(Path(ops.TUNING_FILE).read_text())["gemm"] if Path(ops.TUNING_FILE).exists() else {}
    added = {}
    for key in list(by_key) + list(ln_keys):
        ks = ",".join(str(x) for x in key)
        if ks not in tab_now and ks not in changed and key in ops._TUNED:
            added[ks] = list(ops._TUNED[key])
    if added:
        print(f"{len(added)} keys of this step are new to the table:", ", ".join(f"{k} -> {v[0]}/{v[1]}" for k, v in sorted(added.items())), flush=True)
    if args.write and (changed or added):
        tab = json.loads(Path(ops.TUNING_FILE).read_text())
        tab["gemm"].update(added)
        tab["gemm"].update(changed)
        tab["note"] = (tab.get("note", "") + " | round 6: entries re-tuned IN THE STEP by tools/tune_in_step.py").strip()
        Path(ops.TUNING_FILE).write_text(json.dumps(tab, indent=0))
        print("wrote", ops.TUNING_FILE)


if __name__ == "__main__":
    main()
