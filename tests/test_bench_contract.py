"""bench.py output contract: the committed bench lines (profiles/r1_bench.json, profiles/r2_bench.json, produced on an MI355X by
`python bench.py`) carry every field the driver parses, and bench.py's flags / defaults are the ones the driver passes."""
from __future__ import annotations

import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


import pytest


@pytest.mark.parametrize("name", ["r1_bench.json", "r2_bench.json", "r2_bench_fast_box.json", "r3_bench.json", "r3_bench_fast_box.json", "r3_bench_final_sources.json", "r3_bench_slow_box.json",
                                  "r4_bench.json", "r4_bench_with_traffic.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    line = (ROOT / "profiles" / name).read_text().strip().splitlines()[-1]
    d = json.loads(line)
    base = json.loads((ROOT / "BASELINE.json").read_text())
    assert d["metric"].startswith("images/sec") and d["unit"] == "images/s" and base["metric"].startswith("images/sec")
    for k, t in (("value", float), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float)):
        assert isinstance(d[k], t) and d[k] > 0, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "352x512" in cfg["workload"] and "batch=4" in cfg["workload"] and "50 DDIM" in cfg["workload"] and "model" not in cfg
    # value is whole-job throughput: images = n_gpus * batch * steps over the timed region
    assert abs(d["value"] - d["n_gpus"] * cfg["global_batch"] / d["n_gpus"] * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    # achieved = algorithmic FLOPs per launch / average launch duration
    assert abs(r["achieved"] - r["alg_gflop_per_launch"] / r["avg_launch_us"] * 1e3)   # GFLOP / us = 1000 TFLOP/s < 0.02 * r["achieved"]
    if name == "r4_bench_with_traffic.json":   # (the line that quotes the traffic passes of its session: run with --no-cpu-baseline)
        assert r["traffic"] and r["traffic"] > 1.5 * 48.6e6 and "cpu_baseline" not in d
        return
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["unit"] == "images/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    if name.startswith("r4"):   # round 4: CPU baseline from the pinned child process (thread counts of the host, NUMA map recorded)
        assert 0 < r["e2e_frac_executed"] < r["e2e_frac"] < 1 and c["config1"]["seconds"] > 0
        assert str(c["cores"]) in c["per_thread_count"] and c["host"]["numa_nodes"] and "pinned" in c["sample"]
        assert all(v["tflops_fp32"] > 0 for v in c["per_thread_count"].values())
        return
    if name.startswith(("r2", "r3")):   # round 2: whole-path fraction, the RCCL world size, the full configs[0] CPU run
        assert 0 < r["e2e_frac"] < 1 and cfg["rccl_world_size"] == d["n_gpus"] and c["config1"]["seconds"] > 0
    if name.startswith("r3"):   # round 3: executed-FLOP fraction, CPU baseline at physical-core thread counts, traffic only with a source stamp
        assert 0 < r["e2e_frac_executed"] < r["e2e_frac"] and "process_group" in cfg
        assert c["cores"] in (c["host"]["cores_per_socket"], c["host"]["cores_per_socket"] // 2) and str(c["cores"]) in c["per_thread_count"]
        assert r["traffic"] is None or "kernel_src_sha256" in json.loads((ROOT / "profiles" / "gemm_traffic.json").read_text())


def test_bench_flags_and_defaults():
    src = (ROOT / "bench.py").read_text()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src, flag
    assert re.search(r'"--gpus", type=int, default=1', src)
    for env in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        assert env in src
    assert "dist.barrier()" in src and "torch.cuda.synchronize()" in src and "ReduceOp.MAX" in src
    # the oracle is only the cpu_baseline leg / synthetic data generator, never the measured path
    # (round 4: the timing runs in a pinned child process, bench.py --cpu-baseline-worker)
    worker = src.split("def cpu_baseline_worker")[1].split("\ndef ")[0]
    assert "unet_forward" in worker and "from oracle.unet import" in worker
    timed = src.split("def timed(")[1].split("elapsed = timed(")[0]
    assert "oracle" not in timed


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 8` from a bare shell (no WORLD_SIZE): bench.py becomes the launcher -- one
    torch.distributed.run rank per GPU on 127.0.0.1 with the same flags."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _run_bench(args, timeout=900):
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PCDM_BENCH_FORCE_DIST")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


_EMU_FLAGS = ["--backend", "gloo", "--emu", "--tiny", "--batch", "1", "--height", "64", "--width", "32", "--ddim-steps", "1", "--steps", "1", "--warmup", "0",
              "--no-cpu-baseline", "--no-roofline", "--no-vae"]


def test_bench_world2_through_its_own_launcher_on_the_emulator(tmp_path):
    """The code that will run on the 8-GPU node, executed with MORE THAN ONE RANK before the driver does: a bare
    ``python bench.py --gpus 2`` becomes the launcher (no monkeypatch: real ``torch.distributed.run`` children on 127.0.0.1), every rank draws
    its replicated weights with ``device_state_dict``, samples its own pair, the final latents are gathered, the timed region is bracketed by
    barriers + a MAX all-reduce, and -- with ``--configs2-world 2`` -- the branch that times BASELINE configs[2]'s own per-GPU batch runs too.
    Here the process group is gloo and the kernels are the lane-emulator build of the same sources (tiny UNet, one DDIM step): a plumbing
    test, no measurement.  Checked: ONE JSON line; world size 2 in it; both ranks hold identical weights; the gathered tensor is identical
    on both ranks, and its r-th slice equals a SINGLE-process run of the same file with pair seed 1000 + r (same device-drawn weights)."""
    import torch
    d2 = tmp_path / "w2"
    lines = _run_bench(["--gpus", "2", *_EMU_FLAGS, "--configs2-world", "2", "--configs2-batch", "2", "--dump", str(d2)])
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_world_size"] == 2 and d["config"]["process_group"] == "gloo" and d["config"]["global_batch"] == 2
    assert d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak" and d["value"] > 0 and "EMULATOR" in d["data"]
    c2 = d["config"]["configs2"]
    assert c2["value"] > 0 and "batch=4 (2 per GPU" in c2["workload"] and "dp2" in c2["workload"]
    r0, r1 = torch.load(d2 / "rank0.pt"), torch.load(d2 / "rank1.pt")
    assert r0["world"] == r1["world"] == 2 and r0["sd_hash"] == r1["sd_hash"]
    assert torch.equal(r0["gathered"], r1["gathered"]) and r0["gathered"].shape[0] == 2
    assert torch.equal(r0["gathered"][0:1], r0["lat"]) and torch.equal(r0["gathered"][1:2], r1["lat"])
    assert not torch.equal(r0["lat"], r1["lat"])                      # two different pairs
    g0, g1 = torch.load(d2 / "rank0_configs2.pt"), torch.load(d2 / "rank1_configs2.pt")
    assert torch.equal(g0["gathered"], g1["gathered"]) and g0["gathered"].shape[0] == 4 and torch.equal(g0["gathered"][2:4], g1["lat"])
    # single-process runs of the same file, one per pair seed
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(2) as ex:
        futs = [ex.submit(_run_bench, ["--gpus", "1", *_EMU_FLAGS, "--device-weights", "--pair-seed", str(1000 + r), "--dump", str(tmp_path / f"s{r}")])
                for r in range(2)]
        for f in futs:
            assert len(f.result()) == 1
    for r in range(2):
        s = torch.load(tmp_path / f"s{r}" / "rank0.pt")
        assert s["sd_hash"] == r0["sd_hash"] and s["world"] == 1 and s["gathered"] is None
        assert torch.equal(s["lat"], r0["gathered"][r:r + 1]), r


def test_bench_refuses_the_emulator_without_gloo_and_a_world_mismatch():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--emu"], capture_output=True, text=True, timeout=120, cwd=str(ROOT))
    assert r.returncode != 0 and "go together" in (r.stderr + r.stdout)


def test_entry_points_exist():
    src = (ROOT / "__graft_entry__.py").read_text()
    assert "def build(" in src and "def smoke(" in src and "gfx950" in (ROOT / "pcdms_amd" / "build.py").read_text()


@pytest.mark.gpu
def test_bench_runs_with_a_one_rank_rccl_group():
    """`PCDM_BENCH_FORCE_DIST=1 python bench.py --gpus 1`: the code path the driver's N > 1 runs take -- RCCL process group on
    127.0.0.1, hipGraph capture with the group alive, barrier + MAX all-reduce around the timed region, the rank gather -- executed on
    the one GPU a test box has.  Short schedule (10 DDIM steps): this checks the plumbing, not the number."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PCDM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--ddim-steps", "10",
                        "--no-cpu-baseline", "--no-roofline", "--no-vae"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["value"] > 0
    assert d["config"]["rccl_world_size"] == 1 and d["config"]["process_group"] == "nccl" and d["config"]["hipgraph"] is True


def test_cpu_baseline_thread_counts_follow_the_cgroup_quota(tmp_path):
    """The host leg of bench.py: the container's CPU-bandwidth quota (cgroup v2 ``cpu.max`` / v1 ``cpu.cfs_quota_us``) is read and bounds
    the thread-count sweep -- the GPU boxes show 256 CPUs and grant 16 cores, which is why 64 threads measured slower than 32."""
    import bench
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert bench.cgroup_cpu_quota(tmp_path) == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench.cgroup_cpu_quota(tmp_path) is None
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cgroup_cpu_quota(v1) is None
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("350000\n")
    assert bench.cgroup_cpu_quota(v1) == 3.5
    assert bench.cgroup_cpu_quota(tmp_path / "missing") is None
    topo = {"physical_cores": 128, "cgroup_cpu_quota_cores": 16.0}
    assert bench.baseline_thread_counts(topo) == [8, 16, 32]
    assert bench.baseline_thread_counts({"physical_cores": 128, "cgroup_cpu_quota_cores": None}) == [8, 16, 32, 64]
    assert bench.baseline_thread_counts({"physical_cores": 8, "cgroup_cpu_quota_cores": None}) == [8]
    assert bench.baseline_thread_counts({"physical_cores": 64, "cgroup_cpu_quota_cores": 3.5}) == [3]
    assert bench._parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
