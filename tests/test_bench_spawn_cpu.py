"""bench.py's multi-rank plumbing on CPU (gloo, world_size 2): `python bench.py --gpus 2` with no torchrun environment
must re-launch itself under torch.distributed.run (one process per rank on 127.0.0.1), every rank must take its shard
of the global batch, and rank 0 alone prints the JSON line.  `--selftest-spawn` swaps the GPU work for a fake timing so
that the launch path - the part the driver's `python bench.py --gpus N` depends on - runs without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-spawn"] + extra, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    got = _run(["--gpus", "2", "--batch", "5"])
    assert got == {"selftest": "spawn", "n_gpus": 2, "clips_total": 5, "max_time": 2.0, "rank0_rows": [0, 1, 2]}


def test_single_rank_needs_no_launcher():
    got = _run(["--gpus", "1", "--batch", "3"])
    assert got["n_gpus"] == 1 and got["rank0_rows"] == [0, 1, 2]


def test_rooflines_pick_the_dominant_dit_gemm_symbol():
    sys.path.insert(0, ROOT)
    import bench
    stats = [
        dict(name="dit/gemm8_bf16_256x256_8phase", launches=10, flops=1.0e13, bytes=1e9, ms=10.0),
        dict(name="dit/gemm5_bf16_256x128_ld_s3_pf_persist", launches=30, flops=0.5e13, bytes=2e9, ms=8.0),
        dict(name="codec/gemm5_bf16_256x128_ld_s3_pf_persist", launches=5, flops=2.0e12, bytes=4e10, ms=20.0),
        dict(name="dit/rmsnorm_mod", launches=4, flops=0.0, bytes=8e9, ms=2.0),
    ]
    r = bench.rooflines(stats)
    assert r["roofline"]["kernel"] == "dit/gemm8_bf16_256x256_8phase"
    assert abs(r["roofline"]["achieved"] - 1000.0) < 1e-6 and abs(r["roofline"]["frac"] - 0.4) < 1e-9
    assert abs(r["roofline"]["dit_gemm_all"]["achieved"] - 1.5e13 / 18e-3 / 1e12) < 0.01
    codec = r["roofline_hbm"][0]
    assert codec["kernel"].startswith("codec/") and codec["bound"] == "hbm" and abs(codec["achieved"] - 2000.0) < 1e-6
    rms = [g for g in r["roofline_hbm"] if g["kernel"] == "dit/rmsnorm_mod"][0]
    assert abs(rms["achieved"] - 4000.0) < 1e-6 and abs(rms["frac"] - 0.5) < 1e-9
