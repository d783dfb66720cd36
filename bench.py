#!/usr/bin/env python
"""bench.py - throughput of the SAMAudio.separate() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one `model.separate(batch)` over one batch of synthetic 10 s / 48 kHz mono clips that is already
resident in HBM: DAC-VAE encode -> 32 DiT evaluations of the 16-step midpoint ODE -> DAC-VAE decode of target
and residual -> unbatch (reference sam_audio/model/model.py:247-338).  Metric (BASELINE.json):
seconds of audio separated per second per node = N * B * 10 s * K / wall.

Workload: BASELINE.json configs[2] - the configuration the metric is quoted on ("sam-audio-large bf16,
batch=32x10 s clips, text prompt"); it fits one GPU.  The checkpoint's real config.json is not reachable
offline, so the dims are the labelled stand-in `large*` (D=2816, H=22, L=22, F=7552);
weights are seeded random, text features are synthetic T5-shaped tensors (no tokenizer offline).  Each rank
processes its own batch of B clips (the path shards over clips with no data-path collective: "weak").
Rank 0 creates the weights and broadcasts them over RCCL/xGMI before the timed region.

The JSON line also carries
  roofline     - bf16 MFMA roofline of the dominant kernel (the 128x128-tile GEMM): algorithmic flops of its
                 launches / their HIP-event time, measured on the launch stream in one extra instrumented step;
  cpu_baseline - the CPU oracle (oracle/samaudio_oracle.py, a torch fp32 restatement of the reference
                 algorithm) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CLIP_SECONDS = 10.0
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", default="large*", help="dims preset (sam_audio_amd.config.SIZE_PRESETS)")
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step")
    ap.add_argument("--text-len", type=int, default=8)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--streams", type=int, default=1, help="row groups of the batch solved concurrently on separate HIP streams")
    ap.add_argument("--candidates", type=int, default=1,
                    help="> 1: BASELINE.json configs[3] - reranking_candidates per clip, scored by the HIP Judge "
                         "(pe-av-large stand-in dims, random weights); the default bench line stays configs[2]")
    return ap.parse_args()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """Host cores this process may really use: min(affinity mask, cgroup v2/v1 CPU quota).  os.cpu_count()
    reports the machine (256 on the MI355X box) even when the container is capped at a few cores, and 256
    OpenMP threads on a 16-core quota spin against each other instead of computing."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg, sd_cpu, clip, text, tmask, noise, threads):
    """The oracle on the host cores, bounded: one 10 s clip; DAC encode, ONE of the 16 midpoint steps (2 of the
    32 DiT evaluations) and DAC decode are each timed once; ODE time is scaled x16."""
    from oracle import samaudio_oracle as O
    torch.set_num_threads(threads)
    codec = cfg.audio_codec
    with torch.inference_mode():
        t0 = time.perf_counter()
        z = O.dac_encode(sd_cpu, codec, clip).transpose(1, 2)
        t_enc = time.perf_counter() - t0
        log(f"cpu baseline: DAC encode {t_enc:.2f} s on {threads} threads")
        feats = torch.cat([z, z], dim=2)
        T = feats.shape[1]
        pad = torch.ones(1, T, dtype=torch.bool)
        ids, align = O.anchors_to_ids(None, pad, codec.hop_length, codec.sample_rate)
        video = feats.new_zeros(1, cfg.vision_encoder.dim, T)

        def field(t, y):
            return O.samaudio_forward(sd_cpu, cfg, y, feats, text, t.expand(1), video=video, text_mask=tmask,
                                      anchor_ids=ids, anchor_alignment=align, pad_mask=pad)

        t0 = time.perf_counter()
        lat = O.ode_fixed_grid(field, noise, method="midpoint", step_size=1.0)  # one step = 2 evaluations
        t_step = time.perf_counter() - t0
        log(f"cpu baseline: one midpoint step {t_step:.2f} s")
        gen = lat.transpose(1, 2).reshape(2, lat.shape[2] // 2, T)
        t0 = time.perf_counter()
        O.dac_decode(sd_cpu, codec, gen)
        t_dec = time.perf_counter() - t0
        log(f"cpu baseline: DAC decode x2 {t_dec:.2f} s")
    per_clip = t_enc + 16 * t_step + t_dec
    return {
        "value": CLIP_SECONDS / per_clip, "unit": "s-audio/s", "cores": threads, "kind": "port",
        "sample": (f"1 clip x 10 s, same dims, fp32 torch oracle: DAC encode {t_enc:.2f} s + 1 of 16 midpoint steps "
                   f"(2 of 32 DiT evals) {t_step:.2f} s (scaled x16) + DAC decode x2 {t_dec:.2f} s "
                   f"=> {per_clip:.1f} s per clip"),
    }


class _HashTokenizer:
    """Stand-in for the Judge's ModernBERT tokenizer (no tokenizer files offline): word hashes -> ids, pad-to-longest."""

    def __call__(self, text, return_tensors="pt", padding="longest", max_length=512, truncation=True):
        rows = [[1] + [3 + (hash(w) % 30000) for w in t.split()][: max_length - 1] for t in text]
        width = max(len(r) for r in rows)
        ids = torch.zeros(len(rows), width, dtype=torch.long)
        att = torch.zeros(len(rows), width, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r)
            att[i, : len(r)] = 1
        return {"input_ids": ids, "attention_mask": att}


def build_judge_ranker(cfg, precision, dev):
    """SAMAudioJudgeModel with the pe-av-large defaults for both PE-AV transformers (the judge checkpoint's real
    config.json is not reachable offline), seeded random weights, the SAMAudio codec dims, ModernBERT-base random init."""
    from sam_audio_amd.config import SAMAudioJudgeConfig
    from sam_audio_amd.judge import SAMAudioJudgeModel
    from sam_audio_amd.processor import SAMAudioJudgeProcessor
    from sam_audio_amd.ranking import JudgeRanker
    from sam_audio_amd.synthetic import init_judge_state_dict
    jcfg = SAMAudioJudgeConfig(audio_codec=vars(cfg.audio_codec))  # text tower: ModernBertConfig defaults (base)
    judge = SAMAudioJudgeModel(jcfg, precision=precision, device=str(dev))
    judge.load_state_dict(init_judge_state_dict(jcfg, seed=1, device=dev), strict=False)
    proc = SAMAudioJudgeProcessor(jcfg.audio_codec.hop_length, jcfg.audio_codec.sample_rate, tokenizer=_HashTokenizer())
    return JudgeRanker(model=judge, processor=proc)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (the hot path has no CPU fallback)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
    from sam_audio_amd.dist import broadcast_state_dict
    from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_text_features

    cfg = preset_config(args.size)
    tcfg = cfg.transformer
    B = args.batch

    # ---- weights: rank 0 creates them, RCCL broadcast over xGMI to the other ranks -----------------------
    log(f"world {world}, preset {args.size}, {B} clips per GPU, usable host cores {usable_cores()}")
    sd = init_state_dict(cfg, seed=0, device=dev) if rank == 0 else None
    sd = broadcast_state_dict(sd, src=0, device=dev)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    sd_cpu = {k: v.cpu() for k, v in sd.items()} if want_cpu else None
    model = SAMAudio(cfg, precision=args.precision, device=str(dev), streams=args.streams)
    model.load_state_dict(sd, strict=False)
    del sd
    torch.cuda.empty_cache()
    log("weights loaded")

    # ---- inputs: this rank's B synthetic clips, resident in HBM before the timed region ------------------
    n_samples = int(CLIP_SECONDS * cfg.audio_codec.sample_rate)
    clips = [synthetic_clip(rank * B + i, n_samples) for i in range(B)]
    text, tmask = synthetic_text_features(B, args.text_len, seed=7 + rank)
    proc = SAMAudioProcessor.from_config(cfg)
    batch = proc(descriptions=["sound"] * B, audios=clips, text_features=text, text_mask=tmask).to(dev)
    os.environ.setdefault("SAMAUDIO_CODEC_CHUNK", "32")

    if args.candidates > 1:
        model.text_ranker = build_judge_ranker(cfg, args.precision, dev)

    def step():
        # noise=None: drawn on device inside, like the reference (model.py:274-275)
        return model.separate(batch, reranking_candidates=args.candidates)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("inputs resident; warm-up")
    for i in range(args.warmup):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {time.perf_counter() - t0:.3f} s")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(torch.isfinite(w).all() for w in res.target), "non-finite output"
    value = world * B * CLIP_SECONDS * args.steps / elapsed
    log(f"timed {args.steps} steps in {elapsed:.3f} s -> {value:.2f} s-audio/s")

    # ---- roofline of the dominant kernel: one extra, instrumented step (HIP events on the launch stream) --
    roofline = None
    if rank == 0 and not args.no_roofline:
        model.streams = 1  # events bracket single launches: keep the GPU to one stream while they are recorded
        model.profile_begin()
        step()
        stats = model.profile_end()
        model.streams = args.streams
        dom = max(stats, key=lambda k: k["ms"])
        if dom["ms"] > 0:
            achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            roofline = {
                "bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
                "launches_per_step": dom["launches"], "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
                "flops_per_step": dom["flops"],
                "other_gemm_variants": [
                    {"kernel": k["name"], "launches": k["launches"], "ms": round(k["ms"], 3),
                     "tflops": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2) if k["ms"] > 0 else None}
                    for k in stats if k is not dom],
                "gemm_ms_per_step": round(sum(k["ms"] for k in stats), 2),
            }

    # HBM traffic of the dominant kernel: PMC numbers come from separate profiled runs (tools/profile_bench.sh ->
    # tools/pmc_traffic.py -> profiles/r1_traffic.json); they cannot be collected inside a timed run.
    if roofline is not None:
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))["kernels"]
            key = {"gemm2_bf16_256x128_s2": "gemm2_kernel<256, 128, 4, 2, 2, 64, 0>",
                   "gemm3_bf16_256x256_pp2": "gemm3_kernel<256, 256, 2, 4, 2, 2, 0>"}.get(roofline["kernel"], "")
            hit = [v for k, v in traffic.items() if key and key in k]
            if hit:
                roofline["traffic"] = round(hit[0]["traffic_bytes_per_launch"])
                roofline["traffic_note"] = "HBM bytes per launch, rocprofv3 PMC passes (profiles/r1_traffic.json)"
        except (OSError, KeyError, ValueError):
            pass

    cpu = None
    if want_cpu:
        threads = args.cpu_threads or usable_cores()
        g = torch.Generator().manual_seed(99)
        noise = torch.randn(1, n_samples // cfg.audio_codec.hop_length, tcfg.out_channels, generator=g)
        cpu = cpu_baseline(cfg, sd_cpu, clips[0].unsqueeze(0), text[:1], tmask[:1], noise, threads)

    if rank == 0:
        line = {
            "metric": "seconds-of-audio separated/sec/node", "value": round(value, 3), "unit": "s-audio/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic (seeded random weights, synthetic 10 s/48 kHz clips, synthetic T5-shaped text features)",
            "config": {
                "workload": (f"sam-audio-{args.size} (stand-in dims D={tcfg.dim} H={tcfg.n_heads} L={tcfg.n_layers} "
                             f"F={tcfg.ffn_hidden}) {args.precision}, batch={B}x10 s clips per GPU, text prompt "
                             f"Lt={args.text_len}, midpoint ODE 16 steps = 32 DiT evals, DAC-VAE encode + decode x2"),
                "clips_per_gpu": B, "global_batch": B * world, "parallelism": f"clip-sharded x{world}",
                "streams_per_gpu": args.streams, "reranking_candidates": args.candidates,
            },
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
