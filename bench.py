#!/usr/bin/env python
"""bench.py - throughput of the SAMAudio.separate() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) and as a plain `python bench.py --gpus N`, which re-executes
itself under torch.distributed.run on 127.0.0.1 (one process per GPU, RCCL over xGMI).

One "step" = one `model.separate(batch)` over one batch of synthetic 10 s / 48 kHz mono clips that is already
resident in HBM: DAC-VAE encode -> 32 DiT evaluations of the 16-step midpoint ODE -> DAC-VAE decode of target
and residual -> unbatch (reference sam_audio/model/model.py:247-338).  Metric (BASELINE.json):
seconds of audio separated per second per node = (clips processed by all ranks) * 10 s * K / wall.

Workload: BASELINE.json configs[2] - the configuration the metric is quoted on ("sam-audio-large bf16,
batch=32x10 s clips, text prompt"); it fits one GPU.  The checkpoint's real config.json is not reachable
offline, so the dims are the labelled stand-in `large*` (D=2816, H=22, L=22, F=7552);
weights are seeded random.  The prompt goes through the T5 encoder INSIDE every timed step, as in the reference
(model.py:208-210, text_encoder.py:19-37): a t5-base-shaped stack on the HIP library (random init - no checkpoint offline - and
a hash tokenizer: no sentencepiece file offline; `--no-t5` = resident synthetic T5-shaped features instead).
Rank 0 creates the weights and broadcasts them over RCCL/xGMI before the timed region; the steady state has no collective.

Scaling.  BASELINE configs[2] is ONE batch of 32 clips sharded over the GPUs ("batch=32x10 s clips ... 1->8 MI355X
batch-sharded"; north_star: >= 7.5x at 8 vs 1 GPU on that batch), so with N > 1 the line's `value` is `--scaling strong`: the
global batch of --batch clips is split contiguously over the ranks (32 -> 8 x 4).  `--scaling weak`:
every rank processes its own batch of --batch clips.  With N > 1 the run also times the other mode and reports it under
"other_scaling", so one driver invocation per N yields both curves.  At N = 1 the two coincide.

The JSON line also carries
  roofline       - bf16 MFMA roofline of the dominant DiT GEMM kernel symbol: algorithmic flops of its launches / their
                   HIP-event time, measured on the launch stream in one extra instrumented step (single stream); plus the
                   aggregate over every DiT GEMM launch ("dit_gemm_all"), the whole-step figure ("whole_step": executed
                   flops of all kernels / the TIMED step, concurrent streams included) and the per-kernel table;
  roofline_hbm   - the DAC-VAE convolutions (their own kernel symbols) and the streaming kernels of the DiT against
                   max(flops / MFMA peak, algorithmic bytes / HBM peak);
  cpu_baseline   - the CPU oracle (oracle/samaudio_oracle.py, a torch fp32 restatement of the reference algorithm)
                   timed on this box's host cores on a bounded sample (2 clips, the whole path; rank 0, N=1 only);
  parity_check   - the same sample (2 clips, fixed noise, the full 16-step solve) run through the HIP path in the benchmarked
                   precision and compared with the oracle's result: encode latent, ODE latent and waveform max-abs error;
  hostile_check  - the same comparison at `small*` dims on TRAINED-LIKE weights (synthetic.make_hostile: residual-stream outlier
                   channels, O(3) adaLN tables, Snake alphas over two decades, gains on the DAC convolutions), for the headline
                   precision and for plain fp16: the headline is the fastest mode that holds 1e-3 on BOTH weight sets;
  fp16_mode, bf16_mode - the same steps timed with plain 16-bit GEMM operands (IEEE half / bfloat16 = BASELINE's nominal dtype), each
                   with its own parity_check: one MFMA product per multiply, 2.8x the headline's throughput, inside 1e-3 on the
                   benign seeded weights (fp16) but not on trained-like ones (6.6e-3 at best) - reported, not the headline.
                   The headline "fp16x3" keeps fp32 storage as the reference (README.md:48) and multiplies hi/lo-split IEEE-half
                   operands: 3 MFMA products per multiply, fp32-grade results;
  other_configs  - short lines of BASELINE configs[1], [3], [4] and of one GPU's share of configs[2] under strong scaling, each
                   a sub-process after the main measurement, with its own roofline and its OWN parity_check: configs[4] compares
                   the PE-Core tower's features and the visually conditioned solve, configs[3] the candidate solve, the Judge's
                   scores and the argmax (on 2.56 s clips, so that their oracle passes stay within the run's budget).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CLIP_SECONDS = 10.0
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0      # HBM3E peak, same table (6.29 TB/s measured achievable)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", default="large*", help="dims preset (sam_audio_amd.config.SIZE_PRESETS)")
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step (weak) / global batch (strong)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="default: strong for N > 1 (BASELINE configs[2] is ONE batch of 32 clips split over the GPUs: the "
                         "north_star's 8-vs-1 figure), weak for N = 1 (the same thing there)")
    ap.add_argument("--no-other-scaling", "--no-strong", dest="no_other", action="store_true",
                    help="N > 1: skip the extra measurement of the other scaling mode")
    ap.add_argument("--text-len", type=int, default=8)
    ap.add_argument("--precision", default="fp16x3", choices=["bf16", "fp16", "mixed", "fp32", "fp16x3", "bf16x3"],
                    help="precision of the timed model: fp16x3 (default) = fp32 storage, every big GEMM / convolution / attention "
                         "contraction on hi/lo-split IEEE-half operands (3 MFMA products per multiply, fp32-grade results): the "
                         "fastest mode that holds the 1e-3 parity bound on the benign AND the trained-like weights | fp16 = IEEE "
                         "half operands (one product per multiply; inside the bound on benign weights only) | mixed = bfloat16 on "
                         "the five big GEMM classes, fp16 elsewhere | bf16 everywhere (BASELINE's nominal dtype; outside the bound) "
                         "| fp32 (exact-fp32 MFMA) | bf16x3.  Plain fp16 and bf16 are timed side by side (--no-parity-mode skips it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip parity_check (it shares the oracle run with cpu_baseline)")
    ap.add_argument("--no-parity-mode", action="store_true",
                    help="skip the side-by-side timing of the other 16-bit mode (fp16 runs: \"mixed_mode\"; mixed / bf16 runs: "
                         "\"fp16_mode\")")
    ap.add_argument("--verify", action="store_true", help="run parity_check even with --no-cpu-baseline")
    ap.add_argument("--no-hostile", action="store_true", help="skip hostile_check (the main line of the default workload runs it)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--streams", type=int, default=0,
                    help="row groups of the batch solved concurrently on separate HIP streams (default 0 = auto: 2 with >= 16 "
                         "clips on the rank, else 1; with 2, one group's GEMM "
                         "tile tails - 352 tiles of 256x256 on 256 CUs at N = D - and epilogues are filled by the other's "
                         "workgroups; bitwise equal to 1 stream; 200.5 vs 181.1 s-audio/s, profiles/r2_call3/)")
    ap.add_argument("--serial-groups", action="store_true",
                    help="measurement aid: the row groups of --streams 2 run one after the other on one stream (the launches "
                         "of the timed configuration without their overlap) - the command whose rocprofv3 --kernel-trace "
                         "--stats summary the roofline's per-launch durations must agree with")
    ap.add_argument("--candidates", type=int, default=1,
                    help="> 1: BASELINE.json configs[3] - reranking_candidates per clip, scored by the HIP Judge "
                         "(pe-av-large stand-in dims, random weights); the default bench line stays configs[2]")
    ap.add_argument("--predict-spans", action="store_true",
                    help="configs[3]: run the PE-A-Frame span predictor first (random weights, stand-in dims)")
    ap.add_argument("--t5", dest="t5", action="store_true", default=True,
                    help="(default) row a3 inside the step, as the reference runs it: descriptions go through a t5-base-shaped "
                         "T5 encoder stack on the HIP library (weights of a random-init transformers T5EncoderModel, hash "
                         "tokenizer - no tokenizer files offline)")
    ap.add_argument("--no-t5", dest="t5", action="store_false",
                    help="resident synthetic T5-shaped text features instead of the T5 encoder inside the step")
    ap.add_argument("--visual", action="store_true",
                    help="BASELINE.json configs[4]: visual prompting - every clip comes with a 250-frame 336x336 uint8 video "
                         "(left half masked out), encoded by the PE-Core-L14-336 tower on the HIP library inside the step "
                         "(random weights); use with --batch 4")
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional check of the N-rank path on a box with ONE GPU: every rank uses cuda:0 and the process "
                         "group runs on gloo (RCCL refuses two ranks on one device); the ranks time-share the GPU, so the "
                         "value is NOT a scaling number")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default invocation (N = 1, every workload flag at its default) also runs short lines of BASELINE "
                         "configs[1], [3], [4] and the 4-clips-per-GPU share of configs[2] as sub-processes and reports them "
                         "under other_configs; this switch skips them")
    ap.add_argument("--verify-seconds", type=float, default=2.56,
                    help="clip length of the parity samples of --visual / --candidates runs (their oracle passes carry the "
                         "vision tower / the Judge on top of the solve)")
    ap.add_argument("--oracle-cache", default=None,
                    help="file holding / receiving the CPU oracle's result for this (size, sample): sub-runs of other_configs "
                         "on the same weights reuse the main run's oracle pass for their parity_check")
    ap.add_argument("--selftest-spawn", action="store_true",
                    help="CPU-only: exercise the self-launch + sharding + gather plumbing on gloo (tests/test_bench_spawn_cpu.py)")
    return ap.parse_args(argv)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_self_launch(args) -> None:
    """`python bench.py --gpus N` outside a torchrun environment: become `python -m torch.distributed.run ... bench.py`
    (exec, so stdout / stderr / the exit code are this process's)."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


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


def selftest_spawn(args) -> None:
    """No GPU: every rank joins a gloo group, takes its shard of the global batch exactly as the strong-scaling run
    does, all-reduces a fake timing with MAX and rank 0 prints one JSON line.  Covers maybe_self_launch(), the env
    contract and the collectives bench.py uses, with world_size > 1 on CPU."""
    import torch
    import torch.distributed as dist
    from sam_audio_amd.dist import shard_range
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    rows = list(shard_range(args.batch, rank, world))
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    n = torch.tensor([len(rows)], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": "spawn", "n_gpus": world, "clips_total": int(n.item()), "max_time": float(t.item()),
                          "rank0_rows": rows}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def oracle_sample(cfg, sd_cpu, clips, text, tmask, noise, threads):
    """The oracle on the host cores, bounded to R clips of 10 s but otherwise the WHOLE path: DAC encode, the full 16-step
    midpoint solve (32 DiT evaluations) and DAC decode of target + residual, each timed once.  Returns (cpu_baseline
    dict, the oracle's tensors for parity_check)."""
    import torch
    from oracle import samaudio_oracle as O
    torch.set_num_threads(threads)
    codec = cfg.audio_codec
    R = clips.size(0)
    with torch.inference_mode():
        t0 = time.perf_counter()
        z = O.dac_encode(sd_cpu, codec, clips).transpose(1, 2)
        t_enc = time.perf_counter() - t0
        log(f"cpu baseline: DAC encode of {R} clips {t_enc:.2f} s on {threads} threads")
        feats = torch.cat([z, z], dim=2)
        T = feats.shape[1]
        pad = torch.ones(R, T, dtype=torch.bool)
        ids, align = O.anchors_to_ids(None, pad, codec.hop_length, codec.sample_rate)
        video = feats.new_zeros(R, cfg.vision_encoder.dim, T)

        def field(t, y):
            return O.samaudio_forward(sd_cpu, cfg, y, feats, text, t.expand(R), video=video, text_mask=tmask,
                                      anchor_ids=ids, anchor_alignment=align, pad_mask=pad)

        t0 = time.perf_counter()
        lat = O.ode_fixed_grid(field, noise, method="midpoint", step_size=2 / 32)  # reference model.py:22: 16 steps
        t_ode = time.perf_counter() - t0
        log(f"cpu baseline: 16 midpoint steps (32 DiT evaluations) {t_ode:.2f} s")
        gen = lat.transpose(1, 2).reshape(2 * R, lat.shape[2] // 2, T)
        t0 = time.perf_counter()
        wav = O.dac_decode(sd_cpu, codec, gen)
        t_dec = time.perf_counter() - t0
        log(f"cpu baseline: DAC decode x{2 * R} {t_dec:.2f} s")
    per_clip = (t_enc + t_ode + t_dec) / R
    base = {
        "value": CLIP_SECONDS / per_clip, "unit": "s-audio/s", "cores": threads, "kind": "port",
        "sample": (f"{R} clip(s) x 10 s, same dims, fp32 torch oracle (restatement of the reference, pinned to its own "
                   f"classes; the reference package itself cannot be imported on the GPU box), the whole path once: DAC "
                   f"encode {t_enc:.2f} s + 16 midpoint steps = 32 DiT evaluations {t_ode:.2f} s + DAC decode "
                   f"x{2 * R} {t_dec:.2f} s => {per_clip:.1f} s per clip"),
    }
    return base, {"z": z, "lat": lat, "wav": wav.reshape(R, 2, -1)}


def parity_check(model, sub, noise, ref, R, dev, precision):
    """The oracle's sample (`sub`: the same R clips as a device batch) through the HIP path in the benchmarked precision:
    the full default solve (16 midpoint steps) exactly as the timed steps run it."""
    import torch
    with torch.inference_mode():
        z = model.encode_audio(sub.audios)
        res = model.separate(sub, noise=noise.to(dev))
        lat = model.last_latent
        wav = torch.stack([torch.stack(res.target), torch.stack(res.residual)], 1)
        torch.cuda.synchronize()

    def err(a, b):
        return float((a.float().cpu() - b).abs().max())

    e_lat, e_wav = err(lat, ref["lat"]), err(wav, ref["wav"])
    return {
        "precision": precision, "f32_classes": [c for c in hip_classes() if model.f32_classes & hip_cls(c)],
        "bf16_operand_classes": [c for c in hip_classes() if getattr(model, "alt16_classes", 0) & hip_cls(c)] or None, "rows": R,
        "what": "DAC encode -> the full 16-step midpoint solve (32 DiT evaluations) on fixed CPU noise -> DAC decode of "
                "target+residual, i.e. separate() as timed; HIP path vs the fp32 CPU oracle, max-abs",
        "tolerance": 1e-3, "within_tolerance": bool(e_lat <= 1e-3 and e_wav <= 1e-3),
        "encode_latent_err": err(z, ref["z"]), "encode_latent_ref_max": float(ref["z"].abs().max()),
        "ode_latent_err": e_lat, "ode_latent_ref_max": float(ref["lat"].abs().max()),
        "waveform_err": e_wav, "waveform_ref_max": float(ref["wav"].abs().max()),
    }


def hostile_check(precision, dev, threads, size="small*"):
    """separate() as timed (DAC encode -> 16 midpoint steps -> decode) on TRAINED-LIKE weights (synthetic.make_hostile) at `size`
    dims, 2 clips x 10 s, against the fp32 CPU oracle: the headline precision and plain fp16 (tests/test_hostile_gpu.py and
    tests/test_x3_gpu.py assert the same figures; large* dims: SAMAUDIO_HOSTILE_SIZE there)."""
    import torch
    from oracle import samaudio_oracle as O
    from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
    from sam_audio_amd.synthetic import init_state_dict, make_hostile, synthetic_clip, synthetic_noise, synthetic_text_features
    cfg = preset_config(size)
    sd = make_hostile(init_state_dict(cfg, seed=0, device=dev), cfg, seed=0)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    R = 1
    hop = cfg.audio_codec.hop_length
    n = int(CLIP_SECONDS * cfg.audio_codec.sample_rate) // hop * hop
    clips = [synthetic_clip(i, n) for i in range(R)]
    text, tmask = synthetic_text_features(R, 8, seed=7)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["sound"] * R, audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(R, n // hop)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    with torch.inference_mode():
        t_ref, r_ref, lat_ref = O.separate(sd_cpu, cfg, batch.audios, batch.sizes.long(), text, tmask, noise)
    out = {"what": f"trained-like ('hostile') weights at {size} dims, {R} clips x 10 s, separate() as timed vs the fp32 CPU oracle, max-abs",
           "tolerance": 1e-3, "latent_ref_max": float(lat_ref.abs().max()), "waveform_ref_max": float(max(w.abs().max() for w in t_ref + r_ref)),
           "oracle_seconds": round(time.perf_counter() - t0, 1), "modes": {}}
    for prec in dict.fromkeys([precision, "fp16"]):
        model = SAMAudio(cfg, precision=prec, device=str(dev))
        model.load_state_dict(sd, strict=False)
        with torch.inference_mode():
            res = model.separate(batch.to(dev), noise=noise.to(dev))
            torch.cuda.synchronize()
        e_lat = float((model.last_latent.cpu() - lat_ref).abs().max())
        e_wav = max(float((a.cpu() - b).abs().max()) for a, b in zip(res.target + res.residual, t_ref + r_ref))
        out["modes"][prec] = {"ode_latent_err": e_lat, "waveform_err": e_wav, "within_tolerance": bool(e_lat <= 1e-3 and e_wav <= 1e-3)}
        del model
        torch.cuda.empty_cache()
    return out


def _max_err(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


def _verify_inputs(cfg, R, seconds, text_len, seed_noise=99, rows_per_clip=1):
    """R synthetic clips of `seconds` (a whole number of latent frames), text features and fixed CPU noise."""
    import torch
    from sam_audio_amd.synthetic import synthetic_clip, synthetic_text_features
    hop, sr = cfg.audio_codec.hop_length, cfg.audio_codec.sample_rate
    T = max(8, int(round(seconds * sr / hop)))
    clips = [synthetic_clip(100 + i, T * hop) for i in range(R)]
    text, tmask = synthetic_text_features(R, text_len, seed=17)
    g = torch.Generator().manual_seed(seed_noise)
    noise = torch.randn(R * rows_per_clip, T, cfg.transformer.out_channels, generator=g)
    return clips, text, tmask, noise, T


def parity_visual(model, cfg, sd_cpu, proc, dev, precision, vsd_cpu, pe_cfg, seconds, text_len, threads):
    """configs[4]'s own check, two parts.  (1) The PE-Core tower: the features `separate()` itself computes for 2 masked videos
    (processor frame sampling -> resize / normalise -> HIP tower) against the CPU tower oracle on 4 frames of each video.
    (2) The visually conditioned solve: DAC encode -> 16 midpoint steps with the video term live -> decode, HIP path against the
    oracle's separate() fed with THOSE features (so (2) isolates the DiT's video path from (1))."""
    import torch
    from oracle import samaudio_oracle as O
    from oracle import vit_oracle as V
    torch.set_num_threads(threads)
    R = 2
    clips, text, tmask, noise, T = _verify_inputs(cfg, R, seconds, text_len)
    S = cfg.vision_encoder.image_size
    vids = []
    for i in range(R):
        g = torch.Generator().manual_seed(8765 + i)
        v = torch.randint(0, 256, (T, 3, S, S), generator=g, dtype=torch.uint8)
        m = torch.zeros(T, 1, S, S, dtype=torch.uint8)
        m[..., : S // 2] = 1
        vids.append((v, m))
    masked = proc.mask_videos([v for v, _ in vids], [m for _, m in vids])
    sub = proc(descriptions=["sound"] * R, audios=clips, masked_videos=masked, text_features=text, text_mask=tmask)
    sizes = sub.sizes.long().clone()
    audios_cpu = sub.audios.clone()
    video_cpu = [v.clone() for v in sub.masked_video]
    sub = sub.to(dev)
    with torch.inference_mode():
        feats_hip = model.vision_encoder(sub.masked_video).float().cpu()            # [R, T, dim], what separate() uses
        res = model.separate(sub, noise=noise.to(dev))
        lat = model.last_latent.float().cpu()
        wav = torch.stack([torch.stack(res.target), torch.stack(res.residual)], 1).float().cpu()
        idx = torch.linspace(0, T - 1, 4).round().long()
        t0 = time.perf_counter()
        want_f = torch.stack([V.encode_image(vsd_cpu, pe_cfg, model.vision_encoder.transform(v[idx]), normalize=True)
                              for v in video_cpu])
        t_vit = time.perf_counter() - t0
        got_f = feats_hip[:, idx]
        t0 = time.perf_counter()
        t_ref, r_ref, lat_ref = O.separate(sd_cpu, cfg, audios_cpu, sizes, text, tmask, noise, video=feats_hip.transpose(1, 2))
        t_sep = time.perf_counter() - t0
    wav_ref = torch.stack([torch.stack(t_ref), torch.stack(r_ref)], 1)
    e_f, e_lat, e_wav = _max_err(got_f, want_f), _max_err(lat, lat_ref), _max_err(wav, wav_ref)
    cos = float(torch.nn.functional.cosine_similarity(got_f, want_f, dim=-1).min())
    return {
        "precision": precision, "rows": R, "clip_seconds": round(T * cfg.audio_codec.hop_length / cfg.audio_codec.sample_rate, 3),
        "what": "configs[4]: (1) PE-Core tower features of 2 masked videos as separate() computes them vs the CPU tower oracle on 4 "
                "frames each (L2-normalised features: max-abs and min cosine); (2) DAC encode -> full 16-step solve WITH the video "
                "term -> decode, HIP vs the oracle's separate() fed with the same features; max-abs",
        "tolerance": 1e-3, "within_tolerance": bool(e_lat <= 1e-3 and e_wav <= 1e-3),
        "tower_feature_err": e_f, "tower_feature_min_cosine": cos, "tower_frames_compared": int(2 * idx.numel()),
        "ode_latent_err": e_lat, "ode_latent_ref_max": float(lat_ref.abs().max()),
        "waveform_err": e_wav, "waveform_ref_max": float(wav_ref.abs().max()),
        "oracle_seconds": {"tower": round(t_vit, 1), "separate": round(t_sep, 1)},
    }


def parity_rerank(model, cfg, sd_cpu, proc, dev, precision, judge_sd_cpu, seconds, text_len, threads, cand=2):
    """configs[3]'s own check: 1 clip x `cand` candidates.  The HIP path's separate(reranking_candidates=cand) with the HIP Judge
    as text_ranker against the oracle: candidate solve (latent of every candidate), the Judge's overall scores (oracle Judge on
    the ORACLE's candidate waveforms, HIP Judge on the HIP ones) and the index it picks."""
    import torch
    from oracle import judge_oracle as J
    from oracle import samaudio_oracle as O
    torch.set_num_threads(threads)
    clips, text, tmask, noise, T = _verify_inputs(cfg, 1, seconds, text_len, rows_per_clip=cand)
    prompt = [PROMPTS[0]]
    sub = proc(descriptions=prompt, audios=clips, text_features=text, text_mask=tmask)
    sizes, audios_cpu = sub.sizes.long().clone(), sub.audios.clone()
    sub = sub.to(dev)
    ranker = model.text_ranker
    seen = {}

    def spy(**kw):
        seen["scores"] = ranker(**kw)
        return seen["scores"]

    model.text_ranker = spy
    try:
        with torch.inference_mode():
            res = model.separate(sub, noise=noise.to(dev), reranking_candidates=cand)
            lat = model.last_latent.float().cpu()
            torch.cuda.synchronize()
    finally:
        model.text_ranker = ranker
    scores_hip = seen["scores"].float().cpu().reshape(1, cand)
    codec, jcfg, judge = cfg.audio_codec, ranker.model.config, ranker.model
    with torch.inference_mode():
        t0 = time.perf_counter()
        _, _, lat_ref = O.separate(sd_cpu, cfg, audios_cpu, sizes, text, tmask, noise, candidates=cand, decode=False)
        gen = lat_ref.transpose(1, 2).reshape(2 * cand, lat_ref.shape[2] // 2, T)
        wav_ref = O.dac_decode(sd_cpu, codec, gen).view(cand, 2, -1)
        t_sep = time.perf_counter() - t0
        # the Judge oracle on the oracle's candidates, inputs formed as ranking/judge.py:21-42 + processor.py:337-363 form them
        n = int(sizes[0]) * codec.hop_length
        pj = ranker.processor(text=prompt, input_audio=[audios_cpu[0, :, :n]], separated_audio=[wav_ref[c, 0, :n][None] for c in range(cand)],
                              sampling_rate=codec.sample_rate)
        tm = judge.text_model                       # the transformers module that carries the text tower's weights (CPU)
        layers = tm.config.num_hidden_layers
        grabbed = []
        hook = tm.final_norm.register_forward_pre_hook(lambda mod, a: grabbed.append(a[0]))
        out = tm(input_ids=pj["input_ids"], attention_mask=pj.get("attention_mask"),
                 output_hidden_states=jcfg.nth_text_layer is not None)
        hook.remove()
        nth = jcfg.nth_text_layer
        hs = (out.last_hidden_state if nth is None else
              (grabbed[0] if (nth == layers and jcfg.last_text_layer_prenorm) else
               (out.last_hidden_state if nth == layers else out.hidden_states[nth])))
        pooled = hs[:, 0].repeat_interleave(cand, 0)
        t0 = time.perf_counter()
        want = J.judge_forward(judge_sd_cpu, jcfg, pooled, pj["input_values"].repeat_interleave(cand, 0), pj["separated_values"],
                               pj["padding_mask"].repeat_interleave(cand, 0))[:, 0].reshape(1, cand)
        t_judge = time.perf_counter() - t0
    pick_hip, pick_ref = int(scores_hip.argmax(dim=1)), int(J.rerank_select(want)[0])
    e_lat, e_score = _max_err(lat, lat_ref), _max_err(scores_hip, want)
    # The selection is compared at the Judge's own resolution: the two paths pick the same candidate, or the HIP pick is - by the
    # ORACLE's scores - within twice the measured score error of the oracle's best (with random-init weights the eight candidates of a
    # clip score within 2e-4 of each other at large* dims, as far apart as the 16-bit Judge is accurate: profiles/r6_final4/).  The
    # waveform is that of the HIP pick against the oracle's waveform of the SAME candidate.
    margin = float(want[0, pick_ref] - want[0, pick_hip])
    consistent = pick_hip == pick_ref or margin <= 2.0 * e_score
    e_wav = _max_err(res.target[0], wav_ref[pick_hip, 0, :n])
    return {
        "precision": precision, "rows": cand, "clip_seconds": round(T * codec.hop_length / codec.sample_rate, 3),
        "what": f"configs[3]: 1 clip x {cand} candidates - DAC encode -> full 16-step solve of every candidate -> decode -> Judge "
                "scores -> argmax; HIP path (HIP Judge as text_ranker) vs the oracle (separate(candidates) + oracle Judge on the "
                "oracle's candidates); max-abs",
        "tolerance": 1e-3, "within_tolerance": bool(e_lat <= 1e-3 and consistent and e_wav <= 1e-3),
        "ode_latent_err": e_lat, "ode_latent_ref_max": float(lat_ref.abs().max()),
        "judge_overall_scores_hip": [round(float(v), 5) for v in scores_hip[0]],
        "judge_overall_scores_oracle": [round(float(v), 5) for v in want[0]],
        "judge_score_err": e_score, "argmax_hip": pick_hip, "argmax_oracle": pick_ref, "argmax_equal": pick_hip == pick_ref,
        "argmax_margin_oracle": margin, "argmax_consistent": bool(consistent),
        "selected_waveform_err": e_wav, "oracle_seconds": {"separate": round(t_sep, 1), "judge": round(t_judge, 1)},
    }


def default_workload(args):
    return (args.size == "large*" and args.batch == 32 and args.candidates == 1 and not args.predict_spans and not args.visual
            and args.t5 and args.streams == 0 and not args.serial_groups and args.text_len == 8)


def other_configs(args):
    """Short lines of the other BASELINE.json configurations, each a sub-process of this script on the same GPU after the
    main measurement (the main run's models have been released): configs[1] small* 8 clips, configs[3] large* 8 clips x 8
    candidates with the span predictor and the Judge reranker, configs[4] large* 4 clips with visual prompts through the
    PE-Core tower, and the 4-clips-per-GPU share of configs[2] (what one of 8 GPUs runs under strong scaling).  Every
    sub-line carries its own `roofline` and `parity_check`: the text-only ones the 2-clip full solve of their dims against the
    CPU oracle (the 4-clip share IS the main line's model and sample, so it reads the main run's oracle pass from
    --oracle-cache and says so); configs[3] and [4] run their own samples (parity_rerank / parity_visual)."""
    import subprocess
    large = args.oracle_cache
    common = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
              "--verify", "--no-other-configs", "--no-parity-mode", "--precision", args.precision]
    runs = [
        ("configs[1] small* 8 clips", ["--size", "small*", "--batch", "8"]),
        ("configs[2] share of one of 8 GPUs: 4 clips", ["--batch", "4"] + (["--oracle-cache", large] if large else [])),
        ("configs[3] 8 clips x 8 candidates, span predictor + Judge", ["--batch", "8", "--candidates", "8", "--predict-spans", "--steps", "2",
                                                                       "--verify-seconds", "1.28"]),
        ("configs[4] 4 clips, visual prompts", ["--batch", "4", "--visual"]),
    ]
    out = []
    for name, extra in runs:
        t0 = time.perf_counter()
        try:
            p = subprocess.run(common + extra, capture_output=True, text=True, timeout=900)
            rows = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not rows:
                out.append({"config": name, "error": (p.stderr or p.stdout)[-400:]})
                continue
            sub = json.loads(rows[-1])
            keep = {k: sub.get(k) for k in ("value", "unit", "ms_per_step", "steps", "dtype", "config", "parity_check",
                                            "rerank_breakdown", "vision_tower")}
            r = sub.get("roofline") or {}
            keep["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches_per_step",
                                                       "avg_launch_us")} if r else None
            if r.get("whole_step"):
                keep["whole_step_frac"] = r["whole_step"].get("frac")
            keep["config_name"] = name
            keep["wall_s"] = round(time.perf_counter() - t0, 1)
            out.append(keep)
            log(f"other_configs: {name}: {keep['value']} {keep['unit']} in {keep['wall_s']} s")
        except Exception as exc:   # a sub-line must never cost the main line
            out.append({"config": name, "error": repr(exc)[:400]})
    return out


def hip_classes():
    from sam_audio_amd import hip
    return hip.CLASSES


def hip_cls(name):
    from sam_audio_amd import hip
    return hip.CLS[name]


PROMPTS = ["a dog barking", "man speaking", "rain on a tin roof", "acoustic guitar strumming chords", "car engine idling",
           "glass breaking", "a woman singing softly over a piano", "birds"]


class _HashTokenizer:
    """Stand-in for the Judge's ModernBERT tokenizer (no tokenizer files offline): word hashes -> ids, pad-to-longest."""

    def __call__(self, text, return_tensors="pt", padding="longest", max_length=512, truncation=True, **_):
        import torch
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
    jsd = init_judge_state_dict(jcfg, seed=1, device=dev)
    judge.load_state_dict(jsd, strict=False)
    proc = SAMAudioJudgeProcessor(jcfg.audio_codec.hop_length, jcfg.audio_codec.sample_rate, tokenizer=_HashTokenizer())
    return JudgeRanker(model=judge, processor=proc), jsd


def build_span_predictor(cfg, precision, dev):
    """PE-A-Frame with the pe-a-frame-large stand-in dims (PEAudioFrameConfig defaults: PE-AV audio tower 1792 x 6 layers,
    ModernBERT-large-shaped text tower, random init; reference model.py:96-102) + the hash tokenizer as its transform."""
    from sam_audio_amd.config import PEAudioFrameConfig
    from sam_audio_amd.judge import PEAudioFrame
    from sam_audio_amd.synthetic import init_frame_state_dict
    fcfg = PEAudioFrameConfig(codebook_dim=cfg.audio_codec.codebook_dim)
    predictor = PEAudioFrame(fcfg, precision=precision, device=str(dev), hop_length=cfg.audio_codec.hop_length,
                             sample_rate=cfg.audio_codec.sample_rate)
    predictor.load_state_dict(init_frame_state_dict(fcfg, seed=2, device=dev), strict=False)
    tok = _HashTokenizer()
    return predictor, (lambda text: tok(text))


# rocprofv3 kernel symbols of the profile names (profiles/r2_traffic.json is keyed by symbol)
SYMBOLS = {
    "gemm8_bf16_256x256_8phase": "sa::gemm8_kernel<false",         # plain GEMMs: every Linear of the DiT (<false, true>: the
                                                                   # bf16-operand instantiation of the mixed mode)
    "gemm8_bf16_256x256_8phase_conv": "sa::gemm8_kernel<true",     # implicit convolutions (patcher, wide codec stages)
    "gemm8s_bf16_128x128": "sa::gemm8s_kernel<",
    # the K' = 3K launches of the compensated mode on K-concatenated operands: the operand-sharing kernel (common.h GEMM_FLAG_X3_SHARE)
    "gemm8_bf16_256x256_8phase_x3": "sa::gemm8x_kernel",
}


def traffic_of(kernel: str, split: bool = False):
    """HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/r2_final.sh -> tools/pmc_traffic.py); they
    cannot be collected inside a timed run.  r2_traffic.json: launches not split into whole rounds + tail (what two
    concurrent row groups run); r2_traffic_split.json: the single-group form."""
    for fname in ("r6_traffic_x3.json", "r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json") + (("r2_traffic_split.json",) if split else ()) + ("r2_traffic.json", "r1_traffic.json"):
        try:
            table = json.load(open(os.path.join(ROOT, "profiles", fname)))["kernels"]
        except (OSError, KeyError, ValueError):
            continue
        name = kernel.split("/")[-1]
        # launches over K' = 3K have their own PMC passes (profiles/r6_traffic_x3.json): never mixed with the one-product figures
        if name.endswith("_x3") != fname.startswith("r6_traffic_x3"):
            continue
        key = SYMBOLS.get(name, "")
        hit = [v for k, v in table.items() if key and k.startswith(key)]
        if not hit and name.endswith("_x3"):   # (the convolutions and every x3 launch of a table older than gemm8x_kernel: gemm8_kernel itself)
            key = SYMBOLS.get(name[:-3], "")
            hit = [v for k, v in table.items() if key and k.startswith(key)]
        if hit:   # several instantiations of the symbol: the one with the most launches is the one the roofline is about
            best = max(hit, key=lambda v: v.get("launches", 0))
            return round(best["traffic_bytes_per_launch"]), f"HBM bytes per launch, rocprofv3 PMC passes (profiles/{fname})"
    return None, None


SPLIT_MODE = [False]   # set by main(): whether the instrumented step runs launches split into whole rounds + tail


def reference_flops(cfg, clips, text_len):
    """Algorithmic FLOPs of one separate() over `clips` 10 s clips as the REFERENCE executes it:
    per evaluation L x (2T D^2 x 6 + 4 T^2 D + 6 T D F + cross K,V and scores) + patcher 12 T D^2 + the 768/1024/128/256-wide
    projections, x 32 evaluations; DAC-VAE encode 0.487 TF + decode 2 x 1.096 TF per clip (default codec dims)."""
    t, codec = cfg.transformer, cfg.audio_codec
    T = int(round(CLIP_SECONDS * codec.sample_rate / codec.hop_length))   # 250 latent frames per 10 s clip
    D, F, L, Lt = t.dim, t.ffn_hidden, t.n_layers, text_len
    text_d, video_d, anchor_d, latent = cfg.text_encoder.dim, cfg.vision_encoder.dim, cfg.anchor_embedding_dim, t.out_channels
    layer = 2 * T * D * D * 6 + 4 * T * T * D + 6 * T * D * F + 2 * Lt * D * D * 2 + 4 * T * Lt * D
    # input projection (3 x latent wide, model.py:116-125), video conv1x1, anchor projection, output projection (widths from cfg)
    per_eval = (L * layer + 12 * T * D * D + 2 * T * D * (3 * latent + video_d + anchor_d + latent) + 2 * Lt * D * text_d
                + 6 * Lt * D * D)
    return clips * (32.0 * per_eval + 0.487e12 + 2 * 1.096e12)   # codec figures: SURVEY.md section 8(a2, a15), default codec dims


def tower_precision(precision):
    """the towers beside the DiT (Judge, span predictor, vision tower: SURVEY.md section 8 "next" rows) have no compensated mode: beside
    an x3 DiT they run on the library's plain 16-bit operands, as they do beside an fp16 one"""
    from sam_audio_amd import hip
    return hip.tower_precision(precision)


X3_PRODUCTS = 3   # MFMA products per algorithmic multiply of a "_x3" launch (lo*hi + hi*lo + hi*hi)


def rooflines(stats):
    """stats: [{name 'class/kernel', launches, flops, bytes, ms}] of ONE instrumented step."""
    rows = []
    for k in stats:
        if k["ms"] <= 0:
            continue
        sec = k["ms"] * 1e-3
        tf, gbs = k["flops"] / sec / 1e12, k["bytes"] / sec / 1e9
        t_mfma, t_hbm = k["flops"] / (PEAK_BF16_TFLOPS * 1e12), k["bytes"] / (PEAK_HBM_GBS * 1e9)
        rows.append({"kernel": k["name"], "launches": k["launches"], "ms": round(k["ms"], 3), "tflops": round(tf, 2),
                     "gbs": round(gbs, 1), "bound": "mfma" if t_mfma >= t_hbm else "hbm",
                     "frac": round(max(t_mfma, t_hbm) / sec, 4), "_flops": k["flops"], "_bytes": k["bytes"]})
    rows.sort(key=lambda r: -r["ms"])
    dit_gemm = [r for r in rows if r["kernel"].startswith("dit/gemm")]
    out = {"roofline": None, "roofline_hbm": None}
    if dit_gemm:
        dom = dit_gemm[0]
        ms_all = sum(r["ms"] for r in dit_gemm)
        fl_all = sum(r["_flops"] for r in dit_gemm)
        traffic, note = traffic_of(dom["kernel"], split=SPLIT_MODE[0])
        x3 = dom["kernel"].endswith("_x3")
        ex_all = sum(r["_flops"] * (X3_PRODUCTS if r["kernel"].endswith("_x3") else 1) for r in dit_gemm)
        out["roofline"] = {
            "bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": round(dom["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
            "launches_per_step": dom["launches"], "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
            "flops_per_step": dom["_flops"],
            "dit_gemm_all": {"achieved": round(fl_all / (ms_all * 1e-3) / 1e12, 2),
                             "frac": round(fl_all / (ms_all * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                             "mfma_frac": round(ex_all / (ms_all * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                             "ms_per_step": round(ms_all, 2), "flops_per_step": fl_all,
                             "launches_per_step": sum(r["launches"] for r in dit_gemm)},
        }
        if x3:
            out["roofline"]["mfma"] = {
                "what": f"`achieved` / `frac` count the ALGORITHMIC flops of the launch (2 M N K: one product per multiply, as the "
                        f"reference's fp32 GEMM); the kernel issues {X3_PRODUCTS} 16-bit MFMA products per multiply (x_lo W_hi + x_hi W_lo + "
                        "x_hi W_hi over K' = 3K) to deliver fp32-grade results, so the matrix cores run at `achieved` TFLOP/s below",
                "products_per_multiply": X3_PRODUCTS, "achieved": round(dom["tflops"] * X3_PRODUCTS, 2),
                "frac": round(dom["tflops"] * X3_PRODUCTS / PEAK_BF16_TFLOPS, 4),
                "fp32_mfma_peak": 157.3, "vs_fp32_mfma_peak": round(dom["tflops"] / 157.3, 2)}
        if note:
            out["roofline"]["traffic_note"] = note
    groups = []
    codec = [r for r in rows if r["kernel"].startswith("codec/")]
    if codec:
        ms, fl, by = sum(r["ms"] for r in codec), sum(r["_flops"] for r in codec), sum(r["_bytes"] for r in codec)
        t_mfma, t_hbm = fl / (PEAK_BF16_TFLOPS * 1e12), by / (PEAK_HBM_GBS * 1e9)
        groups.append({"kernel": "codec/* (DAC-VAE convolutions, all symbols)", "ms": round(ms, 2),
                       "tflops": round(fl / (ms * 1e-3) / 1e12, 2), "achieved": round(by / (ms * 1e-3) / 1e9, 1),
                       "peak": PEAK_HBM_GBS, "unit": "GB/s", "bound": "mfma" if t_mfma >= t_hbm else "hbm",
                       "frac": round(max(t_mfma, t_hbm) / (ms * 1e-3), 4), "algorithmic_bytes": by})
    for r in rows:
        if r["kernel"].startswith("dit/") and not r["kernel"].startswith("dit/gemm"):
            groups.append({"kernel": r["kernel"], "ms": r["ms"], "launches": r["launches"], "achieved": r["gbs"],
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "bound": r["bound"], "frac": r["frac"],
                           "algorithmic_bytes": r["_bytes"]})
    out["roofline_hbm"] = groups or None
    out["kernels"] = [{k: v for k, v in r.items() if not k.startswith("_")} for r in rows]
    out["_rows"] = rows
    return out


def main():
    args = parse()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    maybe_self_launch(args)
    if args.selftest_spawn:
        return selftest_spawn(args)
    scratch = None
    if args.oracle_cache is None and default_workload(args) and args.gpus == 1 and not args.no_other_configs:
        # the oracle's result travels to the 4-clip sub-run through a file in a private directory (mode 0700), removed afterwards
        import tempfile
        scratch = tempfile.mkdtemp(prefix="samaudio_bench_")
        args.oracle_cache = os.path.join(scratch, "oracle_large.pt")
    try:
        return run(args)
    finally:
        if scratch is not None:
            import shutil
            shutil.rmtree(scratch, ignore_errors=True)


def run(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (the hot path has no CPU fallback)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
    from sam_audio_amd.dist import broadcast_state_dict, shard_range
    from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_text_features

    cfg = preset_config(args.size)
    tcfg = cfg.transformer
    if os.environ.get("SAMAUDIO_DEBUG_FLAGS"):   # A/B switches, applied by sam_audio_amd.hip.lib() at load
        log(f"debug flags {os.environ['SAMAUDIO_DEBUG_FLAGS']}")

    # ---- weights: rank 0 creates them, RCCL broadcast over xGMI to the other ranks -----------------------
    log(f"world {world} (backend {('gloo, ranks share cuda:0' if args.share_gpu else 'nccl/RCCL') if world > 1 else 'none'}), preset {args.size}, scaling {args.scaling}, "
        f"batch {args.batch}, usable host cores {usable_cores()}")
    sd = init_state_dict(cfg, seed=0, device=dev) if rank == 0 else None
    sd = broadcast_state_dict(sd, src=0, device=dev)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    want_verify = rank == 0 and world == 1 and not args.no_verify and (want_cpu or args.verify)
    sd_cpu = {k: v.cpu() for k, v in sd.items()} if (want_cpu or want_verify) else None
    def auto_streams(n_clips):
        return args.streams if args.streams > 0 else (2 if n_clips >= 16 else 1)

    model = SAMAudio(cfg, precision=args.precision, device=str(dev), streams=max(args.streams, 2))
    model.load_state_dict(sd, strict=False)
    # plain 16-bit operand modes timed side by side with the headline
    sides = {"fp16x3": ["fp16", "bf16"], "bf16x3": ["bf16"], "fp16": ["mixed"], "bf16": ["fp16"], "mixed": ["fp16"]}.get(args.precision, [])
    want_parity_mode = (bool(sides) and not args.no_parity_mode and not args.visual and args.candidates == 1
                        and not args.predict_spans)
    sd_keep = sd if want_parity_mode else None   # the parity-mode model is built from the same weights after the timed run
    del sd
    torch.cuda.empty_cache()
    log("weights loaded")
    os.environ.setdefault("SAMAUDIO_CODEC_CHUNK", "32")
    n_samples = int(CLIP_SECONDS * cfg.audio_codec.sample_rate)
    proc = SAMAudioProcessor.from_config(cfg)

    def make_batch(clip_ids):
        """This rank's synthetic clips, resident in HBM before the timed region."""
        clips = [synthetic_clip(i, n_samples) for i in clip_ids]
        text, tmask = synthetic_text_features(len(clip_ids), args.text_len, seed=7 + rank)
        videos = None
        if args.visual:  # uint8 video [250, 3, 336, 336] uniform random, mask = left half
            S = cfg.vision_encoder.image_size
            vids, masks = [], []
            for i in clip_ids:
                g = torch.Generator().manual_seed(4321 + i)
                vids.append(torch.randint(0, 256, (250, 3, S, S), generator=g, dtype=torch.uint8))
                m = torch.zeros(250, 1, S, S, dtype=torch.uint8)
                m[..., : S // 2] = 1
                masks.append(m)
            videos = proc.mask_videos(vids, masks)
        if args.t5:   # the processor hands the descriptions through; SAMAudio.separate() calls model.text_encoder
            b = proc(descriptions=[PROMPTS[i % len(PROMPTS)] for i in clip_ids], audios=clips, masked_videos=videos)
        else:
            b = proc(descriptions=["sound"] * len(clip_ids), audios=clips, masked_videos=videos, text_features=text,
                     text_mask=tmask)
        return b.to(dev), clips, text, tmask

    if args.t5:
        import transformers
        from sam_audio_amd.text_encoder import T5TextEncoder
        t5cfg = transformers.T5Config(vocab_size=32128, d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_heads=12,
                                      feed_forward_proj="relu")   # t5-base (reference text_encoder.py:11-17)
        torch.manual_seed(11)
        model.text_encoder = T5TextEncoder(cfg.text_encoder, model=transformers.T5EncoderModel(t5cfg), tokenizer=_HashTokenizer(),
                                           device=dev)
        log(f"T5 text encoder attached (t5-base dims, random init, hash tokenizer, backend {model.text_encoder.backend})")

    vision = vsd_cpu = judge_sd_cpu = None
    if args.visual:
        from sam_audio_amd.config import PE_VISION_CONFIGS
        from sam_audio_amd.synthetic import init_vision_state_dict
        from sam_audio_amd.vision_encoder import PerceptionEncoder
        pe_cfg = PE_VISION_CONFIGS[cfg.vision_encoder.name]
        model.vision_encoder = PerceptionEncoder(cfg.vision_encoder, device=dev, precision=tower_precision(args.precision))
        vsd = init_vision_state_dict(pe_cfg, seed=5, device=dev)
        if want_verify:
            vsd_cpu = {k: v.float().cpu() for k, v in vsd.items()}
        model.vision_encoder.load_state_dict({"model.visual." + k: v for k, v in vsd.items()})
        del vsd
        log(f"vision tower {cfg.vision_encoder.name} attached ({pe_cfg.layers} layers, width {pe_cfg.width}, "
            f"{pe_cfg.tokens} tokens per frame)")

    if args.candidates > 1:
        model.text_ranker, judge_sd = build_judge_ranker(cfg, tower_precision(args.precision), dev)
        if want_verify:
            judge_sd_cpu = {k: v.float().cpu() for k, v in judge_sd.items()}
        del judge_sd
    if args.predict_spans:
        model.span_predictor, model.span_predictor_transform = build_span_predictor(cfg, tower_precision(args.precision), dev)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(batch, steps, warmup, label):

        def step():
            # noise=None: drawn on device inside, like the reference (model.py:274-275)
            if args.predict_spans:
                # predict_spans() writes the predicted anchors into the batch (reference model.py:245, processor.py:122-123)
                # and a batch that already has anchors skips the predictor: every step starts from the un-anchored batch
                batch.process_anchors(None)
                batch.to(dev)   # (Batch.to keeps the host-side range guarantee of the anchor tensors; assigning them would drop it)
            return model.separate(batch, reranking_candidates=args.candidates, predict_spans=args.predict_spans)

        for i in range(warmup):
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            log(f"{label}: warm-up step {i}: {time.perf_counter() - t0:.3f} s")
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert all(torch.isfinite(w).all() for w in res.target), "non-finite output"
        return elapsed, step

    # ---- the timed run ------------------------------------------------------------------------------------
    if args.scaling == "weak":
        my_ids = list(range(rank * args.batch, (rank + 1) * args.batch))
        clips_total = world * args.batch
    else:
        my_ids = list(shard_range(args.batch, rank, world))
        clips_total = args.batch
        assert my_ids, f"strong scaling: rank {rank} got no clip (global batch {args.batch} < {world} ranks)"
    batch, clips, text, tmask = make_batch(my_ids)
    n_streams = model.streams = auto_streams(len(my_ids))
    model._serial_groups = bool(args.serial_groups)
    # pinned, so that the instrumented (single-stream) step launches the timed kernels; SAMAUDIO_BENCH_TAIL_SPLIT=0/1: A/B
    model.tail_split = (n_streams == 1) if os.environ.get("SAMAUDIO_BENCH_TAIL_SPLIT") is None else bool(int(os.environ["SAMAUDIO_BENCH_TAIL_SPLIT"]))
    SPLIT_MODE[0] = n_streams == 1
    log(f"inputs resident ({len(my_ids)} clips on this rank, {n_streams} stream(s)); warm-up")
    elapsed, step = timed(batch, args.steps, args.warmup, args.scaling)
    value = clips_total * CLIP_SECONDS * args.steps / elapsed
    log(f"timed {args.steps} steps in {elapsed:.3f} s -> {value:.2f} s-audio/s")

    # ---- configs[3]: what the span predictor and the Judge reranker cost, by difference ---------------------------------
    breakdown = None
    if rank == 0 and (args.candidates > 1 or args.predict_spans):
        ranker, predictor = model.text_ranker, model.span_predictor
        b_steps = max(1, min(args.steps - 1, 4))

        def run(spans, rerank):
            model.text_ranker = ranker if rerank else None
            model.span_predictor = predictor if spans else None
            import warnings

            def once():
                batch.process_anchors(None)
                batch.to(dev)   # (Batch.to keeps the host-side range guarantee of the anchor tensors; assigning them would drop it)
                model.separate(batch, reranking_candidates=args.candidates, predict_spans=spans)

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                once()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(b_steps):
                    once()
                torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / b_steps

        ms_core = run(False, False)
        ms_spans = run(args.predict_spans, False) if args.predict_spans else ms_core
        ms_all = run(args.predict_spans, True)
        model.text_ranker, model.span_predictor = ranker, predictor
        breakdown = {"what": "ms per step by difference: separate() with candidates but no span predictor and no reranker (DAC "
                             "encode, ODE over batch x candidates rows, DAC decode of every candidate) | + PE-A-Frame span "
                             "predictor | + Judge reranker (DAC encodes of mixture and candidates, two PE-AV transformers, "
                             "ModernBERT text tower)",
                     "encode_ode_decode_ms": round(ms_core, 2), "span_predictor_ms": round(ms_spans - ms_core, 2),
                     "judge_reranker_ms": round(ms_all - ms_spans, 2), "steps": b_steps}
        log(f"breakdown: {breakdown}")

    # ---- N > 1: the other scaling mode in the same invocation (strong is the line's value, weak the side key, or v.v.) ----
    strong = None
    if world > 1 and not args.no_other and args.batch >= world:
        other = "weak" if args.scaling == "strong" else "strong"
        o_ids = (list(range(rank * args.batch, (rank + 1) * args.batch)) if other == "weak"
                 else list(shard_range(args.batch, rank, world)))
        o_total = world * args.batch if other == "weak" else args.batch
        o_batch = make_batch(o_ids)[0]
        o_steps = max(2, min(args.steps, 5))
        model.streams = auto_streams(len(o_ids))
        o_elapsed, _ = timed(o_batch, o_steps, 1, other)
        strong = {"scaling": other, "global_batch": o_total, "clips_per_gpu": len(o_ids), "steps": o_steps,
                  "ms_per_step": round(1e3 * o_elapsed / o_steps, 2), "streams_per_gpu": model.streams,
                  "value": round(o_total * CLIP_SECONDS * o_steps / o_elapsed, 3), "unit": "s-audio/s"}
        log(f"{other}: {o_steps} steps in {o_elapsed:.3f} s -> {strong['value']:.2f} s-audio/s")
        model.streams = n_streams
        del o_batch

    # ---- rooflines: one extra, instrumented step (HIP events on the launch stream) --------------------------------
    # The instrumented step issues EXACTLY the launches of the timed steps (same row groups, same tile policy and options);
    # only their scheduling differs: the row groups run one after the other on one stream, so that an event pair brackets a
    # kernel that has the GPU to itself.
    roof = {"roofline": None, "roofline_hbm": None, "kernels": None}
    if rank == 0 and not args.no_roofline:
        model.profile_begin(serial_groups=True)
        step()
        roof = rooflines(model.profile_end())
        if roof["roofline"]:
            fl = sum(k["_flops"] for k in roof["_rows"])
            ref_fl = reference_flops(cfg, len(my_ids), args.text_len) * (args.candidates if args.candidates > 1 else 1)
            roof["roofline"]["measured"] = (
                f"one extra instrumented step issuing the launches of the timed steps ({n_streams} row group(s) of "
                f"{len(my_ids) // n_streams} clips: M = {len(my_ids) // n_streams * 250} rows per DiT GEMM launch), the groups one "
                "after the other on one stream: HIP events bracket every launch on its launch stream, so a duration is that "
                "of a kernel that has the GPU to itself (= what rocprofv3 --kernel-trace of the serialised command reports, "
                "profiles/); in the timed steps the row groups overlap - see whole_step for the figure that includes it")
            roof["roofline"]["whole_step"] = {
                "what": "flops of a whole step / the TIMED ms_per_step (concurrent row groups, codec and streaming kernels "
                        "included): `executed` = what the kernels of this build run (hoisted conditioning, folded cross-"
                        "attention), `as_reference` = the algorithmic flops of the reference's own op sequence (SURVEY.md 8d)",
                "executed_flops_per_step": fl, "executed": round(fl / (elapsed / args.steps) / 1e12, 2),
                "as_reference_flops_per_step": ref_fl, "as_reference": round(ref_fl / (elapsed / args.steps) / 1e12, 2),
                "unit": "TFLOP/s", "frac_executed": round(fl / (elapsed / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
                "frac": round(ref_fl / (elapsed / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4)}

    # ---- visual prompting: the tower alone, HIP events around the frames of ONE video (250 frames) ---------------
    if rank == 0 and args.visual:
        from sam_audio_amd.vision_tower import tower_flops
        frames = model.vision_encoder.transform(batch.masked_video[0])
        model.vision_encoder.encode(frames)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            model.vision_encoder.encode(frames)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        fl = tower_flops(pe_cfg, frames.shape[0])
        vision = {"tower": cfg.vision_encoder.name, "frames": int(frames.shape[0]), "ms": round(ms, 2),
                  "flops": fl, "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                  "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "bound": "mfma",
                  "frames_per_s": round(frames.shape[0] / (ms * 1e-3), 1),
                  "what": "encode_image of one 250-frame video after resize/normalise: algorithmic flops of the tower "
                          "(all GEMMs + attention) / HIP-event time on the current stream"}
        log(f"vision tower: {vision['ms']} ms per 250 frames, {vision['achieved']} TFLOP/s")

    # ---- the other 16-bit operand format, side by side: the same steps ----------------------------------------------------
    pmodels, pmodes = {}, {}
    for side in (sides if want_parity_mode else []):
        pmodel = SAMAudio(cfg, precision=side, device=str(dev), streams=max(args.streams, 2))
        pmodel.load_state_dict(sd_keep, strict=False)
        torch.cuda.empty_cache()
        pmodel.streams = n_streams
        pmodel.tail_split = model.tail_split
        pmodel.text_encoder = model.text_encoder   # the same T5 stack inside its steps
        p_steps = max(2, min(args.steps, 10))

        def pstep():
            return pmodel.separate(batch)

        pstep()
        fence()
        t0 = time.perf_counter()
        for _ in range(p_steps):
            pstep()
        fence()
        p_elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([p_elapsed], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            p_elapsed = float(t.item())
        pmode = {"precision": side, "what": f"the same workload and streams in precision {side!r}: plain 16-bit GEMM operands, ONE MFMA "
                 "product per multiply.  On the benign seeded weights fp16 and mixed are inside the 1e-3 bound and bf16 is not "
                 "(tests/test_large_gpu.py::test_full_solve_and_decode); on trained-like weights no plain 16-bit mode is "
                 "(tests/test_hostile_gpu.py, DESIGN.md section 4) - which is why none of them is the headline",
                 "value": round(clips_total * CLIP_SECONDS * p_steps / p_elapsed, 3), "unit": "s-audio/s", "steps": p_steps,
                 "ms_per_step": round(1e3 * p_elapsed / p_steps, 2), "parity_check": None}
        log(f"side by side ({side}): {p_steps} steps in {p_elapsed:.3f} s -> {pmode['value']:.2f} s-audio/s")
        if rank == 0 and not args.no_roofline:   # the one-product kernel's own roofline: one instrumented step of this mode
            pmodel.profile_begin(serial_groups=True)
            pstep()
            r = rooflines(pmodel.profile_end())["roofline"]
            pmode["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step",
                                                       "avg_launch_us", "flops_per_step")} if r else None
        pmodels[side], pmodes[side] = pmodel, pmode
    sd_keep = None

    cpu = parity = None
    if want_verify and args.visual:        # configs[4]: its own sample - tower features + the visually conditioned solve
        parity = parity_visual(model, cfg, sd_cpu, proc, dev, args.precision, vsd_cpu, pe_cfg, args.verify_seconds, args.text_len,
                               args.cpu_threads or usable_cores())
        log(f"parity_check (visual): {parity}")
    elif want_verify and args.candidates > 1:   # configs[3]: its own sample - candidate solve + Judge scores + argmax
        parity = parity_rerank(model, cfg, sd_cpu, proc, dev, args.precision, judge_sd_cpu, args.verify_seconds, args.text_len,
                               args.cpu_threads or usable_cores())
        log(f"parity_check (rerank): {parity}")
    elif want_cpu or want_verify:
        R = 1   # the bounded CPU sample: ONE 10 s clip through the whole path (~50 s on 16 cores; 2 clips until round 5)
        threads = args.cpu_threads or usable_cores()
        g = torch.Generator().manual_seed(99)
        noise = torch.randn(R, n_samples // cfg.audio_codec.hop_length, tcfg.out_channels, generator=g)
        cache = args.oracle_cache
        key = [args.size, R, float(torch.stack(clips[:R]).double().sum()), float(text[:R].double().sum()), float(noise.double().sum())]
        ref = None
        reused = False
        if cache and os.path.exists(cache) and not want_cpu:
            ref = torch.load(cache, weights_only=True)
            if ref.get("key") != key:   # another sample / size: not this run's oracle result
                ref = None
            else:
                cpu = None
                reused = True
                log(f"oracle result of this sample read from {cache}")
        if ref is None:
            cpu, ref = oracle_sample(cfg, sd_cpu, torch.stack(clips[:R]), text[:R], tmask[:R], noise, threads)
            if cache:
                ref["key"] = key
                torch.save(ref, cache)
        if want_verify:
            sub = proc(descriptions=["sound"] * R, audios=clips[:R], text_features=text[:R], text_mask=tmask[:R]).to(dev)
            parity = parity_check(model, sub, noise, ref, R, dev, args.precision)
            if reused:
                parity["oracle_pass"] = ("reused from the main line's run (--oracle-cache): the same dims, weights, precision and "
                                         "sample - this sub-run differs from it only in the number of clips it TIMES")
            log(f"parity_check: {parity}")
            for side, pmodel in (pmodels.items() if rank == 0 else ()):
                pmodes[side]["parity_check"] = parity_check(pmodel, sub, noise, ref, R, dev, side)
                log(f"parity_check ({side}): {pmodes[side]['parity_check']}")
        if not want_cpu:
            cpu = None

    hostile = None
    if rank == 0 and world == 1 and want_verify and default_workload(args) and not args.no_hostile:
        hostile = hostile_check(args.precision, dev, args.cpu_threads or usable_cores())
        log(f"hostile_check: {hostile}")
    if rank == 0:
        line = {
            "metric": "seconds-of-audio separated/sec/node", "value": round(value, 3), "unit": "s-audio/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": {"fp16x3": "fp16x3 (fp32 storage as the reference; every big contraction on IEEE-half hi + lo split operands: 3 MFMA "
                                "products per multiply, fp32 accumulation - fp32-grade results on the 16-bit matrix cores)",
                      "mixed": "bf16 (five big GEMM classes of the DiT layers = 96 % of the flops) + fp16 (other GEMMs), fp32 accumulation",
                      "fp16": "fp16 (IEEE half GEMM operands: 16 bits as BASELINE's nominal bf16, the same MFMA rate), fp32 accumulation"
                      }.get(args.precision, args.precision),
            "data": ("synthetic (seeded random weights, synthetic 10 s/48 kHz clips, "
                     + ("prompts through a random-init t5-base-shaped encoder inside the step)" if args.t5
                        else "resident synthetic T5-shaped text features)")),
            "config": {
                "workload": (f"sam-audio-{args.size} (stand-in dims D={tcfg.dim} H={tcfg.n_heads} L={tcfg.n_layers} "
                             f"F={tcfg.ffn_hidden}) {args.precision}, batch={args.batch}x10 s clips "
                             f"{'per GPU' if args.scaling == 'weak' else 'global, split over the GPUs'}, text prompt "
                             f"Lt={args.text_len}, midpoint ODE 16 steps = 32 DiT evals, DAC-VAE encode + decode x2"
                             + (", visual prompt: 250 masked video frames per clip through the PE-Core tower" if args.visual
                                else "")),
                "clips_per_gpu": len(my_ids), "global_batch": clips_total, "parallelism": f"clip-sharded x{world}" + (" (ranks time-share ONE GPU over gloo: functional check, not a scaling number)"
                                                                         if args.share_gpu and world > 1 else ""),
                "streams_per_gpu": n_streams, "reranking_candidates": args.candidates,
                "predict_spans": bool(args.predict_spans), "world_size_seen": world,
                "text_encoder_in_step": ("t5-base dims (12 layers, d_model 768), random init, hash tokenizer, T5 stack on the HIP "
                                         "library (fp32), run on the descriptions inside every timed step") if args.t5 else None,
                "visual_prompt": (f"{cfg.vision_encoder.name} tower, 250 frames x 336x336 per clip, encoded inside the step"
                                  if args.visual else None),
            },
            "vision_tower": vision,
            "roofline": roof["roofline"], "roofline_hbm": roof["roofline_hbm"], "cpu_baseline": cpu,
            "parity_check": parity, "hostile_check": hostile, "other_scaling": strong, "rerank_breakdown": breakdown,
            "kernels": roof.get("kernels"),
        }
        for side in sides:
            line[f"{side}_mode"] = pmodes.get(side)
        if default_workload(args) and world == 1 and not args.no_other_configs:
            del model, pmodels, batch, step   # the sub-processes build their own models on this GPU
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
