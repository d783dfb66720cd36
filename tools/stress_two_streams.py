#!/usr/bin/env python
"""The scenario of tests/test_path_gpu.py::test_concurrent_streams_are_bitwise_equal_to_one_stream, `--reps` times in one
process: `SAMAudio(streams=2).separate()` (two row groups on two engine contexts / HIP streams, two host threads) against
the one-stream latent, bit for bit.  Between repetitions the allocator is churned (blocks of varying size filled with
finite garbage are allocated and dropped, so that whatever lies behind the tensors the kernels are handed changes) and
every `--rebuild` repetitions the two-stream model is built anew (new lanes, new streams from torch's pool, new weight
conversions).  Run it under both allocator modes:

    python tools/stress_two_streams.py --reps 300
    PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1 python tools/stress_two_streams.py --reps 300

With SAMAUDIO_TRACE_HASH=1 the per-stage checksums of a differing repetition are left in --trace-dir for tools/diag_hash.py.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config  # noqa: E402
from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_noise, synthetic_text_features  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--rebuild", type=int, default=25)
    ap.add_argument("--config", default="mini")
    ap.add_argument("--clips", type=int, default=5)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--precision", default="bf16")
    args = ap.parse_args()
    gpu = torch.device(os.environ.get("SAMAUDIO_TOOL_DEVICE", "cuda:0"))   # ("cpu": dry run on the emulation, tests/conftest.py)
    cfg = preset_config(args.config)
    sd = init_state_dict(cfg, seed=11)
    hop = cfg.audio_codec.hop_length
    n, T = args.clips, args.frames
    clips = [synthetic_clip(i, T * hop) for i in range(n)]
    text, tmask = synthetic_text_features(n, 6, ragged=True)
    proc = SAMAudioProcessor.from_config(cfg)
    batch = proc(descriptions=["x"] * n, audios=clips, text_features=text, text_mask=tmask).to(gpu)
    noise = synthetic_noise(n, T).to(gpu)
    opt = {"method": "midpoint", "options": {"step_size": 0.25}}

    def build(streams):
        m = SAMAudio(cfg, precision=args.precision, device=str(gpu), streams=streams)
        m.load_state_dict(sd, strict=False)
        return m

    one = build(1)
    one.separate(batch, noise=noise, ode_opt=opt)
    ref = one.last_latent.clone()
    g = torch.Generator().manual_seed(0)
    two, bad, junk = None, 0, []
    for rep in range(args.reps):
        if two is None or rep % args.rebuild == 0:
            two = build(2)
        # allocator churn: finite garbage of varying sizes, some kept for a few repetitions
        sizes = torch.randint(1, 1 << 20, (6,), generator=g).tolist()
        junk.append([torch.full((s,), 0.37 + rep, device=gpu) for s in sizes])
        if len(junk) > 3:
            junk.pop(int(torch.randint(0, len(junk), (1,), generator=g)))
        two.separate(batch, noise=noise, ode_opt=opt)
        torch.cuda.synchronize()
        if not torch.equal(two.last_latent, ref):
            bad += 1
            d = (two.last_latent.float() - ref.float()).abs()
            print(f"  repetition {rep}: max |diff| per clip {['%.3g' % x for x in d.flatten(1).max(dim=1).values.tolist()]}, "
                  f"{int((d > 0).sum())} of {d.numel()} elements, finite={bool(torch.isfinite(two.last_latent).all())}", flush=True)
        if rep % 50 == 49:   # the one-stream model again, too: it must still reproduce itself
            one.separate(batch, noise=noise, ode_opt=opt)
            if not torch.equal(one.last_latent, ref):
                bad += 1
                print(f"  repetition {rep}: the ONE-stream model differs from its first result", flush=True)
    mode = "no caching" if os.environ.get("PYTORCH_NO_HIP_MEMORY_CACHING") or os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") else "caching"
    print(f"two streams vs one: {bad} of {args.reps} repetitions differ ({args.config}, {n} clips x {T} frames, "
          f"{args.precision}, allocator: {mode}, poison: {os.environ.get('SAMAUDIO_POISON', '0')})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
