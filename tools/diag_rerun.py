#!/usr/bin/env python
"""Run-to-run determinism of one prepare + solve at the reference's default width (the scenario of
tests/test_path_gpu.py::test_batch_sharding_is_bitwise_invariant_at_full_width), with per-stage checksums
(SAMAUDIO_TRACE_HASH=1): prints the first stage whose per-clip checksum differs between consecutive runs.
    python tools/diag_rerun.py [--runs 4] [--flags 27=1]"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import torch
from sam_audio_amd import SAMAudio, hip, preset_config
from sam_audio_amd.synthetic import init_state_dict, synthetic_noise
gpu = torch.device("cuda:0")
for fl in %(flags)r:
    k, v = fl.split("=")
    hip.lib().samaudio_debug_set_flag(int(k), int(v))
cfg = preset_config("default", transformer=dict(n_layers=2))
sd = init_state_dict(cfg, seed=10, device=gpu, with_codec=False)
B, T = 4, 250
g = torch.Generator().manual_seed(4)
z = torch.randn(B, T, 128, generator=g)
feats, text = torch.cat([z, z], 2), torch.randn(B, 8, 768, generator=g)
noise = synthetic_noise(B, T)
model = SAMAudio(cfg, precision="bf16", device=str(gpu))
model.load_state_dict(sd, strict=False)
opt = {"method": "midpoint", "options": {"step_size": 0.5}}
ref = None
for r in range(%(runs)d):
    print(f"RUN {r}", file=sys.stderr, flush=True)
    model._prepare(feats, text, None, None, None, None, None)
    out = model.solve(noise.to(gpu), opt)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    d = (out - ref).abs().flatten(1).max(dim=1).values.tolist()
    print(f"RESULT {r} " + " ".join("%%.3g" %% x for x in d), file=sys.stderr, flush=True)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--flags", nargs="*", default=[])
    args = ap.parse_args()
    env = dict(os.environ, SAMAUDIO_TRACE_HASH="1")
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, runs=args.runs, flags=args.flags)], env=env,
                       capture_output=True, text=True)
    runs, cur, results = {}, None, {}
    for ln in p.stderr.splitlines():
        if ln.startswith("RUN "):
            cur = int(ln.split()[1]); runs[cur] = []
        elif ln.startswith("RESULT "):
            parts = ln.split(); results[int(parts[1])] = parts[2:]
        else:
            m = re.match(r"\[samaudio hash\] (\S+) (\d+) (.*) item (\d+) of (\d+) ([0-9a-f]{16})", ln)
            if m and cur is not None:
                runs[cur].append((m.group(3), int(m.group(4)), m.group(6)))
    if not runs:
        print(p.stderr[-3000:]); sys.exit(1)
    print("flags", args.flags, "max |diff| per clip vs run 0:", results)
    base = runs[0]
    for r in sorted(runs)[1:]:
        first = next(((i, a[0], a[1]) for i, (a, b) in enumerate(zip(base, runs[r])) if a != b), None)
        n_diff = sum(1 for a, b in zip(base, runs[r]) if a != b)
        print(f"run {r} vs run 0: {len(runs[r])} records, {n_diff} differ; first differing record: {first}")


if __name__ == "__main__":
    main()
