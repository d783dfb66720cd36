#!/usr/bin/env python
"""Where does a two-stream solve first differ from the one-stream solve?  Runs the configuration of
tests/test_path_gpu.py::test_concurrent_streams_are_bitwise_equal_to_one_stream with SAMAUDIO_TRACE_HASH=1 (engine.hip: one
checksum per stage and batch item, written on the launch stream WITHOUT synchronising - the concurrency under test stays
intact), `--reps` times, and compares the per-clip checksum sequence of every two-stream run with the one-stream run.
Prints the first stage / clip whose checksum differs for each repetition that ends with different latents.

    SAMAUDIO_TRACE_HASH=1 python tools/diag_hash.py [--reps 40] 2> hash.log     (stderr carries the checksums; this script
    re-reads them through a pipe it sets up itself when run without the redirect)
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, os
sys.path.insert(0, %(root)r)
if os.environ.get("SAMAUDIO_EMU_DRYRUN"):   # shake-out of this tool on the CPU emulation (tests/conftest.py)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import conftest
import torch
from sam_audio_amd import SAMAudio, SAMAudioProcessor, hip, preset_config
from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_noise, synthetic_text_features
gpu = torch.device(os.environ.get("SAMAUDIO_TOOL_DEVICE", "cuda:0"))
cfg = preset_config("mini")
sd = init_state_dict(cfg, seed=11)
hop = cfg.audio_codec.hop_length
clips = [synthetic_clip(i, 12 * hop) for i in range(5)]
text, tmask = synthetic_text_features(5, 6, ragged=True)
proc = SAMAudioProcessor.from_config(cfg)
batch = proc(descriptions=["x"] * 5, audios=clips, text_features=text, text_mask=tmask).to(gpu)
noise = synthetic_noise(5, 12).to(gpu)
opt = {"method": "midpoint", "options": {"step_size": 0.25}}
def model(streams):
    m = SAMAudio(cfg, precision="bf16", device=str(gpu), streams=streams)
    m.load_state_dict(sd, strict=False)
    return m
one = model(1)
print("RUN one", file=sys.stderr, flush=True)
one.separate(batch, noise=noise, ode_opt=opt); torch.cuda.synchronize()
ref = one.last_latent.clone()
two = None
junk = []
for rep in range(%(reps)d):
    if two is None or (%(rebuild)d and rep %% %(rebuild)d == 0):
        # a fresh model: new lanes, new workspaces - on memory that last held finite garbage of another magnitude
        junk = [torch.full((s,), 200.37 + rep, device=gpu) for s in (1 << 18, 1 << 20, 3 << 20)]
        del junk
        two = model(2)
    print(f"RUN two {rep}", file=sys.stderr, flush=True)
    two.separate(batch, noise=noise, ode_opt=opt); torch.cuda.synchronize()
    d = (two.last_latent.float() - ref.float()).abs().flatten(1).max(dim=1).values.tolist()
    print(f"RESULT {rep} " + " ".join("%%.3g" %% x for x in d), file=sys.stderr, flush=True)
'''


def parse(lines):
    """-> {run name: {context: [(seq, stage, item, items, hash)]}}, {rep: diffs}"""
    runs, results, cur = {}, {}, None
    for ln in lines:
        if ln.startswith("RUN "):
            cur = ln.split(None, 1)[1].strip()
            runs[cur] = {}
        elif ln.startswith("RESULT "):
            parts = ln.split()
            results[int(parts[1])] = [float(x) for x in parts[2:]]
        else:
            m = re.match(r"\[samaudio hash\] (\S+) (\d+) (.*) item (\d+) of (\d+) ([0-9a-f]{16})", ln)
            if m and cur is not None:
                ctx, seq, stage, item, items, h = m.groups()
                runs[cur].setdefault(ctx, []).append((int(seq), stage, int(item), int(items), h))
    return runs, results


def per_clip(run):
    """{clip: {(stage, occurrence): (seq, hash)}}: contexts ordered by size (the 3-clip group holds clips 0-2, the 2-clip
    group clips 3-4); only stages recorded once per batch item of the context are comparable between shardings"""
    ctxs = sorted(run.items(), key=lambda kv: -max(r[3] for r in kv[1]))
    out, base = {}, 0
    for _, recs in ctxs:
        n = max(r[3] for r in recs)
        seen = {}
        for seq, stage, item, items, h in recs:
            if items != n:
                continue
            k = seen.get((stage, item), 0)
            seen[(stage, item)] = k + 1
            out.setdefault(base + item, {})[(stage, k)] = (seq, h)
        base += n
    return out


def first_difference(ref, got):
    first = None
    for clip in sorted(ref):
        a, b = ref[clip], got.get(clip, {})
        for key in sorted(a, key=lambda k: a[k][0]):   # in the order the reference run recorded them
            if key in b and a[key][1] != b[key][1]:
                cand = (a[key][0], clip, key)
                if first is None or cand[0] < first[0]:
                    first = cand
                break
    return first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--rebuild", type=int, default=0, help="build the two-stream model anew every N repetitions (0: once)")
    ap.add_argument("--keep", default="", help="directory for the raw checksum lines of every differing repetition (+ the reference)")
    args = ap.parse_args()
    env = dict(os.environ, SAMAUDIO_TRACE_HASH="1")
    # the child's stderr is read as it comes and a repetition's records are dropped once it has been compared: thousands of
    # repetitions (a difference shows up once in several hundred) fit in memory
    p = subprocess.Popen([sys.executable, "-c", CHILD % dict(root=ROOT, reps=args.reps, rebuild=args.rebuild)], env=env,
                         stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
    ref, ref_lines, cur_name, cur_lines, n, bad, tail = None, [], None, [], 0, 0, []
    for ln in p.stderr:
        ln = ln.rstrip("\n")
        tail = (tail + [ln])[-30:]
        if ln.startswith("RUN "):
            cur_name, cur_lines = ln.split(None, 1)[1].strip(), [ln]
            continue
        cur_lines.append(ln)
        if cur_name == "one" and ln.startswith("[samaudio hash]"):
            ref_lines.append(ln)
        if not ln.startswith("RESULT "):
            continue
        if ref is None:
            ref = per_clip(parse(["RUN one"] + ref_lines)[0]["one"])
        runs, results = parse(cur_lines)
        (rep, diffs), = results.items()
        first = first_difference(ref, per_clip(runs.get(cur_name, {})))
        n += 1
        if any(d > 0 for d in diffs) or first is not None:
            bad += 1
            status = "DIFF" if any(d > 0 for d in diffs) else "same"
            print(f"rep {rep}: latents {status} {diffs}; first differing checksum (sequence number, clip, (stage, occurrence)): {first}", flush=True)
            if args.keep:
                os.makedirs(args.keep, exist_ok=True)
                open(os.path.join(args.keep, f"rep{rep}.txt"), "w").write("\n".join(cur_lines) + "\n")
                open(os.path.join(args.keep, "reference.txt"), "w").write("\n".join(ref_lines) + "\n")
        cur_lines = []
    p.wait()
    if ref is None:
        print("\n".join(tail))
        sys.exit(1)
    print(f"{n} two-stream repetitions, {bad} with differences; stages per clip in the reference: {len(next(iter(ref.values())))}")


if __name__ == "__main__":
    main()
