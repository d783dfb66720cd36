#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes
DIR/NAME_results.db on ROCm 7.2) into the per-kernel table `--stats` would print: calls, total / average /
min / max duration and share of GPU time.  Usage: python tools/rocpd_stats.py results.db > profiles/x.md"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)            # drop the argument list
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main(path: str) -> None:
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"source: {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % GPU time |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, mn, mx in rows:
        print(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} |"
              f" {100.0 * tot / total:.2f} |")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    # Is the GPU waiting for the host between launches?  Idle gaps between consecutive dispatches (a gap = the next start minus
    # the latest end so far; overlapping dispatches of concurrent streams give none), split at 100 us: short gaps are launch
    # gaps inside a stream of launches, long ones are host phases (model building, H2D copies, the end of a step).
    iv = db.execute(f"select start, end, {name_col} from kernels order by start").fetchall()
    short_gaps, long_gaps, latest, prev = [], [], iv[0][1] if iv else 0, iv[0][2] if iv else ""
    where = []   # (gap, kernel before, kernel after) of the long gaps: a blocking host read inside separate() shows up here
    for st, en, nm in iv[1:]:
        if st > latest:
            (short_gaps if st - latest < 100_000 else long_gaps).append(st - latest)
            if st - latest >= 100_000:
                where.append((st - latest, short(prev)[:60], short(nm)[:60]))
        if en >= latest:
            latest, prev = en, nm
    if short_gaps:
        sg = sorted(short_gaps)
        busy = total / (total + sum(sg))
        print(f"idle gaps < 100 us between dispatches: {len(sg)} totalling {sum(sg) / 1e6:.3f} ms (median {sg[len(sg) // 2] / 1e3:.2f} us, "
              f"p90 {sg[len(sg) * 9 // 10] / 1e3:.2f} us); gaps >= 100 us: {len(long_gaps)} totalling {sum(long_gaps) / 1e6:.3f} ms; "
              f"GPU busy inside the launch streams = kernel time / (kernel time + short gaps) = {busy:.3f}")
    if where:
        print("\nidle gaps >= 100 us (host phases: model building, the end of a step; a blocking host read INSIDE a step would be listed "
              "here between two kernels of the step), largest first:\n")
        print("| gap ms | last kernel before | first kernel after |")
        print("|---:|---|---|")
        for g, a, b in sorted(where, reverse=True)[:16]:
            print(f"| {g / 1e6:.3f} | `{a}` | `{b}` |")


if __name__ == "__main__":
    main(sys.argv[1])
