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


if __name__ == "__main__":
    main(sys.argv[1])
