#!/usr/bin/env python
"""SQ counter summary per kernel from the rocprofv3 --pmc passes of tools/pmc_gemm.sh (csv), as fractions of the kernel's
wave cycles.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD with an MFMA in flight (MI355X_MICROARCH.md "rocprofv3 PMC slots").
GRBM_GUI_ACTIVE comes summed over the 8 XCDs (6.85 M per ~394 us launch = 8 x 0.857 M cycles at 2.17 GHz), so
MFMA busy = MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8).  usage: python tools/pmc_sq.py gpurun_out/pmc_gemm"""
import collections
import csv
import glob
import re
import sys


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
            if not name.startswith("sa::"):
                continue
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name][r["Counter_Name"]] += 1
    for name in acc:   # per-launch averages: a counter collected in several passes must not be counted twice
        for k in acc[name]:
            acc[name][k] /= launches[name][k]
        print("raw per launch:", name[:50], {k: round(v) for k, v in acc[name].items()}, file=sys.stderr)
    print("| kernel | launches | MFMA busy | wave cycles: waiting (s_waitcnt / barrier) | issue stall | issuing | LDS bank-conflict cycles / LDS active |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, c in acc.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        f = lambda k: f"{100 * c[k] / wc:.1f} %" if wc and k in c else "-"   # noqa: E731
        mfma = f"{100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * 256 * gui / 8):.1f} %" if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in c else "-"
        lds = (f"{100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} %"
               if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c else "-")
        n = max(launches[name].values())
        print(f"| `{name[:70]}` | {n} | {mfma} | {f('SQ_WAIT_ANY')} | {f('SQ_WAIT_INST_ANY')} | {f('SQ_ACTIVE_INST_ANY')} | {lds} |")


if __name__ == "__main__":
    main(sys.argv[1])
