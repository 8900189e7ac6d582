"""Summarise rocprofv3 --pmc CSV passes (one directory per pass) per (kernel, grid size) into markdown.
usage: python tools/pmc_summary.py gpurun_out/pmc out.md
Corrections follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B and on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads (x2); GRBM_GUI_ACTIVE is summed over the 8 XCDs;
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs."""
import collections
import csv
import glob
import re
import sys

root, out = sys.argv[1], sys.argv[2]
data = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        if not name.startswith(("gemm", "attn", "peg", "layernorm")):
            continue
        key = (name, int(r["Grid_Size"]))
        data[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        data[key]["_dur_" + r["Counter_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["| kernel | grid | launches | avg us (profiled) | FETCH MB (x2 corrected) | WRITE MB | HBM GB/s | MFMA busy % |", "|---|---:|---:|---:|---:|---:|---:|---:|"]
for (name, grid), c in sorted(data.items()):
    avg = lambda k: sum(c[k]) / len(c[k]) if k in c else None
    f, w = avg("FETCH_SIZE"), avg("WRITE_SIZE")
    dur = avg("_dur_FETCH_SIZE") or avg("_dur_SQ_VALU_MFMA_BUSY_CYCLES")
    fm = f * 1024 * 2 / 1e6 if f is not None else None
    wm = w * 1024 / 1e6 if w is not None else None
    bw = (fm + wm) / dur * 1e3 if (fm is not None and wm is not None and dur) else None
    mf, ga = avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("GRBM_GUI_ACTIVE")
    util = 100 * mf / (ga / 8 * 1024) if mf and ga else None
    if util is None and mf:   # no GRBM_GUI_ACTIVE pass: busy cycles per SIMD over wall time at the nominal 2.4 GHz (a lower bound)
        dm = avg("_dur_SQ_VALU_MFMA_BUSY_CYCLES")
        util = 100 * (mf / 1024) / (dm * 1e-6 * 2.4e9) if dm else None
    fmt = lambda v, p=1: "-" if v is None else f"{v:.{p}f}"
    lines.append(f"| `{name}` | {grid} | {len(c.get('FETCH_SIZE', c.get('SQ_VALU_MFMA_BUSY_CYCLES', [])))} | {fmt(dur)} | {fmt(fm)} | {fmt(wm)} | {fmt(bw, 0)} | {fmt(util)} |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
