"""Per-queue timeline of the LAST training step in a rocprofv3 kernel trace (tools/gpu_trace_streams.sh): which HSA queue every stream's
kernels ran on, how busy each queue was, and where the main queue (the one with the image tower's GEMMs) sat idle.

    python tools/trace_timeline.py <dir with *kernel_trace.csv> <label> [compact.csv.gz]

Writes a markdown summary to stdout; the optional third argument receives one row per dispatch of the last step (short name, queue, stream,
start us relative to the step, duration us) for offline inspection."""
import collections
import csv
import glob
import gzip
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).split("(")[0]
    return n[:70]


def main():
    d, label = sys.argv[1], sys.argv[2]
    rows = []
    for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), int(r["Stream_Id"]), short(r["Kernel_Name"])))
    rows.sort()
    # the optimiser kernel closes a step
    ends = [i for i, r in enumerate(rows) if r[4].startswith("adam_kernel")]
    assert len(ends) >= 2, "need two optimiser launches to bracket a step"
    a, b = ends[-2] + 1, ends[-1] + 1
    step = rows[a:b]
    t0, t1 = rows[ends[-2]][1], rows[ends[-1]][1]
    print(f"## {label}: last step = {(t1 - t0) / 1e6:.3f} ms (optimiser end to optimiser end), {len(step)} dispatches\n")
    byq = collections.defaultdict(list)
    for r in step:
        byq[(r[2], r[3])].append(r)
    print("| queue | stream | dispatches | busy ms | first kernel at ms | last end at ms | top kernels |\n|---:|---:|---:|---:|---:|---:|---|")
    for (q, s), v in sorted(byq.items(), key=lambda kv: -sum(x[1] - x[0] for x in kv[1])):
        busy = sum(x[1] - x[0] for x in v) / 1e6
        top = collections.Counter()
        for x in v:
            top[x[4]] += x[1] - x[0]
        names = ", ".join(f"{n[:36]} {t / 1e6:.1f}" for n, t in top.most_common(3))
        print(f"| {q} | {s} | {len(v)} | {busy:.2f} | {(v[0][0] - t0) / 1e6:.2f} | {(max(x[1] for x in v) - t0) / 1e6:.2f} | {names} |")
    # the main queue: the one with the most busy time
    mq = max(byq, key=lambda k: sum(x[1] - x[0] for x in byq[k]))
    v = byq[mq]
    gaps = []
    for x, y in zip(v, v[1:]):
        g = y[0] - x[1]
        if g > 0:
            gaps.append((g, x, y))
    tot = sum(g for g, _, _ in gaps)
    print(f"\nmain queue {mq}: idle between consecutive kernels {tot / 1e6:.2f} ms in {len(gaps)} gaps; gaps > 20 us: "
          f"{sum(g for g, _, _ in gaps if g > 20000) / 1e6:.2f} ms in {sum(1 for g, _, _ in gaps if g > 20000)}\n")
    print("| gap us | at ms | after | before | other queues' kernels inside the gap |\n|---:|---:|---|---|---|")
    for g, x, y in sorted(gaps, key=lambda t: -t[0])[:25]:
        inside = collections.Counter()
        for r in step:
            if (r[2], r[3]) != mq and r[0] < y[0] and r[1] > x[1]:
                inside[f"q{r[2]}/s{r[3]} {r[4][:28]}"] += 1
        ins = "; ".join(f"{k} x{c}" for k, c in inside.most_common(4))
        print(f"| {g / 1e3:.1f} | {(x[1] - t0) / 1e6:.2f} | {x[4][:40]} | {y[4][:40]} | {ins} |")
    if len(sys.argv) > 3:
        with gzip.open(sys.argv[3], "wt") as f:
            f.write("name,queue,stream,start_us,dur_us\n")
            for r in step:
                f.write(f"\"{r[4]}\",{r[2]},{r[3]},{(r[0] - t0) / 1e3:.1f},{(r[1] - r[0]) / 1e3:.1f}\n")      # (kernel names contain commas)


if __name__ == "__main__":
    main()
