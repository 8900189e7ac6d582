#!/bin/bash
# HBM traffic of every kernel of the training step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, the TCC block cannot
# hold both) over a 3-step bench (12+12 layers, B = 8, weight-gradient stream off) -> gpurun_out/pmc_step/traffic.md
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_step; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  CTCLIP_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-pmc --no-attn-block --no-reference-depth > /dev/null 2> $GRAFT_REPO_ROOT/$O/$c.err
done
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/traffic.md 2>&1
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"gpurun_out/pmc_step/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != c:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            n = re.sub(r"^void ", "", n).split("(")[0]
            acc[n][c].append(float(r["Counter_Value"]))
            acc[n]["us_" + c].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for n, d in acc.items():
    if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
        continue
    calls = len(d["FETCH_SIZE"])
    f = sum(d["FETCH_SIZE"]) / calls * 1024 * 2          # KiB units; x2: gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM)
    w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024
    us = sum(d["us_FETCH_SIZE"]) / calls
    rows.append((calls * (f + w), n, calls, us, f, w))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("| kernel | launches (3 steps) | avg µs (under the counter pass) | fetched MB / launch (×2) | written MB / launch | GB/s | share of all HBM bytes |\n|---|---:|---:|---:|---:|---:|---:|")
for t, n, calls, us, f, w in rows[:32]:
    print(f"| `{n[:100]}` | {calls} | {us:.1f} | {f/1e6:.1f} | {w/1e6:.1f} | {(f+w)/us/1e3:.0f} | {100*t/tot:.1f} % |")
print(f"\nall kernels: {tot/3/1e9:.1f} GB of HBM traffic per optimisation step")
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
cat $O/traffic.md | head -40; tail -n 3 $O/FETCH_SIZE.err
