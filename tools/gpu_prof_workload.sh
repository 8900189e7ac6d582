#!/bin/bash
# rocprofv3 --kernel-trace over bench.py with the arguments given ("--workload vocabfine", "--layers 4 ..."): per-kernel table -> gpurun_out/pw/stats.md
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/pw; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-attn-block --no-reference-depth --no-text512 "$@" > $GRAFT_REPO_ROOT/$O/b.json 2> $GRAFT_REPO_ROOT/$O/b.err)
python - <<'PY' > $O/stats.md
import csv, glob, re, collections
rows = collections.defaultdict(list)
for path in glob.glob("gpurun_out/pw/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
        rows[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values()); nl = sum(len(v) for v in rows.values())
print(f"total kernel time {tot/1e3:.2f} ms over 6 steps, {nl} launches\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:70]:
    print(f"| `{n[:110]}` | {len(v)} | {sum(v)/1e3:.2f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100*sum(v)/tot:.1f} |")
PY
rm -rf $O/prof
