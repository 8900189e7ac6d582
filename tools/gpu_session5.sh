#!/bin/bash
# evidence session: long determinism trace on the final library, rocprofv3 kernel statistics of the two fine-tuning workloads -> gpurun_out/s5
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/trace_determinism.py --runs 600 2>&1 | grep TRACE_SUMMARY >> $O/summary.log
timeout 900 python tools/trace_determinism.py --runs 60 --config bench 2>&1 | grep TRACE_SUMMARY >> $O/summary.log
for wl in lipro vocabfine; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/$wl.json 2> $GRAFT_REPO_ROOT/$O/$wl.err
  cd $GRAFT_REPO_ROOT
  python - $wl <<'PY' > $O/prof_stats_$wl.md 2>&1
import csv, glob, re, collections, sys
rows = collections.defaultdict(list)
for path in glob.glob(f"gpurun_out/s5/prof_{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n).split("(")[0]
        rows[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"| `{n[:110]}` | {len(v)} | {sum(v)/1e3:.2f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100*sum(v)/tot:.1f} |")
print(f"\ntotal kernel time {tot/1e3:.1f} ms over {sum(len(v) for v in rows.values())} dispatches (6 steps incl. warm-up)")
PY
  rm -rf $O/prof_$wl
done
cat $O/summary.log; head -16 $O/prof_stats_vocabfine.md; tail -n 2 $O/prof_stats_vocabfine.md
