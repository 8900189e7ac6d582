#!/bin/bash
# round-2 GPU session E: deterministic reductions, dBias v2, full test suite, timings, kernel statistics of the step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all gpu tests rc=$?" >> $O/summary.log
timeout 120 python tools/bench_attn2_fwd.py 20 all > $O/attn2.json 2>> $O/abl.err
timeout 200 python tools/bench_ops.py attn2 10 > $O/ops_attn2.json 2> $O/ops_attn2.err
timeout 300 python tools/bench_ops.py peg 10 > $O/ops_peg.json 2> $O/ops_peg.err
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-attn-block > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/prof_stats.md 2>&1
import csv, glob, re, collections
rows = collections.defaultdict(list)
for path in glob.glob("gpurun_out/r2e/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n).split("(")[0]
        rows[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:60]:
    print(f"| `{n[:110]}` | {len(v)} | {sum(v)/1e3:.2f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100*sum(v)/tot:.1f} |")
print(f"\ntotal kernel time {tot/1e3:.1f} ms over {sum(len(v) for v in rows.values())} dispatches")
PY
rm -rf $O/prof/*/*.db
tail -n 12 $O/t_all.log; cat $O/summary.log $O/attn2.json; python -c "
import json;d=json.load(open('$O/ops_attn2.json'));print({k:v['avg_us'] for k,v in d.items()})
d=json.load(open('$O/ops_peg.json'));print({k:v['avg_us'] for k,v in d.items()})
b=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(b['ms_per_step'],b['value'],{k:b['attn_block'][k] for k in ('fwd_us','fwd_bwd_us','mfma_util_fwd','mfma_util_fwd_bwd')})"
head -30 $O/prof_stats.md; tail -n 3 $O/bench.err $O/prof.err
