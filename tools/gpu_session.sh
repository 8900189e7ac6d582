#!/bin/bash
# One GPU session through gpurun: the whole GPU suite, the default bench (50 steps, PMC traffic, attention block, CPU baseline), a short
# bench under rocprofv3 --kernel-trace for the per-kernel statistics -> gpurun_out/session
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/session; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > $O/t_all.log 2>&1; echo "all gpu tests rc=$? $(tail -n 1 $O/t_all.log)" >> $O/summary.log
timeout 600 python tools/trace_determinism.py --runs ${TRACE_RUNS:-200} 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/trace200.log; grep TRACE_SUMMARY $O/trace200.log >> $O/summary.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?" >> $O/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
timeout 300 python tools/bench_gemm_shapes.py 10 > $O/shapes.json 2> $O/shapes.err; echo "shapes rc=$?" >> $O/summary.log
cd /tmp
# (per-kernel durations: with the weight-gradient stream off, so that no two big kernels share the chip inside one duration)
CTCLIP_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-attn-block > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/prof_stats.md 2>&1
import csv, glob, re, collections
rows = collections.defaultdict(list)
for path in glob.glob("gpurun_out/session/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n).split("(")[0]
        rows[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:75]:
    print(f"| `{n[:110]}` | {len(v)} | {sum(v)/1e3:.2f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100*sum(v)/tot:.1f} |")
print(f"\ntotal kernel time {tot/1e3:.1f} ms over {sum(len(v) for v in rows.values())} dispatches")
PY
rm -rf $O/prof/*/*.db $O/prof/*/*_agent_info.csv
grep -h "FAILED\|Error" $O/t_all.log | head; cat $O/summary.log; python -c "
import json
b=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(b['ms_per_step'],b['value'],b['roofline']['kernel'],b['roofline']['frac'],b['roofline'].get('traffic_over_algorithmic'),b['attn_block'].get('mfma_util_fwd'),b['cpu_baseline'].get('value'))"
head -12 $O/prof_stats.md
