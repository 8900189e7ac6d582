#!/bin/bash
# Round-end GPU session: whole GPU suite, smoke, determinism trace, the default bench (PMC traffic, attention block, CPU baseline, 4+4 depth),
# the fine-tuning workloads, a short bench under rocprofv3 --kernel-trace, attention PMC passes -> gpurun_out/final
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > $O/t_all.log 2>&1; echo "all gpu tests rc=$? $(tail -n 1 $O/t_all.log)" >> $O/summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.log
timeout 600 python tools/trace_determinism.py --runs 200 2>&1 | grep TRACE_SUMMARY >> $O/summary.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?" >> $O/summary.log
timeout 600 python bench.py --workload lipro > $O/bench_lipro.json 2> $O/bench_lipro.err; echo "lipro bench rc=$?" >> $O/summary.log
timeout 600 python bench.py --workload vocabfine > $O/bench_vocabfine.json 2> $O/bench_vocabfine.err; echo "vocabfine bench rc=$?" >> $O/summary.log
cd /tmp
CTCLIP_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-attn-block --no-reference-depth > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/prof_stats.md 2>&1
import csv, glob, re, collections
rows = collections.defaultdict(list)
for path in glob.glob("gpurun_out/final/prof/**/*kernel_trace.csv", recursive=True):
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
rm -rf $O/prof/*/*.db $O/prof/*/*_agent_info.csv $O/prof/*/*kernel_trace.csv
bash tools/gpu_pmc_attn2.sh > $O/attn_pmc.txt 2>&1
cat $O/summary.log; grep -h "FAILED\|^E  " $O/t_all.log | head; tail -n 5 $O/smoke.log; for f in default lipro vocabfine; do python -c "
import json
b=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]);print('$f',b['ms_per_step'],b['value'],b['roofline']['kernel'],b['roofline']['frac'],b.get('reference_depth_4+4',{}).get('value'),(b.get('attn_block') or {}).get('mfma_util_fwd'),b['cpu_baseline'].get('value'))"; done
head -8 $O/prof_stats.md; tail -n 12 $O/attn_pmc.txt | cut -c1-400
