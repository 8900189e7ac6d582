#!/bin/bash
# refresh of the round-end bench line and kernel statistics on the final library (the full round-end session is tools/gpu_session_final.sh) -> gpurun_out/final2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?" >> $O/summary.log
cd /tmp
CTCLIP_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-attn-block --no-reference-depth > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/prof_stats.md 2>&1
import csv, glob, re, collections
rows = collections.defaultdict(list)
for path in glob.glob("gpurun_out/final2/prof/**/*kernel_trace.csv", recursive=True):
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
rm -rf $O/prof
cat $O/summary.log; python -c "
import json
b=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(b['ms_per_step'],b['value'],b['roofline']['frac'],b.get('reference_depth_4+4',{}).get('value'),(b.get('attn_block') or {}).get('fwd_us'),(b.get('attn_block') or {}).get('fwd_bwd_us'))"
head -6 $O/prof_stats.md; grep "patch_ln\|peg_" $O/prof_stats.md
