#!/bin/bash
# One GPU session through gpurun: kernel tests, the whole GPU suite, a short bench, rocprofv3 kernel statistics of the step -> gpurun_out/<tag>
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/session; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "geglu or gemm or latent or vlat or visual" > $O/t_geglu.log 2>&1; echo "geglu tests rc=$? $(tail -n 1 $O/t_geglu.log)" >> $O/summary.log
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all gpu tests rc=$? $(tail -n 1 $O/t_all.log)" >> $O/summary.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc --no-attn-block > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-attn-block > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
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
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:70]:
    print(f"| `{n[:110]}` | {len(v)} | {sum(v)/1e3:.2f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100*sum(v)/tot:.1f} |")
print(f"\ntotal kernel time {tot/1e3:.1f} ms over {sum(len(v) for v in rows.values())} dispatches")
PY
rm -rf $O/prof/*/*.db
grep -h "FAILED\|Error" $O/t_geglu.log $O/t_all.log | head; cat $O/summary.log; python -c "
import json
b=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(b['ms_per_step'],b['value'])"
head -14 $O/prof_stats.md
