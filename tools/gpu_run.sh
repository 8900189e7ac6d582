#!/bin/bash
# ONE parametrised GPU session (through gpurun) instead of the per-round scripts of rounds 1-3:
#   tools/gpu_run.sh [tests[=PYTEST_EXPR]] [soak[=REPS[:NAME]]] [bench[=EXTRA_ARGS]] [prof] [smoke] [attnbwd]
# Every stage writes under gpurun_out/run/ and adds a line to summary.log; stages run in the order given.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/run; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
for stage in "$@"; do
  name=${stage%%=*}; arg=""; [ "$name" != "$stage" ] && arg=${stage#*=}
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout 1800 python -m pytest tests -q -m gpu -s -k "$arg" > $O/tests.log 2>&1
      else timeout 1800 python -m pytest tests -q -m gpu -s > $O/tests.log 2>&1; fi
      echo "tests[$arg] rc=$? $(tail -n 1 $O/tests.log)" >> $O/summary.log
      grep -h "^\[\|^FAILED\|^ERROR" $O/tests.log | head -60 >> $O/summary.log ;;
    soak)
      reps=${arg%%:*}; nm=""; [ "$reps" != "$arg" ] && nm=${arg#*:}
      timeout 900 python tools/soak_kernels.py ${reps:-100} "$nm" > $O/soak.log 2>&1; echo "soak rc=$?" >> $O/summary.log; grep -h "deviating" $O/soak.log >> $O/summary.log ;;
    bench)
      timeout 1200 python bench.py $arg > $O/bench.json 2> $O/bench.err; echo "bench[$arg] rc=$?" >> $O/summary.log
      python - <<PY >> $O/summary.log 2>&1
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r=b.get("roofline",{}); a=b.get("attn_block",{})
print("  ms/step", b["ms_per_step"], "value", b["value"], "| roofline", r.get("kernel"), r.get("frac"), "| attn_block", a.get("fwd_us"), a.get("fwd_bwd_us"), a.get("mfma_util_fwd"), a.get("mfma_util_fwd_bwd"), "| 4+4", (b.get("reference_depth_4+4") or {}).get("ms_per_step"), "| cpu", (b.get("cpu_baseline") or {}).get("value"))
PY
      ;;
    prof)   # per-kernel durations with the weight-gradient stream off (one kernel per duration)
      (cd /tmp && CTCLIP_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-attn-block --no-reference-depth --no-text512 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err)
      python - <<'PY' > $O/prof_stats.md 2>&1
import csv, glob, re, collections
rows = collections.defaultdict(list)
for path in glob.glob("gpurun_out/run/prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n).split("(")[0]
        rows[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:80]:
    print(f"| `{n[:110]}` | {len(v)} | {sum(v)/1e3:.2f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100*sum(v)/tot:.1f} |")
print(f"\ntotal kernel time {tot/1e3:.1f} ms over {sum(len(v) for v in rows.values())} dispatches (6 steps: 2 warm-up + 4 timed)")
PY
      rm -rf $O/prof/*/*.db $O/prof/*/*_agent_info.csv; echo "prof done" >> $O/summary.log; head -14 $O/prof_stats.md >> $O/summary.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? $(grep -h 'bf16\|f32' $O/smoke.log | tail -n 3 | tr '\n' ' ')" >> $O/summary.log ;;
    attnbwd)
      timeout 300 python tools/bench_attn2_bwd.py 20 > $O/attn2_bwd.json 2>> $O/attn2_bwd.err; echo "attnbwd $(cat $O/attn2_bwd.json)" >> $O/summary.log ;;
    *) echo "unknown stage $stage" >> $O/summary.log ;;
  esac
done
cat $O/summary.log
