#!/bin/bash
# per-kernel cost of the compensated residual stream in an inference forward (LiPro workload) + a few tests -> gpurun_out/s4
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_ddp_gpu.py -q -k "adam or finetune_workloads or self_launches" > $O/t.log 2>&1; echo "tests rc=$? $(tail -n 1 $O/t.log)" >> $O/summary.log
for mode in 1 0; do
  cd /tmp
  CTCLIP_RESIDUAL_COMP=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$mode -- python $GRAFT_REPO_ROOT/bench.py --workload lipro --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/$O/lipro$mode.json 2> $GRAFT_REPO_ROOT/$O/prof$mode.err
  cd $GRAFT_REPO_ROOT
  python - $mode <<'PY' > $O/prof_stats$mode.md 2>&1
import csv, glob, re, collections, sys
rows = collections.defaultdict(list)
for path in glob.glob(f"gpurun_out/s4/prof{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n).split("(")[0]
        rows[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|")
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:30]:
    print(f"| `{n[:110]}` | {len(v)} | {sum(v)/1e3:.2f} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100*sum(v)/tot:.1f} |")
print(f"\ntotal kernel time {tot/1e3:.1f} ms over {sum(len(v) for v in rows.values())} dispatches")
PY
  rm -rf $O/prof$mode
done
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-attn-block --profile-steps 0 --no-reference-depth"
timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('train step', r['ms_per_step'], r['loss'])" >> $O/summary.log
cat $O/summary.log; head -22 $O/prof_stats1.md; head -22 $O/prof_stats0.md
