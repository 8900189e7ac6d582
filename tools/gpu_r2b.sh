#!/bin/bash
# GPU session (round 2, second half): tests of the changed kernels, GEMM shape timings with / without the counted epilogue wait,
# step-time A/B of the GEGLU recomputation and the batched shadow refresh, rocprofv3 kernel statistics of the new default.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or geglu or shadow" > $O/t_kernels.log 2>&1; echo "kernel tests rc=$? $(tail -n 1 $O/t_kernels.log)" >> $O/summary.log
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_finetune_gpu.py -q -x > $O/t_e2e.log 2>&1; echo "e2e tests rc=$? $(tail -n 1 $O/t_e2e.log)" >> $O/summary.log
timeout 300 python tools/bench_gemm_shapes.py 10 > $O/shapes_new.json 2> $O/shapes_new.err; echo "shapes new rc=$?" >> $O/summary.log
CTCLIP_LIB=ct_clip_amd/libctclip_abl0_NT_COUNTED_EPI0.so timeout 300 python tools/bench_gemm_shapes.py 10 > $O/shapes_uncounted.json 2> $O/shapes_uncounted.err; echo "shapes uncounted rc=$?" >> $O/summary.log
B="--steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-attn-block"
timeout 600 python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "bench new rc=$?" >> $O/summary.log
CTCLIP_GEGLU_RECOMPUTE=0 timeout 600 python bench.py $B > $O/bench_storedu.json 2> $O/bench_storedu.err; echo "bench stored-u rc=$?" >> $O/summary.log
CTCLIP_SHADOW_BATCH=0 timeout 600 python bench.py $B > $O/bench_lazyshadow.json 2> $O/bench_lazyshadow.err; echo "bench lazy-shadow rc=$?" >> $O/summary.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-attn-block > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/prof_stats.md 2>&1
import csv, glob, re, collections
rows = collections.defaultdict(list)
for path in glob.glob("gpurun_out/r2b/prof/**/*kernel_trace.csv", recursive=True):
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
grep -h "FAILED\|Error" $O/t_kernels.log $O/t_e2e.log | head; cat $O/summary.log
python - <<'PY'
import json
for n in ("new", "storedu", "lazyshadow"):
    try:
        b = json.loads(open(f"gpurun_out/r2b/bench_{n}.json").read().strip().splitlines()[-1]); print(n, b["ms_per_step"], b["value"], b["loss"])
    except Exception as e:
        print(n, "failed", e)
PY
head -30 $O/prof_stats.md
