#!/bin/bash
# round-5 session F: the round's record run -- full GPU suite, smoke, default bench (roofline + PMC + attn_block + 4+4 + T=512 + cpu baseline),
# rocprofv3 kernel statistics, the two fine-tuning workloads, what the text tower costs the step
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_run.sh tests smoke bench prof
O=gpurun_out/run
cp $O/summary.log $O/summary_main.log
timeout 600 python bench.py --workload lipro > $O/bench_lipro.json 2> $O/bench_lipro.err; echo "lipro rc=$?" >> $O/summary_main.log
timeout 600 python bench.py --workload vocabfine > $O/bench_vocabfine.json 2> $O/bench_vocabfine.err; echo "vocabfine rc=$?" >> $O/summary_main.log
SHORT="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 --no-attn-block"
for skip in 0 1 0 1; do
  PROBE_SKIP_TEXT=$skip timeout 600 python tools/probe_text_cost.py $SHORT > $O/probe_$skip.json 2> $O/probe_$skip.err
  python - <<PY >> $O/summary_main.log
import json
try:
    b=json.loads(open("$O/probe_$skip.json").read().strip().splitlines()[-1]); print("probe skip_text=$skip", b["ms_per_step"], "ms/step")
except Exception as e:
    print("probe $skip failed", e); print(open("$O/probe_$skip.err").read()[-800:])
PY
done
for skip in 0 1; do
  PROBE_SKIP_TEXT=$skip timeout 600 python tools/probe_text_cost.py $SHORT --spatial-depth 4 --temporal-depth 4 > $O/probe44_$skip.json 2> $O/probe44_$skip.err
  python - <<PY >> $O/summary_main.log
import json
try:
    b=json.loads(open("$O/probe44_$skip.json").read().strip().splitlines()[-1]); print("probe 4+4 skip_text=$skip", b["ms_per_step"], "ms/step")
except Exception as e:
    print("probe44 $skip failed", e); print(open("$O/probe44_$skip.err").read()[-800:])
PY
done
python - <<'PY' >> $O/summary_main.log
import json
for w in ("lipro", "vocabfine"):
    try:
        b = json.loads(open(f"gpurun_out/run/bench_{w}.json").read().strip().splitlines()[-1])
        print(w, b["value"], "volumes/s", b["ms_per_step"], "ms/step roofline", b.get("roofline", {}).get("kernel"), b.get("roofline", {}).get("frac"))
    except Exception as e:
        print(w, "failed", e)
PY
cat $O/summary_main.log
