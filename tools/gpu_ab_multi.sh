#!/bin/bash
# Same-box comparison of the training step under several environment settings (two interleaved rounds):
#   tools/gpu_ab_multi.sh "ENV_A" "ENV_B" ...      ("-" = no extra environment; ARGS="--spatial-depth 4 ..." = extra bench.py arguments)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abm; mkdir -p $O; rm -f $O/ab.log
for r in 1 2; do
  i=0
  for E in "$@"; do
    i=$((i + 1)); [ "$E" = "-" ] && E="X_NONE=1"
    env $E timeout 600 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --no-attn-block --profile-steps 0 ${ARGS:-} > $O/c${i}_${r}.json 2> $O/c${i}_${r}.err
    python - <<PY >> $O/ab.log
import json
try:
    b=json.loads(open("$O/c${i}_${r}.json").read().strip().splitlines()[-1])
    print("round $r [$E]", b["ms_per_step"], "ms/step  loss", b.get("loss"))
except Exception as e:
    print("round $r [$E] failed", e); print(open("$O/c${i}_${r}.err").read()[-800:])
PY
  done
done
cat $O/ab.log
