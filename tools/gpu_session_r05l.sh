#!/bin/bash
# round-5 session L: Adam clearing the gradients itself against the overlapped fill (bench.py knob CTCLIP_ADAM_ZERO)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/run; mkdir -p $O; rm -f $O/summary_l.log
SHORT="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 --no-attn-block"
i=0
for E in "CTCLIP_ADAM_ZERO=1" "CTCLIP_ADAM_ZERO=0" "CTCLIP_ADAM_ZERO=1" "CTCLIP_ADAM_ZERO=0" "CTCLIP_ADAM_ZERO=1" "CTCLIP_ADAM_ZERO=0"; do
  i=$((i+1))
  env $E timeout 600 python bench.py $SHORT > $O/l_ab$i.json 2> $O/l_ab$i.err
  python - <<PY >> $O/summary_l.log
import json
try:
    b=json.loads(open("$O/l_ab$i.json").read().strip().splitlines()[-1]); print("ab$i [$E]", b["ms_per_step"], "ms/step", b["value"], "loss", b["loss"])
except Exception as e:
    print("ab$i failed", e); print(open("$O/l_ab$i.err").read()[-1500:])
PY
done
cat $O/summary_l.log
