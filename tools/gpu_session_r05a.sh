#!/bin/bash
# round-5 session A: parity of the new default precision policy (tests + smoke), the default bench line, and same-box A/Bs of the policy's cost
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_run.sh tests smoke bench
O=gpurun_out/run
SHORT="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 --no-attn-block"
i=0
for E in "CTCLIP_NOOP=1" "CTCLIP_TEXT_DTYPE=bf16 CTCLIP_HEAD_DTYPE=bf16" "CTCLIP_TEXT_DTYPE=bf16" "CTCLIP_HEAD_DTYPE=bf16" "CTCLIP_RESIDUAL_COMP=1" "CTCLIP_NOOP=1" "CTCLIP_TEXT_DTYPE=bf16 CTCLIP_HEAD_DTYPE=bf16"; do
  i=$((i+1))
  env $E timeout 600 python bench.py $SHORT > $O/ab$i.json 2> $O/ab$i.err
  python - <<PY >> $O/summary.log
import json
try:
    b=json.loads(open("$O/ab$i.json").read().strip().splitlines()[-1]); print("ab$i [$E]", b["ms_per_step"], "ms/step", b["value"], "loss", b["loss"])
except Exception as e:
    print("ab$i failed", e); print(open("$O/ab$i.err").read()[-1500:])
PY
done
cat $O/summary.log
