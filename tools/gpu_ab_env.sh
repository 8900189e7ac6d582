#!/bin/bash
# Same-box A/B of the training step under two environment settings:  tools/gpu_ab_env.sh "A_ENV=.." "B_ENV=.." [steps]
# (each side twice, interleaved: boxes differ by 3-5 %, only same-session pairs rank a change)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab; mkdir -p $O
A="$1"; B="$2"; N=${3:-20}
for r in 1 2; do
  for side in A B; do
    E="${!side}"
    env $E timeout 600 python bench.py --steps $N --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 > $O/${side}${r}.json 2> $O/${side}${r}.err
    python - <<PY
import json
try:
    b=json.loads(open("$O/${side}${r}.json").read().strip().splitlines()[-1])
    print("$side$r [$E]", b["ms_per_step"], "ms/step", b["value"], b.get("attn_block",{}).get("fwd_us"), b.get("attn_block",{}).get("fwd_bwd_us"))
except Exception as e:
    print("$side$r failed", e); print(open("$O/${side}${r}.err").read()[-1500:])
PY
  done
done
