#!/bin/bash
# same-box A/B of two trees: _old (a git worktree of the previous state) against the working tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab; mkdir -p $O
B="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-attn-block"
for rep in 1 2; do
  (cd _old && timeout 600 python bench.py $B > ../$O/old_$rep.json 2> ../$O/old_$rep.err)
  timeout 600 python bench.py $B > $O/new_$rep.json 2> $O/new_$rep.err
done
python - <<'PY'
import json
for n in ("old_1", "new_1", "old_2", "new_2"):
    try:
        b = json.loads(open(f"gpurun_out/ab/{n}.json").read().strip().splitlines()[-1]); print(n, b["ms_per_step"], b["value"], b["loss"])
    except Exception as e:
        print(n, "failed", e)
PY
