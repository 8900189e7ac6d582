#!/bin/bash
# the text tower's cost under the image tower (side-stream interference) -> gpurun_out/s10
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s10; mkdir -p $O
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --profile-steps 0 --no-reference-depth"
for i in 1 2; do
  timeout 300 python tools/probe_text_cost.py $AB 2>$O/err_a.log | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('with text tower   ', r['ms_per_step'])" >> $O/ab.log
  PROBE_SKIP_TEXT=1 timeout 300 python tools/probe_text_cost.py $AB 2>$O/err_b.log | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('text tower stubbed', r['ms_per_step'])" >> $O/ab.log
  CTCLIP_TEXT_STREAM=0 timeout 300 python tools/probe_text_cost.py $AB 2>$O/err_c.log | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('text tower serial ', r['ms_per_step'])" >> $O/ab.log
done
cat $O/ab.log; tail -n 3 $O/err_b.log
