#!/bin/bash
# round-5 session D: LayerNorm-backward fold on the side stream + vectorised gemm_sm epilogue: full suite, kernel stats, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/run; mkdir -p $O; rm -f $O/summary.log
timeout 1500 python -m pytest tests -q -m gpu -x -s > $O/tests.log 2>&1
echo "tests rc=$? $(tail -n 1 $O/tests.log)" >> $O/summary.log
grep -h "^FAILED\|^ERROR\|Error" $O/tests.log | head -20 >> $O/summary.log
timeout 300 python tools/bench_gemm_sm.py 128 > $O/sm_shapes_v.jsonl 2> $O/sm_shapes_v.err; tail -n 1 $O/sm_shapes_v.jsonl >> $O/summary.log
SHORT="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 --no-attn-block"
i=0
for E in "CTCLIP_LN_REDUCE_SIDE=1" "CTCLIP_LN_REDUCE_SIDE=0" "CTCLIP_LN_REDUCE_SIDE=1" "CTCLIP_LN_REDUCE_SIDE=0"; do
  i=$((i+1))
  env $E timeout 600 python bench.py $SHORT > $O/ln_ab$i.json 2> $O/ln_ab$i.err
  python - <<PY >> $O/summary.log
import json
try:
    b=json.loads(open("$O/ln_ab$i.json").read().strip().splitlines()[-1]); print("ab$i [$E]", b["ms_per_step"], "ms/step", b["value"], "loss", b["loss"])
except Exception as e:
    print("ab$i failed", e); print(open("$O/ln_ab$i.err").read()[-1500:])
PY
done
bash tools/gpu_run.sh prof > /dev/null 2>&1
cat $O/summary.log
