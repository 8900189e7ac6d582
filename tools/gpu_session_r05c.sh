#!/bin/bash
# round-5 session C: gemm_sm.hip (text-tower GEMM sizes): full GPU suite, isolated shapes with / without, in-step A/B incl. T = 512
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/run; mkdir -p $O; rm -f $O/summary.log
timeout 1500 python -m pytest tests -q -m gpu -x -s > $O/tests.log 2>&1
echo "tests rc=$? $(tail -n 1 $O/tests.log)" >> $O/summary.log
grep -h "^FAILED\|^ERROR\|Error" $O/tests.log | head -20 >> $O/summary.log
for sm in 1 0; do
  CTCLIP_GEMM_SM=$sm timeout 300 python tools/bench_gemm_sm.py 128 > $O/sm_shapes_$sm.jsonl 2> $O/sm_shapes_$sm.err; echo "shapes sm=$sm rc=$?" >> $O/summary.log; cat $O/sm_shapes_$sm.jsonl >> $O/summary.log
done
CTCLIP_GEMM_SM=1 timeout 300 python tools/bench_gemm_sm.py 512 > $O/sm_shapes_512.jsonl 2>> $O/sm_shapes_1.err; tail -n 1 $O/sm_shapes_512.jsonl >> $O/summary.log
SHORT="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 --no-attn-block"
i=0
for E in "CTCLIP_GEMM_SM=1" "CTCLIP_GEMM_SM=0" "CTCLIP_GEMM_SM=1" "CTCLIP_GEMM_SM=0" "CTCLIP_GEMM_SM=1 CTCLIP_ZERO_OVERLAP=0"; do
  i=$((i+1))
  env $E timeout 600 python bench.py $SHORT > $O/sm_ab$i.json 2> $O/sm_ab$i.err
  python - <<PY >> $O/summary.log
import json
try:
    b=json.loads(open("$O/sm_ab$i.json").read().strip().splitlines()[-1]); print("ab$i [$E]", b["ms_per_step"], "ms/step", b["value"], "loss", b["loss"])
except Exception as e:
    print("ab$i failed", e); print(open("$O/sm_ab$i.err").read()[-1500:])
PY
done
for E in "CTCLIP_GEMM_SM=1" "CTCLIP_GEMM_SM=0"; do
  i=$((i+1))
  env $E timeout 600 python bench.py $SHORT --text-len 512 > $O/sm_ab$i.json 2> $O/sm_ab$i.err
  python - <<PY >> $O/summary.log
import json
try:
    b=json.loads(open("$O/sm_ab$i.json").read().strip().splitlines()[-1]); print("ab$i T=512 [$E]", b["ms_per_step"], "ms/step", b["value"], "loss", b["loss"])
except Exception as e:
    print("ab$i failed", e); print(open("$O/sm_ab$i.err").read()[-1500:])
PY
done
cat $O/summary.log
