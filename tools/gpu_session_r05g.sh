#!/bin/bash
# round-5 session G: dW + db of the text tower's Linear layers in one launch (ctclip_gemm_dw_db): full suite + A/B at 12+12 and 4+4
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/run; mkdir -p $O; rm -f $O/summary_g.log
timeout 1500 python -m pytest tests -q -m gpu -x -s > $O/tests.log 2>&1
echo "tests rc=$? $(tail -n 1 $O/tests.log)" >> $O/summary_g.log
grep -h "^FAILED\|^ERROR\|Error" $O/tests.log | head -20 >> $O/summary_g.log
SHORT="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 --no-attn-block"
i=0
for E in "CTCLIP_DW_DB=1" "CTCLIP_DW_DB=0" "CTCLIP_DW_DB=1" "CTCLIP_DW_DB=0"; do
  for D in "12" "4"; do
    i=$((i+1))
    env $E timeout 600 python bench.py $SHORT --spatial-depth $D --temporal-depth $D > $O/g_ab$i.json 2> $O/g_ab$i.err
    python - <<PY >> $O/summary_g.log
import json
try:
    b=json.loads(open("$O/g_ab$i.json").read().strip().splitlines()[-1]); print("ab$i [$E] $D+$D", b["ms_per_step"], "ms/step", b["value"], "loss", b["loss"])
except Exception as e:
    print("ab$i failed", e); print(open("$O/g_ab$i.err").read()[-1500:])
PY
  done
done
cat $O/summary_g.log
