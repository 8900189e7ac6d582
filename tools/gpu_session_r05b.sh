#!/bin/bash
# round-5 session B: the second form of the NT GEMM (gemm_nt2.hip): parity, isolated A/B per shape, in-step A/B per family mask
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/run; mkdir -p $O; rm -f $O/summary.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -s -k "nt2 or geglu or gemm_nt" > $O/tests_nt2.log 2>&1
echo "tests rc=$? $(tail -n 1 $O/tests_nt2.log)" >> $O/summary.log
grep -h "^FAILED\|^ERROR\|differ\|Error" $O/tests_nt2.log | head -20 >> $O/summary.log
timeout 600 python tools/bench_gemm_nt2.py 12 > $O/nt2_shapes.jsonl 2> $O/nt2_shapes.err
echo "shapes rc=$?" >> $O/summary.log; cat $O/nt2_shapes.jsonl >> $O/summary.log
SHORT="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 --no-attn-block"
i=0
for E in "CTCLIP_GEMM_NT2=0" "CTCLIP_GEMM_NT2=7" "CTCLIP_GEMM_NT2=6" "CTCLIP_GEMM_NT2=2" "CTCLIP_GEMM_NT2=1" "CTCLIP_GEMM_NT2=0" "CTCLIP_GEMM_NT2=7"; do
  i=$((i+1))
  env $E timeout 600 python bench.py $SHORT > $O/nt2_ab$i.json 2> $O/nt2_ab$i.err
  python - <<PY >> $O/summary.log
import json
try:
    b=json.loads(open("$O/nt2_ab$i.json").read().strip().splitlines()[-1]); print("ab$i [$E]", b["ms_per_step"], "ms/step", b["value"], "loss", b["loss"])
except Exception as e:
    print("ab$i failed", e); print(open("$O/nt2_ab$i.err").read()[-1500:])
PY
done
cat $O/summary.log
