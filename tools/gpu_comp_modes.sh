#!/bin/bash
# Round 6: partial forms of the compensated residual stream in TRAINING -- step time (same box, 20 steps each) and the
# free-running training-forward code agreement on the real-reference fixtures (tests/test_full_size_gpu.py -k free_running) per mode.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/comp; mkdir -p $O; rm -f $O/summary.log
for mode in auto last3 last6 temporal 1 auto; do
  CTCLIP_RESIDUAL_COMP=$mode timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-reference-depth --no-text512 --profile-steps 0 --no-attn-block > $O/b_$mode.json 2> $O/b_$mode.err
  python - <<PY >> $O/summary.log
import json
try:
    b=json.loads(open("$O/b_$mode.json").read().strip().splitlines()[-1]); print("CTCLIP_RESIDUAL_COMP=$mode", b["ms_per_step"], "ms/step")
except Exception as e:
    print("$mode failed", e)
PY
done
for mode in auto last3 last6 temporal 1; do
  CTCLIP_RESIDUAL_COMP=$mode timeout 900 python -m pytest tests/test_full_size_gpu.py -q -m gpu -s -k "free_running" 2>&1 | grep "free-running\|passed\|failed" | sed "s/^/[$mode] /" >> $O/summary.log
done
cat $O/summary.log
