#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abl; mkdir -p $O
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_e2e_gpu.py -q -x > $O/t_e2e.log 2>&1; echo "tests rc=$? $(tail -n 1 $O/t_e2e.log)"; grep -h "Error\|FAILED\|assert" $O/t_e2e.log | head -20
B="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-attn-block --profile-steps 0"
for rep in 1 2; do
timeout 600 python bench.py $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
CTCLIP_BRANCH_STREAM=0 timeout 600 python bench.py $B > $O/bench_off_$rep.json 2> $O/bench_off_$rep.err
done
python - <<'PY'
import json
for n in ("new_1", "off_1", "new_2", "off_2"):
    try:
        b = json.loads(open(f"gpurun_out/abl/bench_{n}.json").read().strip().splitlines()[-1]); print(n, b["ms_per_step"], b["value"], b["loss"], b["peak_mem_gib"])
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/abl/bench_{n}.err").read()[-1500:])
PY
