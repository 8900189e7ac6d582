#!/bin/bash
# round-4 session: the one-pass attention backward -- parity tests, timing against the three-pass path, phase clocks
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s1; mkdir -p $O; rm -f $O/summary.log
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attn2_bwd_fused" > $O/t_fused.log 2>&1; echo "fused tests rc=$? $(tail -n 1 $O/t_fused.log)" >> $O/summary.log
timeout 300 python tools/bench_attn2_bwd.py 20 > $O/bench_attn2_bwd.json 2> $O/bench_attn2_bwd.err; echo "bench rc=$?" >> $O/summary.log
timeout 300 python tools/bench_attn2_bwd.py 5 --stamps > $O/bench_attn2_bwd_stamps.json 2>> $O/bench_attn2_bwd.err; echo "stamps rc=$?" >> $O/summary.log
cat $O/summary.log; grep -E "^E  |^FAILED|passed|failed" $O/t_fused.log | head -40; cat $O/bench_attn2_bwd.json; python -c "
import json;d=json.load(open('$O/bench_attn2_bwd_stamps.json'));print({k:v for k,v in d.items() if k!='phases_us_per_item'})
for r in d.get('phases_us_per_item',[]): print(r)"
