#!/bin/bash
# fused k/v un-prep validation: kernel test, attention + e2e + full-size tests, attention block timing, same-box A/B of the step -> gpurun_out/s7
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s7; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "headnorm or attn2 or attention" > $O/t_k.log 2>&1; echo "kernel tests rc=$? $(tail -n 1 $O/t_k.log)" >> $O/summary.log
timeout 1200 python -m pytest tests/test_full_size_gpu.py tests/test_e2e_gpu.py -q -s > $O/t_full.log 2>&1; echo "full-size + e2e rc=$? $(tail -n 1 $O/t_full.log)" >> $O/summary.log
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --profile-steps 0 --no-reference-depth"
for i in 1 2; do
  timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('A fused unprep', r['ms_per_step'], r['loss'], r['attn_block']['fwd_us'], r['attn_block']['fwd_bwd_us'])" >> $O/ab.log
  CTCLIP_ATTN_FUSED_UNPREP=0 timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('B unprep kern  ', r['ms_per_step'], r['loss'], r['attn_block']['fwd_us'], r['attn_block']['fwd_bwd_us'])" >> $O/ab.log
done
timeout 600 python tools/trace_determinism.py --runs 150 2>&1 | grep TRACE_SUMMARY >> $O/summary.log
cat $O/summary.log $O/ab.log; grep -h "FAILED\|^E  " $O/t_k.log $O/t_full.log | head -20; grep -E "\[full" $O/t_full.log | cut -c1-250
