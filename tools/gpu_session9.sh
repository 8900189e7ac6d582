#!/bin/bash
# PEG plane-scatter march: kernel tests, stand-alone timing against the round-2 gather kernel and the prefetch-depth variants, step A/B,
# full-size parity -> gpurun_out/s9
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s9; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "peg" > $O/t_k.log 2>&1; echo "peg kernel tests rc=$? $(tail -n 1 $O/t_k.log)" >> $O/summary.log
for lib in hip pegold pegd1 pegd2w1; do
  for i in 1 2; do
    CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so timeout 120 python tools/bench_ops.py peg 30 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', {k.split(' (')[0]: v['avg_us'] for k, v in r.items()})" >> $O/ops.log
  done
done
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --profile-steps 0 --no-reference-depth"
for i in 1 2; do
  for lib in hip pegold; do
    CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', r['ms_per_step'], r['loss'])" >> $O/ab.log
  done
done
timeout 1200 python -m pytest tests/test_full_size_gpu.py tests/test_e2e_gpu.py -q -s > $O/t_full.log 2>&1; echo "full-size + e2e rc=$? $(tail -n 1 $O/t_full.log)" >> $O/summary.log
cat $O/summary.log $O/ops.log $O/ab.log; grep -h "FAILED\|^E  " $O/t_k.log $O/t_full.log | head -20; grep -E "\[full. bf16" $O/t_full.log | cut -c1-250
