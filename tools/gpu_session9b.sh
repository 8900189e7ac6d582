#!/bin/bash
# PEG stand-alone timing: product against the round-2 gather kernel and the prefetch-depth variants -> gpurun_out/s9
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s9; mkdir -p $O; rm -f $O/ops.log
for i in 1 2; do
  for lib in hip pegold pegd1 pegd2w1; do
    CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so timeout 120 python tools/bench_ops.py peg 30 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', {k.split(' (')[0]: v['avg_us'] for k, v in r.items()})" >> $O/ops.log
  done
done
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --profile-steps 0 --no-reference-depth"
for lib in pegd1 pegd2w1 hip; do
  CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', r['ms_per_step'], r['loss'])" >> $O/ab2.log
done
cat $O/ops.log $O/ab2.log
