#!/bin/bash
# round-2 GPU session L: short-sequence attention -- prefetch / occupancy variants
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
for v in product; do
  if [ $v = product ]; then unset CTCLIP_LIB; else export CTCLIP_LIB=ct_clip_amd/libctclip_$v.so; fi
  timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attn_short" > $O/t_$v.log 2>&1; echo "$v tests rc=$? $(tail -n 1 $O/t_$v.log)" >> $O/summary.log
  timeout 300 python tools/bench_ops.py tattn 10 > $O/ops_$v.json 2>> $O/ops.err
  python -c "
import json;d=json.load(open('$O/ops_$v.json'));print('$v', {k:v['avg_us'] for k,v in d.items() if 'short' in k})" >> $O/summary.log
done
cat $O/summary.log
