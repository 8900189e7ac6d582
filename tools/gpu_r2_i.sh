#!/bin/bash
# round-2 GPU session I: PEG marching kernels -- prefetch depth / counted-wait ablations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
for v in product peg_lds_abl1 peg_lds_abl4; do
  if [ $v = product ]; then unset CTCLIP_LIB; else export CTCLIP_LIB=ct_clip_amd/libctclip_$v.so; fi
  [ $v = peg_lds_abl4 ] || timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "peg" > $O/t_peg_$v.log 2>&1; echo "$v peg tests rc=$? $(tail -n 1 $O/t_peg_$v.log)" >> $O/summary.log
  timeout 300 python tools/bench_ops.py peg 10 > $O/ops_peg_$v.json 2>> $O/ops_peg.err
  python -c "
import json;d=json.load(open('$O/ops_peg_$v.json'));print('$v', {k:v['avg_us'] for k,v in d.items()})" >> $O/summary.log
done
cat $O/summary.log; grep -h "FAILED" $O/t_peg_*.log | head
