#!/bin/bash
# same-box timing of attn2_bwd1.hip ablation builds (tools/build_variant.py abl<mask> attn2_bwd1.hip:BWD1_ABL=<mask>; results of those builds are
# WRONG by construction, only the time is read): one-pass kernel chain in us per layer at the bench shape -> gpurun_out/abl/ab.log
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abl; mkdir -p $O; rm -f $O/ab.log
for v in product $VARIANTS product; do
  lib=""; [ $v != product ] && lib=ct_clip_amd/libctclip_$v.so
  CTCLIP_LIB=$lib timeout 120 python tools/bench_attn2_bwd.py 20 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['one_pass_us'])" >> $O/ab.log
done
cat $O/ab.log
