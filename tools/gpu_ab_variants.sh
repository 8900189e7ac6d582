#!/bin/bash
# same-box A/B of library variants (tools/build_variant.py): ms per step of the default bench, two rounds -> gpurun_out/abv/ab.log
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abv; mkdir -p $O
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --no-attn-block --profile-steps 0 --no-reference-depth"
for i in 1 2; do
  for v in product $VARIANTS; do
    lib=""; [ $v != product ] && lib=ct_clip_amd/libctclip_$v.so
    CTCLIP_LIB=$lib timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['loss'])" >> $O/ab.log
  done
done
cat $O/ab.log
