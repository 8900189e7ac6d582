#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abl; mkdir -p $O
for lib in hip $ABL_LIBS; do
  CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so SHAPES_ONLY=ff timeout 200 python tools/bench_gemm_shapes.py 20 > $O/$lib.json 2> $O/$lib.err
  echo "== $lib"; python -c "
import json; d=json.load(open('$O/$lib.json'))
for k,v in d.items(): print(f'{k:55s} {v[\"us\"]:8.1f} us')" 2>&1 | tail -12
done
