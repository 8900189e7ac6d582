#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in hip $ABL_LIBS hip; do echo "== $lib"; CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so timeout 200 python tools/bench_ln.py 30 2>&1 | tail -2; done
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm or patch_ln" 2>&1 | tail -1
