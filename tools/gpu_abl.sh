#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/abl; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or geglu" > $O/t_kernels.log 2>&1; echo "kernel tests rc=$? $(tail -n 1 $O/t_kernels.log)"
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_full_size_gpu.py -q -x > $O/t_e2e.log 2>&1; echo "e2e tests rc=$? $(tail -n 1 $O/t_e2e.log)"
for lib in hip $ABL_LIBS; do
  CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so SHAPES_ONLY=fused timeout 200 python tools/bench_gemm_shapes.py 20 > $O/$lib.json 2> $O/$lib.err
  echo "== $lib"; python -c "
import json; d=json.load(open('$O/$lib.json'))
for k,v in d.items(): print(f'{k:55s} {v[\"us\"]:8.1f} us')" 2>&1 | tail -12
done
B="--steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-attn-block"
for rep in 1 2; do
timeout 600 python bench.py $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
CTCLIP_FF_NODE=0 timeout 600 python bench.py $B > $O/bench_nonode_$rep.json 2> $O/bench_nonode_$rep.err
done
python - <<'PY'
import json
for n in ("new_1", "nonode_1", "new_2", "nonode_2"):
    try:
        b = json.loads(open(f"gpurun_out/abl/bench_{n}.json").read().strip().splitlines()[-1]); print(n, b["ms_per_step"], b["value"], b["loss"], [ (t["kernel"][-40:], t["avg_us"]) for t in b["roofline"]["top5"]])
    except Exception as e:
        print(n, "failed", e)
PY
