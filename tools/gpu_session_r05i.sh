#!/bin/bash
# round-5 session I: de-phased LDS-DMA issue of the two waves per SIMD in gemm_nt / gemm_tn (build variants): parity, isolated shapes, in-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/run; mkdir -p $O; rm -f $O/summary_i.log
CTCLIP_LIB=ct_clip_amd/libctclip_dephase.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_properties_bench_size_gpu.py -q -m gpu -x -k "gemm or geglu or headnorm or residual_comp or argmax or identity or tile or adjoint" > $O/tests_i.log 2>&1
echo "tests(dephase) rc=$? $(tail -n 1 $O/tests_i.log)" >> $O/summary_i.log
for v in product dephase; do
  lib=""; [ $v != product ] && lib=ct_clip_amd/libctclip_$v.so
  CTCLIP_LIB=$lib timeout 300 python tools/bench_gemm_shapes.py 10 > $O/shapes_$v.json 2> $O/shapes_$v.err
  python - <<PY >> $O/summary_i.log
import json
d=json.load(open("$O/shapes_$v.json"))
print("$v:", "; ".join(f"{k.split(' ')[0]} {k.split(') ')[1][:12]}: {v_['us']}" for k, v_ in d.items() if ') ' in k))
print("$v fused:", "; ".join(f"{k[:28]}: {v_['us']}" for k, v_ in d.items() if ') ' not in k))
PY
done
rm -f gpurun_out/abv/ab.log
VARIANTS="dephase dephase_nt dephase_tn" bash tools/gpu_ab_variants.sh > /dev/null 2>&1
cat gpurun_out/abv/ab.log >> $O/summary_i.log
cat $O/summary_i.log
