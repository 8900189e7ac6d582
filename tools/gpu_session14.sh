#!/bin/bash
# patch-embedding front end with a token row per XCD: kernel test, stand-alone time, step A/B -> gpurun_out/s14
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s14; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -k "patch_ln or tiny" > $O/t_k.log 2>&1; echo "patch_ln + e2e tests rc=$? $(tail -n 1 $O/t_k.log)" >> $O/summary.log
cat > /tmp/pl.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from ct_clip_amd import backend
be = backend.get()
v = torch.rand(8, 1, 240, 480, 480, device="cuda") * 2 - 1
for _ in range(3): be.patch_ln(v, 10, 20, 20, 4032, 1e-5, torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): be.patch_ln(v, 10, 20, 20, 4032, 1e-5, torch.bfloat16)
e1.record(); torch.cuda.synchronize()
print("patch_ln us", round(e0.elapsed_time(e1) / 20 * 1e3, 1))
PY
for i in 1 2; do
  for lib in hip plnoff; do echo "$lib $(CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so timeout 120 python /tmp/pl.py 2>/dev/null | tail -n 1)" >> $O/ops.log; done
done
AB="--steps 30 --warmup 3 --no-cpu-baseline --no-pmc --profile-steps 0 --no-reference-depth"
for i in 1 2; do
  for lib in hip plnoff; do
    CTCLIP_LIB=ct_clip_amd/libctclip_$lib.so timeout 300 python bench.py $AB 2>/dev/null | tail -n 1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', r['ms_per_step'], r['loss'])" >> $O/ab.log
  done
done
cat $O/summary.log $O/ops.log $O/ab.log; grep -h "FAILED\|^E  " $O/t_k.log | head
