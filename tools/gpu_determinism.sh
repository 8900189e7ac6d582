#!/bin/bash
# Determinism check on one GPU box: in-situ launch trace of the product library (tools/trace_determinism.py) at the
# fixture geometry and at the bench configuration, then the per-kernel soak.  If any run deviates and strict-wait builds exist
# (tools/build_variant.py strict_* ...), the same trace with them and with the side streams off.  -> gpurun_out/det
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/det; mkdir -p $O
RUNS=${RUNS:-200}
tr() { timeout 900 python tools/trace_determinism.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids"; }
{
echo "== $(date -u +%H:%M:%S)"
tr --runs $RUNS
tr --runs $((RUNS / 2)) --noise 1
tr --runs ${BENCH_RUNS:-25} --config bench
} > $O/trace_product.log 2>&1
if grep -q '"deviating_runs": [1-9]' $O/trace_product.log; then
  {
  for lib in ct_clip_amd/libctclip_strict_*.so; do [ -f $lib ] && CTCLIP_LIB=$lib tr --runs $RUNS; done
  CTCLIP_TEXT_STREAM=0 CTCLIP_WGRAD_STREAM=0 tr --runs $RUNS
  tr --runs $RUNS --forward-only
  } > $O/trace_bisect.log 2>&1
fi
timeout 900 python tools/soak_kernels.py ${SOAK_REPS:-300} > $O/soak.log 2>&1
grep -h "TRACE_SUMMARY\|^run \|deviating" $O/*.log | cut -c1-300 | head -60
