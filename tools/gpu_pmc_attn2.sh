#!/bin/bash
# PMC passes over tools/bench_ops.py attn2 (slab kernels incl. dBias)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_attn2; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py attn2 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py attn2 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc2.err
cd $GRAFT_REPO_ROOT
rm -rf $O/pmc*/*/*.db
python - <<'PY'
import csv, glob, collections, re
for d in ("gpurun_out/pmc_attn2/pmc1","gpurun_out/pmc_attn2/pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(\w+_slab_kernel|dbias_\w+|attn_prep_kernel|attn_unprep_kernel)", r["Kernel_Name"])
            if not m: continue
            n = m.group(1)
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[n]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n,c in acc.items():
        print(n, {k: f"{sum(v)/len(v):.4g}" for k,v in sorted(c.items())})
PY
