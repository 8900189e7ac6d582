#!/bin/bash
# round-2 GPU session M: PMC passes over the short-sequence attention kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py tattn 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py tattn 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc2.err
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc3 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py tattn 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc3.err
cd $GRAFT_REPO_ROOT
rm -rf $O/pmc*/*/*.db
python - <<'PY'
import csv, glob, collections, re
for d in ("gpurun_out/r2m/pmc1","gpurun_out/r2m/pmc2","gpurun_out/r2m/pmc3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(attn_short_\w+|attn_fwd_kernel|attn_bwd_d\w+)", r["Kernel_Name"])
            if not m: continue
            n = m.group(1)
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[n]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n,c in acc.items():
        print(n, {k: f"{sum(v)/len(v):.4g}" for k,v in sorted(c.items())})
PY
tail -n 2 $O/pmc3.err
