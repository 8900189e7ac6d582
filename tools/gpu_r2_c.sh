#!/bin/bash
# round-2 GPU session C: where does the slab-resident forward spend its time?  ablation builds + PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attn2" > $O/t_attn2.log 2>&1; echo "attn2 tests rc=$?" >> $O/summary.log
timeout 120 python tools/bench_attn2_fwd.py 20 all > $O/abl_product.json 2>> $O/abl.err
for m in 1 2 3 4 8 12 15 16 32; do
  CTCLIP_LIB=ct_clip_amd/libctclip_attn2_abl$m.so timeout 120 python tools/bench_attn2_fwd.py 20 fwd >> $O/abl.jsonl 2>> $O/abl.err
done
timeout 200 python tools/bench_ops.py attn2 10 > $O/ops_attn2.json 2> $O/ops_attn2.err
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_attn2_fwd.py 3 all > $GRAFT_REPO_ROOT/$O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- python $GRAFT_REPO_ROOT/tools/bench_attn2_fwd.py 3 all > $GRAFT_REPO_ROOT/$O/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/pmc_summary.txt 2>&1
import csv, glob, collections
for d in ("gpurun_out/r2c/pmc1", "gpurun_out/r2c/pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")
            if "attn2" not in n and "dbias" not in n: continue
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[n]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, c in acc.items():
        print(d, n, {k: round(sum(v) / len(v), 1) for k, v in c.items()})
PY
tail -n 4 $O/t_attn2.log; cat $O/summary.log $O/abl_product.json $O/abl.jsonl $O/pmc_summary.txt; grep -A2 "unprep\|prep" $O/ops_attn2.json | head -12; tail -3 $O/abl.err $O/pmc2.log
rm -rf $O/pmc1/*/*.db $O/pmc2/*/*.db 2>/dev/null
