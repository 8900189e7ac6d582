#!/bin/bash
# round-2 GPU session G: vector instruction rates, PMC passes over the PEG marching kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/ubench/valu_rates > $O/valu_rates.txt 2>&1
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "peg" > $O/t_peg.log 2>&1; echo "peg tests rc=$?" >> $O/summary.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py peg 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py peg 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc2.err
cd $GRAFT_REPO_ROOT
rm -rf $O/pmc*/*/*.db
cat $O/valu_rates.txt; tail -n 3 $O/t_peg.log; tail -n 3 $O/pmc1.err
