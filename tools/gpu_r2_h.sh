#!/bin/bash
# round-2 GPU session H: PEG marching kernels with three planes in flight
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "peg" > $O/t_peg.log 2>&1; echo "peg tests rc=$?" >> $O/summary.log
timeout 300 python tools/bench_ops.py peg 10 > $O/ops_peg_lds.json 2> $O/ops_peg.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_ops.py peg 3 > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc1.err
cd $GRAFT_REPO_ROOT
rm -rf $O/pmc*/*/*.db
tail -n 3 $O/t_peg.log; cat $O/ops_peg_lds.json
