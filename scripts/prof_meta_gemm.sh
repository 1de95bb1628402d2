#!/bin/bash
# PMC passes over the bench for one kernel (default: the batched meta weight-gradient GEMM)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_meta; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/p1 -o p -- python $ROOT/bench.py --steps 3 --warmup 5 --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d $OUT/p2 -o p -- python $ROOT/bench.py --steps 3 --warmup 5 --no-cpu-baseline > $OUT/p2.log 2>&1
python $ROOT/scripts/pmc_summary.py $OUT ${1:-gemm_bw_kernel} | cut -c1-150
rm -rf $OUT/p1 $OUT/p2
