#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fused_$1
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python $ROOT/scripts/bench_fused.py 150
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/scripts/bench_fused.py 150 > $OUT/trace.log 2>&1
python $ROOT/scripts/top_kernels.py $OUT/trace/trace_results.db 8
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc1 -- python $ROOT/scripts/bench_fused.py 150 > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc2 -o pmc2 -- python $ROOT/scripts/bench_fused.py 150 > $OUT/pmc2.log 2>&1
python $ROOT/scripts/pmc_summary.py $OUT ${2:-gemm_kernel} | grep -v "^  kernel" | cut -c1-150
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2
