#!/bin/bash
# Which rocprofv3 SQ_INSTS_VALU_* class counter does each opcode tick, and what does it cost to issue?
# Runs lib/issue_calib (one dispatch per opcode of the shipped kernels, known instruction count) plainly and under
# two --pmc passes, then writes <out>/opcode_classes.json (tools/opcode_classes.py).
#   scripts/pmc_classify.sh <outdir-under-gpurun_out>
out=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/ray-tracing-series-rust_amd/lib/issue_calib 4 100 3 > $O/issue_calib_w4.json || exit 1
$R/ray-tracing-series-rust_amd/lib/issue_calib 3 100 3 > $O/issue_calib_w3.json || exit 1
$R/ray-tracing-series-rust_amd/lib/issue_calib 2 100 3 > $O/issue_calib_w2.json || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
  --output-format csv -d $O/cls_a -- $R/ray-tracing-series-rust_amd/lib/issue_calib 4 20 1 > $O/cls_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU \
  --output-format csv -d $O/cls_b -- $R/ray-tracing-series-rust_amd/lib/issue_calib 4 20 1 > $O/cls_b.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU \
  --output-format csv -d $O/cls_c -- $R/ray-tracing-series-rust_amd/lib/issue_calib 4 20 1 > $O/cls_c.log 2>&1 || exit 1
cp $(ls $O/cls_c/*/*_counter_collection.csv | head -1) $O/classify_c.csv
cp $(ls $O/cls_a/*/*_counter_collection.csv | head -1) $O/classify_a.csv
cp $(ls $O/cls_b/*/*_counter_collection.csv | head -1) $O/classify_b.csv
rm -rf $O/cls_a $O/cls_b $O/cls_c
python3 $R/tools/opcode_classes.py $O/classify_a.csv $O/classify_b.csv $O/issue_calib_w4.json $O/opcode_classes.json $O/classify_c.csv
