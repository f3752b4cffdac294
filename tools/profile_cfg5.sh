#!/bin/bash
# BASELINE configs[4] (cfg 5) on the GPU box: kernel trace + stats, then SQ counter passes of tools/cfg5_step.py.
# usage: tools/profile_cfg5.sh <tag>      (outputs: gpurun_out/<tag>/)
set -u
TAG=${1:-cfg5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 tools/cfg5_step.py 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -o p -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $OUT/sq3 -o p -- $CMD > $OUT/sq3.log 2>&1
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_INSTS_VALU[A-Z0-9_]*" $OUT/counters.txt | sort -u > $OUT/valu_counters.txt
python3 tools/prof_summary.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) 30 > $OUT/kernel_stats.md 2> $OUT/kernel_stats.err
S1=$(find $OUT/sq1 -name "*counter_collection.csv" | head -1)
S3=$(find $OUT/sq3 -name "*counter_collection.csv" | head -1)
python3 tools/pmc_sq.py $OUT/pmc_sq.json $S1 > $OUT/pmc_sq.md 2> $OUT/pmc_sq.err
python3 tools/pmc_valu_mix.py $S3 $S1 > $OUT/valu_mix.md 2> $OUT/valu_mix.err
find $OUT -name "*counter_collection.csv" -size +6M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -5 $OUT/sq3.log
ls -la $OUT
