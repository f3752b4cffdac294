#!/bin/bash
# SQ counter passes (same sets as tools/profile_round.sh) around ONE python script.  usage: tools/debug/pmc_one.sh <tag> <script.py>
set -u
TAG=$1; SCRIPT=$2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -o p -- python3 $SCRIPT > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/sq2 -o p -- python3 $SCRIPT > $OUT/sq2.log 2>&1
S1=$(find $OUT/sq1 -name "*counter_collection.csv" | head -1); S2=$(find $OUT/sq2 -name "*counter_collection.csv" | head -1)
python3 tools/pmc_sq.py $OUT/pmc_sq.json $S1 $S2 > $OUT/pmc_sq.md 2> $OUT/pmc_sq.err
find $OUT -name "*counter_collection.csv" -size +6M -delete
cat $OUT/pmc_sq.md
