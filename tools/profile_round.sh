#!/bin/bash
# One profiling round on the GPU box: kernel trace + stats, SQ counter passes, FETCH/WRITE passes of
# `bench.py --steps 3 --warmup 2 --no-cpu-baseline [extra bench flags]`.
# usage: tools/profile_round.sh <tag> ["--dtype bf16"]   (outputs: gpurun_out/<tag>/)
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
EXTRA=${2:-}
CMD="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs $EXTRA"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs $EXTRA > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -o p -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/sq2 -o p -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
S1=$(find $OUT/sq1 -name "*counter_collection.csv" | head -1); S2=$(find $OUT/sq2 -name "*counter_collection.csv" | head -1)
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_sq.py $OUT/pmc_sq.json $S1 $S2 > $OUT/pmc_sq.md 2> $OUT/pmc_sq.err
python3 tools/pmc_traffic.py $F $W > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
python3 tools/prof_summary.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) 40 > $OUT/kernel_stats.md 2> $OUT/kernel_stats.err
# keep the merged-back payload small: drop the raw per-dispatch csv of the counter passes (tens of MB)
find $OUT -name "*counter_collection.csv" -size +6M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT
