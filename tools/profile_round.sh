#!/bin/bash
# Collects the rocprofv3 evidence the bench line cites (run on the GPU box through gpurun; outputs under gpurun_out/prof_$1).
#   1. --kernel-trace --stats of the DEFAULT bench command (per-kernel average durations)
#   2. PMC passes at 4096 points, each in its own run: FETCH_SIZE | WRITE_SIZE | SQ pass 1 | SQ pass 2
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o s -- python bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
grep "^{\"metric\"" $OUT/stats.log > $OUT/bench_line.json
B="python bench.py --points 4096 --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- $B > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- $B > $OUT/write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/sq1 -o p -- $B > $OUT/sq1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d $OUT/sq2 -o p -- $B > $OUT/sq2.log 2>&1
python tools/profile_summarize.py $OUT
