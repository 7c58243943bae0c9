#!/bin/bash
# One profiling session on the GPU box: rocprofv3 kernel stats of the bench + SQ counter passes (own runs, --kernel-trace only).
# usage: bash tools/gpu_profile_session.sh <outdir under gpurun_out>
set -u
OUT=$PWD/gpurun_out/${1:-prof}
mkdir -p $OUT
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
B="SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA"
timeout 170 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/attn_a -- python $REPO/tools/pmc_attn.py > $OUT/attn_a.log 2>&1
timeout 170 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/attn_b -- python $REPO/tools/pmc_attn.py > $OUT/attn_b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/step_a -- python $REPO/tools/profile_step.py > $OUT/step_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/step_b -- python $REPO/tools/profile_step.py > $OUT/step_b.log 2>&1
cd $REPO
python tools/pmc_summary.py $OUT/attn_a $OUT/attn_b --match flash_attn > $OUT/pmc_attn.json 2>$OUT/pmc_attn.err
python tools/pmc_summary.py $OUT/step_a $OUT/step_b > $OUT/pmc_step.json 2>$OUT/pmc_step.err
# keep only the summaries (the raw CSVs of a whole step are tens of MB)
rm -rf $OUT/attn_a $OUT/attn_b $OUT/step_a $OUT/step_b
head -c 3000 $OUT/pmc_attn.json
