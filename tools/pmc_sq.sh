#!/bin/bash
# issue-side counters of the bench's kernels (own passes, no tracing beside --kernel-trace): gpurun -- 'bash tools/pmc_sq.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
SETS=("SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM")
[ -n "$PMC_QUICK" ] && SETS=("SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS")
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-student --no-roofline > /dev/null 2> $O/sq$i.err
done
python3 $R/tools/pmc_agg.py $(find $O/sq1 $O/sq2 $O/sq3 $O/sq4 -name "*counter_collection.csv" 2>/dev/null) 2>/dev/null | head -14
rm -rf $O/sq1 $O/sq2 $O/sq3 $O/sq4
