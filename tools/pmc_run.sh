set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); rm -rf $O/pmcl$i
  rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/pmcl$i -o c -- python3 tools/fwd_once.py 256 3 > $O/pmcl$i.log 2>&1
  python3 tools/pmc_layers.py $O/pmcl$i
done
