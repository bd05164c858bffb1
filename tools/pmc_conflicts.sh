set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 -m pytest tests/test_gpu_resnet.py tests/test_gpu_predict.py tests/test_gpu_train.py -x -q 2>&1 | tail -2
rm -rf $O/pmcl1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/pmcl1 -o c -- python3 tools/fwd_once.py 256 3 > $O/pmcl1.log 2>&1
python3 tools/pmc_layers.py $O/pmcl1
bash tools/ab_lib.sh base exp 2>&1 | grep -v "^row\|^s[12] "
