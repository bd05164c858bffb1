# phase stamps of the conv kernels with the stage barrier split into "own DMA wait" and "s_barrier skew" (diagnostic build). Tooling only.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
cp deephisto_amd/libdeephisto_hip.so /tmp/dh_keep.so
echo "== shipped library (barr% = vmcnt(0) wait + s_barrier)" > $O/stamp_split.txt
python3 tools/conv_stamps.py 1024 2>/dev/null >> $O/stamp_split.txt
cp deephisto_amd/libdeephisto_hip_split.so deephisto_amd/libdeephisto_hip.so
echo "== -DDH_STAMP_SPLIT: barr% = the wave's own vmcnt(0)/lgkmcnt(0) wait only; the s_barrier wait is added to cursor%" >> $O/stamp_split.txt
python3 tools/conv_stamps.py 1024 2>/dev/null >> $O/stamp_split.txt
cp /tmp/dh_keep.so deephisto_amd/libdeephisto_hip.so
cat $O/stamp_split.txt
