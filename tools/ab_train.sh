# A/B builds of the library inside ONE gpurun call (same box): wall time per training step of the three engines.
# usage: bash tools/ab_train.sh base exp ...  (expects deephisto_amd/libdeephisto_hip_<name>.so files from tools/build_variant.sh)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
cp deephisto_amd/libdeephisto_hip.so /tmp/dh_keep.so
: > $O/ab_train.txt
for rep in 1 2; do
for v in "$@"; do
  cp deephisto_amd/libdeephisto_hip_$v.so deephisto_amd/libdeephisto_hip.so
  echo "== $v (pass $rep)" >> $O/ab_train.txt
  python3 tools/train_time.py resnet18 resnet50 resnet18bf16 2>/dev/null >> $O/ab_train.txt
done
done
cp /tmp/dh_keep.so deephisto_amd/libdeephisto_hip.so
cat $O/ab_train.txt
