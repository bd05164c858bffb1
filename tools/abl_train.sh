# timing-only ablations of the bf16 engine's step (csrc/train2.inc, DH_T2_ABL bits): upper bounds of what removing a kernel group can give.
# DH_T2_ABL is a COMPILE-TIME macro (round 5): the shipped library has no such switch.  Build one variant per value HERE, before gpurun:
#     for v in 1 2 4; do tools/build_variant.sh abl$v -DDH_T2_ABL=$v; done; tools/build_variant.sh base
# then (through gpurun): bash tools/abl_train.sh ARCH BITS [BITS ...]       Tooling only; results of the variants are garbage by design.
set -e
cd $GRAFT_REPO_ROOT
A=$1; shift
O=gpurun_out/abl_train_$A.txt; : > $O
cp deephisto_amd/libdeephisto_hip.so /tmp/dh_keep.so
echo "== base" >> $O; python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O
for v in "$@"; do
  cp deephisto_amd/libdeephisto_hip_${PFX:-abl}$v.so deephisto_amd/libdeephisto_hip.so
  echo "== -DDH_T2_ABL=$v" >> $O; python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O
done
cp /tmp/dh_keep.so deephisto_amd/libdeephisto_hip.so
echo "== base" >> $O; python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O
cat $O
