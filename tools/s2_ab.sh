# A/B of the stride-2 data gradient variants (run through gpurun from the repository root). Tooling only.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_ab.txt; : > $O
for rep in 1 2; do
echo "== f32 merged classes (default)" >> $O; python3 tools/train_time.py resnet18 2>/dev/null >> $O
echo "== f32 four launches (DH_DGRAD_S2_SERIAL=1)" >> $O; DH_DGRAD_S2_SERIAL=1 python3 tools/train_time.py resnet18 2>/dev/null >> $O
echo "== bf16 upsampled copy (default)" >> $O; python3 tools/train_time.py resnet50 resnet18bf16 2>/dev/null >> $O
echo "== bf16 merged classes (DH_DGRAD_S2_CLASSES=1)" >> $O; DH_DGRAD_S2_CLASSES=1 python3 tools/train_time.py resnet50 resnet18bf16 2>/dev/null >> $O
done
cat $O
