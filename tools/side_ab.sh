# A/B of the side-stream settings (run through gpurun from the repository root). Tooling only.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/side_ab.txt; : > $O
for s in 1 0; do echo "== resnet18 (f32) DH_T1_SIDE=$s" >> $O; DH_T1_SIDE=$s python3 tools/train_time.py resnet18 2>/dev/null >> $O; done
for a in resnet50 resnet18bf16; do
for s in 1 0; do echo "== $a DH_T2_SIDE=$s" >> $O; DH_T2_SIDE=$s python3 tools/train_time.py $a 2>/dev/null >> $O; done
done
cat $O
