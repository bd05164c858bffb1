# sweeps of the training engines' environment knobs on one box (ResNet-50 bf16 step, ms). Tooling only.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep_train_knobs.txt; : > $O
t() { echo "== $*" >> $O; env "$@" python3 tools/train_time.py resnet50 --steps 40 2>/dev/null >> $O; }
echo "== default" >> $O; python3 tools/train_time.py resnet50 --steps 40 2>/dev/null >> $O
t DH_G2_NS3_K=192
t DH_G2_NS3_K=256
t DH_G2_NS3_K=512
t DH_G2_NS3_K=1024
t DH_G2_NS3_K=2048
echo "== default" >> $O; python3 tools/train_time.py resnet50 --steps 40 2>/dev/null >> $O
t DH_G2_NS3_K=192
t DH_G2_NS3_K=512
cat $O
