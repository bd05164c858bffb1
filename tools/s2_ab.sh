# A/B of an env switch on all three engines (run through gpurun from the repository root). Tooling only.  usage: s2_ab.sh VAR=VALUE
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2_ab.txt; : > $O
for rep in 1 2; do
echo "== default" >> $O; python3 tools/train_time.py resnet18 resnet50 resnet18bf16 2>/dev/null >> $O
echo "== $1" >> $O; env $1 python3 tools/train_time.py resnet18 resnet50 resnet18bf16 2>/dev/null >> $O
done
cat $O
