# wall time of the three training engines at HEAD (run through gpurun from the repository root). Tooling only.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/side_ab.txt; : > $O
python3 tools/train_time.py resnet18 resnet50 resnet18bf16 2>/dev/null >> $O
cat $O
