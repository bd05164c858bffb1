# timing-only ablations of the bf16 engine's step (DH_T2_ABL bits, csrc/train2.inc): upper bounds of what removing a kernel group can give.
# Tooling only.  usage (through gpurun): bash tools/abl_train.sh ARCH BITS [BITS ...]
set -e
cd $GRAFT_REPO_ROOT
A=$1; shift
O=gpurun_out/abl_train_$A.txt; : > $O
echo "== base" >> $O; python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O
for v in "$@"; do echo "== DH_T2_ABL=$v" >> $O; DH_T2_ABL=$v python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O; done
echo "== base" >> $O; python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O
cat $O
