# A/B of env switches on one engine (run through gpurun from the repository root). Tooling only.  usage: s2_ab.sh ARCH VAR=VALUE [VAR=VALUE ...]
set -e
cd $GRAFT_REPO_ROOT
A=$1; shift
O=gpurun_out/s2_ab.txt; : > $O
for rep in 1 2; do
echo "== default" >> $O; python3 tools/train_time.py $A 2>/dev/null >> $O
for v in "$@"; do echo "== $v" >> $O; env $v python3 tools/train_time.py $A 2>/dev/null >> $O; done
done
cat $O
