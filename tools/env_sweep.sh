# sweep of one environment switch over the training engines (through gpurun). Tooling only.  usage: env_sweep.sh "ARCH [ARCH..]" VAR v1 v2 ...
set -e
cd $GRAFT_REPO_ROOT
A=$1; V=$2; shift; shift
O=gpurun_out/env_sweep_$V.txt; : > $O
echo "== default" >> $O; python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O
for v in "$@"; do echo "== $V=$v" >> $O; env $V=$v python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O; done
echo "== default" >> $O; python3 tools/train_time.py $A --steps 40 2>/dev/null >> $O
cat $O
