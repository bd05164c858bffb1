# fabric-side traffic of one training step per engine (two --pmc passes each; through gpurun).  usage: bash tools/pmc_train.sh [round]
set -e
RND=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; P=$O/profiles_$RND; mkdir -p $P
cd $R
for a in resnet18 resnet18bf16 resnet50; do
  rm -rf $O/pmc_tf_$a $O/pmc_tw_$a
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_tf_$a -o f -- python3 tools/train_profile.py $a > $O/pmc_tf_$a.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_tw_$a -o w -- python3 tools/train_profile.py $a > $O/pmc_tw_$a.log 2>&1
  python3 tools/pmc_train_traffic.py $O/pmc_tf_$a $O/pmc_tw_$a $a $P/${RND}_pmc_train_traffic_$a.txt
  echo "$a done"
done
find $O -name '*counter_collection.csv' -size +20M -delete
