# Layer ablations at the bench's micro-batch (1024) inside ONE gpurun call: per-layer kernel times of each library variant.
# usage: bash tools/ab_l1.sh base a2 a4 ...   (deephisto_amd/libdeephisto_hip_<name>.so built by tools/build_variant.sh)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
cp deephisto_amd/libdeephisto_hip.so /tmp/keep.so
for v in "$@"; do
  cp deephisto_amd/libdeephisto_hip_$v.so deephisto_amd/libdeephisto_hip.so
  rm -rf $O/abl1_$v
  rocprofv3 --output-format csv --kernel-trace -d $O/abl1_$v -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 --slide 30000 > $O/abl1_$v.json 2> $O/abl1_$v.err
  echo "== $v"; python3 tools/trace_summary.py $O/abl1_$v 1024 | grep -v "accum\|argmax\|synth\|avgpool"
  find $O/abl1_$v -name '*.csv' -size +30M -delete
done
cp /tmp/keep.so deephisto_amd/libdeephisto_hip.so
