# per-layer kernel times of the bench command for several library builds (same box; kernel trace). Tooling only.
# usage: bash tools/layers_ab_lib.sh name name ...   (deephisto_amd/libdeephisto_hip_<name>.so from tools/build_variant.sh; two passes each)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
cp deephisto_amd/libdeephisto_hip.so /tmp/dh_keep.so
for rep in 1 2; do
for v in "$@"; do
  cp deephisto_amd/libdeephisto_hip_$v.so deephisto_amd/libdeephisto_hip.so
  rm -rf $O/lay_$v
  rocprofv3 --output-format csv --kernel-trace -d $O/lay_$v -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/lay_$v.json 2> $O/lay_$v.err
  echo "== $v (pass $rep)"
  python3 tools/trace_summary.py $O/lay_$v 3968 | grep -v "accum\|argmax\|synth\|avgpool"
  find $O/lay_$v -name '*.csv' -size +5M -delete
done
done
cp /tmp/dh_keep.so deephisto_amd/libdeephisto_hip.so
