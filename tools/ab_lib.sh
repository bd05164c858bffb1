# A/B builds of the library inside ONE gpurun call (same box): per-layer kernel times + phase stamps of each.
# usage: bash tools/ab_lib.sh base exp ...  (expects deephisto_amd/libdeephisto_hip_<name>.so files)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in "$@"; do
  cp deephisto_amd/libdeephisto_hip_$v.so deephisto_amd/libdeephisto_hip.so
  rm -rf $O/ab_$v
  rocprofv3 --output-format csv --kernel-trace -d $O/ab_$v -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 --slide 30000 --micro-batch 256 > $O/ab_$v.json 2> $O/ab_$v.err
  echo "== $v"; python3 tools/trace_summary.py $O/ab_$v | grep -v "accum\|argmax\|synth\|avgpool"
  python3 tools/conv_stamps.py 256 2>/dev/null | grep -v "^stem"
  find $O/ab_$v -name '*.csv' -size +30M -delete
done
