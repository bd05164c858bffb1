# A/B builds of the library inside ONE gpurun call (same box): headline inference line + the three training engines.
# usage: bash tools/ab_bench.sh base exp ...  (expects deephisto_amd/libdeephisto_hip_<name>.so files from tools/build_variant.sh)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
cp deephisto_amd/libdeephisto_hip.so /tmp/dh_keep.so
: > $O/ab_bench.txt
for rep in 1 2; do
for v in "$@"; do
  cp deephisto_amd/libdeephisto_hip_$v.so deephisto_amd/libdeephisto_hip.so
  echo "== $v (pass $rep)" >> $O/ab_bench.txt
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inference', round(d['value']), 'patches/s  dominant', round(d['roofline']['frac'],4))" >> $O/ab_bench.txt
  python3 tools/train_time.py resnet18 resnet50 resnet18bf16 2>/dev/null >> $O/ab_bench.txt
done
done
cp /tmp/dh_keep.so deephisto_amd/libdeephisto_hip.so
cat $O/ab_bench.txt
