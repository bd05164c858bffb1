# A/B of library builds (and/or one env switch) on the headline inference line, same box. Tooling only.
# usage: bash tools/ab_infer.sh [-e VAR=VALUE] base exp ...   (expects deephisto_amd/libdeephisto_hip_<name>.so from tools/build_variant.sh; "head" = the shipped library)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
EV=""
if [ "$1" = "-e" ]; then EV=$2; shift; shift; fi
cp deephisto_amd/libdeephisto_hip.so /tmp/dh_keep.so
: > $O/ab_infer.txt
run() { python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inference', round(d['value']), 'patches/s  dominant', round(d['roofline']['frac'],4), ' avg launch us', round(d['roofline']['avg_launch_us'],1))" >> $O/ab_infer.txt; }
for rep in 1 2 3; do
for v in "$@"; do
  if [ "$v" = "head" ]; then cp /tmp/dh_keep.so deephisto_amd/libdeephisto_hip.so; else cp deephisto_amd/libdeephisto_hip_$v.so deephisto_amd/libdeephisto_hip.so; fi
  echo "== $v (pass $rep)" >> $O/ab_infer.txt; run
  if [ -n "$EV" ]; then echo "== $v $EV (pass $rep)" >> $O/ab_infer.txt; export $EV; run; unset ${EV%%=*}; fi
done
done
cp /tmp/dh_keep.so deephisto_amd/libdeephisto_hip.so
cat $O/ab_infer.txt
