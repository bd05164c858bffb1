# A/B of one environment switch on the headline inference line (same box, three passes). Tooling only.  usage: ab_env_infer.sh VAR=VALUE
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/ab_env_infer.txt
run() { python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inference', round(d['value']), 'patches/s  ms/slide', round(d['ms_per_step'],2), ' dominant', round(d['roofline']['frac'],4))" >> $O/ab_env_infer.txt; }
for rep in 1 2 3; do
  echo "== default (pass $rep)" >> $O/ab_env_infer.txt; run
  echo "== $1 (pass $rep)" >> $O/ab_env_infer.txt; env $1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inference', round(d['value']), 'patches/s  ms/slide', round(d['ms_per_step'],2), ' dominant', round(d['roofline']['frac'],4))" >> $O/ab_env_infer.txt
done
cat $O/ab_env_infer.txt
