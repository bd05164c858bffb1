set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for mb in 256 512 1024; do
  rm -rf $O/ab_mb$mb
  rocprofv3 --output-format csv --kernel-trace -d $O/ab_mb$mb -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 --slide 30000 --micro-batch $mb > $O/ab_mb$mb.json 2> $O/ab_mb$mb.err
  echo "== micro-batch $mb"; python3 tools/trace_summary.py $O/ab_mb$mb $mb | grep -v "accum\|argmax\|synth"
  python3 -c "
import json; d=json.loads(open('$O/ab_mb$mb.json').read().splitlines()[-1]); print('value', round(d['value']))"
  find $O/ab_mb$mb -name '*.csv' -size +30M -delete
done
