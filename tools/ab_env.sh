# A/B an environment switch inside ONE gpurun call: bash tools/ab_env.sh VAR   (runs with VAR unset, then VAR=1)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in off on off on; do
  if [ $v = on ]; then export $1=1; else unset $1; fi
  rm -rf $O/ab_$v
  rocprofv3 --output-format csv --kernel-trace -d $O/ab_$v -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 --slide 30000 > $O/ab_$v.json 2> $O/ab_$v.err
  echo "== $1 $v"; python3 tools/trace_summary.py $O/ab_$v | grep -v "accum\|argmax\|synth\|avgpool"
  find $O/ab_$v -name '*.csv' -size +30M -delete
done
