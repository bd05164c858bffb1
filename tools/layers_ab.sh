# per-layer kernel times of the bench command for an environment switch off / on (same box; kernel trace). Tooling only.  usage: layers_ab.sh VAR=VALUE
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for v in "" "$1"; do
  tag=$(echo "${v:-default}" | tr '=' '_')
  rm -rf $O/lay_$tag
  env $v rocprofv3 --output-format csv --kernel-trace -d $O/lay_$tag -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > /dev/null 2> $O/lay_$tag.err
  echo "== ${v:-default}"
  python3 tools/trace_summary.py $O/lay_$tag 3968 | grep -v "accum\|argmax\|synth\|avgpool"
  find $O/lay_$tag -name '*.csv' -size +5M -delete
done
