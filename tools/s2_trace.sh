# per-kernel averages of the stride-2 conv kernels (wide vs 128-pixel) from a kernel trace of the bench command. Tooling only.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for v in 1 0; do
  rm -rf $O/s2t_$v
  DH_CONV_S2_WIDE=$v rocprofv3 --output-format csv --kernel-trace --stats -d $O/s2t_$v -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > /dev/null 2> $O/s2t_$v.err
  echo "== DH_CONV_S2_WIDE=$v"
  f=$(find $O/s2t_$v -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
  find $O/s2t_$v -name '*kernel_trace.csv' -delete
done
