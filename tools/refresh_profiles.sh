set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats -o st -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/pmc_f.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/pmc_w.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 1024 $O/pmc.json
python3 tools/trace_summary.py $O/prof_stats 1024 > $O/per_layer.txt
find $O/pmc_fetch $O/pmc_write -name '*counter_collection.csv' -size +20M -delete
find $O/prof_stats -name '*kernel_trace.csv' -size +20M -delete
cat $O/bench_default.json
