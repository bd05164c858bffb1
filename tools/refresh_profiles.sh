# Regenerates everything under profiles/ for one round on the GPU box (run through gpurun from the repository root):
#   bash tools/refresh_profiles.sh r02      -> gpurun_out/profiles_r02/*  (copy to profiles/ afterwards: tools/collect_profiles.sh r02)
# default bench line, rocprofv3 kernel stats of the same command, the two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs,
# kernel trace only), per-layer table, MFMA utilisation per layer (SQ counters), conv phase stamps, training kernel stats.
set -e
RND=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; P=$O/profiles_$RND; mkdir -p $P
cd $R
python3 bench.py > $P/${RND}_bench_line.json 2> $O/bench_default.err
echo "bench done"
rm -rf $O/prof_stats $O/pmc_fetch $O/pmc_write
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats -o st -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/bench_prof.json 2> $O/bench_prof.err
cp $(find $O/prof_stats -name 'st_kernel_stats.csv' | head -1) $P/${RND}_bench_kernel_stats.csv
python3 tools/trace_summary.py $O/prof_stats 3968 > $P/${RND}_bench_per_layer.txt
echo "stats done"
# the reference's own geometry (224 / 112): kernel stats of the same command the line's `p224` object times
rm -rf $O/prof_stats224
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stats224 -o st -- python3 bench.py --patch 224 --stride 112 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/bench_prof224.json 2> $O/bench_prof224.err
cp $(find $O/prof_stats224 -name 'st_kernel_stats.csv' | head -1) $P/${RND}_bench_p224_kernel_stats.csv
echo "p224 stats done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/pmc_f.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --train-steps 0 > $O/pmc_w.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 4096 $P/${RND}_pmc_dominant_kernel.json 3842 > /dev/null
echo "pmc traffic done"
rm -rf $O/pmc_mfma
rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $O/pmc_mfma -o c -- python3 tools/fwd_once.py 256 3 > $O/pmc_mfma.log 2>&1
python3 tools/pmc_mfma_util.py $O/pmc_mfma $P/${RND}_pmc_mfma_util.json
echo "pmc mfma done"
python3 tools/conv_stamps.py 256 > $P/${RND}_conv_phase_stamps_mb256.txt 2>/dev/null
for a in resnet18 resnet18bf16 resnet50; do
  rm -rf $O/prof_train_$a
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_train_$a -o t -- python3 tools/train_profile.py $a > $O/train_prof_$a.log 2>&1
  cp $(find $O/prof_train_$a -name 't_kernel_stats.csv' | head -1) $P/${RND}_train_${a}_kernel_stats.csv
  python3 tools/step_timeline.py $O/prof_train_$a $P/${RND}_timeline_${a}.txt
done
find $O -name '*counter_collection.csv' -size +20M -delete
find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $P
