cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4prof
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
grep '^{"metric"' $O/bench_under_rocprof.log > $O/bench_under_rocprof_f16.json
cp $(find $O/tr -name "*kernel_stats.csv" | head -1) $O/kernel_stats_f16.csv
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $f 4 > $O/step_timeline.txt 2>&1
rm -rf $O/tr
head -12 $O/kernel_stats_f16.csv | cut -c1-160
head -4 $O/step_timeline.txt
