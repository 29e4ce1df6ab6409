set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_f16.json 2> $O/bench_f16.err; echo "A rc=$?"; tail -c 600 $O/bench_f16.json
timeout -k 10 300 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_f16_batch8.json 2>/dev/null; echo "B rc=$?"
timeout -k 10 300 python bench.py --dtype fp8 --no-cpu-baseline > $O/bench_fp8.json 2>/dev/null; echo "C1 rc=$?"
timeout -k 10 300 python bench.py --dtype fp8 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fp8_batch8.json 2>/dev/null; echo "C2 rc=$?"
timeout -k 10 300 python bench.py --chain --no-cpu-baseline > $O/bench_chain.json 2>/dev/null; echo "D1 rc=$?"
timeout -k 10 300 python bench.py --chain --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_chain_batch8.json 2>/dev/null; echo "D2 rc=$?"
timeout -k 10 300 python bench.py --graph --no-cpu-baseline > $O/bench_graph.json 2>/dev/null; echo "E rc=$?"
for f in bench_f16_batch8 bench_fp8 bench_fp8_batch8 bench_chain bench_chain_batch8 bench_graph; do python -c "import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 18 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof_f16.json 2>/dev/null; echo "F rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_f16.csv
t=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 tools/step_timeline.py $t 2 > $O/step_timeline.txt 2>&1
rm -rf $O/prof
head -12 $O/kernel_stats_f16.csv | cut -c1-160
head -3 $O/step_timeline.txt
timeout -k 10 200 python3 tools/node_write_ceiling.py /dev/shm 110 3 > $O/node_write_ceiling.txt 2>&1; echo "G rc=$?"
cat $O/node_write_ceiling.txt
