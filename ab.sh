cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final5
mkdir -p $O
timeout -k 10 300 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_f16_batch8.json 2>/dev/null; echo "B rc=$?"
timeout -k 10 300 python bench.py --dtype fp8 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fp8_batch8.json 2>/dev/null; echo "C2 rc=$?"
timeout -k 10 300 python bench.py --chain --no-cpu-baseline > $O/bench_chain.json 2>/dev/null; echo "D1 rc=$?"
timeout -k 10 300 python bench.py --chain --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_chain_batch8.json 2>/dev/null; echo "D2 rc=$?"
timeout -k 10 300 python bench.py --graph --no-cpu-baseline > $O/bench_graph.json 2>/dev/null; echo "E rc=$?"
for f in bench_f16_batch8 bench_fp8_batch8 bench_chain bench_chain_batch8 bench_graph; do python -c "import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$f', d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic_source'))"; done
