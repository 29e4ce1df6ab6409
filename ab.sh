set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final4
mkdir -p $O
timeout -k 10 1100 python3 tools/pmc_collect.py $O/pmc_kernels.json qkv_tall:10 fc1_tall:10 proj_tall:10 fc2_tall:10 conv768:9 attn:0 qkv8:0 fc1_8:0 fc2_8:0 > $O/pmc_collect.log 2>&1; echo "pmc rc=$?"
cp $O/pmc_kernels.json profiles/r04_pmc_kernels.json
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "gpu tests rc=$?"
tail -4 $O/pytest_gpu.txt
timeout -k 10 600 python bench.py > $O/bench_f16.json 2> $O/bench_f16.err; echo "A rc=$?"
timeout -k 10 300 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_f16_batch8.json 2>/dev/null; echo "B rc=$?"
timeout -k 10 300 python bench.py --dtype fp8 --no-cpu-baseline > $O/bench_fp8.json 2>/dev/null; echo "C1 rc=$?"
timeout -k 10 300 python bench.py --dtype fp8 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fp8_batch8.json 2>/dev/null; echo "C2 rc=$?"
timeout -k 10 300 python bench.py --chain --no-cpu-baseline > $O/bench_chain.json 2>/dev/null; echo "D1 rc=$?"
timeout -k 10 300 python bench.py --chain --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_chain_batch8.json 2>/dev/null; echo "D2 rc=$?"
timeout -k 10 300 python bench.py --graph --no-cpu-baseline > $O/bench_graph.json 2>/dev/null; echo "E rc=$?"
set +x
for f in bench_f16 bench_f16_batch8 bench_fp8 bench_fp8_batch8 bench_chain bench_chain_batch8 bench_graph; do python -c "import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$f', d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic'), r.get('mfma_util_pmc'))"; done
