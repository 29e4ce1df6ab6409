set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -k "attention" > gpurun_out/r4i/pytest_ops.txt 2>&1; echo "ops rc=$?"
tail -3 gpurun_out/r4i/pytest_ops.txt
ME_ATT_HALVES=1 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -k "attention" > gpurun_out/r4i/pytest_ops_halves.txt 2>&1; echo "ops halves rc=$?"
tail -3 gpurun_out/r4i/pytest_ops_halves.txt
timeout -k 10 300 python tools/attn_ab.py 37 10 > gpurun_out/r4i/attn_ab.txt 2>&1; echo "ab rc=$?"
cat gpurun_out/r4i/attn_ab.txt
for rep in 1 2; do
for v in "ME_ATT_HALVES=0" "ME_ATT_HALVES=1"; do
  echo "== $v" >> gpurun_out/r4i/bench_ab.txt
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [(k['kernel'][:44],k['launches_per_step'],k['ms_per_step']) for k in d['kernels'][:5]])" >> gpurun_out/r4i/bench_ab.txt 2>&1
done; done
cat gpurun_out/r4i/bench_ab.txt
