set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
rm -f gpurun_out/r4h/bench_ab.txt
for rep in 1 2; do
for v in "ME_LN_FUSE=0" "ME_LN_FUSE=1"; do
  echo "== $v" >> gpurun_out/r4h/bench_ab.txt
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [(k['kernel'][:44],k['launches_per_step'],k['ms_per_step']) for k in d['kernels'][:8]])" >> gpurun_out/r4h/bench_ab.txt 2>&1
done; done
cat gpurun_out/r4h/bench_ab.txt
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4h/pytest_gpu.txt 2>&1; echo "gpu tests rc=$?"
tail -8 gpurun_out/r4h/pytest_gpu.txt
