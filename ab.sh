set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout -k 10 120 python tools/ln_debug.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "layernorm_fused" > gpurun_out/r4g/pytest_ops.txt 2>&1; echo "ops rc=$?"
tail -3 gpurun_out/r4g/pytest_ops.txt
rm -f gpurun_out/r4g/bench_ab.txt
for rep in 1 2; do
for v in "ME_LN_FUSE=0" "ME_LN_FUSE=1"; do
  echo "== $v" >> gpurun_out/r4g/bench_ab.txt
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [(k['kernel'][:44],k['launches_per_step'],k['ms_per_step']) for k in d['kernels'][:6]])" >> gpurun_out/r4g/bench_ab.txt 2>&1
done; done
cat gpurun_out/r4g/bench_ab.txt
