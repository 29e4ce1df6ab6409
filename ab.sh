set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "layernorm_fused or tall_tile" > gpurun_out/r4l/pytest_ops.txt 2>&1; echo "ops rc=$?"
tail -3 gpurun_out/r4l/pytest_ops.txt
for v in "ME_LN_FUSE=0" "ME_LN_FUSE=1"; do
  echo "== batch 8 $v" >> gpurun_out/r4l/bench_ab.txt
  env $v timeout -k 10 400 python bench.py --steps 6 --warmup 2 --batch 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [(k['kernel'][:44],k['launches_per_step'],k['ms_per_step']) for k in d['kernels'][:6]])" >> gpurun_out/r4l/bench_ab.txt 2>&1
  echo "== batch 1 $v" >> gpurun_out/r4l/bench_ab.txt
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [(k['kernel'][:44],k['launches_per_step'],k['ms_per_step']) for k in d['kernels'][:6]])" >> gpurun_out/r4l/bench_ab.txt 2>&1
done
cat gpurun_out/r4l/bench_ab.txt
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -q -x -k "full_size_batch" > gpurun_out/r4l/pytest_batch.txt 2>&1; echo "batch tests rc=$?"
tail -3 gpurun_out/r4l/pytest_batch.txt
