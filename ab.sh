set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -k "attention or scaled_cols" > gpurun_out/r4e/pytest_ops.txt 2>&1; echo "ops rc=$?"
tail -3 gpurun_out/r4e/pytest_ops.txt
timeout -k 10 500 python -m pytest tests/test_gpu_pipeline.py -q -x -k "tiny or graph or batch_equals or determinism or u8_entry" > gpurun_out/r4e/pytest_pipe.txt 2>&1; echo "pipe rc=$?"
tail -5 gpurun_out/r4e/pytest_pipe.txt
for rep in 1 2; do
for v in "ME_OVERLAP_TAIL=0" "ME_OVERLAP_TAIL=1" "ME_OVERLAP_TAIL=1 ME_OVERLAP_CAP=128" "ME_OVERLAP_TAIL=1 ME_OVERLAP_CAP=224" "ME_OVERLAP_TAIL=1 ME_OVERLAP_CAP=0"; do
  echo "== $v" >> gpurun_out/r4e/bench_ab.txt
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [(k['kernel'][:40],k['ms_per_step']) for k in d['kernels'][:5]])" >> gpurun_out/r4e/bench_ab.txt 2>&1
done; done
cat gpurun_out/r4e/bench_ab.txt
ME_OVERLAP_TAIL=1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --graph > gpurun_out/r4e/bench_graph.txt 2>&1; tail -1 gpurun_out/r4e/bench_graph.txt | cut -c1-300
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -q -x -s -k "full_size_pairs" > gpurun_out/r4e/pytest_pairs.txt 2>&1; echo "pairs rc=$?"
grep "full-size f16 pair\|passed\|failed\|Error" gpurun_out/r4e/pytest_pairs.txt
