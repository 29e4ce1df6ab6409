cd $GRAFT_REPO_ROOT
O=gpurun_out/r4fp8ln
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fp8.py -x -q > $O/pytest_fp8.txt 2>&1; echo "fp8 tests rc=$?"
tail -5 $O/pytest_fp8.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "layernorm" > $O/pytest_ln.txt 2>&1; echo "ln tests rc=$?"
tail -3 $O/pytest_ln.txt
for i in 1 2; do
ME_LN_FUSE=0 timeout -k 10 300 python bench.py --dtype fp8 --no-cpu-baseline > $O/bench_fp8_unfused_$i.json 2>/dev/null; echo "rc=$?"
timeout -k 10 300 python bench.py --dtype fp8 --no-cpu-baseline > $O/bench_fp8_fused_$i.json 2>/dev/null; echo "rc=$?"
done
for f in $O/bench_fp8_*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
