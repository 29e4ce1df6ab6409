cd $GRAFT_REPO_ROOT
O=gpurun_out/r4head
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "head_final" 2>&1 | tail -2
timeout -k 10 300 python tools/head_probe.py 2>&1 | grep -v amdgpu.ids > $O/head_probe4.txt
cat $O/head_probe4.txt
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q > $O/pytest_pipeline.txt 2>&1; echo "pipeline rc=$?"
tail -3 $O/pytest_pipeline.txt
for i in 1 2; do
ME_HEAD_HALO=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_implicit_$i.json 2>/dev/null; echo "rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_halo_$i.json 2>/dev/null; echo "rc=$?"
done
for f in implicit_1 halo_1 implicit_2 halo_2; do python -c "import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
