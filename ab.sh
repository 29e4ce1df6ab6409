cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4trace
for lim in 256 128 64; do
echo "ME_GEMM_GRID_LIMIT=$lim"
BW_PROBE_AUTO=1 ME_GEMM_GRID_LIMIT=$lim timeout -k 10 200 python tools/bw_bound_probe.py 2>&1 | grep -v "amdgpu.ids\|ME_STAGGER"
done > gpurun_out/r4trace/grid_limit.txt
cat gpurun_out/r4trace/grid_limit.txt
