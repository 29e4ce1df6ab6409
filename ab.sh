cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4trace
timeout -k 10 400 python tools/coresident_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4trace/coresident_ab.txt
cat gpurun_out/r4trace/coresident_ab.txt
