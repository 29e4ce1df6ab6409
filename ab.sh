cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4trace
timeout -k 10 300 python tools/convt_vs_store_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4trace/convt_vs_store.txt
cat gpurun_out/r4trace/convt_vs_store.txt
