cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/gelu_ab.py 2>&1 | grep -v amdgpu.ids
