set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final
mkdir -p $O
timeout -k 10 1100 python3 tools/pmc_collect.py $O/pmc_kernels.json qkv_tall:10 fc1_tall:10 proj_tall:10 fc2_tall:10 conv768:9 attn:0 qkv8:0 fc1_8:0 fc2_8:0 > $O/pmc_collect.log 2>&1; echo "pmc rc=$?"
tail -5 $O/pmc_collect.log
rm -rf gpurun_out/pmc
