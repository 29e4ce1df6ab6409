set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -k "attention or scaled_cols" > gpurun_out/r4d/pytest_ops.txt 2>&1; echo "ops rc=$?" 
tail -5 gpurun_out/r4d/pytest_ops.txt
timeout -k 10 120 python tools/attn_debug.py > gpurun_out/r4d/attn_debug.txt 2>&1; echo "debug rc=$?"
cat gpurun_out/r4d/attn_debug.txt
timeout -k 10 300 python tools/attn_ab.py 37 10 > gpurun_out/r4d/attn_ab.txt 2>&1; echo "ab rc=$?"
cat gpurun_out/r4d/attn_ab.txt
MATRIX_EYES_HIP_LIB=$PWD/build_ab/libstamps.so timeout -k 10 300 python tools/attn_stamps.py > gpurun_out/r4d/attn_stamps.txt 2>&1; echo "stamps rc=$?"
grep "==\|whole\|prologue" gpurun_out/r4d/attn_stamps.txt
timeout -k 10 400 python tools/pmc_attention.py attn 0 > gpurun_out/r4d/pmc_attn.txt 2>&1; echo "pmc rc=$?"
tail -8 gpurun_out/r4d/pmc_attn.txt
