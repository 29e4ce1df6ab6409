set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  ME_OVERLAP_TAIL=$mode timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4f/trace$mode -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r4f/bench_trace$mode.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r4f/trace$mode -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $f 2 -v > $GRAFT_REPO_ROOT/gpurun_out/r4f/timeline$mode.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4f/trace$mode
done
cd $GRAFT_REPO_ROOT
grep -n "busy per queue\|step of" gpurun_out/r4f/timeline0.txt gpurun_out/r4f/timeline1.txt
