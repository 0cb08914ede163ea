cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5_kt -o c5 -- python3 bench.py --workload c5 > gpurun_out/c5_kt.log 2>&1 || exit 1
cat gpurun_out/c5_kt/c5_kernel_stats.csv | head -12
