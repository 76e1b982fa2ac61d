cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_p3_gpu.py tests/test_dist_gpu.py -x -q > $O/r02_p3test.log 2>&1; echo "exit $?" >> $O/r02_p3test.log
tail -6 $O/r02_p3test.log
timeout 600 python tools/p3_bench.py > $O/r02_p3_bench.log 2>&1; tail -1 $O/r02_p3_bench.log
bash tools/pmc_p3.sh > $O/r02_pmc_p3.log 2>&1; head -45 $O/r02_pmc_p3.log
