cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_p3_gpu.py -x -q > $O/r02_p3test.log 2>&1; echo "exit $?" >> $O/r02_p3test.log
tail -15 $O/r02_p3test.log
timeout 600 python tools/p3_bench.py > $O/r02_p3_bench.log 2>&1; tail -8 $O/r02_p3_bench.log
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_p3_gpu.py > $O/r02_gputest2.log 2>&1; echo "exit $?" >> $O/r02_gputest2.log
tail -15 $O/r02_gputest2.log
