cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_p3_gpu.py -x -q > $O/r02_p3test.log 2>&1; echo "exit $?" >> $O/r02_p3test.log
tail -4 $O/r02_p3test.log
timeout 600 python tools/p3_bench.py 2>&1 | tail -1 > $O/r02_p3_bench_dma.json; cat $O/r02_p3_bench_dma.json
WSI_P3_DMA=0 timeout 600 python tools/p3_bench.py 2>&1 | tail -1 > $O/r02_p3_bench_nodma.json; cat $O/r02_p3_bench_nodma.json
bash tools/pmc_p3.sh 2>&1 | head -22
