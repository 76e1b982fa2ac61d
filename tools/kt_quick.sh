cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
BENCH="python $R/bench.py --no-cpu-baseline --no-alt-gemm --no-knn --no-full-depth --no-captured --steps 8 --warmup 2 $KT_ARGS"
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- $BENCH > $O/kt_bench.json 2>/dev/null
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
python $R/tools/step_timeline.py $DB $O/step_timeline.txt > /dev/null
head -30 $O/kernel_stats.csv | cut -c1-160
