cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in hub; do
rm -rf /tmp/prof_$d
rocprofv3 --kernel-trace -d /tmp/prof_$d -o p -- python $R/bench.py --no-cpu-baseline --no-alt-gemm --no-knn --no-kernel-timing --steps 8 --warmup 2 --dst-mode $d > /dev/null 2>&1
db=$(find /tmp/prof_$d -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $R/gpurun_out/stats_$d.csv > /dev/null
grep -E "heat_attn|egrad" $R/gpurun_out/stats_$d.csv | sed 's/(wsi::AttnTables.*)"/"/; s/(float const.*)"/"/'
done
