cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${OUT:-hgt_stats.csv}
rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/hgt_bench.py > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $R/gpurun_out/$OUT > /dev/null
head -40 $R/gpurun_out/$OUT | cut -c1-150 | awk -F, '{print $0}' 
