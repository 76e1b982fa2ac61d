# kernel trace of any tool: KT_CMD="python tools/tn16_bench.py" KT_OUT=name bash tools/kt_tool.sh   (GPU box; writes gpurun_out/<name>_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/kt; (cd $R && timeout ${KT_TIMEOUT:-600} rocprofv3 --kernel-trace -d /tmp/kt -o kt -- $KT_CMD > $O/${KT_OUT}_stdout.txt 2>/dev/null)
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB $O/${KT_OUT}_kernel_stats.csv > /dev/null
head -${KT_HEAD:-25} $O/${KT_OUT}_kernel_stats.csv | cut -c1-200
