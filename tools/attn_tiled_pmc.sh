# L2-blocked attention kernels vs the shipped ones on the bench batch: time per launch, fabric-side traffic (FETCH_SIZE x 2, WRITE_SIZE:
# separate passes, no trace domains), L2 hit rate.  usage (GPU box): bash tools/attn_tiled_pmc.sh [tag]  -> gpurun_out/<tag>_attn_tiled_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
OUT=$R/gpurun_out
mkdir -p $OUT
python $R/tools/attn_tiled_probe.py --iters 30 --json $OUT/${TAG}_attn_tiled_probe.json > $OUT/${TAG}_attn_tiled_probe.log 2>&1
tail -5 $OUT/${TAG}_attn_tiled_probe.log
U=${ATTN_U:-4}
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $R/tools/attn_tiled_probe.py --iters 5 --u $U > /dev/null 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_attn_tiled_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pm -o pm -- python $R/tools/attn_tiled_probe.py --iters 2 --u $U > /dev/null 2>&1
  cp $(find /tmp/pm -name "*counter_collection.csv" | head -1) /tmp/pmc_$n.csv
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv $OUT/${TAG}_attn_tiled_traffic.csv | grep -i "heat" 
python $R/tools/pmc_l2.py /tmp/pmc_TCC_HIT_sum.csv $OUT/${TAG}_attn_tiled_l2.csv | grep -i "heat"
