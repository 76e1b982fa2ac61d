# Fabric-side bytes and L2 hit rate of the attention kernels on the WSI-like (kNN) study graphs, per node order (separate PMC passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for MODE in raw locality; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pl_$c; MODE=$MODE rocprofv3 --pmc $c --output-format csv -d /tmp/pl_$c -o pm -- python $R/tools/locality_study.py > /dev/null 2>&1
  done
  python $R/tools/pmc_traffic.py $(find /tmp/pl_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pl_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/r02_locality_traffic_$MODE.csv | grep heat_attn
  rm -rf /tmp/pl_l2; MODE=$MODE rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d /tmp/pl_l2 -o pm -- python $R/tools/locality_study.py > /dev/null 2>&1
  python $R/tools/pmc_l2.py $(find /tmp/pl_l2 -name "*counter_collection.csv" | head -1) $O/r02_locality_l2_$MODE.csv | grep heat_attn
done
