#!/bin/bash
# kernel trace of loader-fed steps (resident data set: a NEW batch per step) - what the 0.9 ms over the resident-batch step is made of
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/loader; mkdir -p $O
python $R/tools/loader_step_probe.py 12 > $O/host_phases.txt 2>&1
rm -rf /tmp/lt; timeout 600 rocprofv3 --kernel-trace -d /tmp/lt -o lt -- python $R/bench.py --pcie --no-cpu-baseline --no-alt-gemm --no-knn --no-full-depth --no-captured --no-training-config --steps 8 --warmup 2 > $O/bench.json 2>/dev/null
DB=$(find /tmp/lt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats.csv > /dev/null
tail -25 $O/host_phases.txt
