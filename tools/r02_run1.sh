# round-2 GPU call 1: full GPU suite after the host-side refactors, L2 hit/miss counters of the attention kernels, baseline bench lines
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/r02_gputest1.log 2>&1
echo "pytest exit $?" >> $O/r02_gputest1.log
tail -5 $O/r02_gputest1.log
cd /tmp
rm -rf /tmp/pm_l2; timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d /tmp/pm_l2 -o pm -- python $R/bench.py --no-cpu-baseline --no-alt-gemm --no-kernel-timing --steps 2 --warmup 1 > $O/pmc_l2.log 2>&1
python $R/tools/pmc_l2.py $(find /tmp/pm_l2 -name "*counter_collection.csv" | head -1) $O/r02_l2_hit_pmc.csv | head -20
timeout 600 python $R/bench.py 2> $O/r02_bench_base.err | tail -1 > $O/r02_bench_base.json
python - <<PY
import json
d = json.load(open("$O/r02_bench_base.json"))
print("base", round(d["ms_per_step"], 3), round(d["value"]), d["roofline"]["achieved"], d["alt_gemm"] and round(d["alt_gemm"]["ms_per_step"], 3), d["cpu_baseline"])
PY
