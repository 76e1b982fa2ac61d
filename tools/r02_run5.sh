cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python bench.py --schema real --no-cpu-baseline 2> $O/r02_bench_real.err | tail -1 > $O/r02_bench_real_schema.json
python - <<PY
import json
d = json.load(open("$O/r02_bench_real_schema.json"))
print("real", round(d["ms_per_step"], 3), round(d["value"]), d["roofline"], d["edge_phase_roofline"], d["alt_gemm"] and round(d["alt_gemm"]["ms_per_step"], 3), d["config"]["relations"])
PY
cd /tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --schema real --no-cpu-baseline --no-alt-gemm --steps 8 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $O/r02_kernel_stats_real_schema.csv | head -40 | cut -c1-180
