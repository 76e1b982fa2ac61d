# Regenerates the round's profile artefacts on the GPU box into gpurun_out/ (copy the ones to keep into profiles/).
# Usage (GPU box): bash tools/profile_round.sh [r06]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=${1:-r06}
mkdir -p $O
BENCH="python $R/bench.py --no-pmc --no-cpu-baseline --no-alt-gemm --no-knn --no-full-depth --no-training-config --no-captured --steps 8 --warmup 2"
# 1. per-kernel durations (rocprofv3 kernel trace of the bench command): the default arithmetic (fp16x3), then the other two
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- $BENCH > $O/kt_bench.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $O/${T}_kernel_stats.csv > /dev/null
python $R/tools/step_timeline.py $(find /tmp/kt -name "*.db" | head -1) $O/${T}_step_timeline.txt > /dev/null
# (the same with every launch in order - no weight gradient running under an attention backward: clean per-kernel durations of the TN launches)
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- $BENCH --background-dw off > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $O/${T}_kernel_stats_in_order.csv > /dev/null
for g in fp32 bf16x6; do
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- $BENCH --gemm $g > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $O/${T}_kernel_stats_$g.csv > /dev/null
done
# 2. fabric-side traffic: FETCH_SIZE and WRITE_SIZE in separate PMC passes (no trace domains); L2 hit/miss in a third
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c; timeout 400 rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -o pm -- python $R/bench.py --no-pmc --no-cpu-baseline --no-alt-gemm --no-knn --no-full-depth --no-captured --no-training-config --no-kernel-timing --steps 2 --warmup 1 > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/${T}_hbm_traffic_pmc.csv > /dev/null
rm -rf /tmp/pm_l2; timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d /tmp/pm_l2 -o pm -- python $R/bench.py --no-pmc --no-cpu-baseline --no-alt-gemm --no-knn --no-full-depth --no-captured --no-training-config --no-kernel-timing --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_l2.py $(find /tmp/pm_l2 -name "*counter_collection.csv" | head -1) $O/${T}_l2_hit_pmc.csv > /dev/null
# 3. bench lines (default incl. CPU baseline and both GEMM arithmetics; hub destinations; PCIe-inclusive legs; the reference's real schema)
mkdir -p $R/profiles; cp $O/${T}_hbm_traffic_pmc.csv $R/profiles/ 2>/dev/null     # bench.py reads the traffic figure from profiles/
python $R/bench.py 2>/dev/null | tail -1 > $O/${T}_bench_default.json      # (roofline.traffic measured by the run itself: the PMC child passes are the default)
python $R/bench.py --dst-mode hub --no-cpu-baseline 2>/dev/null | tail -1 > $O/${T}_bench_hub.json
python $R/bench.py --pcie --no-cpu-baseline --no-alt-gemm 2>/dev/null | tail -1 > $O/${T}_bench_pcie.json
python $R/bench.py --schema real --no-cpu-baseline 2>/dev/null | tail -1 > $O/${T}_bench_real_schema.json
# (launch-bound configurations: the first ~100 steps on a fresh box run slow - 3.38 ms after 5 warm-up steps, 2.69 after 100 - so they get a long warm-up)
python $R/bench.py --model HEATNet2 --hidden 256 --nodes 5000 --no-cpu-baseline --warmup 100 --steps 50 2>/dev/null | tail -1 > $O/${T}_bench_heatnet2_config2.json
python $R/bench.py --dropout 0.2 --no-cpu-baseline --no-alt-gemm --no-knn 2>/dev/null | tail -1 > $O/${T}_bench_dropout.json
python $R/bench.py --batch 2 --dropout 0.2 --pcie --no-cpu-baseline --no-alt-gemm --no-knn --warmup 100 --steps 50 2>/dev/null | tail -1 > $O/${T}_bench_reference_regime.json
WARM=10 STEPS=40 python $R/tools/hgt_bench.py 2>/dev/null | tail -1 > $O/${T}_hgt_config5_auto.json
WARM=10 STEPS=40 ASAP=1 python $R/tools/hgt_bench.py 2>/dev/null | tail -1 > $O/${T}_hgt_asap_config5_auto.json
python $R/tools/tn16_bench.py --json $O/${T}_tn16_bench.json > /dev/null 2>&1
head -14 $O/${T}_kernel_stats.csv | cut -c1-200
cat $O/${T}_hbm_traffic_pmc.csv | head -12
cat $O/${T}_l2_hit_pmc.csv | head -12
python - <<PY
import json
for f in ("bench_default", "bench_hub", "bench_pcie", "bench_real_schema", "bench_heatnet2_config2"):
    d = json.load(open("$O/${T}_%s.json" % f))
    print(f, round(d["ms_per_step"], 3), round(d["value"]), d["roofline"]["achieved"] if d.get("roofline") else None,
          (round(d["alt_gemm"]["ms_per_step"], 3) if d.get("alt_gemm") else None), d.get("cpu_baseline", None) and d["cpu_baseline"]["value"],
          d.get("pcie_inclusive") and {k: round(v["ms_per_step"], 2) for k, v in d["pcie_inclusive"].items() if isinstance(v, dict)})
PY
