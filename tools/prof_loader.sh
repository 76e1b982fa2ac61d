cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kl; rocprofv3 --kernel-trace -d /tmp/kl -o kl -- python $R/tools/loader_loop.py > /tmp/loader_out.txt 2>&1
tail -3 /tmp/loader_out.txt
python $R/tools/rocpd_stats.py $(find /tmp/kl -name "*.db" | head -1) $R/gpurun_out/loader_stats.csv > /dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/loader_stats.csv")))
tot = sum(float(r["total_us"]) for r in rows)
model = sum(float(r["total_us"]) for r in rows if "gemm" in r["name"] or "heat_attn" in r["name"] or "seg_" in r["name"] or "splitk" in r["name"] or "Adam" in r["name"] or "egrad" in r["name"])
print("total kernel us", round(tot), "model-kernel us", round(model), "other", round(tot - model))
for r in rows:
    n = r["name"]
    if any(k in n for k in ("gemm", "heat_attn", "seg_", "splitk", "Adam", "egrad")):
        continue
    if float(r["total_us"]) > 0.004 * tot:
        print(f'{float(r["total_us"]):10.0f} {r["calls"]:>6} {n[:120]}')
PY
