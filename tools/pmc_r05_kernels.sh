# PMC counters of one K|Q|V-shaped launch of the two scaled-fp16 kernels of round 5 (separate passes, no trace domains):
#   g    gemm_fp16x3g_kernel  Y = X W^T, 80000 x 1536 x 512 (three node-type groups)
#   tn   gemm_tn16_kernel     dW = dY^T X, 3 x (512 x 512) per node type over 40000 / 24000 / 16000 rows (statistics pre-pass inside the launch)
# -> matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES), LDS bank conflicts, instruction mix, waits.  usage (GPU box): bash tools/pmc_r05_kernels.sh > gpurun_out/r05_pmc_gemm_kernels.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/gb5.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from wsi_hgnn_amd import ops, _native as N
dev = torch.device("cuda:0")
ops.set_gemm_precision("fp16x3")
rows = [(0, 40000), (40000, 64000), (64000, 80000)]
n, D = 80000, 512
torch.manual_seed(0)
x = torch.randn(n, D, device=dev); ws = [torch.randn(3 * D, D, device=dev) * 0.03 for _ in rows]; y = torch.empty(n, 3 * D, device=dev)
gy = torch.randn(n, 3 * D, device=dev) * 1e-3
dws = [torch.empty(D, D, device=dev) for _ in range(9)]
def g():
    ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(x, r0 * D * 4), lda=D, B=N.ptr(ws[t]), ldb=D, C=N.ptr(y, r0 * 3 * D * 4), ldc=3 * D, M=r1 - r0, N=3 * D, K=D)
                                 for t, (r0, r1) in enumerate(rows)], dev)
def tn():
    ops._gemm(N.WSI_GEMM_TN, 0, [dict(A=N.ptr(gy, (r0 * 3 * D + j * D) * 4), lda=3 * D, B=N.ptr(x, r0 * D * 4), ldb=D, C=N.ptr(dws[3 * t + j]), ldc=D, M=D, N=D, K=r1 - r0)
                                 for t, (r0, r1) in enumerate(rows) for j in range(3)], dev)
f = {"g": g, "tn": tn}[os.environ.get("WHICH", "g")]
for _ in range(3):
    f()
torch.cuda.synchronize()
PY
for W in g tn; do
echo "== $W"
export WHICH=$W
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
rm -rf /tmp/pm; timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pm -o pm -- python /tmp/gb5.py > /dev/null 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "gemm_fp16x3g" in r["Kernel_Name"] or "gemm_tn16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no data", e)
for k, v in agg.items():
    print(k, v[-1])
PY
done
done
