# PMC counters of one projection launch (kqv forward shape) under the three emulated-GEMM kernels (separate passes, no trace domains)
# MODES: x6 (gemm_bf16x6_kernel), f16w (gemm_fp16x3w_kernel).  (profiles/r02_pmc_emu_kernels.log also holds 'f16': the
# variant that split BOTH operands in the kernel, since removed.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/gb3.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from wsi_hgnn_amd import ops, _native as N
dev = torch.device("cuda:0")
n, K, Nout = 80000, 512, 1536
x = torch.randn(n, K, device=dev); w = torch.randn(Nout, K, device=dev) * 0.03; y = torch.empty(n, Nout, device=dev)
ops.set_gemm_precision(os.environ.get("PREC", "bf16x6"))
f = lambda: ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(x), lda=K, B=N.ptr(w), ldb=K, C=N.ptr(y), ldc=Nout, M=n, N=Nout, K=K)], dev)
for _ in range(3):
    f()
torch.cuda.synchronize()
PY
for MODE in ${MODES:-x6 f16w}; do
echo "== $MODE"
case $MODE in
  x6) export PREC=bf16x6;;
  f16w) export PREC=fp16x3;;
esac
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
rm -rf /tmp/pm; timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pm -o pm -- python /tmp/gb3.py > /dev/null 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if ("gemm_bf16x6" in r["Kernel_Name"] or "fp16x3w" in r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no data", e)
for k, v in agg.items():
    print(k, v[-1])
PY
done
done
