# PMC counters of one GEMM launch on the kqv forward shape for the three bf16-class kernels (separate passes, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/gb2.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from wsi_hgnn_amd import ops, _native as N
dev = torch.device("cuda:0")
mode = os.environ.get("MODE", "p3nt")
n, K, Nout = 80000, 512, 1536
x = torch.randn(n, K, device=dev); w = torch.randn(Nout, K, device=dev) * 0.03; y = torch.empty(n, Nout, device=dev)
if mode == "p3nt":
    xp, wp = ops.split_planes(x), ops.split_planes(w)
    f = lambda: ops.gemm_p3(N.WSI_GEMM_NT, 0, [dict(Ap=N.ptr(xp), ldap=xp.stride(0), Bp=N.ptr(wp), ldbp=wp.stride(0), C=N.ptr(y), ldc=Nout, M=n, N=Nout, K=K)], dev)
elif mode == "p3tn":
    dy = torch.randn(n, Nout, device=dev); dyp, xp = ops.split_planes(dy), ops.split_planes(x); gw = torch.empty(Nout, K, device=dev)
    f = lambda: ops.gemm_p3(N.WSI_GEMM_TN, 0, [dict(Ap=N.ptr(dyp), ldap=dyp.stride(0), Bp=N.ptr(xp), ldbp=xp.stride(0), C=N.ptr(gw), ldc=K, M=Nout, N=K, K=n)], dev)
else:
    ops.set_gemm_precision("bf16x6")
    f = lambda: ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(x), lda=K, B=N.ptr(w), ldb=K, C=N.ptr(y), ldc=Nout, M=n, N=Nout, K=K)], dev)
for _ in range(3):
    f()
torch.cuda.synchronize()
PY
for MODE in p3nt p3tn x6nt; do
echo "== $MODE"
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
rm -rf /tmp/pm; MODE=$MODE rocprofv3 --pmc $c --output-format csv -d /tmp/pm -o pm -- python /tmp/gb2.py > /dev/null 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "gemm" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no data", e)
for k, v in agg.items():
    print(k, v[-1])
PY
done
done
