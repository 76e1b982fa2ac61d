cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/gb1.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from wsi_hgnn_amd import ops, _native as N
dev = torch.device("cuda:0")
ops.set_gemm_precision(os.environ.get("MODE", "bf16x6"))
n, K, Nout = 80000, 512, 1536
x = torch.randn(n, K, device=dev); w = torch.randn(Nout, K, device=dev) * 0.03; y = torch.empty(n, Nout, device=dev)
for _ in range(3):
    ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(x), lda=K, B=N.ptr(w), ldb=K, C=N.ptr(y), ldc=Nout, M=n, N=Nout, K=K)], dev)
torch.cuda.synchronize()
PY
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
rm -rf /tmp/pm; rocprofv3 --pmc $c --output-format csv -d /tmp/pm -o pm -- python /tmp/gb1.py > /dev/null 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, v[-1])
PY
done
