# PMC counters of one projection launch (kqv forward shape, 80000 x 1536 x 512) under the scaled-fp16 kernels: g = gemm_fp16x3g_kernel
# (LDS-DMA staged), g1..g5 = its measurement variants (WSI_F16G_ABL), w = gemm_fp16x3w_kernel, h / h2 / h3 = gemm_fp16x3h_kernel and its variants.  Separate --pmc passes, no trace domains.
# usage (GPU box): MODES="g g5 w" bash tools/pmc_f16g.sh > gpurun_out/pmc_f16g.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gb4.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from wsi_hgnn_amd import _native as N
N.use_measurement_library()
from wsi_hgnn_amd import ops
dev = torch.device("cuda:0")
n, K, Nout = 80000, 512, 1536
x = torch.rand(n, K, device=dev); w = torch.randn(Nout, K, device=dev) * 0.03; y = torch.empty(n, Nout, device=dev)
bits = ops.row_absmax(x)
ops.set_gemm_precision("fp16x3")
f = lambda: ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(x), lda=K, B=N.ptr(w), ldb=K, C=N.ptr(y), ldc=Nout, M=n, N=Nout, K=K, a_absmax=N.ptr(bits), a_absmax_parts=1)], dev)
for _ in range(3):
    f()
torch.cuda.synchronize()
PY
for MODE in ${MODES:-g g5 w}; do
echo "== $MODE"
unset WSI_GEMM_F16_KERNEL WSI_F16G_ABL
case $MODE in
  w) export WSI_GEMM_F16_KERNEL=w;;
  h|h2|h3) export WSI_GEMM_F16_KERNEL=$MODE;;      # the 256 x 128 one-wave-per-SIMD kernel (round 4) and its no-store / no-DMA variants
  g) ;;
  g*) export WSI_F16G_ABL=${MODE#g};;
esac
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_INSTS_WAVE32_LDS" "TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
rm -rf /tmp/pm; timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pm -o pm -- python /tmp/gb4.py > /dev/null 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "gemm_fp16x3" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no data", e)
for k, v in agg.items():
    print(k, v[-1])
PY
done
done
