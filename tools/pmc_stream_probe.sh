cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/attn_tiled_probe.py --only stride --graphs 64 --iters 10 2>&1 | grep -v amdgpu.ids
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE" "TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"; do
rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/pm -o pm -- python $R/tools/attn_tiled_probe.py --only stream --u 4 --iters 1 > /dev/null 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "stream_aggregate" in r["Kernel_Name"] and "Li4ELi8E" in r["Kernel_Name"].replace(" ","").replace("<16,4,8>","Li4ELi8E"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no data", e)
for k, v in agg.items():
    print(k, v[-1], len(v))
PY
done
