for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-alt-gemm --no-kernel-timing --pcie 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:(round(v['ms_per_step'],2) if isinstance(v,dict) else '') for k,v in d['pcie_inclusive'].items()})"
done
