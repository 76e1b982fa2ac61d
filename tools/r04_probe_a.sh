#!/bin/bash
# round-4 side measurements: the single-graph step with / without the background weight gradients; configs[4] under auto
cd $GRAFT_REPO_ROOT
for bg in 1 0; do
  echo "== HEATNet4 one 10k graph, WSI_BACKGROUND_DW=$bg"
  WSI_BACKGROUND_DW=$bg python tools/graph_capture_probe.py --model HEATNet4 --hidden 512 --nodes 10000 --batch 1 --steps 100 2>&1 | tail -3
done
for bg in 1 0; do
  echo "== HEATNet2 one 5k graph hidden 256, WSI_BACKGROUND_DW=$bg"
  WSI_BACKGROUND_DW=$bg python tools/graph_capture_probe.py --model HEATNet2 --hidden 256 --nodes 5000 --batch 1 --steps 100 2>&1 | tail -3
done
echo "== configs[4] auto"
WSI_GEMM_PRECISION=auto ASAP=1 python tools/hgt_bench.py 2>/dev/null | tail -1 | tee gpurun_out/r04_hgt_asap_config5_auto.json
WSI_GEMM_PRECISION=auto python tools/hgt_bench.py 2>/dev/null | tail -1 | tee gpurun_out/r04_hgt_config5_auto.json
WSI_GEMM_PRECISION=auto WSI_BACKGROUND_DW=0 ASAP=1 python tools/hgt_bench.py 2>/dev/null | tail -1
