for pad in 0 24000 40000; do echo "== LDS_PAD $pad"; WSI_GEMM_LDS_PAD=$pad python tools/gemm_bench.py 2>&1 | grep -v "^\[wsi" | sed -n 2,11p; done
