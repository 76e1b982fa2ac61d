"""tools/ only: the WSI_* environment variables the measurement scripts have always been driven with, mapped onto the package's setters.
The PACKAGE reads no environment variable (tests/test_boundary.py); a script under tools/ that wants the old command lines
(`WSI_GEMM_PRECISION=auto python tools/hgt_bench.py`, `WSI_BACKGROUND_DW=0 ...`) calls ``apply()`` once after importing the package."""
import os


def apply() -> dict:
    from wsi_hgnn_amd import graph, ops
    from wsi_hgnn_amd.models import heat_net
    e = os.environ
    done = {}
    if e.get("WSI_GEMM_PRECISION"):
        ops.set_gemm_precision(e["WSI_GEMM_PRECISION"]); done["gemm"] = e["WSI_GEMM_PRECISION"]
    if "WSI_BACKGROUND_DW" in e:
        ops.set_background_weight_gradients(e["WSI_BACKGROUND_DW"] != "0"); done["background_dw"] = e["WSI_BACKGROUND_DW"] != "0"
    if "WSI_COLLAPSE_V" in e or "WSI_COLLAPSE_V_MIN_WORK" in e:
        ops.set_value_collapse(e.get("WSI_COLLAPSE_V", "1") != "0", float(e["WSI_COLLAPSE_V_MIN_WORK"]) if "WSI_COLLAPSE_V_MIN_WORK" in e else None)
    if "WSI_LOW_RANK_READOUT_GRAD" in e:
        ops.set_low_rank_readout_grad(e["WSI_LOW_RANK_READOUT_GRAD"] != "0")
    if "WSI_FUSE_READOUT" in e:
        heat_net.HEATTrunk.fuse_readout = e["WSI_FUSE_READOUT"] != "0"
    graph.set_plan_options(heavy_degree=int(e["WSI_HEAVY_DEGREE"]) if "WSI_HEAVY_DEGREE" in e else None,
                           heavy_degree_locality=int(e["WSI_HEAVY_DEGREE_LOCALITY"]) if "WSI_HEAVY_DEGREE_LOCALITY" in e else None,
                           hub_split=(e["WSI_HUB_SPLIT"] != "0") if "WSI_HUB_SPLIT" in e else None,
                           locality=(e["WSI_LOCALITY"] != "0") if "WSI_LOCALITY" in e else None)
    return done
