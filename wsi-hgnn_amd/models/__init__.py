"""nn.Module mirror of the reference's ``models/`` package for the hot path (models/__init__.py:5-29 —
the reference's own ``from .HAN import HAN`` at :10 points at a file that does not exist, so the
reference package does not import as shipped; that name is not mirrored)."""
from .HEATNet2 import HEATNet2  # noqa: F401
from .HEATNet4 import HEATNet4  # noqa: F401
from .HGT import HGT  # noqa: F401
from .HGT_ASAP import HGTASAP  # noqa: F401  (HGT + ASAPPooling readout: BASELINE configs[4]; no reference counterpart)
from .HetRGCN import HeteroRGCN  # noqa: F401
from .GCN import GCN  # noqa: F401
from .GCN_NTPool import NTPoolGCN  # noqa: F401



def from_config(config_gnn):
    """The model a reference ``GNN:`` config block constructs (``parser.parse_gnn_model``, parser.py:48-174)."""
    from ..parser import parse_gnn_model
    return parse_gnn_model(config_gnn)


__all__ = ["HEATNet2", "HEATNet4", "HGT", "HGTASAP", "HeteroRGCN", "GCN", "NTPoolGCN", "from_config"]
