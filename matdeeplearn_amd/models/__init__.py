"""Model registry — the drop-in boundary b1: the reference looks models up with
`getattr(models, model_name)(data=dataset, **model_params)` (matdeeplearn/training/training.py:250-252;
registry matdeeplearn/models/__init__.py:1-16).  SOAP / SM descriptor MLPs are out of scope (no
message passing; need dscribe features)."""
from .cgcnn import CGCNN
from .gcn import GCN
from .megnet import MEGNet
from .mpnn import MPNN
from .schnet import SchNet

__all__ = ["GCN", "MPNN", "SchNet", "CGCNN", "MEGNet"]
