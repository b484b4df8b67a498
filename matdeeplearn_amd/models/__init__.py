"""Model registry — the drop-in boundary b1: the reference looks models up with
`getattr(models, model_name)(data=dataset, **model_params)` (matdeeplearn/training/training.py:250-252;
registry matdeeplearn/models/__init__.py:1-16)."""
from .cgcnn import CGCNN

__all__ = ["CGCNN"]
