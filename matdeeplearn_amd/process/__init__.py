"""Data side of the hot path: splits, structure->graph, flat dataset, device-side batch assembly."""
from .splits import split_data, split_data_CV  # noqa: F401
from . import graph  # noqa: F401
