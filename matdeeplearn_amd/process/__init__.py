"""Data side of the hot path: splits, structure->graph, flat dataset, device-side batch assembly."""
from .splits import split_data, split_data_CV  # noqa: F401
from . import graph  # noqa: F401
from .dataset import StaticBatch, static_capacity  # noqa: F401
from .dataset import GraphDataset, Batch, DeviceLoader, from_graphs, from_structures, synthetic_bulk, synthetic_mof, synthetic_surface  # noqa: F401
