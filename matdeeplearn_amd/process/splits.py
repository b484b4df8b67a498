"""Train/val/test and CV splits — same index lists as the reference for the same seed.

Mirrors split_data / split_data_CV, /root/reference/matdeeplearn/process/process.py:27-79, which call
torch.utils.data.random_split(dataset, lengths, generator=torch.Generator().manual_seed(seed)):
one randperm(sum(lengths)) of the seeded CPU generator, cut into consecutive chunks.  Lengths use
int(n * ratio) (truncation) and the remainder is "unused" (process.py:37-40).  Pinned by
tests/golden/splits.npz.
"""
import numpy as np
import torch


def _random_chunks(lengths, seed):
    perm = torch.randperm(int(sum(lengths)), generator=torch.Generator().manual_seed(int(seed))).numpy()
    out, off = [], 0
    for ln in lengths:
        out.append(perm[off:off + ln].astype(np.int64))
        off += ln
    return out


def split_data(dataset_size, train_ratio, val_ratio, test_ratio, seed, verbose=False):
    """Returns (train_idx, val_idx, test_idx) as int64 numpy arrays."""
    n = int(dataset_size) if not hasattr(dataset_size, "__len__") else len(dataset_size)
    if train_ratio + val_ratio + test_ratio > 1:
        raise ValueError("invalid ratios")
    tr, va, te = int(n * train_ratio), int(n * val_ratio), int(n * test_ratio)
    unused = n - tr - va - te
    chunks = _random_chunks([tr, va, te, unused], seed)
    if verbose:
        print("train length:", tr, "val length:", va, "test length:", te, "unused length:", unused, "seed :", seed)
    return chunks[0], chunks[1], chunks[2]


def split_data_CV(dataset_size, num_folds=5, seed=0, verbose=False):
    n = int(dataset_size) if not hasattr(dataset_size, "__len__") else len(dataset_size)
    fold = int(n / num_folds)
    chunks = _random_chunks([fold] * num_folds + [n - fold * num_folds], seed)
    if verbose:
        print("fold length :", fold, "unused length:", n - fold * num_folds, "seed", seed)
    return chunks[:num_folds]
