#!/usr/bin/env python3
"""Run a MatDeepLearn-style training job on the HIP engine from a reference-format config.yml.

  python tools/run_training.py --config /path/to/config.yml --model CGCNN_demo --data pt10      # Pt10 fixture (1000 graphs)
  python tools/run_training.py --config cfg.yml --model SchNet_demo --data synthetic:4096 --dtype bf16

`--data DIR` reads a reference-style dataset directory (ASE-json structures + targets.csv, as
/root/reference/matdeeplearn/process/process.py:234-269 expects)."""
import argparse
import csv
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matdeeplearn_amd.process import from_structures, synthetic_bulk, graph as pg  # noqa: E402
from matdeeplearn_amd.training import load_config, train_regular  # noqa: E402


def load_dataset(spec, processing):
    r, k = processing.get("graph_max_radius", 8.0), processing.get("graph_max_neighbors", 12)
    if spec.startswith("synthetic"):
        n = int(spec.split(":")[1]) if ":" in spec else 4096
        return synthetic_bulk(n, seed=0, radius=r, max_neighbors=k)
    if spec == "pt10":
        z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pt10_dataset.npz"))
        structs = [dict(positions=z["positions"][s], numbers=z["numbers"][s], cell=z["cell"][s], pbc=z["pbc"][s])
                   for s in range(len(z["ids"]))]
        return from_structures(structs, z["y"], [str(v) for v in z["ids"]], r, k)
    rows = list(csv.reader(open(os.path.join(spec, processing.get("target_path", "targets.csv")))))
    fmt = processing.get("data_format", "json")
    structs = [pg.read_ase_json(os.path.join(spec, "%s.%s" % (row[0], fmt))) for row in rows]
    ys = np.array([[float(v) for v in row[1:]] for row in rows], dtype=np.float32)
    return from_structures(structs, ys, [row[0] for row in rows], r, k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--model", default=None)
    ap.add_argument("--data", default="pt10")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--graph-replay", default=None, choices=["auto", "True", "False"],
                    help="Training.graph_replay: run the training steps as HIP-graph replays (auto: for small batches)")
    a = ap.parse_args()
    job, processing, training, mp = load_config(a.config, "Training", a.model)
    if a.epochs:
        mp["epochs"] = a.epochs
    mp["compute_dtype"] = a.dtype
    if a.graph_replay:
        training["graph_replay"] = a.graph_replay
    ds = load_dataset(a.data, processing).to("cuda")
    edge_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    out = train_regular("cuda", 1, ds, job, training, mp, edge_dtype=edge_dtype)
    h = out["history"]
    edges = sum(x.get("edges", 0) for x in h[1:])
    secs = sum(x["time"] for x in h[1:])
    if secs > 0:
        steps = sum(-(-x.get("graphs", 0) // mp.get("batch_size", 100)) for x in h[1:])
        print("train edges/s (epochs 2..): %.3e   %.3f ms per step incl. validation (%d steps, %s)"
              % (edges / secs, secs / max(steps, 1) * 1e3, steps, "replayed" if "replays" in h[-1] else "eager"))


if __name__ == "__main__":
    main()
