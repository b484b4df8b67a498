"""Job drivers — the outer loops of /root/reference/matdeeplearn/training/training.py restated around the
device-resident dataset:  train_regular (:377-539), train_CV (:587-715), train_repeat (:719-843),
train_ensemble (:1069-1196), predict (:543-583, from a state_dict instead of a pickled full model).
They read the same YAML keys as the reference's config.yml (sections Job / Processing / Training /
Models, string booleans) and write the same CSV artefacts (`*_outputs.csv`, `_errorvalues.csv`).
The drivers are model-agnostic: `model_factory(name)` returns the class; default = the HIP models.
"""
import csv
import os
import time

import numpy as np
import torch
import yaml

from ..process import DeviceLoader, split_data, split_data_CV
from .dp import FlatDataParallel, ddp_cleanup, ddp_setup
from .loops import evaluate, make_optimizer, make_scheduler, optimizer_state_for_checkpoint, trainer


def load_config(path, run_mode="Training", model=None):
    """config.yml -> (job, processing, training, model_params) like main.py:145-211."""
    with open(path) as f:
        cfg = yaml.safe_load(f)
    job = dict(cfg["Job"][run_mode])
    job["run_mode"] = run_mode
    name = model or job.get("model", "CGCNN_demo")
    models = cfg["Models"]
    key = name if name in models else next((k for k in models if models[k].get("model") == name), None)
    if key is None:
        raise KeyError("model %r not in config" % name)
    return job, dict(cfg["Processing"]), dict(cfg["Training"]), dict(models[key])


def default_model_factory(name):
    from .. import models
    return getattr(models, name)


def write_results(rows, path):
    """ids, target, prediction — training.py:211-223 (all rows are written; the reference drops the last)."""
    with open(path, "w", newline="") as f:
        wr = csv.writer(f)
        ncol = (rows.shape[1] - 1) // 2 if len(rows) else 1
        wr.writerow(["ids"] + ["target"] * ncol + ["prediction"] * ncol)
        wr.writerows(rows.tolist())


def _enter_dist(rank, world_size):
    """(distributed, owned): join the process group; `owned` = this call created it (and must destroy it)."""
    import torch.distributed as dist
    was = dist.is_available() and dist.is_initialized()
    distributed = ddp_setup(rank, world_size)
    return distributed, distributed and not was


def resolve_seed(seed, sync=None):
    """config.yml's `seed: 0` means "draw one" (main.py:239-241 draws it ONCE, before the ranks are spawned).  Here the
    ranks are already running, so rank 0 draws and everyone else receives its value: split, loader permutation
    and torch.manual_seed must agree across ranks or the shards overlap and the test set leaks into training."""
    import torch.distributed as dist
    seed = int(seed or 0)
    if sync is None:
        sync = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if sync:
        if seed == 0:
            seed = int(np.random.randint(1, 1e6))
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([seed], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0)
        return int(t.item())
    return seed or int(np.random.randint(1, 1e6))


def _loaders(dataset, splits, batch_size, seed, rank, world_size, edge_dtype, rbf, shard_val=False):
    tr, va, te = splits
    r = rank if isinstance(rank, int) else 0
    mk = lambda idx, sh, ws=1, rk=0: DeviceLoader(dataset, idx, batch_size, shuffle=sh, seed=seed, rank=rk,
                                                     world_size=ws, edge_dtype=edge_dtype, rbf=rbf) if len(idx) else None
    # training.py:291-325 — DistributedSampler on train only; val/test on rank 0.  shard_val: every rank validates ITS slice
    # va[r::world] (an exact partition: no padding duplicates as in the training partition, the error is a plain sum over ranks)
    val = mk(np.asarray(va)[r::world_size], False) if (shard_val and world_size > 1) else (mk(va, False) if r == 0 else None)
    return mk(tr, True, world_size, r), val, (mk(te, False) if r == 0 else None)


def graph_replay_wanted(mode, dataset, model, train_loader, rbf, distributed, optimizer_name, loss_name):
    """Whether train_regular runs its training steps as HIP-graph replays (training.GraphedStep).  `mode` = the Training section's
    optional `graph_replay` key: "True" / "False" / "auto" (default).  auto = a single-process job on a HIP-resident dataset with
    the kernel RBF expansion, one of this package's models, an Adam-family optimizer, a fused loss, and batches small enough for
    the host to be the bottleneck (< 4e5 edges per batch: the reference's batch_size 100 is 3e4) — at the bench batch the eager
    step is device-bound and the padded replay 1-2 % slower; models that pool with Set2Set stay eager."""
    mode = str(mode)
    if mode == "False" or train_loader is None:
        return False
    dev = dataset.device
    from ..models._base import GraphModel
    ok = (dev is not None and dev.type == "cuda" and rbf is None and isinstance(model, GraphModel)
          and optimizer_name in ("Adam", "AdamW") and loss_name in ("l1_loss", "mse_loss") and hasattr(dataset, "edge_ptr")
          and len(train_loader.indices) >= train_loader.batch_size)
    if mode == "True":
        if not ok:
            raise ValueError("graph_replay: True needs a HIP-resident dataset with the kernel RBF expansion, a matdeeplearn_amd model, "
                             "Adam / AdamW, l1_loss / mse_loss and at least one full batch in the training split")
        return True
    if not ok or distributed:
        return False
    if getattr(model, "pool", None) == "set2set":
        return False        # (Set2Set runs torch's LSTM through the library's RNN path: not captured by default; "True" tries it)
    idx = np.asarray(train_loader.indices)
    mean_edges = float((np.asarray(dataset.edge_ptr)[idx + 1] - np.asarray(dataset.edge_ptr)[idx]).mean())
    return mean_edges * train_loader.batch_size < 4e5


def train_regular(rank, world_size, dataset, job, training, model_params, splits=None, model_factory=None,
                  rbf=None, edge_dtype=torch.float32, log=print):
    """One training job.  Returns dict(train_error, val_error, test_error, history, model)."""
    model_factory = model_factory or default_model_factory
    distributed, owned = _enter_dist(rank, world_size)
    if dataset.device is not None and dataset.device.type == "cuda" and dataset.device.index is not None:
        torch.cuda.set_device(dataset.device)                                 # the HIP ops launch on the current device
    params = dict(model_params)
    lr = params.get("lr", 0.001) * (world_size if distributed else 1)        # training.py:388-389
    dataset.target_index = training.get("target_index", 0)
    seed = resolve_seed(job.get("seed", 0), sync=distributed)                 # one seed for split, loader, init on all ranks
    if splits is None:
        splits = split_data(len(dataset), training["train_ratio"], training["val_ratio"], training["test_ratio"], seed)
    # validation of a distributed run: sharded over the ranks + one scalar all-reduce per epoch unless the job asks
    # for the reference's rank-0 form (training.py:132-134) with `shard_validation: "False"`; same error either way
    shard_val = distributed and str(training.get("shard_validation", "True")) != "False"
    train_loader, val_loader, test_loader = _loaders(dataset, splits, params.get("batch_size", 100), seed, rank,
                                                     world_size if distributed else 1, edge_dtype, rbf, shard_val=shard_val)
    torch.manual_seed(seed)
    model = model_factory(params["model"])(data=dataset, **params)
    dev = dataset.device if dataset.device is not None else torch.device("cpu")
    model = model.to(dev)
    if job.get("load_model") == "True" and os.path.exists(job.get("model_path", "")):
        model.load_state_dict(torch.load(job["model_path"], map_location=dev)["model_state_dict"])   # training.py:259
    dp = FlatDataParallel(model) if distributed else None
    replay = graph_replay_wanted(training.get("graph_replay", "auto"), dataset, model, train_loader, rbf, distributed,
                                 params.get("optimizer", "AdamW"), training["loss"])
    opt_args = dict(params.get("optimizer_args", {}))
    if replay and not distributed:
        opt_args.setdefault("capturable", True)        # the optimizer step is part of the captured graph (device-tensor lr)
    opt = make_optimizer(model.parameters(), params.get("optimizer", "AdamW"), lr=lr, **opt_args)
    sch = make_scheduler(opt, params.get("scheduler", "ReduceLROnPlateau"), **params.get("scheduler_args", {}))
    graphed = None
    if replay:
        from .graphed import GraphedStep
        graphed = GraphedStep(dataset, model, opt, train_loader.batch_size, compute_dtype=edge_dtype, loss=training["loss"],
                              indices=train_loader.indices, dp=dp)
    t0 = time.time()
    model, history = trainer(rank, world_size, model, opt, sch, training["loss"], train_loader, val_loader,
                             params.get("epochs", 1), training.get("verbosity", 5), dp=dp, log=log, shard_val=shard_val,
                             graphed=graphed)
    out = dict(history=history, model=model, seed=seed, train_time=time.time() - t0)
    is_root = (not distributed) or rank == 0
    if is_root:
        name = job.get("job_name", "my_train_job")
        for tag, idx in zip(("train", "val", "test"), splits):
            if len(idx) == 0:
                out[tag + "_error"] = float("nan")
                continue
            ld = DeviceLoader(dataset, idx, params.get("batch_size", 100), edge_dtype=edge_dtype, rbf=rbf)
            err, rows = evaluate(ld, model, training["loss"], rank=rank, out=True)   # training.py:460-486
            out[tag + "_error"] = float(err)
            out[tag + "_rows"] = rows
            if job.get("write_output") == "True":
                write_results(rows, "%s_%s_outputs.csv" % (name, tag))
        if job.get("save_model") == "True":                                       # training.py:489-510
            torch.save({"model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer_state_for_checkpoint(opt),
                        "scheduler_state_dict": sch.state_dict()}, job.get("model_path", "my_model.pth"))
        log("Train Error: {:.5f}, Val Error: {:.5f}, Test Error: {:.5f}".format(
            out["train_error"], out["val_error"], out["test_error"]))
    if owned:
        ddp_cleanup()
    return out


def predict(dataset, model_name, model_params, state_path, loss="l1_loss", model_factory=None, rbf=None,
            edge_dtype=torch.float32, batch_size=128, out_csv=None):
    """training.py:543-583 with a state_dict checkpoint (pickled PyG full models cannot be honoured)."""
    model_factory = model_factory or default_model_factory
    dev = dataset.device or torch.device("cpu")
    model = model_factory(model_name)(data=dataset, **model_params).to(dev)
    model.load_state_dict(torch.load(state_path, map_location=dev)["model_state_dict"])
    ld = DeviceLoader(dataset, np.arange(len(dataset)), batch_size, edge_dtype=edge_dtype, rbf=rbf)
    err, rows = evaluate(ld, model, loss, out=True)
    if out_csv:
        write_results(rows, out_csv)
    return float(err), rows


def train_repeat(rank, world_size, dataset, job, training, model_params, **kw):
    """training.py:719-843 — `repeat_trials` trainings with fresh seeds; mean/std of the errors."""
    trials = int(job.get("repeat_trials", 5))
    _, owned = _enter_dist(rank, world_size)
    errs = []
    for i in range(trials):
        # fresh, rank-agreed seed per trial; names and flags as training.py:728-744 sets them (no checkpoint per trial)
        j = dict(job, seed=resolve_seed(0), job_name="%s%d" % (job.get("job_name", "repeat"), i), save_model="False",
                 load_model="False", model_path=_trial_path(job.get("model_path", "my_model.pth"), i))
        r = train_regular(rank, world_size, dataset, j, training, model_params, **kw)
        errs.append([r.get("train_error", np.nan), r.get("val_error", np.nan), r.get("test_error", np.nan)])
    if owned:
        ddp_cleanup()
    errs = np.array(errs)
    return dict(errors=errs, mean=errs.mean(0), std=errs.std(0))


def _gather_objects(obj, world_size):
    import torch.distributed as dist
    bucket = [None] * world_size
    dist.all_gather_object(bucket, obj)
    return bucket


def _trial_path(path, *parts):
    """The reference's per-trial checkpoint names: repeat trial i -> "<i>_<model_path>" (training.py:744), ensemble member k ->
    "<k>_<model name>_<model_path>" (training.py:1086-1088); the prefix goes in front of the FILE name, so a model_path with a
    directory keeps it (the reference's plain string concatenation only works for bare file names)."""
    head, tail = os.path.split(path)
    return os.path.join(head, "_".join([str(x) for x in parts] + [tail]))


def _member_name(job, models_params, k):
    """name of ensemble member k as the reference's ensemble_list holds it (training.py:1086)"""
    names = job.get("ensemble_list")
    return names[k] if names and k < len(names) else models_params[k].get("model", "model")


def train_repeat_replicas(rank, world_size, dataset, job, training, model_params, **kw):
    """Repeat mode sharded as REPLICAS (option 2): trial t runs entirely on rank t % world_size — a
    single-GPU training with no gradient exchange — and the error table is gathered at the end.  Same trials and same
    statistics as train_repeat (training.py:719-843, which runs the trials one after the other, each data-parallel over
    all GPUs); zero communication on the data path, so N GPUs finish N trials in the time of one."""
    trials = int(job.get("repeat_trials", 5))
    distributed, owned = _enter_dist(rank, world_size)
    r_id = int(rank) if distributed else 0
    ws = world_size if distributed else 1
    seeds = [resolve_seed(0) for _ in range(trials)]                     # agreed on all ranks
    mine = {}
    for t in range(r_id, trials, ws):
        # names and flags as training.py:728-744 sets them: "<t>_<model_path>", no checkpoint per trial
        j = dict(job, seed=seeds[t], job_name="%s%d" % (job.get("job_name", "repeat"), t), save_model="False", load_model="False",
                 model_path=_trial_path(job.get("model_path", "my_model.pth"), t))
        r = train_regular("cuda" if dataset.device is not None and dataset.device.type == "cuda" else "cpu", 1, dataset,
                          j, training, model_params, **kw)             # world_size 1: a local, independent training
        mine[t] = [r.get("train_error", np.nan), r.get("val_error", np.nan), r.get("test_error", np.nan)]
    allr = {}
    for part in (_gather_objects(mine, ws) if distributed else [mine]):
        allr.update(part)
    if owned:
        ddp_cleanup()
    errs = np.array([allr[t] for t in range(trials)])
    return dict(errors=errs, mean=errs.mean(0), std=errs.std(0), seeds=seeds)


def train_ensemble_replicas(rank, world_size, dataset, job, training, models_params, **kw):
    """Ensemble mode sharded as replicas: model k trains on rank k % world_size on the SAME split (one agreed seed);
    the test predictions are gathered and averaged exactly as training.py:1153 does."""
    distributed, owned = _enter_dist(rank, world_size)
    r_id = int(rank) if distributed else 0
    ws = world_size if distributed else 1
    seed = resolve_seed(job.get("seed", 0))
    splits = split_data(len(dataset), training["train_ratio"], training["val_ratio"], training["test_ratio"], seed)
    mine = {}
    for k in range(r_id, len(models_params), ws):
        r = train_regular("cuda" if dataset.device is not None and dataset.device.type == "cuda" else "cpu", 1, dataset,
                          dict(job, seed=seed, job_name="%s%d" % (job.get("job_name", "ens"), k), load_model="False",
                               model_path=_trial_path(job.get("model_path", "my_model.pth"), k, _member_name(job, models_params, k))),
                          training, models_params[k], splits=splits, **kw)
        mine[k] = (r.get("test_error", np.nan), r.get("test_rows"))
    allr = {}
    for part in (_gather_objects(mine, ws) if distributed else [mine]):
        allr.update(part)
    if owned:
        ddp_cleanup()
    out = dict(model_errors=np.array([allr[k][0] for k in range(len(models_params))]))
    rows = [allr[k][1] for k in range(len(models_params)) if allr[k][1] is not None]
    if rows:
        ncol = (rows[0].shape[1] - 1) // 2
        ens = np.mean([r[:, 1 + ncol:].astype(np.float64) for r in rows], axis=0)
        target = rows[0][:, 1:1 + ncol].astype(np.float64)
        out["ensemble_error"] = float(np.abs(ens - target).mean())
        out["ensemble_prediction"] = ens
    return out


def train_CV(rank, world_size, dataset, job, training, model_params, **kw):
    """training.py:587-715 — k-fold: fold i is the test set, the rest is training (no validation)."""
    _, owned = _enter_dist(rank, world_size)
    seed = resolve_seed(job.get("seed", 0))
    folds = split_data_CV(len(dataset), int(job.get("cv_folds", 5)), seed)
    errs, rows = [], []
    for i in range(len(folds)):
        tr = np.concatenate([f for k, f in enumerate(folds) if k != i])
        r = train_regular(rank, world_size, dataset, dict(job, seed=seed, job_name="%s_fold%d" % (job.get("job_name", "cv"), i)),
                          training, model_params, splits=(tr, np.array([], dtype=np.int64), folds[i]), **kw)
        errs.append(r.get("test_error", np.nan))
        if "test_rows" in r:
            rows.append(r["test_rows"])
    if owned:
        ddp_cleanup()
    return dict(fold_errors=np.array(errs), cv_error=float(np.mean(errs)), rows=np.concatenate(rows) if rows else None)


def train_ensemble(rank, world_size, dataset, job, training, models_params, **kw):
    """training.py:1069-1196 — one training per listed model on the SAME split; ensemble = mean prediction."""
    _, owned = _enter_dist(rank, world_size)
    seed = resolve_seed(job.get("seed", 0))
    splits = split_data(len(dataset), training["train_ratio"], training["val_ratio"], training["test_ratio"], seed)
    per_model, preds, target = [], [], None
    for k, mp in enumerate(models_params):
        r = train_regular(rank, world_size, dataset,
                          dict(job, seed=seed, job_name="%s%d" % (job.get("job_name", "ens"), k), load_model="False",
                               model_path=_trial_path(job.get("model_path", "my_model.pth"), k, _member_name(job, models_params, k))),
                          training, mp, splits=splits, **kw)
        per_model.append(r.get("test_error", np.nan))
        if "test_rows" in r:
            rows = r["test_rows"]
            ncol = (rows.shape[1] - 1) // 2
            preds.append(rows[:, 1 + ncol:].astype(np.float64))
            target = rows[:, 1:1 + ncol].astype(np.float64)
    if owned:
        ddp_cleanup()
    out = dict(model_errors=np.array(per_model))
    if preds:
        ens = np.mean(preds, axis=0)                                              # training.py:1153
        out["ensemble_error"] = float(np.abs(ens - target).mean())
        out["ensemble_prediction"] = ens
    return out
