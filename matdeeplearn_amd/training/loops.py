"""Train / evaluate / epoch driver — the callers of the hot path.

Mirrors /root/reference/matdeeplearn/training/training.py: `train` (:34-54), `evaluate` (:58-92),
`trainer` (:96-207) and the optimizer/scheduler lookup by name (:429-436):
  * loss metric = sum(batch-mean loss x batch size) / sum(batch size)   (:45,51-53,67,84-86)
  * the LR scheduler is stepped on the TRAINING error (:193)
  * the best-validation weights are kept (:144-166); epoch time printed every `verbosity` epochs
The loops are model-agnostic (any nn.Module whose forward takes a batch) and never force a
host<->device sync inside an epoch: running sums stay on the device.
"""
import contextlib
import copy
import time

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops


def make_optimizer(params, name="AdamW", lr=0.002, **optimizer_args):
    """getattr(torch.optim, name)(params, lr, **optimizer_args) — training.py:429-432.  `capturable=True` (Adam family)
    keeps the step counter on the device so that the step can be part of a captured HIP graph (training.GraphedStep)."""
    kw = dict(optimizer_args)
    plist = list(params)
    if name in ("Adam", "AdamW") and plist and plist[0].is_cuda and "fused" not in kw and "foreach" not in kw:
        kw["fused"] = True   # one kernel for all parameters instead of one per tensor
    if kw.get("capturable") and plist and plist[0].is_cuda and not torch.is_tensor(lr):
        # a python-float lr is a kernel ARGUMENT of the fused step: captured into a HIP graph it is frozen, and a scheduler
        # (the reference steps ReduceLROnPlateau every epoch, training.py:193) would be ignored by every replay.  As a
        # device tensor it is read by the kernel at run time; torch's schedulers update tensor lrs in place.
        lr = torch.tensor(float(lr), dtype=torch.float32, device=plist[0].device)
    return getattr(torch.optim, name)(plist, lr=lr, **kw)


def optimizer_state_for_checkpoint(optimizer):
    """optimizer.state_dict() in the reference's checkpoint format (training.py:489-510): a capturable optimizer keeps its
    learning rate in a device tensor (make_optimizer), which a non-capturable / foreach optimizer refuses to load ("lr as a
    Tensor is not supported") — the checkpoint holds python floats."""
    sd = optimizer.state_dict()
    sd = dict(sd, param_groups=[dict(g, lr=float(g["lr"]) if torch.is_tensor(g.get("lr")) else g.get("lr")) for g in sd["param_groups"]])
    return sd


def load_optimizer_state(optimizer, state_dict):
    """optimizer.load_state_dict for checkpoints written by optimizer_state_for_checkpoint: an optimizer built with a
    device-tensor learning rate (capturable) keeps ITS tensor (captured graphs hold its address) and receives the value."""
    lr_tensors = [g["lr"] if torch.is_tensor(g.get("lr")) else None for g in optimizer.param_groups]
    optimizer.load_state_dict(state_dict)
    for g, t in zip(optimizer.param_groups, lr_tensors):
        if t is not None:
            t.fill_(float(g["lr"]))
            g["lr"] = t
    return optimizer


def make_scheduler(optimizer, name="ReduceLROnPlateau", **scheduler_args):
    return getattr(torch.optim.lr_scheduler, name)(optimizer, **scheduler_args)


def output_device(data):
    x = getattr(data, "x", None)
    return x.device if torch.is_tensor(x) else torch.device("cpu")


def _step_arena(device):
    """ops.zero_arena on a HIP device (one zero fill per step for the kernels' small accumulators), a no-op elsewhere."""
    if device.type == "cuda":
        from .. import ops
        return ops.zero_arena(device)
    return contextlib.nullcontext()


def train(model, optimizer, loader, loss_method, rank=None, dp=None, stats=None, graphed=None):
    """One pass over `loader` in train mode.  Returns the sample-weighted mean loss (device scalar).
    graphed: a training.GraphedStep bound to this model / optimizer / loss and the loader's dataset — every full batch then runs as
    ONE replay of the captured step (batch assembly -> forward -> loss -> backward -> optimizer; at the reference's batch size,
    config.yml:136, 0.41 ms instead of 1.36 ms of launch-bound eager work), the ragged last batch and batches past the static
    capacity eagerly with the same arithmetic."""
    model.train()
    loss_all, count = 0, 0
    edges = 0
    if graphed is not None:
        for ids in loader.batch_ids():
            e, _ = graphed.step(ids)
            loss_all = loss_all + graphed.loss_value * len(ids)          # (stream-ordered read of the step's loss: no sync)
            count += len(ids)
            edges += int(e)
        if stats is not None:
            stats["edges"] = stats.get("edges", 0) + edges
            stats["graphs"] = stats.get("graphs", 0) + count
            stats["replays"] = graphed.replays
            stats["eager_steps"] = graphed.eager_steps
        return loss_all / max(count, 1)
    for data in loader:
        data = data.to(rank)
        if dp is not None:
            dp.zero_grad()
        else:
            optimizer.zero_grad()
        with _step_arena(output_device(data)):
            output = model(data)
            loss = ops.loss(loss_method, output, data.y)
            ops.backward(loss)
        loss_all = loss_all + loss.detach() * output.size(0)
        if dp is not None:
            dp.reduce_grads()
        optimizer.step()
        count += output.size(0)
        edges += int(getattr(data, "num_edges", 0))
    if stats is not None:
        stats["edges"] = stats.get("edges", 0) + edges
        stats["graphs"] = stats.get("graphs", 0) + count
    return loss_all / max(count, 1)


def evaluate_graphed(loader, graphed):
    """evaluate(loader, model, loss) for the trainer's validation pass with the forward-only step replayed (training.GraphedStep.
    eval_loss): the sample-weighted mean loss as a device scalar.  No rows of predictions — the final evaluation of a job, which
    writes them, stays eager."""
    loss_all, count = 0, 0
    for ids in loader.batch_ids():
        loss_all = loss_all + graphed.eval_loss(ids) * len(ids)
        count += len(ids)
    return loss_all / max(count, 1)


def evaluate(loader, model, loss_method, rank=None, out=False, sharded=False):
    """Eval-mode pass; with out=True also returns rows (id, target, prediction) like training.py:68-90.
    sharded=True (every rank of the process group calls it on ITS shard of the split, loader may be None for an empty shard):
    the sample-weighted loss sum and the sample count are added up over the ranks with ONE two-element all-reduce, so every
    rank returns the error of the whole split — the same number the reference's rank-0 evaluation (training.py:132-134) gives,
    in 1 / world_size of the time.  Returns None when the split is empty on EVERY rank: the trainer then keeps the
    latest weights, as the reference does without a validation loader (training.py:130-170)."""
    model.eval()
    loss_all, count = 0, 0
    ids, preds, targets = [], [], []
    with torch.no_grad():
        for data in (loader if loader is not None else ()):
            data = data.to(rank)
            output = model(data)
            loss = getattr(F, loss_method)(output, data.y)
            loss_all = loss_all + loss * output.size(0)
            if out:
                ids += list(data.structure_id)
                preds.append(output.detach().cpu().numpy())
                targets.append(data.y.detach().cpu().numpy())
            count += output.size(0)
    if sharded:
        import torch.distributed as dist
        dev = loss_all.device if torch.is_tensor(loss_all) else next(model.parameters()).device
        t = torch.zeros(2, dtype=torch.float64, device=dev)
        t[0] = loss_all if not torch.is_tensor(loss_all) else loss_all.double()
        t[1] = float(count)
        if dist.get_backend() == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if float(t[1]) == 0.0:
            return None                       # the whole split is empty (val_ratio = 0): "no validation", not an error of 0
        return (t[0] / t[1]).float()
    loss_all = loss_all / max(count, 1)
    if out:
        return loss_all, np.column_stack((np.array(ids, dtype=object), np.concatenate(targets), np.concatenate(preds)))
    return loss_all


def trainer(rank, world_size, model, optimizer, scheduler, loss, train_loader, val_loader, epochs, verbosity=5,
            dp=None, log=print, shard_val=False, graphed=None):
    """Epoch driver (training.py:96-207).  Returns (model with the best-validation weights loaded,
    history list of dicts).  shard_val (distributed runs): `val_loader` is this rank's SHARD of the validation split (None
    for an empty one) and the error comes from evaluate(sharded=True) — every rank then knows it, so every rank keeps the same
    best-validation weights (with the reference's rank-0 form only rank 0 does).  graphed: see train()."""
    import torch.distributed as dist

    distributed = dp is not None and dp.world_size > 1
    best_val, best_state = 1e10, None
    history = []
    t_mark = time.time()
    for epoch in range(1, epochs + 1):
        lr = optimizer.param_groups[0]["lr"]
        lr = float(lr) if torch.is_tensor(lr) else lr           # (a capturable optimizer keeps it in a device tensor it updates in place)
        if hasattr(train_loader, "set_epoch"):
            train_loader.set_epoch(epoch)                       # training.py:120
        stats = {}
        train_error = train(model, optimizer, train_loader, loss, rank=rank, dp=dp, stats=stats, graphed=graphed)
        if distributed:                                         # training.py:124-125, one tiny collective
            dist.all_reduce(train_error, op=dist.ReduceOp.SUM)
            train_error = train_error / world_size
        val_error = None
        if distributed and shard_val:
            for m in model.modules():                           # (host-side BatchNorm step counters -> their buffers)
                if hasattr(m, "_sync_counter"):
                    m._sync_counter()
            dp.broadcast_buffers()                              # every rank validates rank 0's model, as the reference does
            val_error = evaluate(val_loader, model, loss, rank=rank, sharded=True)
            val_error = None if val_error is None else float(val_error)
        elif val_loader is not None and (not distributed or dist.get_rank() == 0):
            if graphed is not None and not distributed and hasattr(val_loader, "batch_ids"):
                val_error = float(evaluate_graphed(val_loader, graphed))
            else:
                val_error = float(evaluate(val_loader, model, loss, rank=rank))
        if val_error is not None:
            if val_error < best_val:                            # training.py:144-166 (NaN never passes)
                best_val = val_error
                best_state = copy.deepcopy(model.state_dict())
        train_error = float(train_error)
        scheduler.step(train_error)                              # training.py:193 — on the TRAIN error
        now = time.time()
        history.append(dict(epoch=epoch, lr=lr, train=train_error, val=val_error, time=now - t_mark, **stats))
        if verbosity and epoch % verbosity == 0 and (not distributed or dist.get_rank() == 0):
            log("Epoch: {:04d}, Learning Rate: {:.6f}, Training Error: {:.5f}, Val Error: {}, Time per epoch (s): {:.5f}"
                .format(epoch, lr, train_error, "n/a" if val_error is None else "%.5f" % val_error, now - t_mark))
        t_mark = now
    if best_state is not None:
        model.load_state_dict(best_state)
    return model, history
