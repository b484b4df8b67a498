"""Training harness around the hot path: the caller side of the drop-in boundary
(/root/reference/matdeeplearn/training/training.py:34-207,227-270) restated for device-resident
batches, plus the data-parallel engine (one flat gradient buffer, one RCCL all-reduce per step)."""
from .loops import (train, evaluate, trainer, make_optimizer, make_scheduler, optimizer_state_for_checkpoint,  # noqa: F401
                    load_optimizer_state)
from .dp import ddp_setup, ddp_cleanup, FlatDataParallel  # noqa: F401
from .graphed import GraphedStep  # noqa: F401
from .driver import (load_config, train_regular, train_repeat, train_CV, train_ensemble, predict,  # noqa: F401
                     write_results, train_repeat_replicas, train_ensemble_replicas, resolve_seed)
