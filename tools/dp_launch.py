#!/usr/bin/env python
"""Data-parallel launcher for the reference's UNCHANGED training script: one process per GPU, RCCL over xGMI.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/dp_launch.py --script /path/to/SemanticLiDARUnc/src/train_semantics.py --mode train --cfg_path cfg.yaml

The reference has no launcher and no DDP (src/train_semantics.py:47-337 builds dataset, model, AdamW and Trainer in one process;
src/models/trainer.py:783-786 is a plain `loss.backward(); optimizer.step()`).  This wrapper keeps both files untouched and installs
three seams per rank before calling the script's `main(args)`:

  1. `DataLoader` as seen by the script (train_semantics.py:4): a `shuffle=True` loader gets a `ShardedSampler` (disjoint shards of
     one seeded permutation, equal sizes); the `shuffle=False` validation loader is left whole -- every rank evaluates the full
     validation set on identical weights, so all ranks log identical metrics and no metric exchange is needed.  With
     `--gpu-projection` the loader is also the device-projecting one of dataset/gpu_pipeline.py.
  2. `Trainer.__init__` (models/trainer.py:167): after the reference's constructor has moved the model to the rank's GPU,
     parameters and buffers are broadcast from rank 0 and ONE flat gradient all-reduce is attached to the optimizer as a pre-step
     hook (`distributed.FlatGradAllReduce`): the Trainer's own step becomes a synchronous data-parallel step.
  3. `Trainer.test_one_epoch` (models/trainer.py:1072): BatchNorm running statistics are averaged over the ranks first
     (`distributed.average_buffers`), so the evaluated -- and then checkpointed -- model is the same everywhere.

Ranks other than 0 run with logging and visualisation off (they would write the same TensorBoard files / checkpoints).
Semantics that differ from one process are those of SURVEY 8(e): BatchNorm batch statistics and the Lovasz loss are per shard.
"""
from __future__ import annotations

import argparse
import importlib
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _str2bool(v: str) -> bool:
    return str(v).lower() not in ("", "0", "false", "no", "off")


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--script", required=True, help="path to the reference's src/train_semantics.py (or any script with the same main(args))")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL with a GPU, gloo without)")
    ap.add_argument("--seed", type=int, default=0, help="seed of the sharded permutation")
    ap.add_argument("--gpu-projection", action="store_true",
                    help="SemanticKitti loaders: workers only read files, the main process decodes / projects whole batches on the GPU "
                         "(semanticlidarunc_amd/dataset/gpu_pipeline.py); other datasets keep the stock loader")
    ap.add_argument("--trainer-module", default="models.trainer", help="module that defines Trainer (imported with the script's directory on sys.path)")
    # the script's own four flags (train_semantics.py:343-364); its `type=bool` flags treat every non-empty string as True
    ap.add_argument("--visualization", type=_str2bool, default=False)
    ap.add_argument("--with_logging", type=_str2bool, default=True)
    ap.add_argument("--cfg_path", type=str, required=True)
    ap.add_argument("--mode", type=str, default="train")
    return ap.parse_args(argv)


def sharded_loader_class(rank: int, world: int, seed: int, base=None):
    """A DataLoader subclass for the script's namespace: same constructor, a shuffling loader becomes a sharded one.
    `base`: the loader class to derive from (default torch's; --gpu-projection passes the device-projecting loader)."""
    import torch.utils.data as tud
    from semanticlidarunc_amd.distributed import ShardedSampler
    base = base or tud.DataLoader

    class ShardedDataLoader(base):
        def __init__(self, dataset=None, *args, **kwargs):
            if dataset is None:
                dataset = kwargs.pop("dataset")
            if kwargs.get("shuffle") and kwargs.get("sampler") is None and world > 1:
                kwargs["shuffle"] = False
                kwargs["sampler"] = ShardedSampler(len(dataset), rank, world, seed=seed, shuffle=True, drop_last=True)
            super().__init__(dataset, *args, **kwargs)

        def set_epoch(self, epoch: int) -> None:
            if isinstance(self.sampler, ShardedSampler):
                self.sampler.set_epoch(epoch)

    return ShardedDataLoader


def install_trainer_hooks(trainer_cls) -> None:
    """Seams 2 and 3 on a Trainer class with the reference's constructor (model, optimizer, cfg, ...)."""
    from semanticlidarunc_amd.distributed import FlatGradAllReduce, average_buffers, broadcast_parameters
    if getattr(trainer_cls, "_slu_dp_hooks", False):
        return
    orig_init, orig_test, orig_epoch = trainer_cls.__init__, trainer_cls.test_one_epoch, trainer_cls.train_one_epoch

    def __init__(self, model, optimizer, *args, **kwargs):
        orig_init(self, model, optimizer, *args, **kwargs)
        broadcast_parameters(self.model)
        self._slu_grad_reducer = FlatGradAllReduce(self.model.parameters())
        self._slu_grad_hook = self._slu_grad_reducer.attach_to_optimizer(self.optimizer)

    def train_one_epoch(self, loader, epoch, *args, **kwargs):
        if hasattr(loader, "set_epoch"):
            loader.set_epoch(int(epoch))                     # a new permutation per epoch, the same on every rank
        return orig_epoch(self, loader, epoch, *args, **kwargs)

    def test_one_epoch(self, *args, **kwargs):
        average_buffers(self.model)
        return orig_test(self, *args, **kwargs)

    trainer_cls.__init__, trainer_cls.train_one_epoch, trainer_cls.test_one_epoch = __init__, train_one_epoch, test_one_epoch
    trainer_cls._slu_dp_hooks = True


def main(argv=None):
    a = parse(argv)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from semanticlidarunc_amd.distributed import init_from_env
    backend = a.backend or ("nccl" if torch.cuda.is_available() else "gloo")
    rank, local_rank, world = init_from_env(backend)          # sets the current device to local_rank's GPU
    script = os.path.abspath(a.script)
    src_dir = os.path.dirname(script)
    # this repo's mirrors ahead of the reference's src/ so the hot path resolves to the HIP classes
    for p in (src_dir, os.path.join(ROOT, "semanticlidarunc_amd"), ROOT):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[0:0] = [ROOT, os.path.join(ROOT, "semanticlidarunc_amd"), src_dir]
    spec = importlib.util.spec_from_file_location("_slu_dp_train_script", script)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                                # __name__ != "__main__": the script's argparse block does not run
    base = None
    if a.gpu_projection:
        # DataLoader workers only read the .bin / .label files; decode, id_map, rotation, spherical projection, flip,
        # range and normals run as HIP kernels on whole batches in this (the rank's main) process
        from semanticlidarunc_amd.dataset import gpu_pipeline
        # duck-typed: in drop-in mode the script's dataset class comes from the module named `dataset.dataloader_semantic_KITTI`
        # (this repo's mirror under the reference's import path), a different module object than the package-qualified one
        base = gpu_pipeline.projecting_loader_class(lambda ds: ds.projector(torch.device("cuda", local_rank) if torch.cuda.is_available() else "cpu"),
                                                    lambda ds: getattr(ds, "_raw", None) if hasattr(ds, "projector") else None)
    mod.DataLoader = sharded_loader_class(rank, world, a.seed, base)  # seam 1
    if a.mode == "train":
        install_trainer_hooks(importlib.import_module(a.trainer_module).Trainer)     # seams 2, 3
    args = argparse.Namespace(visualization=a.visualization and rank == 0, with_logging=a.with_logging and rank == 0,
                              cfg_path=a.cfg_path, mode=a.mode)
    try:
        mod.main(args)
    except BaseException as e:
        # A rank that failed must NOT issue another collective: its peers are inside the gradient all-reduce (or a broadcast /
        # buffer average), a barrier here would pair with a different-sized collective and leave the job hung until the watchdog
        # fires.  Report and leave with a non-zero code; torch.distributed.run then tears the other ranks down.
        import traceback
        traceback.print_exc()
        sys.stderr.write(f"dp_launch: rank {rank} failed ({type(e).__name__}); exiting without joining further collectives\n")
        sys.stderr.flush()
        os._exit(1 if not isinstance(e, SystemExit) or e.code in (None, 0) else (e.code if isinstance(e.code, int) else 1))
    if dist.is_initialized():      # success path only
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
