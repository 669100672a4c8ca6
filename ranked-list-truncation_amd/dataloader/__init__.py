"""Data path of the hot path: the reference's on-disk robust04 pickle
format -> pinned host tensors -> asynchronous copies to the GPU.  Same entry-point names as the
reference's dataloader/__init__.py for the in-scope loaders."""
from .rank_data import (BatchLoader, RankData, attncut_dataloader as at_dataloader, choopy_dataloader as cp_dataloader,
                        mtcut_dataloader as mc_dataloader, shared_seed)
from .synth import write_synthetic_robust04

__all__ = ["BatchLoader", "RankData", "at_dataloader", "cp_dataloader", "mc_dataloader", "shared_seed", "write_synthetic_robust04"]
