"""Data path of the hot path: the reference's on-disk robust04 pickle
format -> pinned host tensors -> asynchronous copies to the GPU.  Same entry-point names as the
reference's dataloader/__init__.py for the in-scope loaders."""
from .rank_data import RankData, attncut_dataloader as at_dataloader, choopy_dataloader as cp_dataloader
from .synth import write_synthetic_robust04

__all__ = ["RankData", "at_dataloader", "cp_dataloader", "write_synthetic_robust04"]
