"""Multi-GPU mode of BASELINE configs[4]: one camera stream per GPU (one process per GPU,
`torch.distributed`), each fused independently by the HIP core, then an all-gather of the
per-GPU surfel slices into a single GlobalModel.

The frame loop has no data-path collective (every camera's conflict / splat / associate /
append touches only its own slice); the only exchange step is the model all-gather, which goes
over RCCL ("nccl" backend, device buffers aliased as torch tensors: plumbing only) or, for
CPU-only runs and tests, over gloo with host arrays.  Slices are concatenated in rank order, so
the global model is deterministic: [camera 0 surfels | camera 1 surfels | ...].
"""
from __future__ import annotations

import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_layout(counts):
    """Global id range [base, base+n) of every rank's slice for per-rank surfel counts."""
    counts = [int(c) for c in counts]
    bases = [0]
    for c in counts[:-1]:
        bases.append(bases[-1] + c)
    return bases, sum(counts)


def gather_model_host(local_model: np.ndarray, group=None) -> tuple[np.ndarray, list[int]]:
    """All-gather AoS slices (n_r, 12) float32 held in host memory (gloo path)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([local_model.shape[0]], dtype=torch.int64)
    ns = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    counts = [int(x[0]) for x in ns]
    pad = max(max(counts), 1)
    buf = torch.zeros((pad, 12), dtype=torch.float32)
    if local_model.shape[0]:
        buf[:local_model.shape[0]] = torch.from_numpy(np.ascontiguousarray(local_model, np.float32))
    outs = [torch.zeros((pad, 12), dtype=torch.float32) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    parts = [outs[r][:counts[r]].numpy() for r in range(world)]
    return (np.concatenate(parts, axis=0) if parts else np.zeros((0, 12), np.float32)), counts


class _DevArray:
    """__cuda_array_interface__ view of memory owned by the HIP core (no copy)."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def gather_model_device(sm, device_index: int, group=None):
    """All-gather the model slices over RCCL without leaving HBM.  Returns (tensor[world, pad, 12]
    on the GPU, counts).  `sm` is a capi.SurfelMap living on cuda:`device_index`."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = torch.device("cuda", device_index)
    ptr, n = sm.export_model_device()              # synchronises the core's stream
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    cnts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    counts = [int(c) for c in cnts.tolist()]
    pad = max(max(counts), 1)
    send = torch.zeros((pad, 12), dtype=torch.float32, device=dev)
    if n:
        send[:n] = torch.as_tensor(_DevArray(ptr, (n, 12)), device=dev)
    out = torch.empty((world, pad, 12), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, send, group=group)
    torch.cuda.synchronize(dev)
    return out, counts


def build_global_model(sm_global, gathered, counts):
    """Append every rank's slice, in rank order, to `sm_global` (GlobalModel::concatenate's
    device copy, src/GlobalModel.cpp:624-629).  `gathered` is the tensor of gather_model_device."""
    for r, n in enumerate(counts):
        if n:
            sm_global.append_model_device(gathered[r].data_ptr(), n)
    return sm_global.counts()["count"]
