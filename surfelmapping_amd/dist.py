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


class RigMapper:
    """BASELINE configs[4]: a rig of G cameras, one per rank (= one per GPU), consolidated into a SINGLE GlobalModel.

    Frame loop: every rank fuses its own camera into its own slice with the ordinary per-frame path -- no collective.

    `consolidate()` defines "single GlobalModel" with the reference's own operations:
      1. the slices concatenated in rank order (GlobalModel::concatenate's order for G append lists), and
      2. that union cleaned against EVERY camera's latest view, one view after the other in rank order, with the
         reference's tool for testing the model against a view it was not fused from: SurfelMapping::cleanPoints
         (src/SurfelMapping.cpp:496-532 -- conflict.vert with maxDepth = far - 15, threshold 0.1, isClean = 1 -- then
         updateConflict + backMapping).  This is the "conflict pass of every camera's depth against the union" of
         SURVEY.md 8e.
    The conflict test is per surfel and per view, so every rank cleans ITS slice against all G views (after an all-gather
    of the G latest depth / semantic images and poses: 3 bytes per pixel per camera) and the cleaned slices are gathered.
    Two rules of the reference couple the slices and are carried explicitly: surfel id 0 never conflicts
    (conflict.geom:15) -- only the rank holding the first surfel of the union applies the exemption -- and at most W*H
    conflicts take effect per view, in the surfel order of the union (src/GlobalModel.cpp:54-57) -- between a view's
    conflict test and its cull the ranks exchange their conflict counts, and every slice lets its first
    (W*H - conflicts of the slices before it) conflicts take effect: exactly the single-model rule.

    `backend` is the rank's mapper (capi.SurfelMap, or an oracle-backed stand-in in CPU tests) with process_frame, counts,
    clean_points_slice(depth, sem, pose, exempt_first, cap_hook), download_model; `comm` has rank, world, allgather(obj),
    allreduce_sum(array) (sharded.ThreadComm / sharded.TorchComm)."""

    def __init__(self, backend, comm, n_pixels: int, conflict_cap: bool = True):
        self.be, self.comm = backend, comm
        self.P = n_pixels
        self.conflict_cap = conflict_cap
        self.last = None

    def process_frame(self, rgb, depth, sem, pose):
        rc = self.be.process_frame(rgb, depth, sem, pose)
        self.last = (np.ascontiguousarray(depth, np.uint16), np.ascontiguousarray(sem, np.uint8), np.ascontiguousarray(pose, np.float32))
        return rc

    def consolidate(self, device_index=None, sm_global=None):
        """-> (single GlobalModel as AoS float32 [n][12], identical on every rank; per-rank counts; conflicts per view).
        With `device_index` and `sm_global` (a second capi.SurfelMap on the same GPU) the cleaned slices are gathered over
        RCCL on device buffers and appended into `sm_global` without a host round trip; the first return value is then the
        surfel count of that GlobalModel."""
        views = self.comm.allgather(self.last)
        per_view = []
        for v, view in enumerate(views):
            if view is None:
                continue
            counts = [int(c) for c in self.comm.allgather(int(self.be.counts()["count"]))]
            first = next((r for r, c in enumerate(counts) if c > 0), None)
            seen = []

            def share(local):
                # between the conflict test and the cull: this slice's share of the union's W*H conflict records -- the buffer
                # fills in the surfel order of the union, i.e. slice after slice (src/GlobalModel.cpp:54-57)
                allc = [int(c) for c in self.comm.allgather(int(local))]
                seen.append(sum(allc))
                if not self.conflict_cap:
                    return 0xFFFFFFFF
                return max(0, min(int(local), self.P - sum(allc[:self.comm.rank])))
            self.be.clean_points_slice(*view, exempt_first=(first == self.comm.rank), cap_hook=share)
            per_view.append(min(seen[0], self.P) if self.conflict_cap else seen[0])
        if device_index is not None and sm_global is not None:
            gathered, counts = gather_model_device(self.be, device_index)
            return build_global_model(sm_global, gathered, counts), counts, per_view
        slices = self.comm.allgather(np.ascontiguousarray(self.be.download_model(), np.float32))
        counts = [int(x.shape[0]) for x in slices]
        model = np.concatenate(slices, axis=0) if slices else np.zeros((0, 12), np.float32)
        return model, counts, per_view


    # -- the same consolidation inside the HIP core (sm_rig_consolidate): views, sizes, conflict totals and slices cross the
    #    ranks through the collective installed in the core (RCCL on its stream, or a callback); nothing is staged through Python
    def enable_native(self, collective=None, rccl_id: bytes | None = None):
        """collective: None (world 1), "rccl" with the id of capi.rccl_unique_id() handed to all ranks, or a callable
        (send_ptr, recv_ptr, count_u64, op) -> 0 (sharded.ThreadCollective for ranks that are threads of one process)"""
        self.be.rig_configure(self.comm.rank, self.comm.world)
        if collective == "rccl":
            self.be.shard_rccl_init(rccl_id)
        elif collective is not None:
            self.be.shard_set_collective(collective)
        self._native = True

    def consolidate_step_native(self, sm_global):
        """the incremental single GlobalModel, one step (every K frames; collective): -> (new surfels exchanged, surfels in `sm_global`)"""
        assert getattr(self, "_native", False), "enable_native() first"
        depth, sem, pose = self.last
        return self.be.rig_consolidate_step(depth, sem, pose, sm_global)

    def consolidate_native(self, sm_global):
        """-> (surfels of the single GlobalModel, now in `sm_global`; conflicts per view); collective"""
        assert getattr(self, "_native", False), "enable_native() first"
        depth, sem, pose = self.last
        return self.be.rig_consolidate(depth, sem, pose, sm_global)
