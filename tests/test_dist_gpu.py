"""GPU check of the RCCL plumbing used by bench.py --gpus N (N>1 needs an 8-GPU node and is run
by the driver): with world_size 1 on one GPU, torch.distributed('nccl') all-gathers the model
through device buffers owned by the HIP core, and the gathered slices are appended into a second
context = the single GlobalModel.  Run in a subprocess so that the import order (torch first,
as bench.py does) is controlled."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, "__ROOT__"); sys.path.insert(0, os.path.join("__ROOT__", "tests"))
    ORDER = sys.argv[1]
    if ORDER == "torch_first":
        import torch, torch.distributed as dist
    from surfelmapping_amd import capi, synth, dist as smd
    capi.load()
    if ORDER != "torch_first":
        import torch, torch.distributed as dist
    import numpy as np
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (20000 + os.getpid() % 20000), rank=0, world_size=1)
    torch.cuda.set_device(0)
    cam = dict(width=320, height=120, fx=180.0, fy=180.0, cx=159.5, cy=59.5)
    seq = synth.make_sequence(cam, synth.kitti_trajectory(4), seed=3)
    over = dict(preprocess=0, stereo_border=20.0, max_sqrt_vertices=600)
    sm = capi.SurfelMap(capi.make_config(**cam, **over))
    for fr in seq:
        sm.process_frame(*fr)
    local = sm.download_model()
    gathered, counts = smd.gather_model_device(sm, 0)
    glob = capi.SurfelMap(capi.make_config(**cam, **over))
    total = smd.build_global_model(glob, gathered, counts)
    g = glob.download_model()
    assert total == local.shape[0] == counts[0] > 0, (total, local.shape, counts)
    assert np.array_equal(g.view(np.uint32), local.view(np.uint32))
    # ONE stream sharded, RCCL bound by the core (world 1): the key map / fused mask / conflict masks all-reduced on its stream
    from surfelmapping_amd import sharded
    import oracle_lib as ol
    sm2 = capi.SurfelMap(capi.make_config(**cam, **over))
    mp = sharded.StreamShard(sm2, 0, 1, "rccl", capi.rccl_unique_id())
    assert sm2.shard_rccl_nranks() == 1
    o = ol.Oracle(ol.make_config(**cam, **over))
    for fr in seq:
        mp.process_frame(*fr); o.process_frame(*fr)
    gm, om = mp.export_dense(), o.download_model()
    sm2.shard_rccl_finalize()
    assert gm.shape == om.shape and np.array_equal(gm.view(np.uint32), om.view(np.uint32))
    dist.destroy_process_group()
    print("DIST_GPU_OK", ORDER, total)
""").replace("__ROOT__", ROOT)


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["torch_first", "lib_first"])   # either order: capi.load() pre-loads torch's bundled HIP runtime
def test_rccl_gather_world1(order, tmp_path):
    f = tmp_path / "w1.py"
    f.write_text(SCRIPT)
    r = subprocess.run([sys.executable, str(f), order], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


CONFLICT = textwrap.dedent("""
    import os, sys, ctypes
    sys.path.insert(0, "__ROOT__")
    os.environ["SM_NO_TORCH_HIP_PRELOAD"] = "1"          # what a C++ host that loads ROCm's runtime first would see
    from surfelmapping_amd import capi
    L = capi.load()                                       # binds /opt/rocm's libamdhip64.so.7
    cfg = capi.make_config(64, 48, 50.0, 50.0, 31.5, 23.5, max_sqrt_vertices=64)
    h1 = L.sm_create(ctypes.byref(cfg))
    assert h1, L.sm_last_error()
    import importlib.util
    tl = os.path.join(list(importlib.util.find_spec("torch").submodule_search_locations)[0], "lib", "libamdhip64.so")
    ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)              # torch's request "libamdhip64.so" does not match the soname: a 2nd runtime
    h2 = L.sm_create(ctypes.byref(cfg))
    msg = L.sm_last_error().decode()
    assert not h2 and "two HIP runtimes" in msg and "torch" in msg, msg
    print("CONFLICT_DETECTED")
""").replace("__ROOT__", ROOT)


@pytest.mark.gpu
def test_second_hip_runtime_is_refused_not_undefined(tmp_path):
    """The cause of round 1's lib_first failure: two libamdhip64 copies in one process.  The core names both and refuses."""
    f = tmp_path / "c.py"
    f.write_text(CONFLICT)
    r = subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CONFLICT_DETECTED" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


HOLDER = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, "__ROOT__")
    from surfelmapping_amd import capi
    sm = capi.SurfelMap(capi.make_config(64, 48, 50.0, 50.0, 31.5, 23.5, max_sqrt_vertices=64))
    print("HOLDING", flush=True)
    sys.stdin.readline()
""").replace("__ROOT__", ROOT)


@pytest.mark.gpu
def test_other_process_on_the_gpu_is_detected(tmp_path):
    """The in-place compaction assumes its grid is resident; another PROCESS on the GPU breaks that silently.  The core
    counts the processes with queues on its GPU in the KFD driver's tables and switches to the ticket-ordered kernel."""
    import time
    sys.path.insert(0, ROOT)
    from surfelmapping_amd import capi
    sm = capi.SurfelMap(capi.make_config(64, 48, 50.0, 50.0, 31.5, 23.5, max_sqrt_vertices=64))
    alone = sm.gpu_process_count()
    if alone < 0:
        pytest.skip("/sys/class/kfd is not readable here")
    assert alone >= 1
    f = tmp_path / "holder.py"
    f.write_text(HOLDER)
    p = subprocess.Popen([sys.executable, str(f)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    try:
        assert "HOLDING" in p.stdout.readline()
        during = sm.gpu_process_count()
        assert during >= alone + 1          # (>=: a monitoring tool may open the device at any time)
    finally:
        try:
            p.communicate(input="\n", timeout=60)
        except subprocess.TimeoutExpired:
            p.kill()
    time.sleep(0.2)
    assert sm.gpu_process_count() <= during - 1


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_starts_two_ranks_and_they_agree_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` as the driver calls it, rehearsed on the one GPU this box has: the parent starts two fresh
    ranks, they rendezvous (gloo here -- RCCL refuses two ranks on one device -- with the core's collectives staged through
    the host, sharded.GlooCollective), run the rig leg, both consolidations and the sharded leg as two PROCESSES, and rank 0
    prints the one line.  What must hold whatever the transport: n_gpus = 2, the sharded stream's counters identical on both
    ranks and equal to the plain single-GPU run of the same frames, a single GlobalModel out of both consolidations."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "6", "--warmup", "3",
                        "--workers", "1", "--launch-timeout", "600"], cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6
    sh = d["sharded_leg"]
    assert sh["counters_identical_on_all_ranks"] is True
    assert sh["same_surfel_count_as_plain"] is True
    mg = d["config"]["multi_gpu"]
    assert mg["global_model_surfels"] > 0 and mg["incremental"]["global_model_surfels"] > 0
    assert mg["incremental"]["new_surfels_exchanged"] > 0
