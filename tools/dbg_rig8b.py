import math, os, sys, threading, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from surfelmapping_amd import capi, synth, sharded
from surfelmapping_amd import dist as smd


def one_rep(rep, G, cam, over, streams, ref, native):
    grp = sharded.ThreadGroup(G)
    out, errs = [None] * G, []

    def work(r):
        try:
            sm = capi.SurfelMap(capi.make_config(**cam, **over))
            glob = capi.SurfelMap(capi.make_config(**cam, **dict(over, max_sqrt_vertices=3200)))
            mp = smd.RigMapper(sm, sharded.ThreadComm(grp, r), cam["width"] * cam["height"])
            cf = []
            for fr in streams[r]:
                mp.process_frame(*fr)
                cf.append(dict(sm.counts()))
            if cf[-1]["count"] != ref[r][1]:
                print("rep", rep, "rank", r, "AFTER FRAMES count", cf[-1]["count"], "oracle", ref[r][1], [(c["count"], c["unstable_count"], c["fused_count"], c["conflict_count"], c["visible_count"]) for c in cf], flush=True)
            if native:
                mp.enable_native(sharded.ThreadCollective(grp, r, sm))
                tot, per_view = mp.consolidate_native(glob)
            else:
                model, counts, per_view = mp.consolidate()
                tot = sum(counts)
            out[r] = (sm.download_model(), tot, per_view)
        except BaseException as e:
            errs.append((r, repr(e))); grp.barrier.abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in ts]; [t.join(600) for t in ts]
    if errs:
        print("errs", errs)
        return
    for r in range(G):
        a, b = ref[r][0], out[r][0]
        same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        if not same or r == 0:
            print("rep", rep, "rank", r, "cleaned slice oracle", a.shape[0], "product", b.shape[0], "identical" if same else "DIFFERENT", "total", out[r][1],
                  "per_view", out[r][2] if not same else "")


if __name__ == "__main__":
    cam = dict(synth.HD)
    over = dict(preprocess=0, stereo_border=12.0, max_sqrt_vertices=1800, conflict_cap=1)
    G, NF = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 3
    native = (sys.argv[2] if len(sys.argv) > 2 else "native") == "native"
    poses = lambda r: [synth.pose_matrix(0.0, 0.0, 0.6 * k, 14.0 * r + 0.4 * math.sin(k)) for k in range(NF)]
    streams = synth.make_sequences_parallel([(cam, poses(r), 41, 4.0 + r, dict(seed=41 + r, n_boxes=12, length=22.0)) for r in range(G)], 8)
    os.environ["OMP_NUM_THREADS"] = "32"
    L = ol.lib(ol.OMP_LIB_PATH)
    L.smo_set_exempt_id.argtypes = [C.c_void_p, C.c_int32]
    ref = []
    for r in range(G):
        o = ol.Oracle(ol.make_config(**cam, **over), libpath=ol.OMP_LIB_PATH)
        oc = []
        for fr in streams[r]:
            o.process_frame(*fr); oc.append(o.counts())
        n_frames = o.counts()["count"]
        print("oracle rank", r, [(c["count"], c["unstable_count"], c["fused_count"], c["conflict_count"], c["visible_count"]) for c in oc])
        L.smo_set_exempt_id(o._h, 0 if r == 0 else -1)
        for v in range(G):
            o.clean_points(*streams[v][-1][1:])
        ref.append((o.download_model(), n_frames)); o.close()
    for rep in range(int(os.environ.get("REPS", "1"))):
        one_rep(rep, G, cam, over, streams, ref, native)
