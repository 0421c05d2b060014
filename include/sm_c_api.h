/*
 * sm_c_api.h -- C-ABI of the MI355X-native surfel-fusion core (libsurfelmapping_hip.so).
 *
 * The reference (SUSTech-SLAM-XYZZY/SurfelMapping) has no plugin/FFI layer: its boundary is
 * the C++ class surface build_map.cpp / load_map.cpp / gui/GUI.cpp compile against.  This
 * header is the thin C boundary the drop-in C++ facade (surfelmapping_amd/csrc/facade/) calls;
 * every entry point names the reference interface it replaces (file:line under
 * /root/reference).  Plain pointers and sizes only; no torch / HIP types in signatures.
 *
 * Threading: one sm_ctx = one HIP device + one stream; calls on a ctx are not re-entrant.
 * Host-buffer entry points are synchronous (the reference glFinish()es after every pass,
 * e.g. src/GlobalModel.cpp:341); *_device / *_async entry points only enqueue.
 */
#ifndef SM_C_API_H
#define SM_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM_API_VERSION 4   /* bumped whenever a struct or an entry point changes (3, 4: round 3 -- asynchronous host path, rig step, sm_timings::k_scan_own, sm_host_alloc_frame; the staged shard entry points are gone) */

/* error codes (reference: void returns + CheckGlDieOnError(); bool for map IO) */
enum {
    SM_OK = 0,
    SM_E_ARG = -1,          /* null / inconsistent argument (reference would segfault: src/SurfelMapping.cpp:130) */
    SM_E_CAPACITY = -2,     /* model would exceed MAX_VERTICES (unchecked in src/GlobalModel.cpp:627-629) */
    SM_E_UNSUPPORTED = -3,
    SM_E_HIP = -4,          /* a HIP runtime call failed; see sm_last_error() */
    SM_E_NO_DEVICE = -5,    /* no gfx950 device visible: the product has NO CPU fallback */
    SM_E_STALL = -6         /* an in-kernel hand-off wait hit its spin bound: the in-place compaction needs its whole grid
                               resident, which another job on the same GPU can prevent.  Contexts of one process that
                               share a GPU are safe (they use the ticket-ordered form of the kernel); for several
                               PROCESSES on one GPU set SM_COMPACT_TICKETS=1 in their environment. */
};

/* Config singleton values (src/Config.cpp:32-37) + the magic numbers of the hot path
 * (src/SurfelMapping.cpp:197,261; src/IndexMap.cpp:21). */
typedef struct sm_config {
    int32_t width, height;         /* Config::W(), Config::H() */
    float fx, fy, cx, cy;          /* Config::fx() ... */
    float near_clip;               /* 1.0  Config::nearClip() */
    float far_clip;                /* 30.0 Config::farClip()  */
    float fuse_thresh;             /* 0.0  Config::surfelFuseDistanceThreshFactor() */
    int32_t max_sqrt_vertices;     /* 5000 Config::maxSqrtVertices(); capacity = its square */
    int32_t time_delta;            /* 200  src/SurfelMapping.cpp:197 */
    float stereo_border;           /* 80   src/SurfelMapping.cpp:261 */
    int32_t preprocess;            /* 0: metricise only (p0a); 1: full chain p0a..p0e */
    int32_t conflict_cap;          /* 1: only the first W*H conflicts take effect (conflictVbo size, src/GlobalModel.cpp:54-57) */
    int32_t device;                /* HIP device ordinal */
    int32_t enable_timing;         /* 1: record hipEvents per stage (sm_stage_timings) */
    int32_t disable_tile_bounds;   /* 1: never skip tiles by their bounding box (A/B switch; results are identical) */
    int32_t compact_period;        /* 24: deferred compaction -- a cull only marks the surfels it removes (they keep their slots)
                                      and every 24th cull squeezes the dead slots out in one in-place pass; also whenever dead
                                      slots could make a frame exceed MAX_VERTICES.  0/1: compact at every cull like the
                                      reference.  Counts, ids and the stored model are identical for every period. */
} sm_config;

/* GlobalModel counters (src/GlobalModel.cpp:860-888) + tick (src/SurfelMapping.h:100) */
typedef struct sm_counts {
    uint32_t count;           /* getModel().second    */
    uint32_t offset;          /* getOffset()          */
    uint32_t data_count;      /* getData().second     */
    uint32_t conflict_count;  /* getConflict().second */
    uint32_t unstable_count;  /* getUnstable().second */
    uint32_t fused_count;     /* data entries that update an existing surfel (F) */
    uint32_t visible_count;   /* surfels that passed the index-map view test (V) */
    int32_t tick;
} sm_counts;

/* Stage timings in milliseconds (HIP events on the ctx stream, averaged over the frames since
 * the previous query), labelled with the reference's TICK/TOCK names
 * (src/SurfelMapping.cpp:120-250, src/GlobalModel.cpp:258,350,519,583). */
typedef struct sm_timings {
    float preprocess;         /* "Preprocess": k_prep (+ p0b..p0e)                         */
    float conflict;           /* "Conflict": k_conflict + k_scan_cull + k_compact (p2..p5) */
    float index_map;          /* "indexMap": 0, the splat is fused into k_compact          */
    float data_association;   /* "Data::Association" + "Update::Fuse": k_associate         */
    float concatenate;        /* "Concatenate": k_scan_new + k_append                      */
    float run;                /* "Run": whole frame                                        */
    /* per kernel */
    float k_prep, k_conflict, k_scan_cull, k_compact, k_associate, k_scan_new, k_append;
    uint32_t frames;          /* frames averaged */
    float event_overhead;     /* measured cost of one event record, already subtracted from the k_* fields */
    /* the cull slot by kernel: k_compact above averages over ALL frames; these two over their own frames */
    float k_compact_own;      /* k_compact, averaged over the frames that compacted */
    float k_cull_lazy;        /* k_cull_lazy, averaged over the frames that only marked the dead */
    uint32_t frames_compact;  /* how many of `frames` compacted */
    /* the frame forms of the default path, each kernel averaged over the frames that ran it:
     * one-pass frames (the cull only marks the dead): k_surfel_pass (conflict + cull + splat) and k_pass_fixup (0 where the
     * fixup step rides on the next frame's preparation launch: the two-launch frame of asynchronous streams);
     * direct-append frames: k_associate_direct (association + fuse + append); the other frames run k_conflict
     * (+ k_scan_cull + k_cull_finalize + k_compact) and k_associate + k_append_scan */
    float k_surfel_pass, k_pass_fixup, k_conflict_own;
    float k_associate_direct, k_associate_own, k_append_own;
    uint32_t frames_one_pass, frames_direct;
    /* asynchronous plain streams: frames whose preparation launch also carried the previous frame's association
     * (k_assoc_prep); k_prep_own / k_associate_direct then average only over the frames that launched those kernels alone */
    float k_assoc_prep, k_prep_own;
    uint32_t frames_merged, frames_assoc_alone;
    float k_scan_own;         /* k_scan_cull + k_cull_finalize, averaged over the frames that ran k_conflict */
} sm_timings;

/* Per-frame counters written by the device at the end of every fusing frame (ring of
 * SM_FRAME_LOG_LEN entries) so that an asynchronous run can be audited without host syncs. */
#define SM_FRAME_LOG_LEN 1024
typedef struct sm_frame_log {
    uint32_t tick;            /* time stamp of the frame */
    uint32_t n_before;        /* N  live surfels at frame start */
    uint32_t n_after_cull;    /* N' */
    uint32_t n_kill;
    uint32_t conflict_count;  /* C */
    uint32_t visible_count;   /* V */
    uint32_t fused_count;     /* F */
    uint32_t unstable_count;  /* U */
    uint32_t n_static;        /* surfels the in-place cull did not have to move */
    uint32_t n_conf_skipped;  /* surfels in tiles the conflict pass skipped by their bounding box */
    uint32_t n_splat_skipped; /* surfels in static tiles the index-map splat skipped by their bounding box */
    uint32_t n_slots;         /* model slots scanned by the cull = n_before + slots of surfels killed since the last compaction */
} sm_frame_log;

enum { SM_TEX_DEPTH_METRIC = 0, SM_TEX_DEPTH_FILTERED = 1, SM_TEX_LAST = 2 };

typedef struct sm_ctx sm_ctx;

int sm_api_version(void);
const char *sm_last_error(void);

/* Config::getInstance(fx,fy,cx,cy,rows,cols) defaults: src/Config.cpp:7-38 */
int sm_default_config(sm_config *c, int width, int height, float fx, float fy, float cx, float cy);

/* SurfelMapping::SurfelMapping() (src/SurfelMapping.cpp:12-25): allocates all device buffers. */
sm_ctx *sm_create(const sm_config *c);
/* SurfelMapping::~SurfelMapping() (src/SurfelMapping.cpp:27-49) */
void sm_destroy(sm_ctx *s);

/* SurfelMapping::processFrame (src/SurfelMapping.h:31-34, src/SurfelMapping.cpp:115-251).
 * rgb H*W*3 u8 (R first), depth_mm H*W u16 (0 invalid), semantic H*W u8, pose = column-major
 * 4x4 camera->world (Eigen::Matrix4f storage).  Inputs are borrowed for the call only. */
int sm_process_frame(sm_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm,
                     const uint8_t *semantic, const float *pose16);
/* Same, inputs already resident in device memory of this ctx's GPU; enqueue only.  Without the depth filter chain the
 * frame's last kernel (association + append) is held back and launched together with the NEXT call's image preparation
 * (one launch less per frame); sm_sync and every entry point that reads the model launch it first, so the only visible
 * effect is that a caller who never calls anything again must call sm_sync to have the last frame finished.
 * SM_DEFER_ASSOC=0 turns this off.  (One bounded exception to "enqueue only": when the model
 * is within one frame of MAX_VERTICES and the host has run ahead of the device, the call waits up to SM_CAPACITY_WAIT_US
 * (default 2000) microseconds for the device's slot count before deciding whether this frame's cull must compact.) */
int sm_process_frame_device(sm_ctx *s, const uint8_t *d_rgb, const uint16_t *d_depth_mm,
                            const uint8_t *d_semantic, const float *pose16);
/* The same for callers whose images live in HOST memory and who do not want to wait (SurfelMapping::processFrame uploads its
 * three images itself, src/SurfelMapping.cpp:122-128): the images are copied on a copy stream into one of three device input
 * sets, so the 2.8 MB host-to-device copy of frame f+1 runs while frame f computes; the call returns once copies and frame are
 * enqueued (it waits for the frame three calls back, which bounds the work in flight).  Images that live in buffers of
 * sm_host_alloc -- pinned memory a reader decodes straight into -- are copied from in place and must stay unchanged until
 * sm_inputs_consumed() or sm_sync() returns; any other pointer is first copied into pinned staging inside the call (one host
 * memcpy) and may be reused at once.  Caller memory is never registered with the runtime. */
int sm_process_frame_async(sm_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16);
int sm_inputs_consumed(sm_ctx *s);                                     /* waits for the copies only, not for the frames */
void *sm_host_alloc(sm_ctx *s, size_t bytes);                          /* hipHostMalloc, owned by the context */
/* The three images of ONE frame in one pinned block (colour | depth | class, each 16-byte aligned): sm_process_frame_async copies
 * such a frame with a single host-to-device transfer -- 58 us for 2.8 MB at 1242 x 375, the PCIe rate, against 87 us for three
 * transfers from separate buffers (tools/h2d_probe.hip).  Free with sm_host_free(s, *rgb). */
int sm_host_alloc_frame(sm_ctx *s, uint8_t **rgb, uint16_t **depth_mm, uint8_t **semantic);
int sm_host_free(sm_ctx *s, void *p);
/* Wait for all enqueued work; refresh counters; returns a sticky device-side error. */
int sm_sync(sm_ctx *s);

/* SurfelMapping::cleanPoints (src/SurfelMapping.cpp:496-532) */
int sm_clean_points(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic,
                    const float *pose16);
/* cleanPoints for a model SLICE of a multi-GPU rig (surfelmapping_amd/dist.py, BASELINE configs[4]): as sm_clean_points,
 * but the "surfel id 0 never conflicts" rule (conflict.geom:15) applies only where exempt_first != 0 -- i.e. on the rank
 * whose slice holds the first surfel of the single GlobalModel; the other slices have no id 0. */
int sm_clean_points_ex(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16, int exempt_first);
/* The same for rigs that stage the exchange outside the core (surfelmapping_amd/dist.py RigMapper.consolidate): between the
 * conflict test -- which changes nothing -- and the cull, `fn` is called with this slice's conflict count and returns how many
 * of them, in surfel order, may take effect: the slice's share of the union's W*H conflict records (src/GlobalModel.cpp:54-57),
 * i.e. W*H minus the conflicts of the slices before it, clamped; a negative return abandons the cull (the model is untouched)
 * and is passed on as the result. */
typedef long long (*sm_cap_fn)(void *user, uint32_t local_conflicts);
int sm_clean_points_cb(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16, int exempt_first,
                       sm_cap_fn fn, void *user);
/* SurfelMapping::reset (src/SurfelMapping.cpp:436-441): empties the model and sets tick = 0; the next processFrame
 * builds the model anew from that frame's raw cloud (GlobalModel::initialize), discarding anything uploaded in between.
 * The index map is left as the last predictIndices drew it. */
int sm_reset(sm_ctx *s);

/* GlobalModel::getModel/getData/getConflict/getUnstable/getOffset counts */
int sm_get_counts(sm_ctx *s, sm_counts *out);
/* GlobalModel::downloadMap payload (src/GlobalModel.cpp:905-911): AoS, 12 floats/surfel */
int sm_download_model_aos(sm_ctx *s, float *dst12, uint32_t cap, uint32_t *n);
/* GlobalModel::uploadMap payload (src/GlobalModel.cpp:995-1002) */
int sm_upload_model_aos(sm_ctx *s, const float *src12, uint32_t n);
/* GlobalModel::downloadMap / uploadMap files (src/GlobalModel.cpp:901-1011) */
int sm_save_map(sm_ctx *s, const char *path, int32_t start_id, int32_t end_id);
int sm_load_map(sm_ctx *s, const char *path, int32_t *start_id, int32_t *end_id);
/* IndexMap::indexTex/vertConfTex/colorTimeTex/normalRadTex read-back (src/IndexMap.h:70-88),
 * row-major H*W; any pointer may be NULL.  Ids are surfel positions in the model at the time of the last
 * predictIndices (a later cleanPoints / reset / upload does not redraw the map, as in the reference); the three
 * attribute planes are re-derived from the surfels those ids address in the model as it is NOW, i.e. they equal the
 * reference's textures only while the model has not changed since. */
int sm_download_index_map(sm_ctx *s, int32_t *id, float *vert_conf4, float *color_time4,
                          float *norm_rad4);
/* FeedbackBuffer "RAW" (SurfelMapping::getFeedbackBuffer(FeedbackBuffer::RAW), src/SurfelMapping.cpp:367-376,
 * src/FeedbackBuffer.cpp:85-145, surfel_feedback.vert:25-63): the raw surfel cloud of the last processed frame -- every
 * checkerboard pixel with 0 < z < farClip as a camera-frame surfel, 12 floats each (pos, 0.9 | colour bits, 0, time, time |
 * normal, radius), in vertex order (x-outer / y-inner).  The reference fills it on every frame after the first; it feeds
 * the GUI's "Draw raw" view (build_map.cpp:177-184) and GlobalModel::initialize after reset().  Computed on demand here.
 * dst12 may be NULL to query the count. */
int sm_download_raw_cloud(sm_ctx *s, float *dst12, uint32_t cap, uint32_t *n);
/* SurfelMapping::getTexture(DEPTH_METRIC / DEPTH_FILTERED / "LAST") read-back, row-major */
int sm_download_depth(sm_ctx *s, int which, float *dst);

/* GlobalModel::setImageSize + renderImage + the two texture downloads of SurfelMapping::acquireImages
 * (src/GlobalModel.cpp:772-833, src/SurfelMapping.cpp:378-434): novel view of the model from camera->world
 * pose `view16`; bgr_out h*w*3 u8 (B,G,R as FragColor = srgb.wzy), sem_out h*w u8 = class + 1, 0 = empty. */
int sm_render_image(sm_ctx *s, const float *view16, int w, int h, float fx, float fy, float cx, float cy,
                    uint8_t *bgr_out, uint8_t *sem_out);

/* ---- per-pass entry points (GlobalModel / IndexMap methods), synchronous ---- */
/* Upload RGB / metric depth / semantic textures directly (bypasses p0). */
int sm_set_frame(sm_ctx *s, const uint8_t *rgb, const float *depth_metric,
                 const uint8_t *semantic);
int sm_set_tick(sm_ctx *s, int32_t tick);
/* GlobalModel::processConflict + updateConflict (src/GlobalModel.cpp:396-515): conflict test,
 * in-place confidence decrement marks; sets conflict_count. */
int sm_stage_conflict(sm_ctx *s, const float *pose16, float min_depth, float max_depth,
                      float fuse_thresh, int is_clean);
/* GlobalModel::backMapping + buildModelMap (src/GlobalModel.cpp:517-579,639-681): stable
 * compaction of surfels with conf > 0; sets count = offset. */
int sm_stage_cull(sm_ctx *s);
/* IndexMap::predictIndices (src/IndexMap.cpp:138-198) */
int sm_stage_splat(sm_ctx *s, const float *pose16, int32_t time, float depth_cutoff,
                   int32_t time_delta);
/* GlobalModel::dataAssociate + updateFuse + backMapping + concatenate + buildModelMap
 * (src/GlobalModel.cpp:246-394,581-637) */
int sm_stage_associate_fuse(sm_ctx *s, const float *pose16, int32_t time, float depth_min,
                            float depth_max);

/* Stopwatch::getTimings() equivalent (src/Utils/Stopwatch.h:85-88) -- needs enable_timing */
int sm_stage_timings(sm_ctx *s, sm_timings *out);
/* Copy the newest `n` (<= SM_FRAME_LOG_LEN) frame-log entries, oldest first; returns the
 * number written in *written.  Synchronises the stream. */
int sm_read_frame_log(sm_ctx *s, sm_frame_log *out, uint32_t n, uint32_t *written);

/* ---- device-memory helpers for callers that stage frames in HBM ---- */
void *sm_device_alloc(sm_ctx *s, size_t bytes);
int sm_device_free(sm_ctx *s, void *p);
int sm_device_upload(sm_ctx *s, void *dst_device, const void *src_host, size_t bytes);
/* Multi-GPU "all-gather into a single GlobalModel" (BASELINE configs[4]): export the model as
 * AoS (12 f32 / surfel) into a device staging buffer owned by the ctx (valid until the next
 * export/download on this ctx) so that RCCL can gather it without a host round trip ... */
int sm_export_model_device(sm_ctx *s, void **d_aos, uint32_t *n);
/* ... and append `n` AoS surfels that already live in this GPU's memory to the model
 * (GlobalModel::concatenate's glCopyBufferSubData, src/GlobalModel.cpp:624-629). */
int sm_append_model_aos_device(sm_ctx *s, const float *d_src12, uint32_t n);
int sm_device_download(sm_ctx *s, void *dst_host, const void *src_device, size_t bytes);

/* diagnostic: how many frames so far took the rare path of the two-launch frame (their association waited, inside its launch, for
 * the publisher and the cap repair: conflicts > W*H, or the "id 0" surfel died).  Synchronises. */
int sm_debug_slow_frames(sm_ctx *s, uint32_t *n);
/* Diagnostic: processes that hold compute queues on this context's GPU according to the KFD driver's tables (>= 1: this
 * one included), or -1 if /sys/class/kfd is not readable.  The in-place compaction switches to its ticket-ordered form
 * (no co-residency assumption) whenever the value is > 1 or a second context of this process shares the GPU; the value is
 * re-read at most once per second.  SM_COMPACT_TICKETS=1 / 0 overrides the detection. */
int sm_gpu_process_count(sm_ctx *s);

/* ---- ONE camera stream sharded over `world` GPUs (BASELINE configs[3]): no host or Python between the stages of a
 * frame.  Every rank addresses surfels by the slot number the single-GPU run uses and stores only the segments it owns (owner
 * of a frame's new surfels = fusing-frame index % world); the key map (min), the fused-pixel mask + 3 counters (sum) and -- when
 * the model has more slots than pixels and the conflict cap is on -- the conflict masks (sum) are all-reduced on the context's
 * own stream through the installed collective -- RCCL's ncclAllReduce, bound at run time by sm_shard_rccl_init, or any callback
 * with the same meaning (tests: several contexts on one GPU).  All ranks hold the same counters (sm_get_counts) after every
 * frame; results are bit-identical to the single-GPU path, the W*H conflict cap included. */
enum { SM_COLL_SUM = 0, SM_COLL_MIN = 1, SM_COLL_GATHER = 2 };
/* SUM / MIN: all-reduce `count` unsigned 64-bit words from `send` to `recv` (may be equal; device memory of this context) over
 * the ranks.  GATHER: all-gather -- every rank contributes `count` words at `send`, `recv` receives world * count words, rank q's
 * at recv + q * count (in place when send == recv + rank * count, as ncclAllGather).  Enqueued on `hip_stream`; returns 0 on
 * success */
typedef int (*sm_collective_fn)(void *user, const void *send, void *recv, size_t count, int op, void *hip_stream);
/* on a new context (no frame yet); allocates the second surfel set the sharded compaction stages through */
int sm_shard_stream_configure(sm_ctx *s, int rank, int world);
int sm_shard_set_collective(sm_ctx *s, sm_collective_fn fn, void *user);
/* RCCL bootstrap: rank 0 makes the 128-byte id, the host program hands it to every rank (MPI / torch.distributed /
 * a file), each rank calls sm_shard_rccl_init -- a collective call (ncclCommInitRank) */
int sm_shard_rccl_unique_id(void *out128);
int sm_shard_rccl_init(sm_ctx *s, const void *id128);
int sm_shard_rccl_finalize(sm_ctx *s);
/* ranks of the context's RCCL communicator as RCCL itself reports them (ncclCommCount), or a negative SM_E_* */
int sm_shard_rccl_nranks(sm_ctx *s);
/* SurfelMapping::processFrame (src/SurfelMapping.cpp:115-251) on every rank with the same arguments; the _device form
 * takes device pointers and only enqueues (sm_sync to wait).  A frame is a sequence of collectives every rank must enter: an
 * error return from one rank (a failed launch or a failed RCCL call -- nothing a frame's data can cause) leaves the others inside
 * a collective and the communicator undefined; treat it as fatal for the stream and destroy the contexts on all ranks.  (The
 * rig's exchanges, which do run fallible local steps between collectives, carry a status word instead: sm_rig_consolidate.) */
int sm_shard_frame_device(sm_ctx *s, const uint8_t *d_rgb, const uint16_t *d_depth_mm, const uint8_t *d_semantic, const float *pose16);
int sm_shard_frame(sm_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16);
/* squeeze the dead slots out now (collective; frames do it every compact_period-th time by themselves) */
int sm_shard_compact(sm_ctx *s);
/* collective: compacts, then exposes this rank's part of the union as `*count` x 12 floats in device memory with zeros
 * in the other ranks' slots -- the integer (u32) sum over the ranks is the single GlobalModel in the reference's order */
int sm_shard_export_dense_device(sm_ctx *s, const float **d_out12, uint32_t *count);

/* ---- BASELINE configs[4]: a rig of `world` cameras, one context per rank (GPU), consolidated into a single GlobalModel.
 * Frames use sm_process_frame[_device] (no collective).  sm_rig_consolidate is collective: every rank passes its camera's
 * latest view; the union of the slices in rank order is cleaned against every view in rank order with
 * SurfelMapping::cleanPoints (src/SurfelMapping.cpp:496-532), each rank cleaning its own slice, and the cleaned slices are
 * appended in rank order to `global` (another context on the same GPU) on every rank.  The collective is the one installed
 * with sm_shard_rccl_init / sm_shard_set_collective after sm_rig_configure (world 1: none needed).
 * view_conflicts[world] (may be null): conflicts that took effect per view over all slices.  The reference's conflict cap -- at
 * most W*H conflicts per view, in the surfel order of the union (src/GlobalModel.cpp:54-57) -- is applied exactly: between a
 * view's conflict test and its cull the ranks exchange their conflict counts and every slice applies what the lower ranks
 * left of the W*H.  Views, counts and slices cross the ranks as all-gathers; every exchange carries a status word, so a rank
 * whose local step fails makes ALL ranks return an error together (nobody is left waiting inside a collective). */
int sm_rig_configure(sm_ctx *s, int rank, int world);
int sm_rig_consolidate(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16, sm_ctx *global,
                       uint32_t *view_conflicts, uint32_t *total);
/* The single GlobalModel DURING a run, one step (collective; call it every K frames): every rank's surfels created since its
 * previous step are all-gathered -- per-GPU new-surfel lists, SURVEY.md 8e -- and appended to `global` in rank order
 * (GlobalModel::concatenate), then `global` is cleaned against every camera's latest view in rank order (SurfelMapping::cleanPoints)
 * -- the same on every rank, on `global`'s stream.  The camera's own model is not changed.  new_surfels / global_count (may be
 * null): surfels exchanged in this step, surfels in `global` after it. */
int sm_rig_consolidate_step(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16, sm_ctx *global,
                            uint32_t *new_surfels, uint32_t *global_count);

#ifdef __cplusplus
}
#endif
#endif /* SM_C_API_H */
