/*
 * smo.h -- CPU ORACLE for the SurfelMapping per-frame fusion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported CPU baseline.  The product (surfelmapping_amd/) never
 * links, imports or calls this code.
 *
 * PARITY UNPINNED: the reference (SUSTech-SLAM-XYZZY/SurfelMapping) ships no tests,
 * golden vectors or fixtures for this path and its GL implementation cannot be built
 * or run here (Pangolin/OpenCV/Eigen absent, GL_NV_transform_feedback, no GL context;
 * SURVEY.md 8c).  This file is a pass-by-pass scalar restatement of the reference's
 * shaders and host code; every ambiguous GL rule is fixed by SURVEY.md Appendix A and
 * restated in DESIGN.md.  Known-answer tests K1..K13 derived by hand
 * from the cited shader lines pin it in tests/.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef SMO_H
#define SMO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/Config.cpp:32-37, src/SurfelMapping.cpp:197,261, src/IndexMap.cpp:21 */
typedef struct smo_config {
    int32_t width, height;
    float fx, fy, cx, cy;
    float near_clip;           /* 1.0  */
    float far_clip;            /* 30.0 */
    float fuse_thresh;         /* 0.0  surfel_fuse_distance_threshold_factor */
    int32_t max_sqrt_vertices; /* 5000 -> MAX_VERTICES = 25e6 */
    int32_t time_delta;        /* 200  */
    float stereo_border;       /* 80   */
    int32_t preprocess;        /* 0: p0a only (metricise); 1: full chain p0a..p0e */
    int32_t conflict_cap;      /* 1: only the first W*H conflicts take effect (A13) */
} smo_config;

typedef struct smo_counts {
    uint32_t count;          /* GlobalModel::count          */
    uint32_t offset;         /* GlobalModel::offset         */
    uint32_t data_count;     /* GlobalModel::dataCount      */
    uint32_t conflict_count; /* GlobalModel::conflictCount  */
    uint32_t unstable_count; /* GlobalModel::unstableCount  */
    uint32_t fused_count;    /* entries of dataVbo with tag >= 0 (F) */
    uint32_t visible_count;  /* surfels that passed the index-map view test (V) */
    int32_t tick;
} smo_counts;

enum { SMO_OK = 0, SMO_E_ARG = -1, SMO_E_CAPACITY = -2, SMO_E_UNSUPPORTED = -3 };

enum { SMO_TEX_DEPTH_METRIC = 0, SMO_TEX_DEPTH_FILTERED = 1, SMO_TEX_LAST = 2 };

typedef struct smo_ctx smo_ctx;

void smo_default_config(smo_config *c, int w, int h, float fx, float fy, float cx, float cy);
smo_ctx *smo_create(const smo_config *c);
void smo_destroy(smo_ctx *s);

/* SurfelMapping::processFrame (src/SurfelMapping.cpp:115-251). pose: column-major 4x4 camera->world. */
int smo_process_frame(smo_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm,
                      const uint8_t *sem, const float *pose);
/* the two halves of processFrame around the fusing passes (src/SurfelMapping.cpp:115-158,244-248) */
int smo_begin_frame(smo_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm, const uint8_t *sem,
                    const float *pose);
int smo_end_frame(smo_ctx *s);
/* SurfelMapping::cleanPoints (src/SurfelMapping.cpp:496-532) */
int smo_clean_points(smo_ctx *s, const uint16_t *depth_mm, const uint8_t *sem, const float *pose);
/* SurfelMapping::reset (src/SurfelMapping.cpp:436-441) */
int smo_reset(smo_ctx *s);

int smo_get_counts(const smo_ctx *s, smo_counts *out);
/* wall seconds smo_process_frame spent per pass since the last reset (preprocess, processConflict, updateConflict, backMapping,
 * buildModelMap, predictIndices, dataAssociate, updateFuse, concatenate): bench.py's CPU baseline reports them */
int smo_stage_seconds(smo_ctx *s, double *out9, int reset);
/* AoS export, 12 floats per surfel (src/Config.cpp:17-32) */
int smo_download_model(const smo_ctx *s, float *dst12, uint32_t cap, uint32_t *n);
/* GlobalModel::uploadMap payload (src/GlobalModel.cpp:995-1002) + mirror rebuild so that
 * processing can continue from the seeded model. */
int smo_upload_model(smo_ctx *s, const float *src12, uint32_t n);
int smo_download_index_map(const smo_ctx *s, int32_t *id, float *vc4, float *ct4, float *nr4);
int smo_download_depth(const smo_ctx *s, int which, float *dst);
int smo_download_data(const smo_ctx *s, float *dst12, uint32_t cap, uint32_t *n);

/* ---- stage-level entry points (one per reference pass) ---- */
/* Bypass upload + preprocessing: set RGB / DEPTH_METRIC / SEMANTIC "textures" directly. */
int smo_set_frame(smo_ctx *s, const uint8_t *rgb, const float *depth_metric, const uint8_t *sem);
int smo_set_tick(smo_ctx *s, int32_t tick);
int smo_stage_process_conflict(smo_ctx *s, const float *pose, float min_depth, float max_depth,
                               float fuse_thresh, int is_clean);   /* p2 */
int smo_stage_update_conflict(smo_ctx *s);                           /* p3 */
int smo_stage_back_mapping(smo_ctx *s);                              /* p4/p10 */
int smo_stage_build_model_map(smo_ctx *s);                           /* p5/p12 */
int smo_stage_predict_indices(smo_ctx *s, const float *pose, int time, float depth_cutoff,
                              int time_delta);                       /* p6 */
int smo_stage_data_associate(smo_ctx *s, const float *pose, int time, float depth_min,
                             float depth_max);                       /* p8 */
int smo_stage_update_fuse(smo_ctx *s);                               /* p9 */
int smo_stage_concatenate(smo_ctx *s);                               /* p11 */
/* computeFeedbackBuffers + GlobalModel::initialize, the tick==0 branch after reset() */
int smo_stage_initialize(smo_ctx *s, const float *pose, int time, float max_depth);

/* FeedbackBuffer::compute (src/FeedbackBuffer.cpp:85-145, surfel_feedback.vert:25-63): the raw camera-frame cloud of the
 * textures as they are now, with time stamp `time`; dst12 may be NULL to query the count */
int smo_raw_cloud(const smo_ctx *s, int time, float *dst12, uint32_t cap, uint32_t *n);

/* GlobalModel::renderImage (src/GlobalModel.cpp:772-833) with draw_image.vert / draw_image_adaptive.geom /
 * draw_image.frag: every surfel as a screen-space disc (two triangles + per-fragment circle test), z-buffered.
 * Outputs: bgr u8[h][w][3] (FragColor = srgb.wzy) and semantic u8[h][w] = class + 1 (0 = nothing drawn).
 * Rasterisation rules fixed by DESIGN.md "Renderer": 24.8 fixed-point vertices, top-left fill rule. */
int smo_render_image(const smo_ctx *s, const float *view, int w, int h, float fx, float fy, float cx, float cy,
                     uint8_t *bgr, uint8_t *sem);

/* ---- helpers for the multi-GPU shard tests: one oracle instance plays one rank ---- */
int smo_set_exempt_id(smo_ctx *s, int32_t id);
int smo_set_conflict_limit(smo_ctx *s, int64_t limit);   /* a rig slice's share of the W*H conflict records; < 0: the config's rule */
int smo_count_clean_conflicts(smo_ctx *s, const uint16_t *depth_mm, const uint8_t *sem, const float *pose, uint32_t *n);
int smo_download_zbuf(const smo_ctx *s, uint32_t *dst);
int smo_upload_index_ids(smo_ctx *s, const int32_t *idx, const uint8_t *has);
int smo_download_data_pixels(const smo_ctx *s, int32_t *dst, uint32_t cap, uint32_t *n);
int smo_filter_data(smo_ctx *s, const uint8_t *keep);

/* ---- pre-processing passes on explicit buffers (p0a..p0e) ---- */
void smo_metricise(const smo_config *c, const uint16_t *raw, float *out);
void smo_filter_depth(const smo_config *c, const float *d, const uint8_t *sem, float diff_thresh,
                      float *out);
void smo_smooth_depth(const smo_config *c, const float *d, const uint8_t *sem, float *out);
void smo_remove_movings(const smo_config *c, const float *d, const uint8_t *sem,
                        const float *last, const float *t_c2l, float *out);

/* ---- scalar helpers exported for known-answer tests ---- */
float smo_encode_color(float r, float g, float b, uint32_t sem); /* color.glsl:19-26 */
float smo_get_radius(float depth, float norm_z, float inv_fx, float inv_fy); /* surfels.glsl:19-32 */
float smo_acosf(float x);
float smo_expf(float x);
void smo_invert4(const float *m, float *out);
void smo_mul4(const float *a, const float *b, float *out);

#ifdef __cplusplus
}
#endif
#endif
