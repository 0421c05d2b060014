/*
 * smo.c -- CPU ORACLE (test infrastructure, see smo.h).  PARITY UNPINNED (smo.h).
 *
 * Scalar, pass-by-pass restatement of the reference's per-frame hot path.  Each function
 * cites the reference file:line it follows (paths under /root/reference).  Floating point:
 * IEEE fp32, round-to-nearest-even, no FMA contraction (build with -ffp-contract=off
 * -fno-fast-math), fixed evaluation order:
 *     mat*vec   r_i = ((m_i0*x + m_i1*y) + m_i2*z) + m_i3
 *     dot       (a.x*b.x + a.y*b.y) + a.z*b.z
 *     normalize v / sqrt(dot(v,v))      (per-component division)
 *     min(a,b)  (b < a) ? b : a         (GLSL definition)
 * The mirror textures modelMap{VertsConfs,ColorsTime,NormsRadii} are kept, but only the
 * texels that can ever be read (index < count) are cleared instead of the full 5000x5000.
 */
#include "smo.h"

#include <math.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
/* All-core build (oracle/Makefile: libsmo_omp.so, -fopenmp): the per-surfel / per-pixel loops run in parallel, every
 * ordered output (conflict records, compactions, data records) is produced through flags + prefix sums so that the
 * result is bit-identical to the serial build (tests/test_oracle_omp.py).  Used for bench.py's all-core CPU baseline. */
#include <omp.h>
#endif

#define SURFEL_F 12

/* ---- sensitivity variants (tools/oracle_sensitivity.py).  The parity contract is the build with NONE of these
 * defined; each switch replaces one of the free choices DESIGN.md 2 lists by another plausible GL behaviour, so that
 * the effect of that choice on counts and fields can be measured (profiles/oracle_sensitivity.md):
 *   SMO_VAR_RCP        a / b evaluated as a * (1 / b)     (shader compilers commonly lower division this way)
 *   SMO_VAR_LIBM       acosf / expf of libm instead of the fixed kernels
 *   SMO_VAR_D24_TRUNC  24-bit depth = trunc(z * (2^24 - 1)) instead of round-half-up
 *   SMO_VAR_INV_DOUBLE pose.inverse() evaluated in double precision, rounded once
 * (FMA contraction is a compiler switch: -ffp-contract=fast -mfma.) */
#ifdef SMO_VAR_RCP
static inline float fdiv(float a, float b) { return a * (1.0f / b); }
#else
static inline float fdiv(float a, float b) { return a / b; }
#endif

struct smo_ctx {
    smo_config c;
    int P;
    /* per-column / per-row coordinates exactly as the shaders see them */
    float *tcx, *tcy;      /* texcoord = float((i+0.5)/(double)(float)W)  src/GlobalModel.cpp:71-72 */
    float *xs, *ys;        /* x = texcoord.x * cols                       data.vert:62-63 */
    int *ixm, *ixc, *ixp;  /* nearest/clamp texel of texcoord.x -1/cols, +0, +1/cols (A1) */
    int *iym, *iyc, *iyp;
    float *xs_fb, *ys_fb;  /* FeedbackBuffer's uvo: float(i/(float)W + 1.0/(2W)) * cols  src/FeedbackBuffer.cpp:47-53 */
    /* "textures" (src/SurfelMapping.cpp:51-87) */
    float *rgb;            /* RGB32F, P*3 */
    uint16_t *depth_raw;
    float *depth_metric, *depth_filtered, *last;
    uint8_t *sem;
    /* GlobalModel buffers (src/GlobalModel.cpp:33-63) */
    float *model;   uint32_t model_cap;
    float *mvc, *mct, *mnr; uint32_t mirror_cap; uint32_t mirror_dirty;
    float *data;       /* P*12 */
    float *unstable;   /* P*12 */
    float *conflict;   uint32_t conflict_cap; /* records of 5 floats */
    /* IndexMap targets (src/IndexMap.cpp:23-39) */
    int32_t *idx; uint32_t *zbuf; float *ivc, *ict, *inr;
    uint32_t count, offset, data_count, conflict_count, unstable_count, fused_count, visible_count;
    int32_t tick;
    int ref_set;
    float curr_pose[16], last_pose[16];
    int32_t exempt_id;        /* surfel that never fuses / conflicts: id 0 (A5); shard tests move it */
    double stage_sec[12];     /* wall seconds per pass since the last smo_reset_stage_seconds (bench.py's CPU baseline: where the time goes) */
    int64_t conflict_limit;   /* < 0: the config's rule (W*H records or none); >= 0: this many records (a rig slice's share, tests/test_rig.py) */
    int32_t *data_pix;        /* column-major pixel index of every dataVbo record */
#ifdef _OPENMP
    uint8_t *omp_flag; uint32_t omp_flag_cap;   /* per-surfel flags of the parallel passes */
    uint64_t *omp_key;                          /* per-pixel (d24 << 32 | id) keys of the parallel index-map splat */
    uint32_t *omp_col;                          /* per-column record counts / offsets of the parallel association */
#endif
};

int smo_end_frame(smo_ctx *s);
int smo_stage_initialize(smo_ctx *s, const float *pose, int time_i, float max_depth);

/* ------------------------------------------------------------------ scalar helpers */

static inline float min_glsl(float a, float b) { return (b < a) ? b : a; }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* nearest + clamp-to-edge texel index of normalised coordinate t on an n-texel axis (A1) */
static inline int tex_idx(float t, int n)
{
    float f = floorf(t * (float)n);
    if (!(f >= 0.0f)) return 0;            /* negative or NaN */
    if (f > (float)(n - 1)) return n - 1;
    return (int)f;
}

static inline void xform(const float *m, float x, float y, float z, float *o)
{
    for (int i = 0; i < 4; ++i)
        o[i] = ((m[i] * x + m[4 + i] * y) + m[8 + i] * z) + m[12 + i];
}

static inline void rot3(const float *m, float x, float y, float z, float *o)
{
    for (int i = 0; i < 3; ++i)
        o[i] = (m[i] * x + m[4 + i] * y) + m[8 + i] * z;
}

static inline float dot3(const float *a, const float *b)
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

static inline void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

static inline void normalize3(float *v)
{
    float l = sqrtf(dot3(v, v));
    v[0] = fdiv(v[0], l); v[1] = fdiv(v[1], l); v[2] = fdiv(v[2], l);
}

/* acos: fixed rational approximation (fdlibm asinf kernel constants), evaluated with
 * +,-,*,/,sqrt only so that any IEEE implementation gives the same bits (A9).
 * |x|>1 or NaN -> NaN (data.vert:54-57 then compares NaN < 0.5 -> false, K13). */
float smo_acosf(float x)
{
#ifdef SMO_VAR_LIBM
    return acosf(x);
#endif
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f;
    const float qS1 = -7.0662963390e-01f;
    const float PIO2 = 1.57079637050628662109375f, PI = 3.1415927410125732421875f;
    float ax = fabsf(x);
    if (ax <= 0.5f) {
        float z = x * x;
        float r = (z * (pS0 + z * (pS1 + z * pS2))) / (1.0f + z * qS1);
        return PIO2 - (x + x * r);
    } else if (x > 0.0f) {
        float z = (1.0f - x) * 0.5f;
        float s = sqrtf(z);
        float r = (z * (pS0 + z * (pS1 + z * pS2))) / (1.0f + z * qS1);
        return 2.0f * (s + s * r);
    } else {
        float z = (1.0f + x) * 0.5f;
        float s = sqrtf(z);
        float r = (z * (pS0 + z * (pS1 + z * pS2))) / (1.0f + z * qS1);
        return PI - 2.0f * (s + s * r);
    }
}

/* exp: k = rint(x*log2e); r = (x - k*ln2hi) - k*ln2lo; degree-6 Taylor Horner; ldexp (A9) */
float smo_expf(float x)
{
#ifdef SMO_VAR_LIBM
    return expf(x);
#endif
    const float LOG2E = 1.44269502162933349609375f;
    const float LN2HI = 0.693145751953125f, LN2LO = 1.428606765330187045037746429443359375e-06f;
    float k = rintf(x * LOG2E);
    float r = (x - k * LN2HI) - k * LN2LO;
    float p = 1.0f / 720.0f;
    p = 1.0f / 120.0f + r * p;
    p = 1.0f / 24.0f + r * p;
    p = 1.0f / 6.0f + r * p;
    p = 0.5f + r * p;
    p = 1.0f + r * p;
    p = 1.0f + r * p;
    return ldexpf(p, (int)k);
}

/* color.glsl:19-26 */
static inline uint32_t round_u8(float v)
{
    float r = roundf(v);
    if (!(r >= 0.0f)) return 0u;
    if (r > 4294967040.0f) return 4294967040u;
    return (uint32_t)r;
}

float smo_encode_color(float r, float g, float b, uint32_t sem)
{
    uint32_t srgb = sem;
    srgb = (srgb << 8) + round_u8(r * 255.0f);
    srgb = (srgb << 8) + round_u8(g * 255.0f);
    srgb = (srgb << 8) + round_u8(b * 255.0f);
    return u2f(srgb);
}

/* surfels.glsl:19-32; cam.z = 1/fx, cam.w = 1/fy */
float smo_get_radius(float depth, float norm_z, float inv_fx, float inv_fy)
{
    float meanFocal = fdiv(fdiv(1.0f, fabsf(inv_fx)) + fdiv(1.0f, fabsf(inv_fy)), 2.0f);
    const float sqrt2 = 1.41421356237f;
    float radius = fdiv(depth, meanFocal) * sqrt2;
    float radius_n = fdiv(radius, fabsf(norm_z));
    radius_n = min_glsl(2.0f * radius, radius_n);
    return radius_n;
}

/* general 4x4 inverse, column-major, cofactor expansion, inv = adj * (1/det).
 * Stands in for Eigen::Matrix4f::inverse() (src/GlobalModel.cpp:419, src/IndexMap.cpp:157). */
#ifdef SMO_VAR_INV_DOUBLE
void smo_invert4(const float *mf, float *out)
{
    double m[16], inv[16];
    for (int i = 0; i < 16; ++i) m[i] = mf[i];
#else
void smo_invert4(const float *m, float *out)
{
    float inv[16];
#endif
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] +
             m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] -
             m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] +
             m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] -
              m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] -
             m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] +
             m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] -
             m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] +
              m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] +
             m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] -
             m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] +
              m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] -
              m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] -
             m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] +
             m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] -
              m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] +
              m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
#ifdef SMO_VAR_INV_DOUBLE
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    for (int i = 0; i < 16; ++i) out[i] = (float)(inv[i] / det);
#else
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    float rdet = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * rdet;
#endif
}

/* column-major 4x4 product, c_ij = ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j) + a_i3 b_3j */
void smo_mul4(const float *a, const float *b, float *out)
{
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r[j * 4 + i] = ((a[i] * b[j * 4] + a[4 + i] * b[j * 4 + 1]) + a[8 + i] * b[j * 4 + 2]) +
                           a[12 + i] * b[j * 4 + 3];
    memcpy(out, r, sizeof r);
}

/* ------------------------------------------------------------------ lifecycle */

void smo_default_config(smo_config *c, int w, int h, float fx, float fy, float cx, float cy)
{
    memset(c, 0, sizeof *c);
    c->width = w; c->height = h;
    c->fx = fx; c->fy = fy; c->cx = cx; c->cy = cy;
    c->near_clip = 1.0f;  c->far_clip = 30.0f;   /* src/Config.cpp:33-34 */
    c->fuse_thresh = 0.0f;                       /* src/Config.cpp:35 */
    c->max_sqrt_vertices = 5000;                 /* src/Config.cpp:37 */
    c->time_delta = 200;                         /* src/SurfelMapping.cpp:197 */
    c->stereo_border = 80.0f;                    /* src/SurfelMapping.cpp:261 */
    c->preprocess = 1;
    c->conflict_cap = 1;
}

/* memset over all cores (the all-core build clears ~50 MB of textures per frame: mirror planes, index-map planes) */
static void par_memset(void *p, int v, size_t n)
{
#ifdef _OPENMP
    const size_t chunk = (size_t)1 << 20;
    const long nch = (long)((n + chunk - 1) / chunk);
#pragma omp parallel for schedule(static)
    for (long c = 0; c < nch; ++c) {
        const size_t o = (size_t)c * chunk;
        memset((char *)p + o, v, n - o < chunk ? n - o : chunk);
    }
#else
    memset(p, v, n);
#endif
}

static void *xcalloc(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz);
    if (!p) abort();
    return p;
}

smo_ctx *smo_create(const smo_config *c)
{
    if (!c || c->width <= 0 || c->height <= 0) return NULL;
    smo_ctx *s = (smo_ctx *)xcalloc(1, sizeof *s);
    s->c = *c;
    int W = c->width, H = c->height;
    s->P = W * H;
    size_t P = (size_t)s->P;
    s->tcx = xcalloc(W, 4); s->xs = xcalloc(W, 4);
    s->tcy = xcalloc(H, 4); s->ys = xcalloc(H, 4);
    s->ixm = xcalloc(W, 4); s->ixc = xcalloc(W, 4); s->ixp = xcalloc(W, 4);
    s->iym = xcalloc(H, 4); s->iyc = xcalloc(H, 4); s->iyp = xcalloc(H, 4);
    s->xs_fb = xcalloc(W, 4); s->ys_fb = xcalloc(H, 4);
    float cols = (float)W, rows = (float)H;
    float px = 1.0f / cols, py = 1.0f / rows;   /* geometry.glsl:14-18 "1.0 / cols" */
    for (int i = 0; i < W; ++i) {
        s->tcx[i] = (float)((i + 0.5) / (double)cols);   /* src/GlobalModel.cpp:71 */
        s->xs[i] = s->tcx[i] * cols;                     /* data.vert:62 */
        s->ixm[i] = tex_idx(s->tcx[i] - px, W);
        s->ixc[i] = tex_idx(s->tcx[i], W);
        s->ixp[i] = tex_idx(s->tcx[i] + px, W);
        s->xs_fb[i] = (float)((double)((float)i / cols) + 1.0 / (double)(2.0f * cols)) * cols;
    }
    for (int j = 0; j < H; ++j) {
        s->tcy[j] = (float)((j + 0.5) / (double)rows);   /* src/GlobalModel.cpp:72 */
        s->ys[j] = s->tcy[j] * rows;                     /* data.vert:63 */
        s->iym[j] = tex_idx(s->tcy[j] - py, H);
        s->iyc[j] = tex_idx(s->tcy[j], H);
        s->iyp[j] = tex_idx(s->tcy[j] + py, H);
        s->ys_fb[j] = (float)((double)((float)j / rows) + 1.0 / (double)(2.0f * rows)) * rows;
    }
    s->rgb = xcalloc(P * 3, 4);
    s->depth_raw = xcalloc(P, 2);
    s->depth_metric = xcalloc(P, 4);
    s->depth_filtered = xcalloc(P, 4);
    s->last = xcalloc(P, 4);
    s->sem = xcalloc(P, 1);
    s->data = xcalloc(P * SURFEL_F, 4);
    s->unstable = xcalloc(P * SURFEL_F, 4);
    s->conflict = xcalloc(P * 5, 4);
    s->conflict_cap = (uint32_t)P;
    s->conflict_limit = -1;
    s->idx = xcalloc(P, 4); s->zbuf = xcalloc(P, 4);
    s->data_pix = xcalloc(P, 4);
    for (size_t q = 0; q < P; ++q) s->zbuf[q] = 16777215u;
    s->ivc = xcalloc(P * 4, 4); s->ict = xcalloc(P * 4, 4); s->inr = xcalloc(P * 4, 4);
    for (int i = 0; i < 16; ++i) s->curr_pose[i] = s->last_pose[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    return s;
}

void smo_destroy(smo_ctx *s)
{
    if (!s) return;
    free(s->tcx); free(s->tcy); free(s->xs); free(s->ys);
    free(s->ixm); free(s->ixc); free(s->ixp); free(s->iym); free(s->iyc); free(s->iyp);
    free(s->xs_fb); free(s->ys_fb);
    free(s->rgb); free(s->depth_raw); free(s->depth_metric); free(s->depth_filtered);
    free(s->last); free(s->sem);
    free(s->model); free(s->mvc); free(s->mct); free(s->mnr);
    free(s->data); free(s->unstable); free(s->conflict);
    free(s->idx); free(s->zbuf); free(s->data_pix); free(s->ivc); free(s->ict); free(s->inr);
#ifdef _OPENMP
    free(s->omp_flag); free(s->omp_key); free(s->omp_col);
#endif
    free(s);
}

static uint32_t max_vertices(const smo_ctx *s)
{
    return (uint32_t)s->c.max_sqrt_vertices * (uint32_t)s->c.max_sqrt_vertices;
}

static void ensure_model(smo_ctx *s, uint32_t n)
{
    if (n <= s->model_cap) return;
    uint32_t cap = s->model_cap ? s->model_cap : 1u << 16;
    while (cap < n) cap = (cap > (1u << 30)) ? n : cap * 2;
    if (cap > max_vertices(s) && n <= max_vertices(s)) cap = max_vertices(s);
    s->model = realloc(s->model, (size_t)cap * SURFEL_F * 4);
    if (!s->model) abort();
    s->model_cap = cap;
}

static void ensure_mirror(smo_ctx *s, uint32_t n)
{
    if (n <= s->mirror_cap) return;
    uint32_t cap = s->mirror_cap ? s->mirror_cap : 1u << 16;
    while (cap < n) cap = (cap > (1u << 30)) ? n : cap * 2;
    if (cap > max_vertices(s) && n <= max_vertices(s)) cap = max_vertices(s);
    s->mvc = realloc(s->mvc, (size_t)cap * 16);
    s->mct = realloc(s->mct, (size_t)cap * 16);
    s->mnr = realloc(s->mnr, (size_t)cap * 16);
    if (!s->mvc || !s->mct || !s->mnr) abort();
    /* new texels start cleared (glClear) */
    memset(s->mvc + (size_t)s->mirror_cap * 4, 0, (size_t)(cap - s->mirror_cap) * 16);
    memset(s->mct + (size_t)s->mirror_cap * 4, 0, (size_t)(cap - s->mirror_cap) * 16);
    memset(s->mnr + (size_t)s->mirror_cap * 4, 0, (size_t)(cap - s->mirror_cap) * 16);
    s->mirror_cap = cap;
}

/* ------------------------------------------------------------------ p0: pre-processing */

/* depth_metric.frag:15-35 ; host src/SurfelMapping.cpp:254-266 */
void smo_metricise(const smo_config *c, const uint16_t *raw, float *out)
{
    int W = c->width, H = c->height;
    uint32_t lo = (uint32_t)(c->near_clip * 1000.0f);
    uint32_t hi = (uint32_t)((c->far_clip - 0.001f) * 1000.0f);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            uint32_t v = raw[(size_t)j * W + i];
            float r = 0.0f;
            /* texcoord.x*cols < stereoBorder, fragment centre i+0.5 */
            if (!((float)i + 0.5f < c->stereo_border)) {
                if (v > lo && v < hi) r = (float)v / 1000.0f;
            }
            out[(size_t)j * W + i] = r;
        }
}

/* depth_filter.frag:16-80 ; host src/SurfelMapping.cpp:271-288,316-332 (maxD = 100) */
void smo_filter_depth(const smo_config *c, const float *d, const uint8_t *sem, float diff_thresh,
                      float *out)
{
    int W = c->width, H = c->height;
    float minD = c->near_clip, maxD = 100.0f;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            size_t p = (size_t)j * W + i;
            float depth = d[p];
            uint32_t cl = sem[p];
            float r = 0.0f;
            if (!(depth <= minD || depth >= maxD || cl == 10u || cl == 11u || cl == 12u)) {
                int support = 0;
                for (int iy = -1; iy <= 1; ++iy)
                    for (int ix = -1; ix <= 1; ++ix) {
                        if (iy == 0 && ix == 0) continue;
                        /* texX<0 || texX>1 -> skip: fragment centres, so exactly the
                         * out-of-image neighbours are skipped */
                        int qi = i + ix, qj = j + iy;
                        if (qi < 0 || qi >= W || qj < 0 || qj >= H) continue;
                        size_t q = (size_t)qj * W + qi;
                        if (fabsf(d[q] - depth) < diff_thresh && cl == sem[q]) support++;
                    }
                if (support >= 7) r = depth;
            }
            out[p] = r;
        }
}

/* depth_smooth.frag:17-82 ; host src/SurfelMapping.cpp:291-313.  NB the host passes
 * sigma_intensity2_inv_half (0.5/30^2) as "sigPix" (:309) -- reproduced. */
void smo_smooth_depth(const smo_config *c, const float *d, const uint8_t *sem, float *out)
{
    int W = c->width, H = c->height;
    float minD = c->near_clip, maxD = 100.0f;
    float sigma_intensity = 30.0f;
    float sigPix = 0.5f / (sigma_intensity * sigma_intensity);
    float wtab[13][13];
    for (int iy = -6; iy <= 6; ++iy)
        for (int ix = -6; ix <= 6; ++ix) {
            float sd2 = (float)(ix * ix + iy * iy);
            wtab[iy + 6][ix + 6] = smo_expf(-(sd2 * sigPix));
        }
    int border = (int)ceilf(c->stereo_border - 0.5f); /* texX < border/cols  <=>  i+0.5 < border */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            size_t p = (size_t)j * W + i;
            float depth = d[p];
            uint32_t cl = sem[p];
            float r = 0.0f;
            if (!(depth <= minD || depth >= maxD || cl == 10u)) {
                float sum1 = 0.0f, sum2 = 0.0f;
                int valid = 0;
                for (int iy = -6; iy <= 6; ++iy)
                    for (int ix = -6; ix <= 6; ++ix) {
                        int qi = i + ix, qj = j + iy;
                        if (qi < border || qi >= W || qj < 0 || qj >= H) continue;
                        size_t q = (size_t)qj * W + qi;
                        float dk = d[q];
                        if (dk <= minD || dk >= maxD || cl != sem[q]) continue;
                        float w = wtab[iy + 6][ix + 6];
                        sum1 += dk * w;
                        sum2 += w;
                        valid++;
                    }
                if (valid > 0) r = sum1 / sum2;
            }
            out[p] = r;
        }
}

/* depth_movings.frag:20-82 ; host src/SurfelMapping.cpp:336-365 (maxD=100, moveThresh=0.5) */
void smo_remove_movings(const smo_config *c, const float *d, const uint8_t *sem,
                        const float *last, const float *t_c2l, float *out)
{
    int W = c->width, H = c->height;
    float cols = (float)W, rows = (float)H;
    float minD = c->near_clip, maxD = 100.0f, moveThresh = 0.5f;
    float fx = c->fx, fy = c->fy, cx = c->cx, cy = c->cy;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            size_t p = (size_t)j * W + i;
            float depth = d[p];
            uint32_t cl = sem[p];
            float r = depth;
            float px = (float)i + 0.5f, py = (float)j + 0.5f;  /* texcoord*cols at fragment centre */
            if (!(px < c->stereo_border || depth <= minD) && (cl >= 13u && cl <= 18u)) {
                /* reproject(): depth_movings.frag:20-27 */
                float vx = (px - cx) * depth / fx, vy = (py - cy) * depth / fy, vz = depth;
                float t[4];
                xform(t_c2l, vx, vy, vz, t);
                float ux = fx * t[0] / t[2] + cx;
                float uy = fy * t[1] / t[2] + cy;
                float uz = t[2];
                if (!(uz <= minD || uz >= maxD || ux < c->stereo_border || ux > cols || uy < 0.0f ||
                      uy > rows)) {
                    int qi = tex_idx(ux / cols, W), qj = tex_idx(uy / rows, H);
                    float depth_last = last[(size_t)qj * W + qi];
                    if (fabsf(uz - depth_last) > moveThresh) r = 0.0f;
                }
            }
            out[p] = r;
        }
}

/* upload (src/SurfelMapping.cpp:122-128): RGB u8 -> RGB32F normalised (A1) */
static void upload_rgb(smo_ctx *s, const uint8_t *rgb)
{
    size_t n = (size_t)s->P * 3;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t k = 0; k < n; ++k) s->rgb[k] = (float)rgb[k] / 255.0f;
}

/* ------------------------------------------------------------------ p2: conflict */

/* conflict.vert:25-73 for surfel k: does the measured depth lie behind it along the ray? */
static int conflict_test(const smo_ctx *s, const float *t_inv, uint32_t k, float min_depth, float max_depth,
                         float fuse_thresh, int is_clean)
{
    const smo_config *c = &s->c;
    int W = c->width, H = c->height;
    float cols = (float)W, rows = (float)H;
    const float *v = s->model + (size_t)k * SURFEL_F;
    float ph[4];
    xform(t_inv, v[0], v[1], v[2], ph);
    float xl = fdiv(ph[0], ph[2]);
    float yl = fdiv(ph[1], ph[2]);
    float u = c->fx * xl + c->cx;
    float vv = c->fy * yl + c->cy;
    if (u < c->stereo_border || u > cols || vv < 0.0f || vv > rows || ph[2] <= min_depth ||
        ph[2] >= max_depth)
        return 0;   /* conf_id = -10 */
    float lambda = sqrtf((xl * xl + yl * yl) + 1.0f);
    int ti = tex_idx(fdiv(u, cols), W), tj = tex_idx(fdiv(vv, rows), H);
    float depth = s->depth_metric[(size_t)tj * W + ti];
    uint32_t sem = s->sem[(size_t)tj * W + ti];
    if (sem == 10u) depth = max_depth + 1.0f;
    if (is_clean == 0 && depth == 0.0f) depth = max_depth + 20.0f;
    return depth * lambda - ph[2] * lambda > fuse_thresh * ph[2];
}

/* GlobalModel::processConflict (src/GlobalModel.cpp:396-476), conflict.vert:25-83,
 * conflict.geom:13-24.  Writes records (idbits, x, y, z, conf-1); A13: transform feedback
 * stops when conflictVbo (W*H records) is full. */
int smo_stage_process_conflict(smo_ctx *s, const float *pose, float min_depth, float max_depth,
                               float fuse_thresh, int is_clean)
{
    const smo_config *c = &s->c;
    float t_inv[16];
    smo_invert4(pose, t_inv);
    uint32_t cap = c->conflict_cap ? (uint32_t)s->P : s->count;
    if (s->conflict_limit >= 0) cap = (uint32_t)(s->conflict_limit > (int64_t)s->count ? (int64_t)s->count : s->conflict_limit);
    if (cap > s->conflict_cap) {
        s->conflict = realloc(s->conflict, (size_t)cap * 5 * 4);
        if (!s->conflict) abort();
        s->conflict_cap = cap;
    }
    uint32_t n = 0;
#ifdef _OPENMP
    /* parallel: the conflict test of every surfel (flag); serial: the records in surfel order, capped */
    if (s->count > s->omp_flag_cap) {
        s->omp_flag = realloc(s->omp_flag, s->count);
        if (!s->omp_flag) abort();
        s->omp_flag_cap = s->count;
    }
#pragma omp parallel for schedule(static)
    for (uint32_t k = 0; k < s->count; ++k)
        s->omp_flag[k] = (uint8_t)conflict_test(s, t_inv, k, min_depth, max_depth, fuse_thresh, is_clean);
    {   /* the records in surfel order, capped: conflicts per chunk, exclusive prefix, every chunk writes its own range */
        enum { MAXT = 256 };
        uint32_t cnt[MAXT + 1];
        int nt = omp_get_max_threads();
        if (nt > MAXT) nt = MAXT;
        const uint32_t N = s->count, per = (N + (uint32_t)nt - 1) / (uint32_t)nt;
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const uint32_t k0 = (uint32_t)t * per < N ? (uint32_t)t * per : N, k1 = k0 + per < N ? k0 + per : N;
            uint32_t c = 0;
            for (uint32_t k = k0; k < k1; ++k) c += s->omp_flag[k] && (int32_t)k != s->exempt_id;
            cnt[t] = c;
#pragma omp barrier
#pragma omp single
            {
                uint32_t run = 0;
                for (int x = 0; x < nt; ++x) { const uint32_t y = cnt[x]; cnt[x] = run; run += y; }
                cnt[nt] = run;
            }
            uint32_t w = cnt[t];
            for (uint32_t k = k0; k < k1 && w < cap; ++k) {
                if (!s->omp_flag[k] || (int32_t)k == s->exempt_id) continue;
                const float *v = s->model + (size_t)k * SURFEL_F;
                float *r = s->conflict + (size_t)w * 5;
                r[0] = u2f(k);
                r[1] = v[0]; r[2] = v[1]; r[3] = v[2];
                r[4] = v[3] - 1.0f;
                w++;
            }
        }
        n = cnt[nt] < cap ? cnt[nt] : cap;
    }
#else
    for (uint32_t k = 0; k < s->count; ++k) {
        if (!conflict_test(s, t_inv, k, min_depth, max_depth, fuse_thresh, is_clean)) continue;
        if ((int32_t)k != s->exempt_id) {    /* conflict.geom:15: conf_id > 0, i.e. every id but 0 */
            if (n < cap) {
                const float *v = s->model + (size_t)k * SURFEL_F;
                float *r = s->conflict + (size_t)n * 5;
                r[0] = u2f(k);
                r[1] = v[0]; r[2] = v[1]; r[3] = v[2];
                r[4] = v[3] - 1.0f;          /* conflict.vert:72 */
                n++;
            }
        }
    }
#endif
    s->conflict_count = n;
    return SMO_OK;
}

/* GlobalModel::updateConflict (src/GlobalModel.cpp:478-515), update_conf.vert:11-27 */
int smo_stage_update_conflict(smo_ctx *s)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)      /* every record addresses its own texel (ids are unique) */
#endif
    for (uint32_t q = 0; q < s->conflict_count; ++q) {
        const float *r = s->conflict + (size_t)q * 5;
        uint32_t id = f2u(r[0]);
        if (id >= s->mirror_cap) continue;
        memcpy(s->mvc + (size_t)id * 4, r + 1, 16);
    }
    return SMO_OK;
}

/* GlobalModel::backMapping (src/GlobalModel.cpp:517-579), back_map.geom:15-28 */
int smo_stage_back_mapping(smo_ctx *s)
{
    uint32_t n = 0;
    ensure_mirror(s, s->count);
    ensure_model(s, s->count);
#ifdef _OPENMP
    {   /* stable compaction in parallel: survivors per chunk, exclusive prefix, every chunk copies to its own range */
        enum { MAXT = 256 };
        uint32_t cnt[MAXT + 1];
        int nt = omp_get_max_threads();
        if (nt > MAXT) nt = MAXT;
        const uint32_t N = s->count, per = (N + (uint32_t)nt - 1) / (uint32_t)nt;
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const uint32_t k0 = (uint32_t)t * per < N ? (uint32_t)t * per : N, k1 = k0 + per < N ? k0 + per : N;
            uint32_t c = 0;
            for (uint32_t k = k0; k < k1; ++k) c += s->mvc[(size_t)k * 4 + 3] > 0.0f;
            cnt[t] = c;
#pragma omp barrier
#pragma omp single
            {
                uint32_t run = 0;
                for (int x = 0; x < nt; ++x) { const uint32_t y = cnt[x]; cnt[x] = run; run += y; }
                cnt[nt] = run;
            }
            uint32_t w = cnt[t];
            for (uint32_t k = k0; k < k1; ++k) {
                const float *vc = s->mvc + (size_t)k * 4;
                if (vc[3] > 0.0f) {
                    float *o = s->model + (size_t)w * SURFEL_F;
                    memcpy(o, vc, 16);
                    memcpy(o + 4, s->mct + (size_t)k * 4, 16);
                    memcpy(o + 8, s->mnr + (size_t)k * 4, 16);
                    o[5] = 0.0f;
                    w++;
                }
            }
        }
        n = cnt[nt];
    }
#else
    for (uint32_t k = 0; k < s->count; ++k) {
        const float *vc = s->mvc + (size_t)k * 4;
        if (vc[3] > 0.0f) {
            float *o = s->model + (size_t)n * SURFEL_F;
            memcpy(o, vc, 16);
            memcpy(o + 4, s->mct + (size_t)k * 4, 16);
            memcpy(o + 8, s->mnr + (size_t)k * 4, 16);
            o[5] = 0.0f;                         /* back_map.geom:23 */
            n++;
        }
    }
#endif
    s->offset = n;
    s->count = n;                                /* src/GlobalModel.cpp:575 */
    return SMO_OK;
}

/* GlobalModel::buildModelMap (src/GlobalModel.cpp:639-681), map.vert:14-34 */
int smo_stage_build_model_map(smo_ctx *s)
{
    ensure_mirror(s, s->count);
    uint32_t clr = s->mirror_dirty;               /* texels that may be non-zero */
    if (clr > s->mirror_cap) clr = s->mirror_cap;
    par_memset(s->mvc, 0, (size_t)clr * 16);
    par_memset(s->mct, 0, (size_t)clr * 16);
    par_memset(s->mnr, 0, (size_t)clr * 16);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (uint32_t k = 0; k < s->count; ++k) {
        const float *v = s->model + (size_t)k * SURFEL_F;
        memcpy(s->mvc + (size_t)k * 4, v, 16);
        memcpy(s->mct + (size_t)k * 4, v + 4, 16);
        s->mct[(size_t)k * 4 + 1] = (float)(int32_t)k;   /* map.vert:32 */
        memcpy(s->mnr + (size_t)k * 4, v + 8, 16);
    }
    s->mirror_dirty = s->count;
    return SMO_OK;
}

/* ------------------------------------------------------------------ p6: index map */

/* IndexMap::predictIndices (src/IndexMap.cpp:138-198), index_map.vert:38-64,
 * index_map.frag:31-37; rasterisation/depth rules A3/A4. */
/* index_map.vert:38-64 for surfel k: camera-frame position ph, target pixel p (row-major), 24-bit depth.
 * Returns 0 if the surfel draws no fragment (view test, clipping, depth test against the clear value). */
static int splat_project(const smo_ctx *s, const float *t_inv, uint32_t k, int time, float depth_cutoff, int time_delta,
                         float *ph, size_t *pix, uint32_t *d24_out)
{
    const smo_config *c = &s->c;
    int W = c->width, H = c->height;
    float cols = (float)W, rows = (float)H;
    const float *v = s->model + (size_t)k * SURFEL_F;
    xform(t_inv, v[0], v[1], v[2], ph);
    if (ph[2] >= depth_cutoff * 1.5f || ph[2] <= 0.0f ||
        (float)time - v[7] > (float)time_delta)
        return 0;                             /* index_map.vert:45-50 (clipped at -10,-10) */
    float xn = fdiv((fdiv(c->fx * ph[0], ph[2]) + c->cx) - (cols * 0.5f), cols * 0.5f);
    float yn = fdiv((fdiv(c->fy * ph[1], ph[2]) + c->cy) - (rows * 0.5f), rows * 0.5f);
    float zn = fdiv(ph[2], depth_cutoff);
    if (!(xn >= -1.0f && xn <= 1.0f && yn >= -1.0f && yn <= 1.0f && zn >= -1.0f && zn <= 1.0f))
        return 0;                             /* clip volume */
    float xw = (cols * 0.5f) * xn + (cols * 0.5f);
    float yw = (rows * 0.5f) * yn + (rows * 0.5f);
    float fxw = floorf(xw), fyw = floorf(yw);
    if (!(fxw >= 0.0f && fxw < cols && fyw >= 0.0f && fyw < rows)) return 0;
    int px = (int)fxw, py = (int)fyw;
    float zw = 0.5f * zn + 0.5f;
#ifdef SMO_VAR_D24_TRUNC
    uint32_t d24 = (uint32_t)((double)zw * 16777215.0);
#else
    uint32_t d24 = (uint32_t)floor((double)zw * 16777215.0 + 0.5);
#endif
    if (d24 >= 16777215u) return 0;           /* fails GL_LESS against the clear value */
    *pix = (size_t)py * W + px;
    *d24_out = d24;
    return 1;
}

/* the fragment of surfel k lands in pixel p: index_map.frag:31-37 outputs */
static void splat_write(smo_ctx *s, const float *t_inv, uint32_t k, const float *ph, size_t p, uint32_t d24)
{
    const float *v = s->model + (size_t)k * SURFEL_F;
    s->zbuf[p] = d24;
    s->idx[p] = (int32_t)k;
    float *o = s->ivc + p * 4;
    o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = v[3];
    memcpy(s->ict + p * 4, v + 4, 16);
    float n[3];
    rot3(t_inv, v[8], v[9], v[10], n);
    normalize3(n);
    o = s->inr + p * 4;
    o[0] = n[0]; o[1] = n[1]; o[2] = n[2]; o[3] = v[11];
}

int smo_stage_predict_indices(smo_ctx *s, const float *pose, int time, float depth_cutoff,
                              int time_delta)
{
    size_t P = (size_t)s->P;
    float t_inv[16];
    smo_invert4(pose, t_inv);
    par_memset(s->idx, 0, P * 4);                 /* glClearColor(0,0,0,0) src/IndexMap.cpp:151 */
    par_memset(s->ivc, 0, P * 16); par_memset(s->ict, 0, P * 16); par_memset(s->inr, 0, P * 16);
    uint32_t vis = 0;
#ifdef _OPENMP
    /* parallel z-buffer: GL_LESS with draw order = lexicographic minimum of (d24, id), kept as one 64-bit key per pixel */
    if (!s->omp_key) { s->omp_key = malloc(P * 8); if (!s->omp_key) abort(); }
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < P; ++p) { s->zbuf[p] = 16777215u; s->omp_key[p] = ~0ull; }
#pragma omp parallel for schedule(static) reduction(+ : vis)
    for (uint32_t k = 0; k < s->count; ++k) {
        float ph[4];
        size_t p;
        uint32_t d24;
        if (!splat_project(s, t_inv, k, time, depth_cutoff, time_delta, ph, &p, &d24)) continue;
        vis++;
        const uint64_t key = ((uint64_t)d24 << 32) | k;
        uint64_t cur = __atomic_load_n(&s->omp_key[p], __ATOMIC_RELAXED);
        while (key < cur && !__atomic_compare_exchange_n(&s->omp_key[p], &cur, key, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    }
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < P; ++p) {
        const uint64_t key = s->omp_key[p];
        if (key == ~0ull) continue;
        float ph[4];
        size_t p2;
        uint32_t d24;
        splat_project(s, t_inv, (uint32_t)key, time, depth_cutoff, time_delta, ph, &p2, &d24);
        splat_write(s, t_inv, (uint32_t)key, ph, p, d24);
    }
#else
    for (size_t p = 0; p < P; ++p) s->zbuf[p] = 16777215u;   /* depth cleared to 1.0 */
    for (uint32_t k = 0; k < s->count; ++k) {
        float ph[4];
        size_t p;
        uint32_t d24;
        if (!splat_project(s, t_inv, k, time, depth_cutoff, time_delta, ph, &p, &d24)) continue;
        vis++;
        if (d24 < s->zbuf[p]) splat_write(s, t_inv, k, ph, p, d24);   /* GL_LESS: earlier (lower id) wins ties */
    }
#endif
    s->visible_count = vis;
    return SMO_OK;
}

/* ------------------------------------------------------------------ p8: data association */

static inline void get_vertex(const smo_ctx *s, int ti, int tj, float x, float y, float *o)
{
    /* geometry.glsl:5-9 ; cam = (cx, cy, 1/fx, 1/fy) src/GlobalModel.cpp:273-276 */
    float z = s->depth_metric[(size_t)tj * s->c.width + ti];
    float camz = (float)(1.0 / (double)s->c.fx), camw = (float)(1.0 / (double)s->c.fy);
    o[0] = (x - s->c.cx) * z * camz;
    o[1] = (y - s->c.cy) * z * camw;
    o[2] = z;
}

/* data.vert:59-234 + data.geom:32-45 for pixel (i, j): returns 0 if the vertex is dropped (tag -10), else writes the
 * dataVbo record to o and returns 1 (new surfel, tag -1) or 2 (fused into surfel `tag`). */
static int associate_one(const smo_ctx *s, const float *pose, float time, float depth_min, float depth_max, int i, int j,
                         float *o)
{
    const smo_config *c = &s->c;
    int W = c->width;
    float camz = (float)(1.0 / (double)c->fx), camw = (float)(1.0 / (double)c->fy);
    float fuseThresh = c->fuse_thresh;            /* src/GlobalModel.cpp:284 */
    const float *D = s->depth_metric;
    float x = s->xs[i], y = s->ys[j];
    float xl = (x - c->cx) * camz;
    float yl = (y - c->cy) * camw;
    float ray[3] = {xl, yl, 1.0f};
    float lambda = sqrtf((xl * xl + yl * yl) + 1.0f);
    int ci = s->ixc[i], cj = s->iyc[j];
    float value = D[(size_t)cj * W + ci];
    /* checkNeighbours data.vert:33-52 */
    if (D[(size_t)cj * W + s->ixm[i]] == 0.0f) return 0;
    if (D[(size_t)s->iym[j] * W + ci] == 0.0f) return 0;
    if (D[(size_t)cj * W + s->ixp[i]] == 0.0f) return 0;
    if (D[(size_t)s->iyp[j] * W + ci] == 0.0f) return 0;
    if (!(value > depth_min && value < depth_max)) return 0;
    if (((int)x + (int)y) % 2 != 1) return 0;

    float vPosLocal[3];
    get_vertex(s, ci, cj, x, y, vPosLocal);
    /* getNormal geometry.glsl:12-24 */
    float xf[3], xb[3], yf[3], yb[3], del_x[3], del_y[3], vNormLocal[3];
    get_vertex(s, s->ixp[i], cj, x + 1.0f, y, xf);
    get_vertex(s, s->ixm[i], cj, x - 1.0f, y, xb);
    get_vertex(s, ci, s->iyp[j], x, y + 1.0f, yf);
    get_vertex(s, ci, s->iym[j], x, y - 1.0f, yb);
    for (int q = 0; q < 3; ++q) { del_x[q] = xb[q] - xf[q]; del_y[q] = yb[q] - yf[q]; }
    cross3(del_x, del_y, vNormLocal);
    normalize3(vNormLocal);

    float c_n = 0.9f;                     /* data.vert:104 */
    size_t p = (size_t)cj * W + ci;
    float color_n[3] = {s->rgb[p * 3], s->rgb[p * 3 + 1], s->rgb[p * 3 + 2]};
    float radii_n = smo_get_radius(vPosLocal[2], vNormLocal[2], camz, camw);
    uint32_t sem_n = s->sem[p];

    int updateCounter = 0;
    int bestID = 0;
    float bestDist = 1000.0f;
    float posLocal_o[3] = {0, 0, 0}, c_o = 0.0f, normRad_o[4] = {0, 0, 0, 0};
    float color_o[3] = {0, 0, 0}, initTime_o = 0.0f;

    /* window loop data.vert:126-172 with scale == IndexMap::FACTOR == 1: one lookup */
    int currentID = s->idx[p];
    /* data.vert:142 `currentID > 0`: a projection exists and it is not surfel 0 (the index
     * texture is cleared to 0, A5); written via the depth buffer so that shard tests can
     * move the exempt id */
    if (s->zbuf[p] != 16777215u && currentID != s->exempt_id) {
        const float *vertConf = s->ivc + p * 4;
        const float *colorTime = s->ict + p * 4;
        uint32_t sc = f2u(colorTime[0]);
        uint32_t sem_o = (sc >> 24) & 0xFFu;
        if (sem_n == sem_o &&
            fabsf(vertConf[2] * lambda - vPosLocal[2] * lambda) <= fuseThresh) {
            float cr[3];
            cross3(ray, vertConf, cr);
            float dist = fdiv(sqrtf(dot3(cr, cr)), sqrtf(dot3(ray, ray)));
            const float *normRad = s->inr + p * 4;
            float ang = smo_acosf(fdiv(dot3(normRad, vNormLocal),
                                       sqrtf(dot3(normRad, normRad)) *
                                       sqrtf(dot3(vNormLocal, vNormLocal))));
            if (dist < bestDist && fabsf(ang) < 0.5f) {
                updateCounter++;
                bestDist = dist;
                bestID = currentID;
                memcpy(posLocal_o, vertConf, 12);
                c_o = vertConf[3];
                memcpy(normRad_o, normRad, 16);
                color_o[0] = (float)((sc >> 16) & 0xFFu) / 255.0f;
                color_o[1] = (float)((sc >> 8) & 0xFFu) / 255.0f;
                color_o[2] = (float)(sc & 0xFFu) / 255.0f;
                initTime_o = colorTime[2];
            }
        }
    }

    float t4[4], n3[3];
    if (updateCounter > 0) {
        if (radii_n < 1.5f * normRad_o[3]) {              /* data.vert:177-194 */
            float w = c_n + c_o;
            float pn[3];
            for (int q = 0; q < 3; ++q)
                pn[q] = fdiv((c_n * vPosLocal[q]) + (c_o * posLocal_o[q]), w);
            xform(pose, pn[0], pn[1], pn[2], t4);
            o[0] = t4[0]; o[1] = t4[1]; o[2] = t4[2]; o[3] = w;
            float avg[3];
            for (int q = 0; q < 3; ++q)
                avg[q] = fdiv((c_n * color_n[q]) + (c_o * color_n[q]), w);   /* sic :183 */
            o[4] = smo_encode_color(avg[0], avg[1], avg[2], sem_n);
            o[5] = u2f((uint32_t)bestID);
            o[6] = initTime_o;
            o[7] = time;
            float nr[4];
            nr[0] = fdiv((c_n * vNormLocal[0]) + (c_o * normRad_o[0]), w);
            nr[1] = fdiv((c_n * vNormLocal[1]) + (c_o * normRad_o[1]), w);
            nr[2] = fdiv((c_n * vNormLocal[2]) + (c_o * normRad_o[2]), w);
            rot3(pose, nr[0], nr[1], nr[2], n3);
            normalize3(n3);
            o[8] = n3[0]; o[9] = n3[1]; o[10] = n3[2];
            o[11] = (radii_n > normRad_o[3]) ? normRad_o[3] : radii_n;
        } else {                                          /* data.vert:195-208 */
            xform(pose, posLocal_o[0], posLocal_o[1], posLocal_o[2], t4);
            o[0] = t4[0]; o[1] = t4[1]; o[2] = t4[2]; o[3] = c_n + c_o;
            o[4] = smo_encode_color(color_o[0], color_o[1], color_o[2], sem_n);
            o[5] = u2f((uint32_t)bestID);
            o[6] = initTime_o;
            o[7] = time;
            rot3(pose, normRad_o[0], normRad_o[1], normRad_o[2], n3);
            normalize3(n3);
            o[8] = n3[0]; o[9] = n3[1]; o[10] = n3[2];
            o[11] = normRad_o[3];
        }
    } else {                                              /* data.vert:210-225 */
        xform(pose, vPosLocal[0], vPosLocal[1], vPosLocal[2], t4);
        o[0] = t4[0]; o[1] = t4[1]; o[2] = t4[2]; o[3] = c_n;
        rot3(pose, vNormLocal[0], vNormLocal[1], vNormLocal[2], n3);
        normalize3(n3);
        o[8] = n3[0]; o[9] = n3[1]; o[10] = n3[2]; o[11] = radii_n;
        o[4] = smo_encode_color(color_n[0], color_n[1], color_n[2], sem_n);
        o[5] = -1.0f;
        o[6] = time;
        o[7] = time;
    }
    return updateCounter > 0 ? 2 : 1;
}

/* GlobalModel::dataAssociate (src/GlobalModel.cpp:246-346), data.vert:59-234, data.geom:32-45 */
int smo_stage_data_associate(smo_ctx *s, const float *pose, int time_i, float depth_min,
                             float depth_max)
{
    const smo_config *c = &s->c;
    int W = c->width, H = c->height;
    float time = (float)time_i;                   /* src/GlobalModel.cpp:271 */
    uint32_t n = 0, nf = 0;
#ifdef _OPENMP
    /* records are emitted in x-outer / y-inner order: count per column first (into a scratch record), then every column
     * writes at its offset */
    if (!s->omp_col) { s->omp_col = malloc(((size_t)W + 1) * 4); if (!s->omp_col) abort(); }
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < W; ++i) {
        float tmp[SURFEL_F];
        uint32_t cnt = 0;
        for (int j = 0; j < H; ++j) cnt += associate_one(s, pose, time, depth_min, depth_max, i, j, tmp) != 0;
        s->omp_col[i] = cnt;
    }
    for (int i = 0; i < W; ++i) { const uint32_t x = s->omp_col[i]; s->omp_col[i] = n; n += x; }
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : nf)
    for (int i = 0; i < W; ++i) {
        uint32_t w = s->omp_col[i];
        for (int j = 0; j < H; ++j) {
            const int r = associate_one(s, pose, time, depth_min, depth_max, i, j, s->data + (size_t)w * SURFEL_F);
            if (r) { s->data_pix[w] = i * H + j; w++; nf += r == 2; }
        }
    }
#else
    for (int i = 0; i < W; ++i) {                 /* x-outer, y-inner: src/GlobalModel.cpp:67-74 */
        for (int j = 0; j < H; ++j) {
            const int r = associate_one(s, pose, time, depth_min, depth_max, i, j, s->data + (size_t)n * SURFEL_F);
            if (r) { s->data_pix[n] = i * H + j; n++; nf += r == 2; }
        }
    }
#endif
    s->data_count = n;
    s->fused_count = nf;
    return SMO_OK;
}

/* GlobalModel::updateFuse (src/GlobalModel.cpp:348-394), fuse.vert:17-49 */
int smo_stage_update_fuse(smo_ctx *s)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)      /* at most one record per surfel id */
#endif
    for (uint32_t q = 0; q < s->data_count; ++q) {
        const float *d = s->data + (size_t)q * SURFEL_F;
        int32_t mark = (int32_t)f2u(d[5]);
        if (mark >= 0) {
            if ((uint32_t)mark >= s->mirror_cap) continue;
            memcpy(s->mvc + (size_t)mark * 4, d, 16);
            memcpy(s->mct + (size_t)mark * 4, d + 4, 16);
            memcpy(s->mnr + (size_t)mark * 4, d + 8, 16);
        }
    }
    return SMO_OK;
}

/* GlobalModel::concatenate (src/GlobalModel.cpp:581-637), unstable.vert:13-34 */
int smo_stage_concatenate(smo_ctx *s)
{
    uint32_t n = 0;
#ifdef _OPENMP
    {   /* ordered compaction in parallel, as in smo_stage_back_mapping */
        enum { MAXT = 256 };
        uint32_t cnt[MAXT + 1];
        int nt = omp_get_max_threads();
        if (nt > MAXT) nt = MAXT;
        const uint32_t N = s->data_count, per = (N + (uint32_t)nt - 1) / (uint32_t)nt;
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const uint32_t q0 = (uint32_t)t * per < N ? (uint32_t)t * per : N, q1 = q0 + per < N ? q0 + per : N;
            uint32_t c = 0;
            for (uint32_t q = q0; q < q1; ++q) c += (int)roundf(s->data[(size_t)q * SURFEL_F + 5]) < 0;
            cnt[t] = c;
#pragma omp barrier
#pragma omp single
            {
                uint32_t run = 0;
                for (int x = 0; x < nt; ++x) { const uint32_t y = cnt[x]; cnt[x] = run; run += y; }
                cnt[nt] = run;
            }
            uint32_t w = cnt[t];
            for (uint32_t q = q0; q < q1; ++q) {
                const float *d = s->data + (size_t)q * SURFEL_F;
                if ((int)roundf(d[5]) < 0) {
                    float *o = s->unstable + (size_t)w * SURFEL_F;
                    memcpy(o, d, SURFEL_F * 4);
                    o[5] = 0.0f;
                    w++;
                }
            }
        }
        n = cnt[nt];
    }
#else
    for (uint32_t q = 0; q < s->data_count; ++q) {
        const float *d = s->data + (size_t)q * SURFEL_F;
        int mark = (int)roundf(d[5]);
        if (mark < 0) {
            float *o = s->unstable + (size_t)n * SURFEL_F;
            memcpy(o, d, SURFEL_F * 4);
            o[5] = 0.0f;
            n++;
        }
    }
#endif
    s->unstable_count = n;
    /* A13: the reference does not check; the oracle reports the overflow instead of
     * reproducing undefined GL state. */
    if ((uint64_t)s->offset + n > max_vertices(s)) return SMO_E_CAPACITY;
    ensure_model(s, s->offset + n);
    memcpy(s->model + (size_t)s->offset * SURFEL_F, s->unstable, (size_t)n * SURFEL_F * 4);
    s->count = s->offset + n;                    /* src/GlobalModel.cpp:629 */
    return SMO_OK;
}

/* ------------------------------------------------------------------ re-initialisation after reset() */

/* SurfelMapping::computeFeedbackBuffers + GlobalModel::initialize (src/SurfelMapping.cpp:161-169):
 * surfel_feedback.vert:25-63 + surfel_feedback.geom:17-26 build the raw per-frame cloud (every
 * checkerboard pixel with 0 < z < maxDepth, NO neighbour test), init_unstable.vert:31-42 moves it to
 * the world frame; it becomes the whole model.  Vertex order: FeedbackBuffer's uvo, x-outer /
 * y-inner (src/FeedbackBuffer.cpp:47-54). */
/* surfel_feedback.vert:25-63 + surfel_feedback.geom:17-26 for pixel (i, j): the raw camera-frame surfel
 * (pos, 0.9 | colour, 0, time, time | normal, radius) or nothing (returns 0) */
static int raw_surfel(const smo_ctx *s, int i, int j, int time_i, float max_depth, float *o)
{
    const smo_config *c = &s->c;
    int W = c->width;
    float camz = 1.0f / c->fx, camw = 1.0f / c->fy;        /* float division here: src/FeedbackBuffer.cpp:93-96 */
    float x = s->xs_fb[i], y = s->ys_fb[j];
    int ci = i, cj = j;
    float z = s->depth_metric[(size_t)cj * W + ci];
    if (!(z > 0.0f && z < max_depth)) return 0;
    if (((int)x + (int)y) % 2 != 1) return 0;
    float vp[3] = {(x - c->cx) * z * camz, (y - c->cy) * z * camw, z};
    float xf[3], xb[3], yf[3], yb[3], dx[3], dy[3], nrm[3];
    float zr = s->depth_metric[(size_t)cj * W + s->ixp[i]], zl = s->depth_metric[(size_t)cj * W + s->ixm[i]];
    float zd = s->depth_metric[(size_t)s->iyp[j] * W + ci], zu = s->depth_metric[(size_t)s->iym[j] * W + ci];
    xf[0] = (x + 1.0f - c->cx) * zr * camz; xf[1] = (y - c->cy) * zr * camw; xf[2] = zr;
    xb[0] = (x - 1.0f - c->cx) * zl * camz; xb[1] = (y - c->cy) * zl * camw; xb[2] = zl;
    yf[0] = (x - c->cx) * zd * camz; yf[1] = (y + 1.0f - c->cy) * zd * camw; yf[2] = zd;
    yb[0] = (x - c->cx) * zu * camz; yb[1] = (y - 1.0f - c->cy) * zu * camw; yb[2] = zu;
    for (int q = 0; q < 3; ++q) { dx[q] = xb[q] - xf[q]; dy[q] = yb[q] - yf[q]; }
    cross3(dx, dy, nrm);
    normalize3(nrm);
    float radius = smo_get_radius(vp[2], nrm[2], camz, camw);
    size_t p = (size_t)cj * W + ci;
    o[0] = vp[0]; o[1] = vp[1]; o[2] = vp[2]; o[3] = 0.9f;           /* surfel_feedback.vert:96 */
    o[4] = smo_encode_color(s->rgb[p * 3], s->rgb[p * 3 + 1], s->rgb[p * 3 + 2], s->sem[p]);
    o[5] = 0.0f;
    o[6] = (float)time_i; o[7] = (float)time_i;
    o[8] = nrm[0]; o[9] = nrm[1]; o[10] = nrm[2]; o[11] = radius;
    return 1;
}

/* FeedbackBuffer::compute (src/FeedbackBuffer.cpp:85-145): the raw cloud of the current textures, vertex order x-outer /
 * y-inner (src/FeedbackBuffer.cpp:47-54).  dst may be NULL to query the count. */
int smo_raw_cloud(const smo_ctx *s, int time_i, float *dst, uint32_t cap, uint32_t *n)
{
    if (!s || !n) return SMO_E_ARG;
    int W = s->c.width, H = s->c.height;
    uint32_t cnt = 0;
    float o[SURFEL_F];
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < H; ++j)
            if (raw_surfel(s, i, j, time_i, s->c.far_clip, o)) {
                if (dst) {
                    if (cnt >= cap) return SMO_E_CAPACITY;
                    memcpy(dst + (size_t)cnt * SURFEL_F, o, SURFEL_F * 4);
                }
                cnt++;
            }
    *n = cnt;
    return SMO_OK;
}

int smo_stage_initialize(smo_ctx *s, const float *pose, int time_i, float max_depth)
{
    const smo_config *c = &s->c;
    int W = c->width, H = c->height;
    uint32_t n = 0;
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < H; ++j) {
            float r[SURFEL_F];
            if (!raw_surfel(s, i, j, time_i, max_depth, r)) continue;
            if ((uint64_t)n + 1 > max_vertices(s)) return SMO_E_CAPACITY;
            ensure_model(s, n + 1);
            float *o = s->model + (size_t)n * SURFEL_F;
            float t4[4], n3[3];
            xform(pose, r[0], r[1], r[2], t4);                         /* init_unstable.vert:33-36 */
            o[0] = t4[0]; o[1] = t4[1]; o[2] = t4[2]; o[3] = r[3];
            o[4] = r[4]; o[5] = 0.0f; o[6] = r[6]; o[7] = r[7];
            rot3(pose, r[8], r[9], r[10], n3);
            normalize3(n3);
            o[8] = n3[0]; o[9] = n3[1]; o[10] = n3[2]; o[11] = r[11];
            n++;
        }
    s->count = n;
    s->offset = 0;
    s->unstable_count = n; s->data_count = n; s->fused_count = 0; s->conflict_count = 0; s->visible_count = 0;
    return SMO_OK;
}

/* ------------------------------------------------------------------ orchestration */

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
/* stage slots: 0 preprocess (p0a..p0e + uploads), 1 processConflict, 2 updateConflict, 3 backMapping (both calls), 4 buildModelMap
 * (all calls), 5 predictIndices, 6 dataAssociate, 7 updateFuse, 8 concatenate */
#define STAGE(slot, call) do { const double t0_ = now_sec(); call; s->stage_sec[slot] += now_sec() - t0_; } while (0)


static void run_preprocess_pre(smo_ctx *s)
{
    /* metriciseDepth + filterDepth  src/SurfelMapping.cpp:136-139 */
    smo_metricise(&s->c, s->depth_raw, s->depth_metric);
    if (s->c.preprocess) {
        smo_filter_depth(&s->c, s->depth_metric, s->sem, 0.15f, s->depth_filtered);
        smo_smooth_depth(&s->c, s->depth_filtered, s->sem, s->depth_metric);
        smo_filter_depth(&s->c, s->depth_metric, s->sem, 0.1f, s->depth_filtered);
    } else {
        memcpy(s->depth_filtered, s->depth_metric, (size_t)s->P * 4);
    }
}

/* First half of SurfelMapping::processFrame (src/SurfelMapping.cpp:115-158): upload, pre-process,
 * reference-frame early-out.  Returns 1 when the fusing passes must follow, 0 when the call ends
 * here (first call), negative on error. */
int smo_begin_frame(smo_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm, const uint8_t *sem,
                    const float *pose)
{
    if (!s || !rgb || !pose) return SMO_E_ARG;
    upload_rgb(s, rgb);
    if (depth_mm) memcpy(s->depth_raw, depth_mm, (size_t)s->P * 2);
    if (sem) memcpy(s->sem, sem, (size_t)s->P);
    memcpy(s->curr_pose, pose, 64);

    run_preprocess_pre(s);

    if (!s->ref_set) {                                   /* :142-154 */
        memcpy(s->last, s->depth_filtered, (size_t)s->P * 4);
        memcpy(s->last_pose, s->curr_pose, 64);
        s->ref_set = 1;
        s->tick++;
        return 0;
    }

    if (s->c.preprocess) {                               /* removeMovings :156, :336-365 */
        float linv[16], t_c2l[16];
        smo_invert4(s->last_pose, linv);
        smo_mul4(linv, s->curr_pose, t_c2l);
        smo_remove_movings(&s->c, s->depth_filtered, s->sem, s->last, t_c2l, s->depth_metric);
    }
    if (s->tick == 0) {
        /* reachable only after reset() (src/SurfelMapping.cpp:161-169, 436-441) */
        int rc = smo_stage_initialize(s, s->curr_pose, s->tick, s->c.far_clip);
        smo_stage_build_model_map(s);
        smo_end_frame(s);
        return rc;      /* 0: this call is complete */
    }
    return 1;
}

/* Tail of SurfelMapping::processFrame (src/SurfelMapping.cpp:244-248) */
int smo_end_frame(smo_ctx *s)
{
    if (!s) return SMO_E_ARG;
    memcpy(s->last, s->depth_filtered, (size_t)s->P * 4);  /* :244 */
    memcpy(s->last_pose, s->curr_pose, 64);
    s->tick++;
    return SMO_OK;
}

/* SurfelMapping::processFrame src/SurfelMapping.cpp:115-251 */
int smo_process_frame(smo_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm,
                      const uint8_t *sem, const float *pose)
{
    int go;
    STAGE(0, go = smo_begin_frame(s, rgb, depth_mm, sem, pose));
    if (go <= 0) return go;
    float nearc = s->c.near_clip, farc = s->c.far_clip;
    int rc;
    STAGE(1, smo_stage_process_conflict(s, s->curr_pose, nearc, farc, s->c.fuse_thresh, 0)); /* :178 */
    STAGE(2, smo_stage_update_conflict(s));                                                   /* :187 */
    STAGE(3, smo_stage_back_mapping(s));                                                      /* :189 */
    STAGE(4, smo_stage_build_model_map(s));                                                   /* :193 */
    STAGE(5, smo_stage_predict_indices(s, s->curr_pose, s->tick, farc, s->c.time_delta));     /* :197 */
    STAGE(6, smo_stage_data_associate(s, s->curr_pose, s->tick, nearc, farc));                /* :212 */
    STAGE(7, smo_stage_update_fuse(s));                                                       /* :227 */
    STAGE(3, smo_stage_back_mapping(s));                                                      /* :229 */
    STAGE(8, rc = smo_stage_concatenate(s));                                                  /* :234 */
    STAGE(4, smo_stage_build_model_map(s));                                                   /* :239 */
    STAGE(0, smo_end_frame(s));
    return rc;
}

/* SurfelMapping::cleanPoints src/SurfelMapping.cpp:496-532 */
int smo_clean_points(smo_ctx *s, const uint16_t *depth_mm, const uint8_t *sem, const float *pose)
{
    if (!s || !depth_mm || !sem || !pose) return SMO_E_ARG;
    memcpy(s->depth_raw, depth_mm, (size_t)s->P * 2);
    memcpy(s->sem, sem, (size_t)s->P);
    memcpy(s->curr_pose, pose, 64);
    smo_metricise(&s->c, s->depth_raw, s->depth_metric);
    smo_stage_process_conflict(s, pose, s->c.near_clip, s->c.far_clip - 15.0f, 0.1f, 1);
    smo_stage_update_conflict(s);
    smo_stage_back_mapping(s);
    smo_stage_build_model_map(s);
    return SMO_OK;
}

/* SurfelMapping::reset src/SurfelMapping.cpp:436-441 ; GlobalModel::resetBuffer :760-770 */
int smo_reset(smo_ctx *s)
{
    if (!s) return SMO_E_ARG;
    s->count = s->offset = s->data_count = s->conflict_count = s->unstable_count = 0;
    s->fused_count = s->visible_count = 0;
    s->tick = 0;
    return SMO_OK;
}

/* wall seconds per pass since the last call with reset != 0 (9 slots: preprocess, processConflict, updateConflict, backMapping,
 * buildModelMap, predictIndices, dataAssociate, updateFuse, concatenate) */
int smo_stage_seconds(smo_ctx *s, double *out9, int reset)
{
    if (!s) return SMO_E_ARG;
    if (out9) memcpy(out9, s->stage_sec, 9 * sizeof(double));
    if (reset) memset(s->stage_sec, 0, sizeof s->stage_sec);
    return SMO_OK;
}

int smo_get_counts(const smo_ctx *s, smo_counts *o)
{
    if (!s || !o) return SMO_E_ARG;
    o->count = s->count; o->offset = s->offset; o->data_count = s->data_count;
    o->conflict_count = s->conflict_count; o->unstable_count = s->unstable_count;
    o->fused_count = s->fused_count; o->visible_count = s->visible_count; o->tick = s->tick;
    return SMO_OK;
}

int smo_download_model(const smo_ctx *s, float *dst, uint32_t cap, uint32_t *n)
{
    if (!s || !n) return SMO_E_ARG;
    *n = s->count;
    if (!dst) return SMO_OK;
    if (cap < s->count) return SMO_E_CAPACITY;
    memcpy(dst, s->model, (size_t)s->count * SURFEL_F * 4);
    return SMO_OK;
}

int smo_upload_model(smo_ctx *s, const float *src, uint32_t n)
{
    if (!s || (!src && n)) return SMO_E_ARG;
    if (n > max_vertices(s)) return SMO_E_CAPACITY;
    ensure_model(s, n);
    memcpy(s->model, src, (size_t)n * SURFEL_F * 4);
    s->count = s->offset = n;
    smo_stage_build_model_map(s);
    return SMO_OK;
}

int smo_download_index_map(const smo_ctx *s, int32_t *id, float *vc, float *ct, float *nr)
{
    if (!s) return SMO_E_ARG;
    size_t P = (size_t)s->P;
    if (id) memcpy(id, s->idx, P * 4);
    if (vc) memcpy(vc, s->ivc, P * 16);
    if (ct) memcpy(ct, s->ict, P * 16);
    if (nr) memcpy(nr, s->inr, P * 16);
    return SMO_OK;
}

int smo_download_depth(const smo_ctx *s, int which, float *dst)
{
    if (!s || !dst) return SMO_E_ARG;
    const float *src = which == SMO_TEX_DEPTH_METRIC ? s->depth_metric
                     : which == SMO_TEX_DEPTH_FILTERED ? s->depth_filtered
                     : which == SMO_TEX_LAST ? s->last : NULL;
    if (!src) return SMO_E_ARG;
    memcpy(dst, src, (size_t)s->P * 4);
    return SMO_OK;
}

int smo_download_data(const smo_ctx *s, float *dst, uint32_t cap, uint32_t *n)
{
    if (!s || !n) return SMO_E_ARG;
    *n = s->data_count;
    if (!dst) return SMO_OK;
    if (cap < s->data_count) return SMO_E_CAPACITY;
    memcpy(dst, s->data, (size_t)s->data_count * SURFEL_F * 4);
    return SMO_OK;
}

int smo_set_frame(smo_ctx *s, const uint8_t *rgb, const float *depth_metric, const uint8_t *sem)
{
    if (!s) return SMO_E_ARG;
    if (rgb) upload_rgb(s, rgb);
    if (depth_metric) memcpy(s->depth_metric, depth_metric, (size_t)s->P * 4);
    if (sem) memcpy(s->sem, sem, (size_t)s->P);
    return SMO_OK;
}

int smo_set_tick(smo_ctx *s, int32_t tick)
{
    if (!s) return SMO_E_ARG;
    s->tick = tick;
    s->ref_set = 1;
    return SMO_OK;
}

/* ---- helpers for the multi-GPU shard tests (tests/test_sharded.py): one oracle instance plays
 * one rank; the test moves the exempt id, exchanges the index map and filters dataVbo ---- */
int smo_set_exempt_id(smo_ctx *s, int32_t id)
{
    if (!s) return SMO_E_ARG;
    s->exempt_id = id;
    return SMO_OK;
}

/* A rig slice's share of the union's W*H conflict records (src/GlobalModel.cpp:54-57: the buffer is filled in the surfel
 * order of the single GlobalModel, i.e. slice after slice): the next conflict passes record at most `limit` conflicts.
 * limit < 0 restores the config's rule. */
int smo_set_conflict_limit(smo_ctx *s, int64_t limit)
{
    if (!s) return SMO_E_ARG;
    s->conflict_limit = limit;
    return SMO_OK;
}

/* The conflict test of SurfelMapping::cleanPoints (src/SurfelMapping.cpp:496-532: p0a + processConflict with the clean-mode
 * parameters) WITHOUT the record limit and without applying anything (no updateConflict / backMapping): how many surfels of
 * this model the view contradicts. */
int smo_count_clean_conflicts(smo_ctx *s, const uint16_t *depth_mm, const uint8_t *sem, const float *pose, uint32_t *n)
{
    if (!s || !depth_mm || !sem || !pose || !n) return SMO_E_ARG;
    memcpy(s->depth_raw, depth_mm, (size_t)s->P * 2);
    memcpy(s->sem, sem, (size_t)s->P);
    smo_metricise(&s->c, s->depth_raw, s->depth_metric);
    const int64_t keep = s->conflict_limit;
    s->conflict_limit = (int64_t)s->count;
    smo_stage_process_conflict(s, pose, s->c.near_clip, s->c.far_clip - 15.0f, 0.1f, 1);
    s->conflict_limit = keep;
    *n = s->conflict_count;
    return SMO_OK;
}

int smo_download_zbuf(const smo_ctx *s, uint32_t *dst)
{
    if (!s || !dst) return SMO_E_ARG;
    memcpy(dst, s->zbuf, (size_t)s->P * 4);
    return SMO_OK;
}

/* replace the ids / "has a projection" state of the index map (attributes stay as splatted) */
int smo_upload_index_ids(smo_ctx *s, const int32_t *idx, const uint8_t *has)
{
    if (!s || !idx || !has) return SMO_E_ARG;
    for (int p = 0; p < s->P; ++p) {
        s->idx[p] = idx[p];
        s->zbuf[p] = has[p] ? 0u : 16777215u;
    }
    return SMO_OK;
}

int smo_download_data_pixels(const smo_ctx *s, int32_t *dst, uint32_t cap, uint32_t *n)
{
    if (!s || !n) return SMO_E_ARG;
    *n = s->data_count;
    if (!dst) return SMO_OK;
    if (cap < s->data_count) return SMO_E_CAPACITY;
    memcpy(dst, s->data_pix, (size_t)s->data_count * 4);
    return SMO_OK;
}

/* keep only the dataVbo records whose flag is non-zero (order preserved) */
int smo_filter_data(smo_ctx *s, const uint8_t *keep)
{
    if (!s || (!keep && s->data_count)) return SMO_E_ARG;
    uint32_t n = 0, nf = 0;
    for (uint32_t q = 0; q < s->data_count; ++q) {
        if (!keep[q]) continue;
        if (n != q) {
            memmove(s->data + (size_t)n * SURFEL_F, s->data + (size_t)q * SURFEL_F, SURFEL_F * 4);
            s->data_pix[n] = s->data_pix[q];
        }
        if ((int32_t)f2u(s->data[(size_t)n * SURFEL_F + 5]) >= 0) nf++;
        n++;
    }
    s->data_count = n;
    s->fused_count = nf;
    return SMO_OK;
}

/* ------------------------------------------------------------------ novel-view renderer */

typedef struct { int64_t X, Y; float zw, tx, ty; } rv_t;

static inline int64_t edge64(const rv_t *a, const rv_t *b, int64_t px, int64_t py)
{
    return (b->X - a->X) * (py - a->Y) - (b->Y - a->Y) * (px - a->X);
}

static inline int top_left(const rv_t *a, const rv_t *b)
{
    int64_t dx = b->X - a->X, dy = b->Y - a->Y;
    return (dy == 0 && dx > 0) || (dy < 0);
}

static void raster_tri(const rv_t *v0, const rv_t *v1, const rv_t *v2, int w, int h, uint32_t id, uint64_t *key)
{
    int64_t area = edge64(v0, v1, v2->X, v2->Y);
    if (area == 0) return;
    if (area < 0) { const rv_t *t = v1; v1 = v2; v2 = t; area = -area; }
    int64_t minX = v0->X, maxX = v0->X, minY = v0->Y, maxY = v0->Y;
    const rv_t *vs[2] = {v1, v2};
    for (int q = 0; q < 2; ++q) {
        if (vs[q]->X < minX) minX = vs[q]->X;
        if (vs[q]->X > maxX) maxX = vs[q]->X;
        if (vs[q]->Y < minY) minY = vs[q]->Y;
        if (vs[q]->Y > maxY) maxY = vs[q]->Y;
    }
    int64_t x0 = (minX - 128) >> 8, x1 = (maxX - 128) >> 8, y0 = (minY - 128) >> 8, y1 = (maxY - 128) >> 8;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > w - 1) x1 = w - 1;
    if (y1 > h - 1) y1 = h - 1;
    const int b0 = top_left(v1, v2) ? 0 : -1, b1 = top_left(v2, v0) ? 0 : -1, b2 = top_left(v0, v1) ? 0 : -1;
    for (int64_t py = y0; py <= y1; ++py)
        for (int64_t px = x0; px <= x1; ++px) {
            const int64_t cx = px * 256 + 128, cy = py * 256 + 128;
            const int64_t e0 = edge64(v1, v2, cx, cy), e1 = edge64(v2, v0, cx, cy), e2 = edge64(v0, v1, cx, cy);
            if (e0 + b0 < 0 || e1 + b1 < 0 || e2 + b2 < 0) continue;
            const float l0 = (float)((double)e0 / (double)area), l1 = (float)((double)e1 / (double)area),
                        l2 = (float)((double)e2 / (double)area);
            const float tx = (l0 * v0->tx + l1 * v1->tx) + l2 * v2->tx;
            const float ty = (l0 * v0->ty + l1 * v1->ty) + l2 * v2->ty;
            if (tx * tx + ty * ty > 1.0f) continue;                         /* draw_image.frag:13-14 */
            const float zw = (l0 * v0->zw + l1 * v1->zw) + l2 * v2->zw;
            if (!(zw >= 0.0f && zw <= 1.0f)) continue;                      /* depth clip */
            const uint32_t d24 = (uint32_t)floor((double)zw * 16777215.0 + 0.5);
            if (d24 >= 16777215u) continue;                                 /* GL_LESS against the clear value */
            const uint64_t kk = ((uint64_t)d24 << 32) | id;
            uint64_t *dst = key + (size_t)py * w + px;
            if (kk < *dst) *dst = kk;
        }
}

int smo_render_image(const smo_ctx *s, const float *view, int w, int h, float fx, float fy, float cx, float cy,
                     uint8_t *bgr, uint8_t *sem)
{
    if (!s || !view || w <= 0 || h <= 0 || !bgr || !sem) return SMO_E_ARG;
    const float maxDepth = 200.0f;                                          /* src/GlobalModel.cpp:797 */
    const float cols = (float)w, rows = (float)h;
    float t_inv[16];
    smo_invert4(view, t_inv);
    uint64_t *key = malloc((size_t)w * h * 8);
    if (!key) return SMO_E_ARG;
    for (size_t p = 0; p < (size_t)w * h; ++p) key[p] = 0x7FFFFFFFFFFFFFFFull;
    for (uint32_t k = 0; k < s->count; ++k) {
        const float *v = s->model + (size_t)k * SURFEL_F;
        float ph[4], n[3];
        xform(t_inv, v[0], v[1], v[2], ph);                                /* draw_image.vert:20-27 */
        rot3(t_inv, v[8], v[9], v[10], n);
        normalize3(n);
        const float r = v[11];
        if (ph[2] >= maxDepth || ph[2] <= 1.0f) continue;                   /* draw_image_adaptive.geom:41 */
        float x[3], y[3];
        if (ph[2] > 5.0f) {                                                 /* :47-52 */
            const float tn[3] = {0.0f, 0.0f, 1.0f};
            float a[3] = {tn[1] - tn[2], -tn[0], tn[0]};
            normalize3(a);
            for (int q = 0; q < 3; ++q) x[q] = a[q] * r * 1.41421356f;
            cross3(tn, x, y);
        } else {                                                            /* :53-63 */
            const float cosAngle = dot3(ph, n) / (sqrtf(dot3(ph, ph)) * sqrtf(dot3(n, n)));
            const float radius = r / (1.0f + 0.5f * fabsf(cosAngle));
            float a[3] = {n[1] - n[2], -n[0], n[0]};
            normalize3(a);
            for (int q = 0; q < 3; ++q) x[q] = a[q] * radius * 1.41421356f;
            cross3(n, x, y);
        }
        const float sx[4] = {x[0], y[0], -y[0], -x[0]}, sy[4] = {x[1], y[1], -y[1], -x[1]}, sz[4] = {x[2], y[2], -y[2], -x[2]};
        const float tcx[4] = {-1.0f, 1.0f, -1.0f, 1.0f}, tcy[4] = {-1.0f, -1.0f, 1.0f, 1.0f};
        rv_t rv[4];
        int ok = 1;
        for (int q = 0; q < 4 && ok; ++q) {
            const float X = ph[0] + sx[q], Y = ph[1] + sy[q], Z = ph[2] + sz[q];
            if (!(Z > 0.0f)) { ok = 0; break; }                             /* would need polygon clipping: not drawn */
            const float xn = ((((fx * X) / Z) + cx) - (cols * 0.5f)) / (cols * 0.5f);   /* projectPoint :31-36 */
            const float yn = ((((fy * Y) / Z) + cy) - (rows * 0.5f)) / (rows * 0.5f);
            const float zn = (2.0f * Z / maxDepth) - 1.0f;
            const float xw = (cols * 0.5f) * xn + (cols * 0.5f), yw = (rows * 0.5f) * yn + (rows * 0.5f);
            if (!(fabsf(xw) < 1.0e6f && fabsf(yw) < 1.0e6f)) { ok = 0; break; }
            rv[q].X = (int64_t)floor((double)xw * 256.0 + 0.5);
            rv[q].Y = (int64_t)floor((double)yw * 256.0 + 0.5);
            rv[q].zw = 0.5f * zn + 0.5f;
            rv[q].tx = tcx[q]; rv[q].ty = tcy[q];
        }
        if (!ok) continue;
        raster_tri(&rv[0], &rv[1], &rv[2], w, h, k, key);                   /* triangle strip */
        raster_tri(&rv[2], &rv[1], &rv[3], w, h, k, key);
    }
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        if (key[p] == 0x7FFFFFFFFFFFFFFFull) { bgr[p * 3] = bgr[p * 3 + 1] = bgr[p * 3 + 2] = 0; sem[p] = 0; continue; }
        const uint32_t id = (uint32_t)(key[p] & 0xFFFFFFFFu);
        const uint32_t sc = f2u(s->model[(size_t)id * SURFEL_F + 4]);
        bgr[p * 3 + 0] = (uint8_t)(sc & 0xFFu);                             /* vBGR = srgb.wzy :38 */
        bgr[p * 3 + 1] = (uint8_t)((sc >> 8) & 0xFFu);
        bgr[p * 3 + 2] = (uint8_t)((sc >> 16) & 0xFFu);
        sem[p] = (uint8_t)(((sc >> 24) & 0xFFu) + 1u);                      /* :39 */
    }
    free(key);
    return SMO_OK;
}
