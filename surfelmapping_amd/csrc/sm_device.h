// sm_device.h -- device-side data layout and scalar helpers of the gfx950 fusion core.
//
// Numeric contract (DESIGN.md "Arithmetic"): IEEE fp32, round-to-nearest-even, no FMA
// contraction (-ffp-contract=off), correctly rounded / and sqrt, fixed evaluation order:
//   mat*vec   r_i = ((m_i0*x + m_i1*y) + m_i2*z) + m_i3
//   dot       (a.x*b.x + a.y*b.y) + a.z*b.z
//   normalize v / sqrt(dot(v,v))
//   min(a,b)  (b < a) ? b : a
// Shader citations are file:line under /root/reference/src/Shaders.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sm {

constexpr uint64_t KEY_EMPTY = 0x7FFFFFFFFFFFFFFFull;  // positive as int64 too (RCCL min on either type)
constexpr int TILE = 1024;                             // surfels per cull tile (16 ballot words)
constexpr int TILE_WORDS = TILE / 64;
constexpr int PIX_BLOCK = 256;                         // pixels per association block (4 ballot words)
// Sub-counters (64 per sum, <= 32 adders each, read with one load per lane): one 128-byte line each -- memory-side atomics
// on one line serialise (64 counters packed into two lines cost the association 3 us at KITTI size)
constexpr int SUB_STRIDE = 32;                         // words between two sub-counters
constexpr int SUB_SET = 64 * SUB_STRIDE;               // words of one set of 64 sub-counters

// One SoA surfel set: 44 B / surfel (the reference's AoS slot [5] "standby" is always 0
// in the stored model: back_map.geom:23, unstable.vert:32).
struct SurfelSet {
    float4 *pos_conf;    // xyz world, confidence
    float4 *norm_rad;    // unit normal world, radius
    uint32_t *color;     // sem<<24 | r<<16 | g<<8 | b    (color.glsl:19-26)
    float *init_time;
    float *time;
};

struct Model {
    SurfelSet s[2];      // ping-pong for the stable compaction
};

// Device-resident frame state: every kernel reads its sizes from here so that a frame needs
// no host read-back (the reference blocks on 6 glGetQueryObjectuiv per frame).
struct DevState {
    uint32_t count;           // occupied slots [0, count): live surfels + `garbage` dead ones (deferred compaction)
    uint32_t offset;          // slots occupied after the last cull (where the frame's new surfels are appended)
    uint32_t cur;             // which SurfelSet holds the model
    uint32_t cull_n;          // count before the pending cull
    uint32_t cull_src, cull_dst;
    uint32_t n_kill;
    uint32_t conflict_count;
    uint32_t data_count, unstable_count, fused_count;
    uint32_t visible_count;
    uint32_t append_n;        // new surfels the append kernel may write
    int32_t error;            // sticky SM_E_*
    uint32_t frames_logged;   // fusing frames completed (frame-log write index)
    uint32_t n_static;        // surfels in tiles the last cull left in place
    uint32_t n_conf_skipped;  // surfels in tiles the conflict pass skipped by their bounds
    uint32_t n_splat_skipped; // surfels in static tiles the splat skipped by their bounds
    // ---- deferred compaction (DESIGN.md "Deferred compaction") ----
    uint32_t garbage;         // dead slots below `offset`; live surfels = count - garbage
    uint32_t garbage_prev;    // its value before the last cull
    uint32_t do_compact;      // the last cull compacts physically (otherwise it only marks the dead in `alive`)
    uint32_t cap_binds;       // the last cull had more conflicts than the conflict cap (tile_allow is in force)
    uint32_t first_moving;    // first tile the pending compaction moves or thins out (tiles below it stay in place)
    uint32_t compact_ticket;  // work queue of k_compact: next moving tile (relative to first_moving) to hand out
    uint32_t stat_frames;     // frames whose append has completed (tag of the host-visible slot statistic)
    uint32_t first_live;      // slot of the first live surfel = the reference's surfel id 0 (conflict.geom:15, data.vert:142)
    uint32_t fl_dirty2[2];    // k_surfel_pass saw that surfel die (word of the frame's parity, FrameParams::par): the publisher looks for its successor
    uint32_t slow_frames;     // diagnostic: frames whose merged publisher found work the association had to wait for (sm_debug_slow_frames)
    uint32_t slow_done[2];    // two-launch frame, rare path (the conflict cap binds / "id 0" died): publisher + repair crew workgroups that are through (by frame parity)
    // ---- direct append (k_associate_direct): the frame's statistics are completed one kernel later
    uint32_t pend;            // 1: the last frame's new / fused counts, dead-slot total and log entry are still to be completed
                              //    (by the next frame's k_pass_fixup, or by k_frame_finalize before anything else reads them)
    uint32_t pend_tick;       // time stamp of that frame
    uint32_t holes_last;      // slots of the last frame's appended range that stayed empty (candidate pixels that fused instead)
};

struct FrameLog { uint32_t tick, n_before, n_after_cull, n_kill, conflict_count, visible_count, fused_count, unstable_count, n_static, n_conf_skipped, n_splat_skipped, n_slots; };
constexpr uint32_t FRAME_LOG_LEN = 1024;

struct FrameParams {
    float pose[16];           // camera -> world, column-major
    float t_inv[16];          // world -> camera
    float fx, fy, cx, cy;
    float inv_fx, inv_fy;     // float(1.0/fx) : src/GlobalModel.cpp:273-276
    float cols, rows;
    int W, H, P;
    float min_depth, max_depth;
    float conflict_thresh;    // processConflict fuseThresh (0.0; 0.1 in clean mode)
    float fuse_thresh;        // dataAssociate fuseThresh
    float stereo_border;
    int is_clean;
    int time;
    int time_delta;
    float depth_cutoff;
    uint32_t conflict_cap;    // W*H or 0xFFFFFFFF
    int splat_follows;        // the cull is followed by the index-map splat (resets visible_count)
    int log_frame;            // append a FrameLog entry at the end of the frame
    uint32_t max_vertices;
    // ---- re-initialisation after reset(): raw feedback cloud (surfel_feedback.vert) ----
    int init_mode;            // 1: every checkerboard pixel with 0 < z < far becomes a surfel, no association
    float inv_fx_fb, inv_fy_fb;   // 1.0f/fx as float division (src/FeedbackBuffer.cpp:93-96)
    int use_bounds;           // 1: whole 1024-surfel tiles are skipped when their bounding box is outside the view
    // ---- deferred compaction ----
    uint32_t compact_now;     // 1: this cull moves the survivors (k_scan_cull + k_compact); 0: it only marks the dead (k_cull_lazy).
                              // Decided by the host (a fixed period + a capacity bound), so that it can launch the matching kernels
    int maintenance;          // 1: compaction outside a frame (no kills): frame statistics are left alone
    int compact_tickets;      // 1: k_compact hands its moving tiles out in order from a ticket counter (no co-residency needed)
    int no_exempt;            // 1: no surfel is exempt from the conflict test (a rig slice that does not hold the global surfel 0)
    int shard_slots;          // 1: slot-addressed sharding of one stream: ids are global slot numbers on every rank
    int par;                  // frame parity (0 / 1): which of the doubled per-frame words of DevState this frame uses
};

__device__ __forceinline__ float min_glsl(float a, float b) { return (b < a) ? b : a; }

// nearest + clamp-to-edge texel
__device__ __forceinline__ int tex_idx(float t, int n)
{
    float f = floorf(t * (float)n);
    if (!(f >= 0.0f)) return 0;
    if (f > (float)(n - 1)) return n - 1;
    return (int)f;
}

__device__ __forceinline__ float3 xform3(const float *m, float x, float y, float z)
{
    float3 r;
    r.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12];
    r.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13];
    r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
    return r;
}

__device__ __forceinline__ float3 rot3(const float *m, float x, float y, float z)
{
    float3 r;
    r.x = (m[0] * x + m[4] * y) + m[8] * z;
    r.y = (m[1] * x + m[5] * y) + m[9] * z;
    r.z = (m[2] * x + m[6] * y) + m[10] * z;
    return r;
}

__device__ __forceinline__ float dot3(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

__device__ __forceinline__ float3 cross3(float3 a, float3 b)
{
    float3 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}

__device__ __forceinline__ float3 normalize3(float3 v)
{
    float l = sqrtf(dot3(v, v));
    float3 r;
    r.x = v.x / l; r.y = v.y / l; r.z = v.z / l;
    return r;
}

// acos by a fixed rational kernel (DESIGN.md "Arithmetic"); NaN outside [-1,1]
// (data.vert:54-57: NaN < 0.5 is false -> no association).
__device__ __forceinline__ float acos_spec(float x)
{
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

__device__ __forceinline__ uint32_t round_u8(float v)
{
    float r = roundf(v);
    if (!(r >= 0.0f)) return 0u;
    if (r > 4294967040.0f) return 4294967040u;
    return (uint32_t)r;
}

// color.glsl:19-26
__device__ __forceinline__ uint32_t encode_color(float r, float g, float b, uint32_t sem)
{
    uint32_t srgb = sem;
    srgb = (srgb << 8) + round_u8(r * 255.0f);
    srgb = (srgb << 8) + round_u8(g * 255.0f);
    srgb = (srgb << 8) + round_u8(b * 255.0f);
    return srgb;
}

// surfels.glsl:19-32
__device__ __forceinline__ float get_radius(float depth, float norm_z, float inv_fx, float inv_fy)
{
    float meanFocal = ((1.0f / fabsf(inv_fx)) + (1.0f / fabsf(inv_fy))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius / fabsf(norm_z);
    return min_glsl(2.0f * radius, radius_n);
}

// exclusive scan of one value per thread over a 1024-thread block; lds needs 17 words
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *total, uint32_t *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < 16 ? lds[lane] : 0u;
        uint32_t wi = w;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            uint32_t t = __shfl_up(wi, o);
            if (lane >= o) wi += t;
        }
        if (lane < 16) lds[lane] = wi - w;
        if (lane == 15) lds[16] = wi;
    }
    __syncthreads();
    uint32_t excl = inc - v + lds[wave];
    *total = lds[16];
    __syncthreads();
    return excl;
}

// ---------------------------------------------------------------------------------------------
// Per-tile bounds (work skipping).  Tile t = surfels [1024 t, 1024 t + 1024).  Because the model is
// kept in creation order, a tile holds surfels born in the same few image columns of one frame, so
// its world-space box is small and whole tiles fall out of view once the camera has moved on.
// tb[t*8 + 0..2] = min xyz, [3] = number of surfels that defeat the box (conf <= 0 / NaN: only an
// uploaded model can contain them), [4..6] = max xyz, [7] = max last-update time; floats are stored
// in an order-preserving unsigned encoding so that atomicMin/atomicMax maintain them.  Bounds are
// conservative (they only ever grow between rebuilds), skipping decisions add a 2-pixel / 1-cm
// margin, so a skipped tile provably contains no surfel the exact per-surfel test would accept.
// ---------------------------------------------------------------------------------------------
// Encoding: every word is maintained with atomicMax only, so that one wave instruction (7 lanes, one word each)
// updates a tile: [0..2] hold max(~ord(x)) = "negated minimum", [4..6] max(ord(x)), [7] max(ord(time));
// an empty tile is all zeros (a memset).
__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

// value of lane `l` (wave-uniform index) for every lane: v_readlane instead of a ds_bpermute round trip
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int l)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(l));
}

// Wave-wide reductions with DPP (gfx9 row_shr / row_bcast): six VALU instructions and no LDS traffic, against six
// ds_bpermute round trips for a __shfl_xor butterfly.  All 64 lanes must be active; lanes that do not take part pass the
// identity (0 for both max and sum of unsigned).  The result is returned wave-uniform (read from lane 63).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_take(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);   // out-of-range source lanes read 0
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, dpp_take<0x111, 0xF>(v));      // row_shr:1
    v = max(v, dpp_take<0x112, 0xF>(v));      // row_shr:2
    v = max(v, dpp_take<0x114, 0xF>(v));      // row_shr:4
    v = max(v, dpp_take<0x118, 0xF>(v));      // row_shr:8   -> lane 15 of every row holds the row's maximum
    v = max(v, dpp_take<0x142, 0xA>(v));      // row_bcast:15 into rows 1 and 3
    v = max(v, dpp_take<0x143, 0xC>(v));      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += dpp_take<0x111, 0xF>(v);
    v += dpp_take<0x112, 0xF>(v);
    v += dpp_take<0x114, 0xF>(v);
    v += dpp_take<0x118, 0xF>(v);
    v += dpp_take<0x142, 0xA>(v);
    v += dpp_take<0x143, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- two-launch frame (sm_k_pass.h, "Two homes"): the rare path's test and wait
// does the frame need its publisher / repair before anybody may use the pass's results?  (wave-uniform; one load per lane)
__device__ __forceinline__ bool slow_frame(const DevState *__restrict__ st, const uint32_t *__restrict__ conf_sub, uint32_t cap, int par, int lane)
{
    const uint32_t ctotal = wave_sum_u32(conf_sub[lane * SUB_STRIDE]);
    return ctotal > cap || st->fl_dirty2[par] != 0u;
}

// ... and the wait of that rare path: publisher + crew are the first workgroups of the launch (dispatched first, waiting for
// nobody), so this terminates whatever is resident; bounded all the same (SM_E_STALL instead of a hang)
__device__ __forceinline__ void wait_slow_frame(DevState *__restrict__ st, int par, uint32_t need)
{
    uint32_t spins = 0;
    while (__hip_atomic_load(&st->slow_done[par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) { st->error = -6; break; }
    }
}


// expand the bounds of the tiles touched by this wave: each active lane contributes one surfel
// (position, time, "bad") to tile `tile`; lanes are grouped by tile with ballots, reduced with
// shuffles, and ONE atomicMax wave instruction (lanes 0..7 -> words 0..7 of the tile) publishes the group.
__device__ __forceinline__ void bounds_expand_wave(uint32_t *__restrict__ tb, bool active, uint32_t tile, float x, float y,
                                                   float z, float t, bool bad)
{
    uint64_t todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t tcur = (uint32_t)__shfl((int)tile, leader);
        const bool mine = active && tile == tcur;
        const uint64_t grp = __ballot(mine);
        uint32_t w0 = mine ? ~f2ord(x) : 0u, w1 = mine ? ~f2ord(y) : 0u, w2 = mine ? ~f2ord(z) : 0u;
        uint32_t w4 = mine ? f2ord(x) : 0u, w5 = mine ? f2ord(y) : 0u, w6 = mine ? f2ord(z) : 0u, w7 = mine ? f2ord(t) : 0u;
        const uint32_t nb = (uint32_t)__popcll(__ballot(mine && (bad || x != x || y != y || z != z)));
        w0 = wave_max_u32(w0); w1 = wave_max_u32(w1); w2 = wave_max_u32(w2); w4 = wave_max_u32(w4);
        w5 = wave_max_u32(w5); w6 = wave_max_u32(w6); w7 = wave_max_u32(w7);
        if (lane < 8 && lane != 3) {
            const uint32_t val = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : lane == 4 ? w4 : lane == 5 ? w5 : lane == 6 ? w6 : w7;
            atomicMax(&tb[(size_t)tcur * 8 + lane], val);
        }
        if (nb && lane == 3) atomicAdd(&tb[(size_t)tcur * 8 + 3], nb);
        todo &= ~grp;
    }
}

// true if no point of the tile's box can pass the view test  zlo < Z < zhi, ulo <= u <= uhi, vlo <= v <= vhi
// (conflict.vert:35 / index_map.vert:45-55), with a 1 cm / 2 pixel safety margin.  An empty, non-finite or
// "bad" box is never reported outside unless it is empty.
__device__ __forceinline__ bool box_outside_view(const uint32_t *__restrict__ b, const float *t_inv, float fx, float fy, float cx,
                                                 float cy, float zlo, float zhi, float ulo, float uhi, float vlo, float vhi)
{
    if (b[3] != 0u) return false;
    if (b[0] == 0u && b[4] == 0u) return true;                               // no surfel recorded at all
    const float lx = ord2f(~b[0]), ly = ord2f(~b[1]), lz = ord2f(~b[2]), hx = ord2f(b[4]), hy = ord2f(b[5]), hz = ord2f(b[6]);
    float zmin = 3.0e38f, zmax = -3.0e38f;
    bool all_right = true, all_left = true, all_below = true, all_above = true, finite = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float3 p = xform3(t_inv, (c & 1) ? hx : lx, (c & 2) ? hy : ly, (c & 4) ? hz : lz);
        finite = finite && (p.x - p.x == 0.0f) && (p.y - p.y == 0.0f) && (p.z - p.z == 0.0f);
        zmin = fminf(zmin, p.z); zmax = fmaxf(zmax, p.z);
        all_right = all_right && (fx * p.x + (cx - uhi - 2.0f) * p.z > 0.0f);
        all_left = all_left && (fx * p.x + (cx - ulo + 2.0f) * p.z < 0.0f);
        all_below = all_below && (fy * p.y + (cy - vhi - 2.0f) * p.z > 0.0f);
        all_above = all_above && (fy * p.y + (cy - vlo + 2.0f) * p.z < 0.0f);
    }
    if (!finite) return false;
    if (zmax < zlo - 0.01f || zmin > zhi + 0.01f) return true;
    if (zmin > 1.0e-3f && (all_right || all_left || all_below || all_above)) return true;
    return false;
}

// the first `n` set bits of `m` (n may exceed popcount)
__device__ __forceinline__ uint64_t first_n_bits(uint64_t m, uint32_t n)
{
    if (n >= (uint32_t)__popcll(m)) return m;
    uint64_t rest = m;
    for (uint32_t i = 0; i < n; ++i) rest &= rest - 1;   // clear the n lowest set bits
    return m ^ rest;
}

}  // namespace sm
