// sm_kernels.h -- hand-written gfx950 kernels of the per-frame fusion hot path.
// Included once by sm_api.hip.  Shader citations: /root/reference/src/Shaders/<file>:<line>.
//
// Frame images are held COLUMN-MAJOR (q = i*H + j, "x-outer / y-inner"): that is the order
// in which the reference submits pixels to data.vert (src/GlobalModel.cpp:67-74) and hence the
// order in which new surfels are appended, so association + ordered compaction become a 1-D
// coalesced stream over q.
#pragma once

#include "sm_device.h"

namespace sm {

// The side planes of the two view volumes go through the camera centre, so "all 8 corners of the box are outside plane X" is a
// statement about a linear form g(p) = c1 p.x + c2 p.z (or p.y, p.z): positive at every corner => positive on the whole box
// => every point of it with z > 0 projects outside that image edge (by the 2-pixel margin built into c2) -- whether or not
// part of the box is BEHIND the camera.  (Round 1 applied these tests only to boxes entirely in front, zmin > 1 mm; but the
// boxes are thin slanted slabs -- a few image columns of one past frame, near ground to far facades -- that the camera
// passes for ~37 frames with their near end behind it and everything in front of it already outside the image: on a KITTI
// frame 950 tiles passed the old test, 700 pass this one, 470 hold a surfel in view.)  What a box straddling z = 0 does
// need is a guard against rounding, because there the margin (2 pixels x z) shrinks to nothing: g must clear `guard`, a
// bound on the rounding error of g at a corner (transformed coordinates carry ~4 ulp of S = the sum of the box's bounds and
// the translation; the coefficients are below C = fx + fy + cols + rows).
__device__ __forceinline__ float plane_guard(const FrameParams &fp, float lx, float ly, float lz, float hx, float hy, float hz)
{
    const float S = (fabsf(lx) + fabsf(hx)) + (fabsf(ly) + fabsf(hy)) + (fabsf(lz) + fabsf(hz)) +
                    (fabsf(fp.t_inv[12]) + fabsf(fp.t_inv[13]) + fabsf(fp.t_inv[14]));
    return 2.0e-6f * (((fp.fx + fp.fy) + fp.cols) + fp.rows) * S;
}

// ---------------------------------------------------------------------------------------------
// Per-tile skip flags of the frame (bit 0: outside the conflict view volume, conflict.vert:35; bit 1:
// cannot reach the index map, index_map.vert:45-55 incl. the timeDelta gate), from the tile bounds as they
// stand at frame start.  Evaluated inside k_conflict, 64 tiles at a time (one per lane), which keeps bit 0 in a ballot
// mask and stores bit 1 for the splat in k_compact: no extra launch, no extra pass.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tile_flags_one(uint32_t t, const FrameParams &fp, const uint32_t *__restrict__ tb)
{
    uint32_t f = 0;
    const uint32_t *b = tb + (size_t)t * 8;
    if (fp.use_bounds && b[3] == 0u) {
        if (b[0] == 0u && b[4] == 0u) {
            f = 3u;                                        // no surfel recorded at all
        } else {
            // one transform of the 8 box corners serves both view volumes (they share three of the four side planes)
            const float lx = ord2f(~b[0]), ly = ord2f(~b[1]), lz = ord2f(~b[2]), hx = ord2f(b[4]), hy = ord2f(b[5]), hz = ord2f(b[6]);
            float zmin = 3.0e38f, zmax = -3.0e38f;
            bool right = true, left_c = true, left_s = true, below = true, above = true, finite = true;
            const float gd = plane_guard(fp, lx, ly, lz, hx, hy, hz);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float3 p = xform3(fp.t_inv, (c & 1) ? hx : lx, (c & 2) ? hy : ly, (c & 4) ? hz : lz);
                finite = finite && (p.x - p.x == 0.0f) && (p.y - p.y == 0.0f) && (p.z - p.z == 0.0f);
                zmin = fminf(zmin, p.z); zmax = fmaxf(zmax, p.z);
                right = right && (fp.fx * p.x + (fp.cx - fp.cols - 2.0f) * p.z > gd);
                left_c = left_c && (fp.fx * p.x + (fp.cx - fp.stereo_border + 2.0f) * p.z < -gd);
                left_s = left_s && (fp.fx * p.x + (fp.cx + 2.0f) * p.z < -gd);
                below = below && (fp.fy * p.y + (fp.cy - fp.rows - 2.0f) * p.z > gd);
                above = above && (fp.fy * p.y + (fp.cy + 2.0f) * p.z < -gd);
            }
            if (finite) {
                // bit 0: conflict.vert:35  (min < Z < max, border <= u <= cols, 0 <= v <= rows)
                if (zmax < fp.min_depth - 0.01f || zmin > fp.max_depth + 0.01f || right || left_c || below || above)
                    f |= 1u;
                // bit 1: index_map.vert:45-55  (0 < Z < far, inside the image, updated within timeDelta frames)
                if (zmax < -0.01f || zmin > fp.depth_cutoff + 0.01f || right || left_s || below || above ||
                    (float)fp.time - ord2f(b[7]) > (float)fp.time_delta)
                    f |= 2u;
            }
        }
    }
    return f;
}

// The same flags for the next (up to) 64 tiles of a workgroup, corner-parallel: the 8 box corners of a tile go to 8
// lanes (one transform per lane instead of eight), 8 tiles per wave and pass, 32 per workgroup and pass -- a workgroup
// rarely owns more than a handful of tiles per batch, and the per-lane form above made every wave pay the full
// 8-corner evaluation for them (it was 44 % of k_conflict's VALU instructions at KITTI size).  Results go to s_flags[64]
// (entry b <-> tile first + b*stride); the caller synchronises before reading them.
__device__ __forceinline__ void tile_flags_batch(uint32_t first, uint32_t stride, uint32_t ntiles, const FrameParams &fp,
                                                 const uint32_t *__restrict__ tb, uint8_t *s_flags)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 3, c = lane & 7;                       // tile within the pass, corner
    const uint64_t nb64 = first < ntiles ? ((uint64_t)(ntiles - first) + stride - 1) / stride : 0;
    const uint32_t nb = (uint32_t)(nb64 < 64 ? nb64 : 64);        // tiles in this batch
    for (uint32_t pass = 0; pass * 32u < nb; ++pass) {            // workgroup-uniform
        const uint32_t b = pass * 32u + (uint32_t)wave * 8u + (uint32_t)j;
        const bool in = b < nb;
        const uint32_t t = in ? first + b * stride : first;
        const uint32_t *bd = tb + (size_t)t * 8;
        const uint32_t b0 = bd[0], b1 = bd[1], b2 = bd[2], b3 = bd[3], b4 = bd[4], b5 = bd[5], b6 = bd[6], b7 = bd[7];
        const float3 p = xform3(fp.t_inv, (c & 1) ? ord2f(b4) : ord2f(~b0), (c & 2) ? ord2f(b5) : ord2f(~b1),
                                (c & 4) ? ord2f(b6) : ord2f(~b2));
        const bool fin = (p.x - p.x == 0.0f) && (p.y - p.y == 0.0f) && (p.z - p.z == 0.0f);
        const float gd = plane_guard(fp, ord2f(~b0), ord2f(~b1), ord2f(~b2), ord2f(b4), ord2f(b5), ord2f(b6));
        const bool r_ = fp.fx * p.x + (fp.cx - fp.cols - 2.0f) * p.z > gd;
        const bool lc = fp.fx * p.x + (fp.cx - fp.stereo_border + 2.0f) * p.z < -gd;
        const bool ls = fp.fx * p.x + (fp.cx + 2.0f) * p.z < -gd;
        const bool be = fp.fy * p.y + (fp.cy - fp.rows - 2.0f) * p.z > gd;
        const bool ab = fp.fy * p.y + (fp.cy + 2.0f) * p.z < -gd;
        float zmin = p.z, zmax = p.z;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { zmin = fminf(zmin, __shfl_xor(zmin, o)); zmax = fmaxf(zmax, __shfl_xor(zmax, o)); }
        const int sh = j * 8;                                      // "all 8 corners" = the tile's byte of the ballot is 0xFF
        const bool finite = ((__ballot(fin) >> sh) & 0xFFull) == 0xFFull;
        const bool right = ((__ballot(r_) >> sh) & 0xFFull) == 0xFFull, left_c = ((__ballot(lc) >> sh) & 0xFFull) == 0xFFull;
        const bool left_s = ((__ballot(ls) >> sh) & 0xFFull) == 0xFFull, below = ((__ballot(be) >> sh) & 0xFFull) == 0xFFull;
        const bool above = ((__ballot(ab) >> sh) & 0xFFull) == 0xFFull;
        uint32_t f = 0;
        if (fp.use_bounds && b3 == 0u) {
            if (b0 == 0u && b4 == 0u) {
                f = 3u;                                            // no surfel recorded at all
            } else if (finite) {
                if (zmax < fp.min_depth - 0.01f || zmin > fp.max_depth + 0.01f || right || left_c || below || above) f |= 1u;
                if (zmax < -0.01f || zmin > fp.depth_cutoff + 0.01f || right || left_s || below || above ||
                    (float)fp.time - ord2f(b7) > (float)fp.time_delta)
                    f |= 2u;
            }
        }
        if (in && c == 0) s_flags[b] = (uint8_t)f;
    }
    for (uint32_t b = nb + threadIdx.x; b < 64u; b += blockDim.x) s_flags[b] = 0;     // beyond the last tile
}

// ---------------------------------------------------------------------------------------------
// The tile skip flags of a frame, evaluated by a few extra workgroups of k_prep: the pose is known when the frame's images
// are prepared and the tile bounds are final by then (the previous frame's append precedes k_prep in stream order), so the
// one-pass surfel kernel finds one byte per tile ready (loaded together with DevState) instead of opening with a round
// of bounds loads + box tests + barriers on its critical path.  Tiles the conflict test skips get their (zero) conflict
// counts here.  (A compacted list of the active tiles was tried: the returning atomic and the block scan it needs cost
// k_prep 3 us at KITTI size and 10 us at 20 M surfels, and the list's extra load per tile cost the surfel kernel more
// than the even sharing saved.)
// ---------------------------------------------------------------------------------------------
struct TilePrep {
    const DevState *st;
    const uint32_t *tb;
    uint8_t *tile_flags;
    uint4 *wave_cnt;
    uint2 *prep_part;         // [nfb] (conflict-skipped, splat-skipped) surfels
    uint32_t nfb;             // workgroups of k_prep that do this (0: none)
    // k_assoc_prep: the previous frame's association runs in the SAME launch, so its appends / fuses are not in the bounds yet
    const uint32_t *grp_cand; // candidate pixels of that frame per group (its new slot count = offset + their sum), or null
    uint32_t n_grp;
    int prev_time;            // that frame's time stamp
};

__device__ __forceinline__ void tile_prep_block(const FrameParams &fp, const TilePrep &tp, uint32_t first_block = 0u)
{
    __shared__ uint32_t s_sk[2][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 3, c = lane & 7;                       // tile within the wave's 8, box corner
    // Concurrent with the previous frame's association (k_assoc_prep): the slot count that association will publish is
    // offset + (its candidate pixels); tiles it can still change must not be skipped on stale bounds.  Those are the tiles
    // from the old end on (appends) and the tiles that frame drew into the index map (a fuse moves a surfel: its box may
    // grow) -- k_surfel_pass stamped exactly those with the frame's time, so "stamped last frame" means "do not skip".
    uint32_t N = tp.st->count, first_new_tile = 0xFFFFFFFFu;
    if (tp.grp_cand) {
        uint32_t d = 0;
        for (uint32_t g = lane; g < tp.n_grp; g += 64u) d += tp.grp_cand[g];
        const uint32_t off = tp.st->offset;
        N = off + wave_sum_u32(d);
        first_new_tile = off / (uint32_t)TILE;
    }
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const uint32_t per_wg = (blockDim.x >> 6) * 8u, blk = blockIdx.x - first_block;
    uint32_t cskip = 0, sskip = 0;                               // lane c == 0 of every tile accumulates
    for (uint32_t base = blk * per_wg; base < ntiles; base += tp.nfb * per_wg) {   // workgroup-uniform
        const uint32_t t = base + (uint32_t)wave * 8u + (uint32_t)j;
        const bool in = t < ntiles;
        const uint32_t *bd = tp.tb + (size_t)(in ? t : 0u) * 8;
        const uint32_t b0 = bd[0], b1 = bd[1], b2 = bd[2], b3 = bd[3], b4 = bd[4], b5 = bd[5], b6 = bd[6], b7 = bd[7];
        const float3 p = xform3(fp.t_inv, (c & 1) ? ord2f(b4) : ord2f(~b0), (c & 2) ? ord2f(b5) : ord2f(~b1),
                                (c & 4) ? ord2f(b6) : ord2f(~b2));
        const bool fin = (p.x - p.x == 0.0f) && (p.y - p.y == 0.0f) && (p.z - p.z == 0.0f);
        const float gd = plane_guard(fp, ord2f(~b0), ord2f(~b1), ord2f(~b2), ord2f(b4), ord2f(b5), ord2f(b6));
        const bool r_ = fp.fx * p.x + (fp.cx - fp.cols - 2.0f) * p.z > gd;
        const bool lc = fp.fx * p.x + (fp.cx - fp.stereo_border + 2.0f) * p.z < -gd;
        const bool ls = fp.fx * p.x + (fp.cx + 2.0f) * p.z < -gd;
        const bool be = fp.fy * p.y + (fp.cy - fp.rows - 2.0f) * p.z > gd;
        const bool ab = fp.fy * p.y + (fp.cy + 2.0f) * p.z < -gd;
        float zmin = p.z, zmax = p.z;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { zmin = fminf(zmin, __shfl_xor(zmin, o)); zmax = fmaxf(zmax, __shfl_xor(zmax, o)); }
        const int sh = j * 8;                                      // "all 8 corners" = the tile's byte of the ballot is 0xFF
        const bool finite = ((__ballot(fin) >> sh) & 0xFFull) == 0xFFull;
        const bool right = ((__ballot(r_) >> sh) & 0xFFull) == 0xFFull, left_c = ((__ballot(lc) >> sh) & 0xFFull) == 0xFFull;
        const bool left_s = ((__ballot(ls) >> sh) & 0xFFull) == 0xFFull, below = ((__ballot(be) >> sh) & 0xFFull) == 0xFFull;
        const bool above = ((__ballot(ab) >> sh) & 0xFFull) == 0xFFull;
        uint32_t f = 0;
        if (fp.use_bounds && b3 == 0u) {
            if (b0 == 0u && b4 == 0u) {
                f = 3u;                                            // no surfel recorded at all
            } else if (finite) {
                if (zmax < fp.min_depth - 0.01f || zmin > fp.max_depth + 0.01f || right || left_c || below || above) f |= 1u;
                if (zmax < -0.01f || zmin > fp.depth_cutoff + 0.01f || right || left_s || below || above ||
                    (float)fp.time - ord2f(b7) > (float)fp.time_delta)
                    f |= 2u;
            }
        }
        if (tp.grp_cand && (t >= first_new_tile || ((b0 | b4) != 0u && ord2f(b7) >= (float)tp.prev_time))) f = 0u;
        if (in && c == 0) {
            const uint32_t tn = min((uint32_t)TILE, N - t * TILE);
            tp.tile_flags[t] = (uint8_t)f;
            if (f & 1u) { tp.wave_cnt[t] = make_uint4(0u, 0u, 0u, 0u); cskip += tn; }
            if (f & 2u) sskip += tn;
        }
    }
    cskip = wave_sum_u32(cskip); sskip = wave_sum_u32(sskip);
    if (lane == 0) { s_sk[0][wave] = cskip; s_sk[1][wave] = sskip; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += s_sk[0][w]; b += s_sk[1][w]; }
        tp.prep_part[blk] = make_uint2(a, b);
    }
}

constexpr int CAND_GROUP_MAX = 16;  // association blocks per candidate-counting workgroup: 4, 8 or 16, chosen per image size (sm_create: ~250-500 groups)

// ---------------------------------------------------------------------------------------------
// Slot-addressed sharding, the end of a frame: after the fused masks of all ranks were sum-reduced (gmask),
// every rank counts the frame's fused / new pixels from the same two planes (so DevState stays identical on all ranks), the
// owner of the frame's segment empties the slots of candidates that ANOTHER rank fused (it wrote them speculatively in
// k_associate_direct<true>: each candidate owns its slot, so that write disturbed nothing), and block 0 replaces this
// rank's share of the pass counters by the totals over the ranks.  The W*H conflict cap (src/GlobalModel.cpp:54-57)
// is defined on the conflicts of ALL ranks in slot order and is not evaluated per shard: a frame that exceeds it is
// flagged (sticky SM_E_UNSUPPORTED) instead of producing a model that could differ.
// Nothing but the NEXT frame's surfel pass needs this done, so it normally runs as extra workgroups of that frame's
// k_prep (one launch less per frame); k_shard_settle is the stand-alone form for everything that reads the counters first.
// ---------------------------------------------------------------------------------------------
struct ShardSettle {
    uint32_t n;                       // pixel blocks to settle (0: nothing pending) -- as extra workgroups of the next frame's k_prep, or k_shard_settle
    DevState *st;
    const uint64_t *validmask, *ownmask, *gmask;
    uint32_t nwords;
    const uint32_t *blk_cand, *grp_cand;
    uint32_t *frame_sub;
    uint64_t *alive;
    uint32_t *tile_dead;
    int owner;
    uint32_t cap_pixels, max_vertices;
    uint32_t cg;                      // association blocks per candidate group
};

// NSUB pixel blocks per workgroup (blockDim.x == NSUB * 256): sub-block = threadIdx.x / 256.  No early exit: every thread
// reaches every barrier.
template <int NSUB>
__device__ __forceinline__ void shard_settle_body(const ShardSettle &a, uint32_t wg)
{
    __shared__ uint32_t s_v[NSUB][4], s_any[NSUB];
    __shared__ uint32_t s_hole[NSUB][12], s_dead[NSUB][2];
    const uint32_t sub = threadIdx.x >> 8, tid = threadIdx.x & 255u;
    const int lane = (int)(tid & 63u), wave = (int)(tid >> 6);
    const uint32_t blk = wg * (uint32_t)NSUB + sub;                 // pixel block of k_associate_direct's geometry
    if (tid < 12u) s_hole[sub][tid] = 0u;
    if (tid < 2u) s_dead[sub][tid] = 0u;
    if (tid == 0u) s_any[sub] = 0u;
    const uint32_t word = blk * (PIX_BLOCK / 64) + (uint32_t)wave;
    const bool in = blk < a.n && word < a.nwords;
    const uint64_t vw = in ? a.validmask[word] : 0ull, gw = in ? a.gmask[word] : 0ull, ow = in ? a.ownmask[word] : 0ull;
    const uint64_t foreign = gw & ~ow & vw;                       // fused by another rank
    const uint32_t grp = blk / a.cg, in_grp = blk % a.cg;
    uint32_t pre = 0;
    const bool need = a.owner != 0 && blk < a.n;                  // only the owner has slots to empty
    if (need) {
        pre = (lane < (int)in_grp) ? a.blk_cand[grp * a.cg + lane] : 0u;
        for (uint32_t g = lane; g < grp; g += 64u) pre += a.grp_cand[g];
    }
    const uint32_t offset = a.st->offset;
    __syncthreads();
    if (lane == 0) {
        s_v[sub][wave] = (uint32_t)__popcll(vw);
        if (foreign) s_any[sub] = 1u;
    }
    if (blk == 0u && tid == 0u) {
        const uint64_t conf = a.gmask[a.nwords], vis = a.gmask[a.nwords + 1], kill = a.gmask[a.nwords + 2];
        // (the cap itself was applied before the association -- k_shard_cap_repair, with conflict ordinals over ALL ranks -- so
        //  the counters that arrive here are the effective ones; conflictCount saturates like the reference's query)
        a.st->conflict_count = (uint32_t)(conf > (uint64_t)a.cap_pixels ? (uint64_t)a.cap_pixels : conf);
        a.st->visible_count = (uint32_t)vis;
        a.st->n_kill = (uint32_t)kill;
    }
    if (lane == 0 && in) {
        const uint32_t nf = (uint32_t)__popcll(gw & vw), nn = (uint32_t)__popcll(vw & ~gw);
        if (nn) atomicAdd(&a.frame_sub[2 * SUB_SET + (word & 63u) * SUB_STRIDE], nn);
        if (nf) atomicAdd(&a.frame_sub[3 * SUB_SET + (word & 63u) * SUB_STRIDE], nf);
    }
    __syncthreads();
    const bool holes = need && s_any[sub] != 0u;                  // uniform per sub-block
    pre = wave_sum_u32(pre);
    uint32_t rank = (uint32_t)__popcll(vw & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += s_v[sub][w];
    const uint32_t slot = offset + pre + rank;
    const uint32_t blk_first = offset + pre, w_first = blk_first >> 6, t_first = blk_first / (uint32_t)TILE;
    if (holes && ((foreign >> lane) & 1ull) && (uint64_t)slot < (uint64_t)a.max_vertices) {
        const uint32_t w = (slot >> 6) - w_first, bit = slot & 63u;
        atomicOr(&s_hole[sub][w * 2u + (bit >> 5)], 1u << (bit & 31u));
        atomicAdd(&s_dead[sub][slot / (uint32_t)TILE - t_first], 1u);
    }
    __syncthreads();
    if (holes) {
        if (tid < 6u) {
            const uint64_t m = (uint64_t)s_hole[sub][tid * 2u] | ((uint64_t)s_hole[sub][tid * 2u + 1u] << 32);
            if (m) atomicAnd((unsigned long long *)&a.alive[w_first + tid], ~m);
        } else if (tid < 8u) {
            const uint32_t d = s_dead[sub][tid - 6u];
            if (d) atomicAdd(&a.tile_dead[t_first + tid - 6u], d);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// p0a metricise (depth_metric.frag:15-35) + u8 RGB/semantic pack + LDS-tiled transpose to the
// column-major frame layout + key-map clear.  32x32 pixel tile per 1024-thread workgroup.
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
    const uint8_t *rgb; const uint16_t *depth_raw; const uint8_t *sem; const float *depth_f32;
    float *depthT; uint32_t *rgbsT; uint64_t *keyT; uint2 *dcT; uint32_t *conf_sub;
};

// one 32x32-pixel tile per workgroup of NT = 1024 (one round) or 256 threads (four rounds of 8 rows, unrolled: all loads of
// a thread are in flight together)
template <int NT>
__device__ __forceinline__ void prep_image_block(const PrepArgs &a, const FrameParams &fp, uint32_t bid)
{
    const uint8_t *__restrict__ rgb = a.rgb; const uint16_t *__restrict__ depth_raw = a.depth_raw; const uint8_t *__restrict__ sem = a.sem;
    const float *__restrict__ depth_f32 = a.depth_f32; float *__restrict__ depthT = a.depthT; uint32_t *__restrict__ rgbsT = a.rgbsT;
    uint64_t *__restrict__ keyT = a.keyT; uint2 *__restrict__ dcT = a.dcT; uint32_t *__restrict__ conf_sub = a.conf_sub;
    __shared__ float s_d[32][33];
    if (conf_sub && bid == 0 && threadIdx.x < 64) conf_sub[threadIdx.x * SUB_STRIDE] = 0u;
    __shared__ uint32_t s_c[32][33];
    const int W = fp.W, H = fp.H;
    const int tiles_x = (W + 31) >> 5;
    const int i0 = (bid % tiles_x) << 5, j0 = (bid / tiles_x) << 5;
    constexpr int ROWS = NT / 32, ROUNDS = 32 / ROWS;
    const int tx = threadIdx.x & 31, ty0 = (int)(threadIdx.x >> 5);
    if (ROUNDS == 1) {
        const int ty = ty0;
        const int i = i0 + tx, j = j0 + ty;          // read: lanes along the image row
        float d = 0.0f;
        uint32_t c = 0;
        if (i < W && j < H) {
            const size_t p = (size_t)j * W + i;
            if (depth_f32) {
                d = depth_f32[p];
            } else if (depth_raw) {
                const uint32_t lo = (uint32_t)(fp.min_depth * 1000.0f);
                const uint32_t hi = (uint32_t)((fp.max_depth - 0.001f) * 1000.0f);
                const uint32_t v = depth_raw[p];
                if (!((float)i + 0.5f < fp.stereo_border)) {
                    if (v > lo && v < hi) d = (float)v / 1000.0f;
                }
            }
            uint32_t s = sem ? (uint32_t)sem[p] : 0u;
            uint32_t cr = 0, cg = 0, cb = 0;
            if (rgb) { cr = rgb[p * 3]; cg = rgb[p * 3 + 1]; cb = rgb[p * 3 + 2]; }
            c = (s << 24) | (cr << 16) | (cg << 8) | cb;
        }
        s_d[ty][tx] = d;
        s_c[ty][tx] = c;
    } else {
        // several rows per thread: every load unconditional (clamped address) and issued before the first use
        uint32_t v[ROUNDS], sv[ROUNDS], cr[ROUNDS], cg[ROUNDS], cb[ROUNDS];
        float df[ROUNDS];
        const int ic = min(i0 + tx, W - 1);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const size_t p = (size_t)min(j0 + ty0 + r * ROWS, H - 1) * W + ic;
            v[r] = (depth_raw && !depth_f32) ? depth_raw[p] : 0u;
            df[r] = depth_f32 ? depth_f32[p] : 0.0f;
            sv[r] = sem ? (uint32_t)sem[p] : 0u;
            cr[r] = rgb ? rgb[p * 3] : 0u; cg[r] = rgb ? rgb[p * 3 + 1] : 0u; cb[r] = rgb ? rgb[p * 3 + 2] : 0u;
        }
        const uint32_t lo = (uint32_t)(fp.min_depth * 1000.0f);
        const uint32_t hi = (uint32_t)((fp.max_depth - 0.001f) * 1000.0f);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int ty = ty0 + r * ROWS;
            const int i = i0 + tx, j = j0 + ty;
            float d = 0.0f;
            uint32_t c = 0;
            if (i < W && j < H) {
                if (depth_f32) d = df[r];
                else if (depth_raw && !((float)i + 0.5f < fp.stereo_border) && v[r] > lo && v[r] < hi) d = (float)v[r] / 1000.0f;
                c = (sv[r] << 24) | (cr[r] << 16) | (cg[r] << 8) | cb[r];
            }
            s_d[ty][tx] = d;
            s_c[ty][tx] = c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ty = ty0; ty < 32; ty += ROWS) {
        const int i = i0 + ty, j = j0 + tx;          // write: lanes along the image column
        if (i < W && j < H) {
            const size_t q = (size_t)i * H + j;
            if (depthT) depthT[q] = s_d[tx][ty];
            if (rgbsT) rgbsT[q] = s_c[tx][ty];
            if (depthT) dcT[q] = make_uint2(__float_as_uint(s_d[tx][ty]), s_c[tx][ty]);
            else reinterpret_cast<uint32_t *>(dcT)[2 * q + 1] = s_c[tx][ty];       // depth plane kept
            if (keyT) keyT[q] = KEY_EMPTY;
        }
    }
}

__global__ __launch_bounds__(1024) void k_prep(const uint8_t *__restrict__ rgb,
                                               const uint16_t *__restrict__ depth_raw,
                                               const uint8_t *__restrict__ sem,
                                               const float *__restrict__ depth_f32,  // optional: metric depth given directly
                                               float *__restrict__ depthT, uint32_t *__restrict__ rgbsT,
                                               uint64_t *__restrict__ keyT, FrameParams fp,
                                               uint2 *__restrict__ dcT /* (depth bits, rgbs) per pixel: one 8-byte gather for the conflict test */,
                                               uint32_t *__restrict__ conf_sub /* this frame's 64 conflict sub-counters, or null */,
                                               TilePrep tp /* the first tp.nfb workgroups build the frame's tile flags */,
                                               ShardSettle ss /* then ceil(ss.n / 4) workgroups finish the previous frame of a sharded stream */)
{
    if (blockIdx.x < tp.nfb) { tile_prep_block(fp, tp); return; }       // workgroup-uniform
    const uint32_t nsb = (ss.n + 3u) / 4u;
    if (blockIdx.x < tp.nfb + nsb) { shard_settle_body<4>(ss, blockIdx.x - tp.nfb); return; }
    PrepArgs pa;
    pa.rgb = rgb; pa.depth_raw = depth_raw; pa.sem = sem; pa.depth_f32 = depth_f32; pa.depthT = depthT; pa.rgbsT = rgbsT; pa.keyT = keyT;
    pa.dcT = dcT; pa.conf_sub = conf_sub;
    prep_image_block<1024>(pa, fp, blockIdx.x - tp.nfb - nsb);
}

// ---------------------------------------------------------------------------------------------
// The depth pre-processing chain p0a..p0e of SurfelMapping::processFrame (src/SurfelMapping.cpp:136-156,254-365) as ONE
// LDS-tiled stage: a workgroup produces a 14 x 30-pixel tile of the frame planes and computes everything that tile needs
// from the caller's raw images itself --
//   p0a  metricise            depth_metric.frag:15-35      on the tile + 8 pixels of halo   (30 x 46)
//   p0b  filter, |dz| < 0.15  depth_filter.frag:16-80      on the tile + 7                  (28 x 44)
//   p0c  13 x 13 class-aware weighted mean  depth_smooth.frag:17-82   on the tile + 1       (16 x 32)
//   p0d  filter, |dz| < 0.10                               on the tile
//   p0e  moving-object removal against LAST  depth_movings.frag:20-82  on the tile
// -- so the five dependent launches of rounds 1-2 (k_prep, k_filter_depth, k_smooth_depth, k_filter_depth, k_remove_movings:
// ~55 us at KITTI size, most of it launch floors and boundaries) become block ranges of the frame's one preparation launch
// (k_assoc_prep<., true>), next to the previous frame's association and this frame's tile flags, and a frame with the chain
// has the same three launches as one without.  Every stage is a pure function of the stage before it, so recomputing the halo
// gives the values the separate passes gave: results are bit-identical (tests/test_gpu_parity.py::test_preprocess_*).  The
// halo costs 3.3x of the cheap stages and 512 / 420 = 1.22x of the smooth, which is where the time is (169 taps per pixel);
// 30 x 30 tiles (1.14x) left 546 workgroups for 256 CUs -- some CUs three, most two -- and four pixels per thread.  Texture names as the reference's ping-pong leaves them: DEPTH_METRIC = p0e's output (p0c's on the reference
// frame, which stops before p0e), DEPTH_FILTERED = p0d's, LAST <- DEPTH_FILTERED at the end of the frame.
// 256 threads; the 16 x 32 smooth region is two pixels per thread.
// ---------------------------------------------------------------------------------------------
struct Mat4 { float m[16]; };

struct ChainArgs {
    const float *lastT;       // LAST: the previous frame's DEPTH_FILTERED (column-major)
    float *filteredT;         // out: DEPTH_FILTERED of this frame
    float w[169];             // 13 x 13 weights exp(-(ix^2 + iy^2) sigPix), the host's (src/SurfelMapping.cpp:292-309): kernel arguments, read with scalar loads
    Mat4 t_c2l;               // current camera -> last camera (src/SurfelMapping.cpp:345-349)
    int do_movings;           // 0: the reference frame (src/SurfelMapping.cpp:142-154 returns before removeMovings)
    int border;               // ceil(stereoBorder - 0.5): first column the smooth may read (texX < stereoBorder / cols is skipped)
};

constexpr int CH_TX = 14, CH_TY = 30;                                   // tile: 14 columns x 30 rows (column-major planes: a 30-row run is 120 contiguous bytes)
constexpr int CH_MX = CH_TX + 16, CH_MY = CH_TY + 16;                  // metric region (tile + 8)
constexpr int CH_FX = CH_TX + 14, CH_FY = CH_TY + 14;                  // p0b region (tile + 7)
constexpr int CH_SX = CH_TX + 2, CH_SY = CH_TY + 2;                    // p0c region (tile + 1): 16 x 32 = 512 pixels, two per thread
constexpr int CH_MS = CH_MY + 1, CH_CS = CH_MY + 2, CH_FS = CH_FY + 1, CH_KS = CH_FY + 2, CH_SS = CH_SY + 1;   // row strides (odd word strides: no bank conflicts along a column of lanes)
constexpr int CH_OFF_C = CH_MX * CH_MS * 4, CH_OFF_F = CH_OFF_C + CH_MX * CH_CS, CH_OFF_K = CH_OFF_F + CH_FX * CH_FS * 4;
constexpr int CHAIN_LDS_BYTES = (CH_OFF_K + CH_FX * CH_KS * 2 + 15) / 16 * 16;
static_assert(CH_OFF_F % 4 == 0 && CH_OFF_K % 2 == 0 && CH_SX * CH_SY == 512, "chain tile layout");

// depth_filter.frag:16-80 for the pixel at (ci, cj) of a staged plane `d` (stride ds) whose classes are in `c` (stride cs, at
// (ki, kj)); (gi, gj) is the pixel's position in the image (neighbours outside the image do not count: depth_filter.frag:52)
__device__ __forceinline__ float chain_filter_px(const float *d, int ds, int ci, int cj, const uint8_t *c, int cs, int ki, int kj,
                                                 int gi, int gj, int W, int H, float min_depth, float diff_thresh)
{
    // branch-free: the nine depths and classes are loaded together (one wait), the support is a sum of predicates
    float dn[9];
    uint32_t cn[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ix = t % 3 - 1, iy = t / 3 - 1;
        dn[t] = d[(ci + ix) * ds + cj + iy];
        cn[t] = c[(ki + ix) * cs + kj + iy];
    }
    const float depth = dn[4];
    const uint32_t cl = cn[4];
    int support = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t == 4) continue;
        const int qi = gi + t % 3 - 1, qj = gj + t / 3 - 1;
        const bool in = qi >= 0 && qi < W && qj >= 0 && qj < H;                // depth_filter.frag:52
        support += (in & (fabsf(dn[t] - depth) < diff_thresh) & (cl == cn[t])) ? 1 : 0;
    }
    if (depth <= min_depth || depth >= 100.0f || cl == 10u || cl == 11u || cl == 12u) return 0.0f;
    return support >= 7 ? depth : 0.0f;
}

__device__ __forceinline__ void prep_chain_block(const PrepArgs &a, const ChainArgs &ch, const FrameParams &fp, uint32_t bid,
                                                 unsigned char *lds /* CHAIN_LDS_BYTES, 16-byte aligned */)
{
    float *s_m = reinterpret_cast<float *>(lds);                                   // [CH_MX][CH_MS] metric depth; later [CH_SX][CH_SS] smoothed
    uint8_t *s_c = lds + CH_OFF_C;                                                 // [CH_MX][CH_CS] class (any value outside the image: depth 0 there)
    float *s_f = reinterpret_cast<float *>(lds + CH_OFF_F);                        // [CH_FX][CH_FS] p0b's output; later the tile's packed colour words
    uint16_t *s_k = reinterpret_cast<uint16_t *>(lds + CH_OFF_K);                  // [CH_FX][CH_KS] the class of a pixel p0c may average, 0x100 for one it may not
    const int W = fp.W, H = fp.H;
    const int tiles_x = (W + CH_TX - 1) / CH_TX;
    const int i0 = (int)(bid % (uint32_t)tiles_x) * CH_TX, j0 = (int)(bid / (uint32_t)tiles_x) * CH_TY;
    const int tid = (int)threadIdx.x;
    if (a.conf_sub && bid == 0 && tid < 64) a.conf_sub[tid * SUB_STRIDE] = 0u;
    // ---- p0a: metricise the tile + 8 (lanes along the image row: the caller's images are row-major)
    {
        const uint32_t lo = (uint32_t)(fp.min_depth * 1000.0f);
        const uint32_t hi = (uint32_t)((fp.max_depth - 0.001f) * 1000.0f);
        for (int e = tid; e < CH_MX * CH_MY; e += 256) {
            const int lj = e / CH_MX, li = e - lj * CH_MX;
            const int gi = i0 - 8 + li, gj = j0 - 8 + lj;
            float d = 0.0f;
            uint32_t c = 255u;
            if (gi >= 0 && gi < W && gj >= 0 && gj < H) {
                const size_t p = (size_t)gj * W + gi;
                const uint32_t v = a.depth_raw[p];
                if (!((float)gi + 0.5f < fp.stereo_border) && v > lo && v < hi) d = (float)v / 1000.0f;
                c = a.sem ? (uint32_t)a.sem[p] : 0u;
            }
            s_m[li * CH_MS + lj] = d;
            s_c[li * CH_CS + lj] = (uint8_t)c;
        }
    }
    __syncthreads();
    // ---- p0b: filter, 0.15, on the tile + 7; with it, per pixel, what p0c's taps test -- "inside the columns the smooth may read
    // (texX >= stereoBorder / cols, depth_smooth.frag), depth in (min, 100)" -- folded into one 16-bit word with the class
    for (int e = tid; e < CH_FX * CH_FY; e += 256) {
        const int fi = e / CH_FY, fj = e - fi * CH_FY;
        const int gi = i0 - 7 + fi, gj = j0 - 7 + fj;
        float r = 0.0f;
        if (gi >= 0 && gi < W && gj >= 0 && gj < H)
            r = chain_filter_px(s_m, CH_MS, fi + 1, fj + 1, s_c, CH_CS, fi + 1, fj + 1, gi, gj, W, H, fp.min_depth, 0.15f);
        s_f[fi * CH_FS + fj] = r;
        const bool tap_ok = gi >= ch.border && gi < W && gj >= 0 && gj < H && !(r <= fp.min_depth || r >= 100.0f);
        s_k[fi * CH_KS + fj] = tap_ok ? (uint16_t)s_c[(fi + 1) * CH_CS + fj + 1] : (uint16_t)0x100u;
    }
    __syncthreads();
    // ---- p0c: 13 x 13 class-aware weighted mean on the tile + 1; accumulation order as the shader's (iy outer, ix inner).  A
    // thread takes two pixels side by side in a row, (2c, r) and (2c + 1, r): per window row they share 12 of their 13 columns,
    // so 14 depths + 14 validity-class words serve both (28 LDS reads for 26 taps; lanes run along the image column: odd row
    // strides, no bank conflicts).  The result overwrites the metric plane (dead since p0b).
    float sm[2] = {0.0f, 0.0f};
    {
        const int sc = tid >> 5, sj = tid & 31;                    // column pair, row of the 16 x 32 region
        const int si = 2 * sc;
        const int gi = i0 - 1 + si, gj = j0 - 1 + sj;
        const float dep0 = s_f[(si + 6) * CH_FS + sj + 6], dep1 = s_f[(si + 7) * CH_FS + sj + 6];
        const uint32_t cl0 = s_c[(si + 7) * CH_CS + sj + 7], cl1 = s_c[(si + 8) * CH_CS + sj + 7];
        const bool act0 = gi >= 0 && gi < W && gj >= 0 && gj < H && !(dep0 <= fp.min_depth || dep0 >= 100.0f || cl0 == 10u);
        const bool act1 = gi + 1 >= 0 && gi + 1 < W && gj >= 0 && gj < H && !(dep1 <= fp.min_depth || dep1 >= 100.0f || cl1 == 10u);
        if (act0 || act1) {
            // (a pixel that is not averaged runs along with an impossible class: its sums stay 0 and are not used)
            const uint32_t k0 = act0 ? cl0 : 0x200u, k1 = act1 ? cl1 : 0x200u;
            float s10 = 0.0f, s20 = 0.0f, s11 = 0.0f, s21 = 0.0f;
            // A window row at a time: its LDS reads go out together and the taps are predicated, not branched -- with a branch per
            // tap the compiler put an s_waitcnt behind every single read, three dependent LDS round trips per tap, and a tile took
            // 108 us.  A tap is one compare (the word of s_k: class, or 0x100 where the smooth may not read), one select, one
            // multiply, two adds: `w' = ok ? w : 0; sum1 += dk * w'; sum2 += w'` is the shader's arithmetic -- depths are finite and
            // >= 0, so a skipped tap adds +0, and sums that start at +0 never become -0 (round to nearest): adding +0 changes
            // nothing, and the taken adds come in the shader's order.  Its `valid > 0` is `sum2 > 0`: every weight is positive.
#pragma unroll 1
            for (int iy = -6; iy <= 6; ++iy) {
                float dk[14];
                uint32_t ck[14];
#pragma unroll
                for (int x = 0; x < 14; ++x) {
                    dk[x] = s_f[(si + x) * CH_FS + sj + 6 + iy];
                    ck[x] = s_k[(si + x) * CH_KS + sj + 6 + iy];
                }
#pragma unroll
                for (int ix = 0; ix < 13; ++ix) {
                    const float w = ch.w[(iy + 6) * 13 + ix];               // (wave-uniform index into the kernel arguments: a scalar load)
                    const float w0 = (k0 == ck[ix]) ? w : 0.0f, w1 = (k1 == ck[ix + 1]) ? w : 0.0f;
                    s10 += dk[ix] * w0; s20 += w0;
                    s11 += dk[ix + 1] * w1; s21 += w1;
                }
            }
            if (act0 && s20 > 0.0f) sm[0] = s10 / s20;
            if (act1 && s21 > 0.0f) sm[1] = s11 / s21;
        }
    }
    __syncthreads();                                   // every read of the metric plane (p0b) is long done; p0c's reads of s_f are done
    float *s_s = s_m;                                  // [CH_SX][CH_SS]
    uint32_t *s_rgb = reinterpret_cast<uint32_t *>(s_f);   // [CH_TX][CH_TY + 1] packed class | r | g | b of the tile
    s_s[(2 * (tid >> 5)) * CH_SS + (tid & 31)] = sm[0];
    s_s[(2 * (tid >> 5) + 1) * CH_SS + (tid & 31)] = sm[1];
    // the tile's colour words: read along image rows (coalesced), used along columns below
    for (int e = tid; e < CH_TX * CH_TY; e += 256) {
        const int oj = e / CH_TX, oi = e - oj * CH_TX;
        const int gi = i0 + oi, gj = j0 + oj;
        uint32_t c = 0u;
        if (gi < W && gj < H) {
            const size_t p = (size_t)gj * W + gi;
            uint32_t cr = 0, cg = 0, cb = 0;
            if (a.rgb) { cr = a.rgb[p * 3]; cg = a.rgb[p * 3 + 1]; cb = a.rgb[p * 3 + 2]; }
            c = ((uint32_t)s_c[(oi + 8) * CH_CS + oj + 8] << 24) | (cr << 16) | (cg << 8) | cb;
        }
        s_rgb[oi * (CH_TY + 1) + oj] = c;
    }
    __syncthreads();
    // ---- p0d: filter, 0.10, and p0e: moving objects, on the tile; lanes along the image COLUMN (the planes are column-major)
    for (int e = tid; e < CH_TX * CH_TY; e += 256) {
        const int oi = e / CH_TY, oj = e - oi * CH_TY;
        const int gi = i0 + oi, gj = j0 + oj;
        if (gi >= W || gj >= H) continue;
        const float f2 = chain_filter_px(s_s, CH_SS, oi + 1, oj + 1, s_c, CH_CS, oi + 8, oj + 8, gi, gj, W, H, fp.min_depth, 0.1f);
        const uint32_t rgbs = s_rgb[oi * (CH_TY + 1) + oj];
        const uint32_t cl = rgbs >> 24;
        float out = s_s[(oi + 1) * CH_SS + oj + 1];    // the reference frame stops after p0d: DEPTH_METRIC holds p0c's output
        if (ch.do_movings) {
            // depth_movings.frag:20-82 (host src/SurfelMapping.cpp:336-365): pixels of movable classes (13..18) are reprojected
            // into the previous frame and zeroed if |z_hat - z_last| > 0.5 m
            out = f2;
            const float px = (float)gi + 0.5f, py = (float)gj + 0.5f;
            if (!(px < fp.stereo_border || f2 <= fp.min_depth) && (cl >= 13u && cl <= 18u)) {
                const float vx = (px - fp.cx) * f2 / fp.fx, vy = (py - fp.cy) * f2 / fp.fy;
                const float3 t = xform3(ch.t_c2l.m, vx, vy, f2);
                const float ux = fp.fx * t.x / t.z + fp.cx;
                const float uy = fp.fy * t.y / t.z + fp.cy;
                const float uz = t.z;
                if (!(uz <= fp.min_depth || uz >= 100.0f || ux < fp.stereo_border || ux > fp.cols || uy < 0.0f || uy > fp.rows)) {
                    const int qi = tex_idx(ux / fp.cols, W), qj = tex_idx(uy / fp.rows, H);
                    const float depth_last = ch.lastT[(size_t)qi * H + qj];
                    if (fabsf(uz - depth_last) > 0.5f) out = 0.0f;
                }
            }
        }
        const size_t q = (size_t)gi * H + gj;
        a.depthT[q] = out;
        ch.filteredT[q] = f2;
        a.rgbsT[q] = rgbs;
        a.dcT[q] = make_uint2(__float_as_uint(out), rgbs);
        if (a.keyT) a.keyT[q] = KEY_EMPTY;
    }
}

// column-major -> row-major read-back helper (tests / GUI textures)
__global__ void k_untranspose_f32(const float *__restrict__ srcT, float *__restrict__ dst, int W, int H)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W * H) return;
    const int j = p / W, i = p - j * W;
    dst[p] = srcT[(size_t)i * H + j];
}

__global__ void k_fill_keys(uint64_t *keyT, int P)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < P) keyT[q] = KEY_EMPTY;
}

// ---------------------------------------------------------------------------------------------
// p2 conflict test (conflict.vert:25-83, conflict.geom:13-24) over the SoA model.
// One wave = 64 consecutive surfels = one ballot word per mask:
//   cm  conflict (and id > 0)          dm  would die if decremented: !(conf-1 > 0)
//   zm  dead already: !(conf > 0)      (back_map.geom:17 culls on conf <= 0 / NaN)
// plus per-tile counts (nconf, nkill = popc(zm | cm&dm), nzero).
// ---------------------------------------------------------------------------------------------
#ifndef SM_CONFLICT_WAVES
#define SM_CONFLICT_WAVES 5     // 96 VGPRs, no spills: 5 waves/SIMD measured best (6 and 8 spill and are slower)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SM_CONFLICT_WAVES, 8))) void k_conflict(Model M, const DevState *__restrict__ st, FrameParams fp,
                                                  const uint2 *__restrict__ dcT /* (depth bits, sem<<24|rgb) per pixel */,
                                                  uint64_t *__restrict__ cm, uint64_t *__restrict__ dm,
                                                  uint64_t *__restrict__ zm, uint32_t *__restrict__ tile_cnt,
                                                  const uint32_t *__restrict__ tb, uint8_t *__restrict__ tile_flags,
                                                  uint32_t *__restrict__ blk_part /* [grid][4]: skipped, nconf, nkill, - */,
                                                  const uint64_t *__restrict__ alive,
                                                  uint32_t *__restrict__ conf_sub /* 64 sub-counters of the frame's conflicts (zeroed by k_prep) */)
{
    __shared__ uint32_t s_red[4][3];
    __shared__ uint64_t s_m[3][TILE_WORDS];
    __shared__ uint8_t s_flags[64];
    const uint32_t N = st->count;
    const bool has_dead = st->garbage != 0u;          // slots of surfels killed since the last physical compaction
    const uint32_t exempt = fp.no_exempt ? 0xFFFFFFFFu : (fp.world > 1 ? fp.exempt_local : st->first_live);   // the surfel with (global) id 0
    const float4 *__restrict__ pc = M.s[st->cur].pos_conf;
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t skipped = 0;
    uint32_t acc = 0;                 // thread 0: conflicts, thread 1: kills of this workgroup's tiles
    uint64_t skipmask = 0;
    uint32_t iter = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++iter) {
        uint32_t nconf = 0, nkill = 0, nzero = 0;
        if ((iter & 63u) == 0u) {
            // the skip flags of this workgroup's next 64 tiles (corner-parallel, via LDS): bit 0 stays in a ballot mask,
            // bit 1 (splat) is stored for the cull kernel
            __syncthreads();
            tile_flags_batch(tile, gridDim.x, ntiles, fp, tb, s_flags);
            __syncthreads();
            const uint64_t tl = (uint64_t)tile + (uint64_t)lane * gridDim.x;
            const uint32_t f = s_flags[lane];
            if (wave == 0 && tl < ntiles) tile_flags[tl] = (uint8_t)f;
            skipmask = __ballot((f & 1u) != 0u);
        }
        // whole tile outside the conflict view volume (conflict.vert:35)?  Then nothing conflicts, and a tile
        // without "bad" surfels has nothing dead either: zero masks, zero counts, no surfel read.
        if ((skipmask >> (iter & 63u)) & 1ull) {
            if (threadIdx.x < 3) tile_cnt[tile * 3 + threadIdx.x] = 0u;
            if (threadIdx.x >= 64 && threadIdx.x < 64 + 3 * TILE_WORDS) {
                const int m = (threadIdx.x - 64) / TILE_WORDS, w = (threadIdx.x - 64) % TILE_WORDS;
                const uint32_t word = tile * TILE_WORDS + w;
                if ((uint64_t)word * 64u < N) { uint64_t *dst = m == 0 ? cm : (m == 1 ? dm : zm); dst[word] = 0ull; }
            }
            skipped += min((uint32_t)TILE, N - tile * TILE);
            continue;
        }
        // phase 1: all four 16-byte loads of the lane in flight together
        float4 v[4];
        bool valid[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t k = (tile * TILE_WORDS + r * 4 + wave) * 64u + lane;
            valid[r] = k < N;
            v[r] = pc[min(k, N - 1u)];          // unconditional (clamped): a branch here would serialise the loads
        }
        if (has_dead) {                         // workgroup-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) valid[r] = valid[r] && ((alive[tile * TILE_WORDS + r * 4 + wave] >> lane) & 1ull);
        }
        // phase 2: projection + view test; phase 3: the dependent depth/class gathers, again together
        float zc[4], lam[4], dep[4];
        uint32_t cls[4], qq[4];
        bool inview[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            inview[r] = false; zc[r] = 0.f; lam[r] = 0.f; qq[r] = 0u;
            if (valid[r]) {
                const float3 ph = xform3(fp.t_inv, v[r].x, v[r].y, v[r].z);
                // the depth-range test first: it needs no division (conflict.vert:35 is one || chain)
                if (!(ph.z <= fp.min_depth || ph.z >= fp.max_depth)) {
                    const float xl = ph.x / ph.z;
                    const float yl = ph.y / ph.z;
                    const float u = fp.fx * xl + fp.cx;
                    const float vv = fp.fy * yl + fp.cy;
                    if (!(u < fp.stereo_border || u > fp.cols || vv < 0.0f || vv > fp.rows)) {
                        const int ti = tex_idx(u / fp.cols, fp.W), tj = tex_idx(vv / fp.rows, fp.H);
                        qq[r] = (uint32_t)(ti * fp.H + tj);
                        lam[r] = sqrtf((xl * xl + yl * yl) + 1.0f);
                        zc[r] = ph.z;
                        inview[r] = true;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                 // gathers, unconditional (pixel 0 for out-of-view lanes)
            const uint2 g = dcT[qq[r]];               // depth and class in one 8-byte access
            dep[r] = __uint_as_float(g.x);
            cls[r] = g.y >> 24;
        }
        // phase 4: conflict rule + ballots
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t word = tile * TILE_WORDS + r * 4 + wave;
            const uint32_t k = word * 64u + lane;
            bool conflict = false;
            if (inview[r]) {
                float depth = dep[r];
                if (cls[r] == 10u) depth = fp.max_depth + 1.0f;
                if (fp.is_clean == 0 && depth == 0.0f) depth = fp.max_depth + 20.0f;
                conflict = (depth * lam[r] - zc[r] * lam[r] > fp.conflict_thresh * zc[r]) && (k != exempt);
            }
            const bool dies = valid[r] && !(v[r].w - 1.0f > 0.0f);
            const bool dead = valid[r] && !(v[r].w > 0.0f);
            const uint64_t cw = __ballot(conflict), dw = __ballot(dies), zw = __ballot(dead);
            if (lane == 0) { s_m[0][r * 4 + wave] = cw; s_m[1][r * 4 + wave] = dw; s_m[2][r * 4 + wave] = zw; }
            nconf += __popcll(cw);
            nkill += __popcll(zw | (cw & dw));
            nzero += __popcll(zw);
        }
        if (lane == 0) { s_red[wave][0] = nconf; s_red[wave][1] = nkill; s_red[wave][2] = nzero; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const uint32_t tsum = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
            tile_cnt[tile * 3 + threadIdx.x] = tsum;
            acc += tsum;
        }
        if (threadIdx.x >= 64 && threadIdx.x < 64 + 3 * TILE_WORDS) {
            // one store instruction for the tile's 3 x 16 ballot words (48 lanes, three 128-byte runs)
            const int m = (threadIdx.x - 64) / TILE_WORDS, w = (threadIdx.x - 64) % TILE_WORDS;
            const uint32_t word = tile * TILE_WORDS + w;
            if ((uint64_t)word * 64u < N) {
                uint64_t *dst = m == 0 ? cm : (m == 1 ? dm : zm);
                dst[word] = s_m[m][w];
            }
        }
        __syncthreads();
    }
    // per-workgroup partials, summed by k_cull_finalize (no same-address atomics)
    if (threadIdx.x < 2) blk_part[blockIdx.x * 4 + 1 + threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        blk_part[blockIdx.x * 4] = skipped;
        // the conflict total in a form the next kernel can read in one instruction: 64 counters, <= 32 adders each
        if (acc) atomicAdd(&conf_sub[(blockIdx.x & 63u) * SUB_STRIDE], acc);
    }
}

// ---------------------------------------------------------------------------------------------
// Two-level scan of the per-tile counts -> survivor prefix for the stable compaction (p4,
// back_map.geom:15-28), the new count/offset (src/GlobalModel.cpp:575) and the "first cap conflicts
// only" rule (conflictVbo holds W*H records: src/GlobalModel.cpp:54-57).
//   k_scan_cull      one workgroup per group of 1024 tiles: group-local exclusive prefixes + totals
//   k_cull_finalize  one workgroup: scans the (<= a few hundred) group totals into group bases and
//                    publishes DevState; if the conflict cap binds (rare) it redoes the scan
//                    sequentially with the cap applied tile by tile (exact, straddling tile from masks)
// consumers use  prefix(t) = tile_keep_prefix[t] + group_keep_base[t / 1024].
// ---------------------------------------------------------------------------------------------
constexpr int GROUP = 1024;   // tiles per scan group

__global__ __launch_bounds__(1024) void k_scan_cull(const DevState *__restrict__ st,
                                                    const uint32_t *__restrict__ tile_cnt,
                                                    uint32_t *__restrict__ tile_allow,
                                                    uint32_t *__restrict__ tile_keep_prefix,
                                                    uint32_t *__restrict__ group_tot /* [g][4]: conf, keep, first killing tile, - */,
                                                    const uint32_t *__restrict__ tile_dead)
{
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_first;
    const uint32_t N = st->count;
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const uint32_t t = blockIdx.x * GROUP + threadIdx.x;
    if (threadIdx.x == 0) s_first = 0xFFFFFFFFu;
    uint32_t nconf = 0, keep = 0;
    bool kills = false;
    if (t < ntiles) {
        nconf = tile_cnt[t * 3];
        const uint32_t nkill = tile_cnt[t * 3 + 1];
        keep = min((uint32_t)TILE, N - t * TILE) - tile_dead[t] - nkill;
        kills = nkill != 0 || tile_dead[t] != 0;
        tile_allow[t] = nconf;                  // every conflict takes effect unless the cap binds
    }
    uint32_t ctot, ktot;
    block_scan_1024(nconf, &ctot, s_scan);
    const uint32_t kpre = block_scan_1024(keep, &ktot, s_scan);
    if (kills) atomicMin(&s_first, t);
    if (t < ntiles) tile_keep_prefix[t] = kpre;
    __syncthreads();
    if (threadIdx.x == 0) {
        group_tot[blockIdx.x * 4 + 0] = ctot;
        group_tot[blockIdx.x * 4 + 1] = ktot;
        group_tot[blockIdx.x * 4 + 2] = s_first;
    }
}

// survivors of one 64-surfel word of tile `t` under the effective conflict set (the first `allow` conflicts of the tile)
__device__ __forceinline__ uint64_t keep_word(uint32_t t, int w, uint32_t N, const uint64_t *__restrict__ cm,
                                              const uint64_t *__restrict__ dm, const uint64_t *__restrict__ zm,
                                              const uint64_t *__restrict__ alive, uint32_t allow, uint32_t nconf)
{
    const uint32_t word = t * TILE_WORDS + (uint32_t)w;
    const uint64_t base = (uint64_t)word * 64u;
    if (base >= N) return 0ull;
    const uint64_t rem = (uint64_t)N - base;
    const uint64_t valid = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & alive[word];
    uint64_t ce = cm[word];
    if (allow != nconf) {
        uint32_t before = 0;
        for (int x = 0; x < w; ++x) before += (uint32_t)__popcll(cm[t * TILE_WORDS + x]);
        ce = before >= allow ? 0ull : first_n_bits(ce, allow - before);
    }
    return ~(zm[word] | (ce & dm[word])) & valid;
}

__global__ __launch_bounds__(1024) void k_cull_finalize(DevState *__restrict__ st, FrameParams fp,
                                                        const uint64_t *__restrict__ cm,
                                                        const uint64_t *__restrict__ dm,
                                                        const uint64_t *__restrict__ zm,
                                                        const uint32_t *__restrict__ tile_cnt,
                                                        uint32_t *__restrict__ tile_allow,
                                                        uint32_t *__restrict__ tile_keep_prefix,
                                                        const uint32_t *__restrict__ group_tot,
                                                        uint32_t *__restrict__ group_keep_base,
                                                        const uint32_t *__restrict__ conf_part, uint32_t n_conf_part,
                                                        const uint64_t *__restrict__ alive,
                                                        const uint32_t *__restrict__ tile_dead,
                                                        unsigned long long *__restrict__ host_stat)
{
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_first, s_ft, s_fl, s_keep_first;
    const uint32_t N = st->count;                     // occupied slots
    const uint32_t g0 = st->garbage;                  // dead ones among them
    const uint32_t old_first = st->first_live, old_offset = st->offset;
    const uint32_t holes = st->holes_last;            // dead slots ABOVE old_offset (k_associate_direct: candidates that fused)
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const uint32_t ngroups = (ntiles + GROUP - 1) / GROUP;
    const uint32_t cap = fp.conflict_cap;
    // k_scan_cull ran before this kernel exactly when this cull compacts (the host decides and launches accordingly);
    // a cull that only marks the dead needs no prefixes, its totals come from k_conflict's per-workgroup sums
    const bool have_scan = fp.compact_now != 0u || fp.maintenance != 0;
    if (threadIdx.x == 0) { s_first = 0xFFFFFFFFu; s_ft = 0xFFFFFFFFu; s_fl = 0xFFFFFFFFu; }
    if (threadIdx.x == 1023) {
        // does the surfel that is id 0 today survive this cull?  (almost always: then its slot stays "id 0")
        uint32_t survive = 0;
        if (old_first < N) {
            const uint32_t w = old_first / 64u, bit = old_first % 64u;
            survive = (((zm[w] | (cm[w] & dm[w])) >> bit) & 1ull) ? 0u : 1u;     // all conflicts counted: conservative under the cap
        }
        s_keep_first = survive;
    }
    // totals of the conflict pass (per-workgroup partials instead of same-address atomics)
    uint32_t cskip = 0, cconf = 0, ckill = 0, cskip_tot, cconf_tot, ckill_tot;
    for (uint32_t b = threadIdx.x; b < n_conf_part; b += 1024u) {
        cskip += conf_part[b * 4]; cconf += conf_part[b * 4 + 1]; ckill += conf_part[b * 4 + 2];
    }
    __syncthreads();
    block_scan_1024(cskip, &cskip_tot, s_scan);
    block_scan_1024(cconf, &cconf_tot, s_scan);
    block_scan_1024(ckill, &ckill_tot, s_scan);
    uint32_t ctotal = cconf_tot, ktotal = (N - g0) - ckill_tot, gkpre = 0;
    uint32_t nstatic = N;
    if (have_scan) {
        // scan the group totals of k_scan_cull (ngroups <= 1024 covers 1 G surfels)
        uint32_t gc = 0, gk = 0;
        if (threadIdx.x < ngroups) {
            gc = group_tot[threadIdx.x * 4 + 0];
            gk = group_tot[threadIdx.x * 4 + 1];
            atomicMin(&s_first, group_tot[threadIdx.x * 4 + 2]);
        }
        block_scan_1024(gc, &ctotal, s_scan);
        gkpre = block_scan_1024(gk, &ktotal, s_scan);
        nstatic = (s_first == 0xFFFFFFFFu) ? N : min(N, s_first * (uint32_t)TILE);
    }
    const bool cap_binds = ctotal > cap;
    if (!cap_binds) {
        if (have_scan && threadIdx.x < ngroups) group_keep_base[threadIdx.x] = gkpre;
    } else {
        // ---- slow path: the cap binds; exact sequential-order scan with absolute prefixes
        if (threadIdx.x < ngroups) group_keep_base[threadIdx.x] = 0;
        const uint32_t per = (ntiles + 1023u) / 1024u;
        const uint32_t t0 = min(threadIdx.x * per, ntiles), t1 = min(t0 + per, ntiles);
        uint32_t csum = 0;
        for (uint32_t t = t0; t < t1; ++t) csum += tile_cnt[t * 3];
        uint32_t dummy;
        uint32_t cpre = block_scan_1024(csum, &dummy, s_scan);
        uint32_t ksum = 0;
        for (uint32_t t = t0; t < t1; ++t) {
            const uint32_t nconf = tile_cnt[t * 3];
            uint32_t allow = nconf;
            if (cpre >= cap) allow = 0;
            else if (cap - cpre < nconf) allow = cap - cpre;
            tile_allow[t] = allow;
            uint32_t kills;
            if (allow == nconf) kills = tile_cnt[t * 3 + 1];
            else if (allow == 0) kills = tile_cnt[t * 3 + 2];
            else {   // the one tile straddling the cap
                kills = 0;
                uint32_t rem = allow;
                for (int w = 0; w < TILE_WORDS; ++w) {
                    const uint32_t word = t * TILE_WORDS + w;
                    if ((uint64_t)word * 64u >= N) break;
                    const uint64_t c = cm[word];
                    const uint64_t ce = first_n_bits(c, rem);
                    rem -= (uint32_t)__popcll(ce);
                    kills += (uint32_t)__popcll(zm[word] | (ce & dm[word]));
                }
            }
            tile_keep_prefix[t] = kills;          // parked: rewritten with the prefix below
            ksum += min((uint32_t)TILE, N - t * TILE) - tile_dead[t] - kills;
            cpre += nconf;
        }
        uint32_t kpre = block_scan_1024(ksum, &ktotal, s_scan);
        uint32_t ns = 0;
        for (uint32_t t = t0; t < t1; ++t) {
            const uint32_t kills = tile_keep_prefix[t];
            const uint32_t nv = min((uint32_t)TILE, N - t * TILE) - tile_dead[t];
            if (kpre == t * TILE && kills == 0 && tile_dead[t] == 0u) ns += nv;
            else atomicMin(&s_first, t);             // first tile that moves or thins out
            tile_keep_prefix[t] = kpre;
            kpre += nv - kills;
        }
        block_scan_1024(ns, &nstatic, s_scan);
        __syncthreads();
    }
    // ---- deferred compaction: mark the dead now, move the survivors only once enough slots are dead
    const uint32_t kept = ktotal;                     // live surfels after this cull
    const uint32_t g1 = N - kept;                     // dead slots if nothing moves
    const bool compact = have_scan;
    // slot of the first survivor (the surfel the reference addresses as id 0)
    uint32_t first_live = compact ? 0u : N;
    if (!compact && kept != 0u) {
        __syncthreads();                              // s_keep_first; tile_keep_prefix of the slow path
        if (s_keep_first) {
            first_live = old_first;
        } else {
            for (uint32_t base = min(old_first, N - 1u) / TILE; base < ntiles; base += 1024u) {
                const uint32_t t = base + threadIdx.x;
                if (t < ntiles) {
                    uint32_t keep_t;
                    if (cap_binds) keep_t = ((t + 1 < ntiles) ? tile_keep_prefix[t + 1] : kept) - tile_keep_prefix[t];
                    else keep_t = min((uint32_t)TILE, N - t * TILE) - tile_dead[t] - tile_cnt[t * 3 + 1];
                    if (keep_t != 0u) atomicMin(&s_ft, t);
                }
                __syncthreads();
                const uint32_t found = s_ft;
                __syncthreads();
                if (found != 0xFFFFFFFFu) break;
            }
            const uint32_t ft = s_ft;
            if (ft != 0xFFFFFFFFu && threadIdx.x < TILE_WORDS) {
                const uint32_t nconf = tile_cnt[ft * 3];
                const uint64_t k = keep_word(ft, (int)threadIdx.x, N, cm, dm, zm, alive, cap_binds ? tile_allow[ft] : nconf, nconf);
                if (k) atomicMin(&s_fl, (ft * TILE_WORDS + threadIdx.x) * 64u + (uint32_t)(__ffsll((long long)k) - 1));
            }
            __syncthreads();
            first_live = s_fl;
        }
    }
    if (threadIdx.x == 0) {
        if (!fp.maintenance) {
            st->n_conf_skipped = cskip_tot;
            st->n_static = compact ? nstatic : N;
            st->n_kill = (N - g0) - kept;
            st->conflict_count = min(ctotal, cap);
            if (fp.splat_follows) st->visible_count = 0;
        }
        st->cull_n = N;
        st->cull_src = st->cur;
        st->cull_dst = st->cur;                           // compaction is in place
        st->garbage_prev = g0;
        st->cap_binds = cap_binds ? 1u : 0u;
        st->do_compact = compact ? 1u : 0u;
        st->first_moving = (compact && s_first != 0xFFFFFFFFu) ? min(s_first, ntiles) : ntiles;
        st->compact_ticket = 0u;
        st->first_live = first_live;
        st->holes_last = 0u;
        if (compact) {
            st->count = kept;                             // src/GlobalModel.cpp:575
            st->offset = fp.maintenance ? old_offset - (g0 - holes) : kept;
            st->garbage = 0;
        } else {
            st->count = N;                                // the dead keep their slots until the next compaction
            st->offset = N;
            st->garbage = g1;
        }
        // host-visible (pinned) statistic: occupied slots, tagged with the number of completed appends, so that the host
        // can bound the slot count of a frame it enqueues without waiting for the device
        if (host_stat)
            __hip_atomic_store(host_stat, ((unsigned long long)st->stat_frames << 32) | (unsigned long long)(compact ? kept : N),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// after a physical compaction every slot below the new count is live again: refill the alive mask and clear
// the per-tile dead counts over the range the model occupied before (grid-stride, called by the frame's last
// kernel or by k_post_fill)
__device__ __forceinline__ void post_compact_fill(const DevState *__restrict__ st, uint64_t *__restrict__ alive,
                                                  uint32_t *__restrict__ tile_dead, uint32_t tid, uint32_t nthreads)
{
    if (st->do_compact == 0u || st->garbage_prev == 0u) return;
    const uint32_t n = st->cull_n;
    const uint32_t nwords = (n + 63u) / 64u, ntiles = (n + TILE - 1) / TILE;
    for (uint32_t w = tid; w < nwords; w += nthreads) alive[w] = ~0ull;
    for (uint32_t t = tid; t < ntiles; t += nthreads) tile_dead[t] = 0u;
}

__global__ void k_post_fill(const DevState *__restrict__ st, uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead)
{
    post_compact_fill(st, alive, tile_dead, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ids of the key map: slot -> position among the live surfels (before a compaction outside a frame moves them)
__global__ void k_remap_keys(const DevState *__restrict__ st, uint64_t *__restrict__ keyT, int P,
                             const uint64_t *__restrict__ alive, const uint32_t *__restrict__ tile_keep_prefix,
                             const uint32_t *__restrict__ group_keep_base)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P || st->garbage_prev == 0u) return;
    const uint64_t key = keyT[q];
    if (key == KEY_EMPTY) return;
    const uint32_t id = (uint32_t)(key & 0xFFFFFFFFull);
    if (id >= st->cull_n) return;
    const uint32_t tile = id / TILE, w = (id % TILE) / 64u, bit = id % 64u;
    uint32_t nid = tile_keep_prefix[tile] + group_keep_base[tile / GROUP];
    for (uint32_t x = 0; x < w; ++x) nid += (uint32_t)__popcll(alive[tile * TILE_WORDS + x]);
    nid += (uint32_t)__popcll(alive[tile * TILE_WORDS + w] & ((1ull << bit) - 1ull));
    keyT[q] = (key & 0xFFFFFFFF00000000ull) | (uint64_t)nid;
}

// z-buffered 1-px splat of one surfel (index_map.vert:38-64, index_map.frag:31-37;
// rasterisation + GL_LESS on a 24-bit depth: SURVEY.md A3/A4): 64-bit atomicMin of d24<<32|id.
__device__ __forceinline__ bool splat_one(const FrameParams &fp, float x, float y, float z, float t_last,
                                          uint32_t id, uint64_t *__restrict__ keyT)
{
    const float3 ph = xform3(fp.t_inv, x, y, z);
    if (ph.z >= fp.depth_cutoff * 1.5f || ph.z <= 0.0f || (float)fp.time - t_last > (float)fp.time_delta)
        return false;
    const float xn = ((((fp.fx * ph.x) / ph.z) + fp.cx) - (fp.cols * 0.5f)) / (fp.cols * 0.5f);
    const float yn = ((((fp.fy * ph.y) / ph.z) + fp.cy) - (fp.rows * 0.5f)) / (fp.rows * 0.5f);
    const float zn = ph.z / fp.depth_cutoff;
    if (!(xn >= -1.0f && xn <= 1.0f && yn >= -1.0f && yn <= 1.0f && zn >= -1.0f && zn <= 1.0f)) return false;
    const float xw = (fp.cols * 0.5f) * xn + (fp.cols * 0.5f);
    const float yw = (fp.rows * 0.5f) * yn + (fp.rows * 0.5f);
    const float fxw = floorf(xw), fyw = floorf(yw);
    if (!(fxw >= 0.0f && fxw < fp.cols && fyw >= 0.0f && fyw < fp.rows)) return false;
    const int px = (int)fxw, py = (int)fyw;
    const float zw = 0.5f * zn + 0.5f;
    const uint32_t d24 = (uint32_t)floor((double)zw * 16777215.0 + 0.5);
    if (d24 >= 16777215u) return false;
    const uint64_t key = ((uint64_t)d24 << 32) | (uint64_t)id;
    atomicMin((unsigned long long *)&keyT[(size_t)px * fp.H + py], (unsigned long long)key);
    return true;
}

// ---------------------------------------------------------------------------------------------
// p3+p4+p5(+p6): apply the confidence decrement, stable-compact the survivors IN PLACE and, fused,
// splat each survivor under its NEW id.
//
// In-place stable compaction across workgroups: a survivor never moves to a higher index, so the
// destination range [prefix, prefix+kept) of tile t lies inside the source regions of tiles <= t.
// Every tile that moves or loses surfels first loads all its survivors into registers, then
// publishes tile_flag[t] = epoch ("my source is consumed"), then waits for the flags of the (at
// most two) lower tiles its destination overlaps, then writes.  Tiles with nothing killed in or
// before them are "static": they copy nothing (only the decremented confidences are written), so
// the part of the model the camera has left behind costs 20 B/surfel instead of 88.
// Deadlock-freedom: a tile only waits for lower-numbered tiles, a tile publishes before it waits,
// tiles are assigned round-robin to a grid that is fully co-resident (<= 4 workgroups per CU).
// Flag protocol: agent-scope atomic exchange to publish, sc1 (agent-scope relaxed) load to poll
// (MI355X_MICROARCH.md "hand-offs measured", row 3); nothing but the flag itself is handed over.
// ---------------------------------------------------------------------------------------------
template <bool SPLAT>
__global__ __launch_bounds__(256) void k_compact(Model M, DevState *__restrict__ st, FrameParams fp,
                                                 const uint64_t *__restrict__ cm,
                                                 const uint64_t *__restrict__ dm,
                                                 const uint64_t *__restrict__ zm,
                                                 const uint32_t *__restrict__ tile_cnt,
                                                 const uint32_t *__restrict__ tile_allow,
                                                 const uint32_t *__restrict__ tile_keep_prefix,
                                                 uint64_t *__restrict__ keyT,
                                                 uint32_t *__restrict__ tile_flag, uint32_t epoch,
                                                 const uint32_t *__restrict__ seg_lstart,
                                                 const uint32_t *__restrict__ seg_gbase,
                                                 const uint32_t *__restrict__ group_keep_base,
                                                 uint32_t *__restrict__ tb, const uint8_t *__restrict__ tile_flags,
                                                 uint2 *__restrict__ blk_part /* [grid] (visible, splat-skipped) */,
                                                 uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead)
{
    __shared__ uint64_t s_keep[TILE_WORDS], s_ceff[TILE_WORDS];
    __shared__ uint32_t s_cpop[TILE_WORDS], s_kpre[TILE_WORDS + 1];
    __shared__ uint32_t s_vis[4];
    const uint32_t N = st->cull_n;
    const SurfelSet set = M.s[st->cull_src];
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t vis = 0, skipped = 0, iter = 0;
    uint64_t skipmask = 0;
    uint32_t m_nconf = 0, m_nkill = 0, m_allow = 0, m_base = 0, m_dead = 0;
    // Deferred compaction: this kernel runs on the culls that compact; the slots left dead by the culls in between
    // (k_cull_lazy) are squeezed out together with this cull's own victims.
    const bool had_dead = st->garbage_prev != 0u;
    const bool cap_binds = st->cap_binds != 0u;
    // Two ways to share out the tiles.  Round-robin over a grid that is known to be fully resident (the default: a tile
    // only waits for lower tiles, all of which are then running).  Or, when the GPU is shared and residency cannot be
    // counted on (fp.compact_tickets): the tiles below `fm` stay in place (nothing killed or dead in or before them) and
    // go round-robin, the tiles from `fm` on -- the ones that wait for hand-off flags -- are handed out IN ORDER from a
    // ticket counter: whoever holds a ticket is running, and a running tile publishes its flag before it waits for
    // anything, so progress never depends on how many workgroups the GPU keeps resident.  (+1 returning atomic per
    // moving tile on its critical path: k_compact 50 -> 66 us at KITTI size, hence not the default.)
    const bool use_tickets = fp.compact_tickets != 0;
    const uint32_t fm = use_tickets ? min(st->first_moving, ntiles) : ntiles;
    constexpr uint32_t TICKET = 1;                    // one tile per ticket: a tile must be able to publish without first finishing a lower one
    __shared__ uint32_t s_tk;
    bool ticketing = false;
    uint32_t rr_tile = blockIdx.x, tk_tile = 0, tk_left = 0;
    for (;;) {
        uint32_t tile, allow, nconf, nkill_full, base_id, tdead;
        bool skipbit;
        if (!ticketing && rr_tile >= fm) {                                      // workgroup-uniform
            if (!use_tickets) break;
            ticketing = true;
        }
        if (!ticketing) {
            tile = rr_tile;
            if ((iter & 63u) == 0u) {
                // metadata of this workgroup's next 64 tiles in one round of loads (lane i <-> i-th tile), so that the
                // per-tile critical path holds a single memory latency (the surfel loads themselves)
                const uint64_t tl = (uint64_t)tile + (uint64_t)lane * gridDim.x;
                const bool in = tl < fm;
                const uint32_t tt = in ? (uint32_t)tl : 0u;
                skipmask = __ballot(in && (tile_flags[tt] & 2u));
                m_nconf = tile_cnt[tt * 3]; m_nkill = tile_cnt[tt * 3 + 1];
                m_allow = cap_binds ? tile_allow[tt] : m_nconf;         // every conflict takes effect unless the cap binds
                m_base = tile_keep_prefix[tt] + group_keep_base[tt / GROUP];
                m_dead = had_dead ? tile_dead[tt] : 0u;
            }
            const int sl = (int)(iter & 63u);
            allow = lane_bcast(m_allow, sl); nconf = lane_bcast(m_nconf, sl);
            nkill_full = lane_bcast(m_nkill, sl); base_id = lane_bcast(m_base, sl);
            tdead = lane_bcast(m_dead, sl);
            skipbit = (skipmask >> (iter & 63u)) & 1ull;
            rr_tile += gridDim.x; ++iter;
        } else {
            if (tk_left == 0u) {
                __syncthreads();
                if (threadIdx.x == 0) s_tk = atomicAdd(&st->compact_ticket, TICKET);
                __syncthreads();
                tk_tile = fm + s_tk; tk_left = TICKET;
            }
            tile = tk_tile;
            if (tile >= ntiles) break;
            ++tk_tile; --tk_left;
            nconf = tile_cnt[tile * 3]; nkill_full = tile_cnt[tile * 3 + 1];
            allow = cap_binds ? tile_allow[tile] : nconf;
            base_id = tile_keep_prefix[tile] + group_keep_base[tile / GROUP];
            tdead = had_dead ? tile_dead[tile] : 0u;
            skipbit = false;
        }
        // fast path (workgroup-uniform): nothing of this tile conflicts, dies or moves -- the bulk of the map
        // once the camera has passed.  No masks, no LDS, no barriers: read pos+time, splat.
        if (!ticketing && nconf == 0 && nkill_full == 0 && base_id == tile * (uint32_t)TILE && tdead == 0u) {
            // ... and if its box cannot reach the index map (index_map.vert:45-55: 0 < z < far inside the image,
            // updated within timeDelta frames) it is not even read
            if (SPLAT && skipbit) {
                skipped += min((uint32_t)TILE, N - tile * TILE);
                continue;
            }
            if (SPLAT) {
                float4 pv[4];
                float pt[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t k = (tile * TILE_WORDS + r * 4 + wave) * 64u + lane;
                    const uint32_t kc = min(k, N - 1u);
                    pv[r] = set.pos_conf[kc];
                    pt[r] = set.time[kc];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t k = (tile * TILE_WORDS + r * 4 + wave) * 64u + lane;
                    bool drew = false;
                    if (k < N)
                        drew = splat_one(fp, pv[r].x, pv[r].y, pv[r].z, pt[r], local_to_global(k, seg_lstart, seg_gbase, fp.nseg), keyT);
                    vis += (uint32_t)__popcll(__ballot(drew));
                }
            }
            continue;
        }
        // conservative: a tile classified "moving" that turns out static is handled correctly (it rewrites itself)
        const bool moving = ticketing || (base_id != tile * (uint32_t)TILE) || nkill_full != 0u || tdead != 0u;   // workgroup-uniform
        // ---- issue every surfel load of the tile first (unconditional, clamped: a per-lane branch would serialise
        // them behind s_waitcnt); the mask bookkeeping below overlaps their latency
        float4 v[4], nr[4];
        uint32_t col[4], nid[4];
        float it[4], tl[4];
        bool kept[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t k = (tile * TILE_WORDS + r * 4 + wave) * 64u + lane;
            const uint32_t kc = min(k, N - 1u);
            v[r] = set.pos_conf[kc];
            tl[r] = set.time[kc];
            nr[r] = make_float4(0.f, 0.f, 0.f, 0.f); col[r] = 0; it[r] = 0.f;
            if (moving) { nr[r] = set.norm_rad[kc]; col[r] = set.color[kc]; it[r] = set.init_time[kc]; }
        }
        uint64_t c = 0, d = 0, z = 0, valid = 0;
        if (threadIdx.x < TILE_WORDS) {
            const uint32_t word = tile * TILE_WORDS + threadIdx.x;
            const uint64_t base = (uint64_t)word * 64u;
            if (base < N) {
                c = cm[word]; d = dm[word]; z = zm[word];
                const uint64_t rem = (uint64_t)N - base;
                valid = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
                if (had_dead) valid &= alive[word];
            }
            s_cpop[threadIdx.x] = (uint32_t)__popcll(c);
        }
        __syncthreads();
        if (threadIdx.x < TILE_WORDS) {
            uint64_t ce = c;
            if (allow != nconf) {
                uint32_t before = 0;
                for (int w = 0; w < (int)threadIdx.x; ++w) before += s_cpop[w];
                ce = before >= allow ? 0ull : first_n_bits(c, allow - before);
            }
            const uint64_t keep = ~(z | (ce & d)) & valid;
            s_ceff[threadIdx.x] = ce;
            s_keep[threadIdx.x] = keep;
        }
        __syncthreads();
        if (threadIdx.x <= TILE_WORDS) {
            uint32_t before = 0;
            for (int w = 0; w < (int)threadIdx.x; ++w) before += (uint32_t)__popcll(s_keep[w]);
            s_kpre[threadIdx.x] = before;                 // s_kpre[TILE_WORDS] = survivors of the tile
        }
        __syncthreads();
        const uint32_t kcount = s_kpre[TILE_WORDS];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int w = r * 4 + wave;
            const uint64_t keepw = s_keep[w];
            kept[r] = (keepw >> lane) & 1ull;
            nid[r] = base_id + s_kpre[w] + (uint32_t)__popcll(keepw & ((1ull << lane) - 1ull));
            if (kept[r] && ((s_ceff[w] >> lane) & 1ull)) {
                v[r].w -= 1.0f;                           // conflict.vert:72
                if (!moving) set.pos_conf[(tile * TILE_WORDS + w) * 64u + lane].w = v[r].w;
            }
        }
        if (moving) {
            // This tile's slot is rewritten by the compaction (by this or a higher tile): empty its bounds entry now.
            // Atomic (memory-side) stores, completed by the wait below, so that the atomicMax of any later writer --
            // which first waits for this tile's flag -- is ordered after them on every XCD.
            if (threadIdx.x < 8) atomicExch(&tb[(size_t)tile * 8 + threadIdx.x], 0u);
            // all loads (and the reset) of this workgroup have completed before the flag goes out
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_exchange(&tile_flag[tile], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (kcount && base_id != tile * TILE) {
                    const uint32_t first = base_id / TILE;
                    const uint32_t last = min(tile - 1u, (base_id + kcount - 1u) / TILE);
                    for (uint32_t t = first; t <= last; ++t) {
                        uint32_t spins = 0;
                        while (__hip_atomic_load(&tile_flag[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > (1u << 24)) { st->error = -6; break; }   // SM_E_STALL: never hang the GPU
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kept[r]) {
                    set.pos_conf[nid[r]] = v[r];
                    set.norm_rad[nid[r]] = nr[r];
                    set.color[nid[r]] = col[r];
                    set.init_time[nid[r]] = it[r];
                    set.time[nid[r]] = tl[r];
                }
                bounds_expand_wave(tb, kept[r], nid[r] / (uint32_t)TILE, v[r].x, v[r].y, v[r].z, tl[r], false);
            }
        }
        if (SPLAT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool drew = false;
                if (kept[r])
                    drew = splat_one(fp, v[r].x, v[r].y, v[r].z, tl[r], local_to_global(nid[r], seg_lstart, seg_gbase, fp.nseg), keyT);
                vis += (uint32_t)__popcll(__ballot(drew));
            }
        }
        __syncthreads();
    }
    if (SPLAT) {
        if (lane == 0) s_vis[wave] = vis;
        __syncthreads();
        if (threadIdx.x == 0)      // per-workgroup partials, summed by the append kernel (no same-address atomics)
            blk_part[blockIdx.x] = make_uint2(s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3], skipped);
    }
}

// ---------------------------------------------------------------------------------------------
// The cull of a frame that does not compact (deferred compaction): survivors keep their slots, so nothing depends
// on other tiles or even on the other words of a tile.  Each wave settles four 64-surfel words on its own -- apply
// the confidence decrement, clear the dead from the alive mask, splat the survivors under their slot number -- with
// no LDS, no barriers, no hand-off, and a quarter of k_compact's registers (twice its occupancy).  The grid needs no
// co-residency.  (Slots are ids here: the order of slots is the order of ids, which is all the key map needs.)
// ---------------------------------------------------------------------------------------------
template <bool SPLAT>
__global__ __launch_bounds__(256) void k_cull_lazy(Model M, const DevState *__restrict__ st, FrameParams fp,
                                                   const uint64_t *__restrict__ cm, const uint64_t *__restrict__ dm,
                                                   const uint64_t *__restrict__ zm, const uint32_t *__restrict__ tile_cnt,
                                                   const uint32_t *__restrict__ tile_allow, uint64_t *__restrict__ keyT,
                                                   const uint8_t *__restrict__ tile_flags,
                                                   uint2 *__restrict__ blk_part /* [grid] (visible, splat-skipped) */,
                                                   uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead)
{
    __shared__ uint32_t s_vis[4];
    const uint32_t N = st->cull_n;
    const SurfelSet set = M.s[st->cull_src];
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool had_dead = st->garbage_prev != 0u, cap_binds = st->cap_binds != 0u;
    uint32_t vis = 0, skipped = 0, iter = 0;
    uint64_t skipmask = 0;
    uint32_t m_nconf = 0, m_nkill = 0, m_dead = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++iter) {
        if ((iter & 63u) == 0u) {                    // metadata of this workgroup's next 64 tiles, one per lane
            const uint64_t tl = (uint64_t)tile + (uint64_t)lane * gridDim.x;
            const bool in = tl < ntiles;
            const uint32_t tt = in ? (uint32_t)tl : 0u;
            skipmask = __ballot(in && (tile_flags[tt] & 2u));
            m_nconf = tile_cnt[tt * 3]; m_nkill = tile_cnt[tt * 3 + 1];
            m_dead = had_dead ? tile_dead[tt] : 0u;
        }
        const int sl = (int)(iter & 63u);
        const uint32_t nconf = lane_bcast(m_nconf, sl), nkill = lane_bcast(m_nkill, sl);
        const uint32_t tdead = lane_bcast(m_dead, sl);
        const uint32_t tn = min((uint32_t)TILE, N - tile * TILE);
        const bool touched = nconf != 0u || nkill != 0u;                                  // workgroup-uniform
        const bool nosplat = !SPLAT || ((skipmask >> (iter & 63u)) & 1ull);              // box outside the index map's view
        if (SPLAT && nosplat) skipped += tn;
        if (!touched && nosplat) continue;                                                // the bulk of the map: not even read
        float4 pv[4];
        float pt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                // unconditional, clamped: all loads of the lane in flight together
            const uint32_t kc = min((tile * TILE_WORDS + r * 4 + wave) * 64u + lane, N - 1u);
            pv[r] = set.pos_conf[kc];
            pt[r] = nosplat ? 0.0f : set.time[kc];
        }
        const uint32_t allow = (touched && cap_binds) ? tile_allow[tile] : nconf;
        uint32_t killed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int w = r * 4 + wave;
            const uint32_t word = tile * TILE_WORDS + (uint32_t)w;
            const uint32_t k = word * 64u + lane;
            const uint64_t base = (uint64_t)word * 64u;
            uint64_t range = 0ull;
            if (base < N) { const uint64_t rem = (uint64_t)N - base; range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull); }
            const uint64_t valid = range & ((tdead != 0u) ? alive[word] : ~0ull);
            uint64_t keep = valid, ce = 0ull;
            if (touched) {
                ce = cm[word];
                if (allow != nconf) {                // the conflict cap binds inside this tile: only its first `allow` conflicts count
                    uint32_t before = 0;
                    for (int x = 0; x < w; ++x) before += (uint32_t)__popcll(cm[tile * TILE_WORDS + x]);
                    ce = before >= allow ? 0ull : first_n_bits(ce, allow - before);
                }
                keep = ~(zm[word] | (ce & dm[word])) & valid;
                if (keep != valid && lane == 0) alive[word] = keep | ~range;            // the dead keep their slots
                killed += (uint32_t)__popcll(valid ^ keep);
            }
            const bool kp = (keep >> lane) & 1ull;
            if (kp && ((ce >> lane) & 1ull)) set.pos_conf[k].w = pv[r].w - 1.0f;       // conflict.vert:72
            if (!nosplat) {                                                              // workgroup-uniform
                bool drew = false;
                if (kp) drew = splat_one(fp, pv[r].x, pv[r].y, pv[r].z, pt[r], k, keyT);
                vis += (uint32_t)__popcll(__ballot(drew));
            }
        }
        if (killed && lane == 0) atomicAdd(&tile_dead[tile], killed);
    }
    if (SPLAT) {
        if (lane == 0) s_vis[wave] = vis;
        __syncthreads();
        if (threadIdx.x == 0) blk_part[blockIdx.x] = make_uint2(s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3], skipped);
    }
}

// ---------------------------------------------------------------------------------------------
// k_cull_lazy for the frame path with the finalize step folded in (one launch fewer on 7 of 8 frames).  A cull that
// only marks the dead needs nothing global except "does the W*H conflict cap bind?", which every wave reads from the
// 64 conflict sub-counters k_conflict maintains (one load per lane + a wave reduction).  One extra workgroup (the last
// of the grid, no tiles) publishes DevState; the number of kills is only known when all workgroups are done, so it
// travels as a third per-workgroup partial to the append kernel, which completes DevState::garbage / n_kill.
// DevState fields the workers read (count, cur) are not changed by a lazy cull.  If the cap binds (rare) each
// workgroup rebuilds the "first W*H conflicts" rule from prefix sums of the tile counts.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cull_lazy_frame(Model M, DevState *__restrict__ st, FrameParams fp,
                                                         const uint64_t *__restrict__ cm, const uint64_t *__restrict__ dm,
                                                         const uint64_t *__restrict__ zm, const uint32_t *__restrict__ tile_cnt,
                                                         uint64_t *__restrict__ keyT, const uint8_t *__restrict__ tile_flags,
                                                         uint4 *__restrict__ blk_part /* [workers] (visible, splat-skipped, killed, -) */,
                                                         uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead,
                                                         const uint32_t *__restrict__ conf_part, uint32_t n_conf_part,
                                                         const uint32_t *__restrict__ conf_sub,
                                                         unsigned long long *__restrict__ host_stat)
{
    __shared__ uint32_t s_a[4], s_b[4], s_c[4];
    __shared__ uint32_t s_fl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwg = gridDim.x - 1u;               // workers; workgroup 0 (dispatched first) publishes
    const bool publisher = blockIdx.x == 0u;
    const uint32_t wi = blockIdx.x - 1u;               // worker index
    const uint32_t ctotal = wave_sum_u32(conf_sub[lane * SUB_STRIDE]);
    const uint32_t N = st->count;                      // occupied slots: unchanged by this cull
    const SurfelSet set = M.s[st->cur];
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const uint32_t cap = fp.conflict_cap;
    const bool cap_binds = ctotal > cap;
    if (publisher) {
        const uint32_t g0 = st->garbage, old_first = st->first_live;
        uint32_t cskip = 0;
        for (uint32_t b = threadIdx.x; b < n_conf_part; b += 256u) cskip += conf_part[(size_t)b * 4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cskip += __shfl_xor(cskip, o);
        if (lane == 0) s_a[wave] = cskip;
        if (threadIdx.x == 0) s_fl = 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t cskip_tot = s_a[0] + s_a[1] + s_a[2] + s_a[3];
        // slot of the first live surfel after this cull (the reference's id 0): unchanged if that surfel survives
        uint32_t first_live = N;
        bool keep_first = false;
        if (old_first < N) {
            const uint32_t w = old_first / 64u, bit = old_first % 64u;
            keep_first = !(((zm[w] | (cm[w] & dm[w])) >> bit) & 1ull);     // all conflicts counted: conservative under the cap
        }
        if (keep_first) {
            first_live = old_first;
        } else if (N) {
            // search upwards from its tile, 16 tiles (256 words) per round; under a binding cap the allowance of a
            // tile needs the conflicts of all tiles below it
            const uint32_t t0 = min(old_first, N - 1u) / TILE;
            uint32_t before = 0;
            if (cap_binds) {
                uint32_t part = 0;
                for (uint32_t t = threadIdx.x; t < t0; t += 256u) part += tile_cnt[t * 3];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                if (lane == 0) s_c[wave] = part;
                __syncthreads();
                before = s_c[0] + s_c[1] + s_c[2] + s_c[3];
                __syncthreads();
            }
            for (uint32_t base = t0; base < ntiles; base += 16u) {
                const uint32_t t = base + (threadIdx.x >> 4);
                const int w = (int)(threadIdx.x & 15u);
                if (t < ntiles) {
                    const uint32_t nconf = tile_cnt[t * 3];
                    uint32_t allow = nconf;
                    if (cap_binds) {
                        uint32_t pre = before;
                        for (uint32_t x = base; x < t; ++x) pre += tile_cnt[x * 3];
                        allow = pre >= cap ? 0u : min(nconf, cap - pre);
                    }
                    const uint64_t k = keep_word(t, w, N, cm, dm, zm, alive, allow, nconf);
                    if (k) atomicMin(&s_fl, (t * TILE_WORDS + (uint32_t)w) * 64u + (uint32_t)(__ffsll((long long)k) - 1));
                }
                __syncthreads();
                const uint32_t found = s_fl;
                if (cap_binds && wave == 0) {          // advance the running conflict prefix by this round's 16 tiles
                    uint32_t add = 0;
                    if (lane < 16 && base + (uint32_t)lane < ntiles) add = tile_cnt[(base + (uint32_t)lane) * 3];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) add += __shfl_xor(add, o);
                    if (lane == 0) s_c[0] = add;
                }
                __syncthreads();
                if (cap_binds) before += s_c[0];
                __syncthreads();
                if (found != 0xFFFFFFFFu) { first_live = found; break; }
            }
        }
        if (threadIdx.x == 0) {
            st->n_conf_skipped = cskip_tot;
            st->n_static = N;
            st->conflict_count = min(ctotal, cap);
            if (fp.splat_follows) st->visible_count = 0;
            st->cull_n = N;
            st->cull_src = st->cur;
            st->cull_dst = st->cur;
            st->garbage_prev = g0;
            st->cap_binds = cap_binds ? 1u : 0u;
            st->do_compact = 0u;
            st->first_live = first_live;
            st->holes_last = 0u;
            st->offset = N;                             // the dead keep their slots until the next compaction
            if (host_stat)
                __hip_atomic_store(host_stat, ((unsigned long long)st->stat_frames << 32) | (unsigned long long)N, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // conflicts in all tiles below this workgroup's first one (only needed when the cap binds)
    uint32_t cpre = 0;
    if (cap_binds) {
        uint32_t part = 0;
        for (uint32_t t = threadIdx.x; t < min(wi, ntiles); t += 256u) part += tile_cnt[t * 3];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) s_c[wave] = part;
        __syncthreads();
        cpre = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        __syncthreads();
    }
    // ---- the cull proper (as k_cull_lazy<true>)
    uint32_t vis = 0, skipped = 0, killed_wg = 0, iter = 0;
    uint64_t skipmask = 0;
    uint32_t m_nconf = 0, m_nkill = 0, m_dead = 0;
    for (uint32_t tile = wi; tile < ntiles; tile += nwg, ++iter) {
        if ((iter & 63u) == 0u) {                    // metadata of this workgroup's next 64 tiles, one per lane
            const uint64_t tl = (uint64_t)tile + (uint64_t)lane * nwg;
            const bool in = tl < ntiles;
            const uint32_t tt = in ? (uint32_t)tl : 0u;
            skipmask = __ballot(in && (tile_flags[tt] & 2u));
            m_nconf = tile_cnt[tt * 3]; m_nkill = tile_cnt[tt * 3 + 1];
            m_dead = tile_dead[tt];
        }
        const int sl = (int)(iter & 63u);
        const uint32_t nconf = lane_bcast(m_nconf, sl), nkill = lane_bcast(m_nkill, sl);
        const uint32_t tdead = lane_bcast(m_dead, sl);
        const uint32_t tn = min((uint32_t)TILE, N - tile * TILE);
        uint32_t allow = nconf;
        if (cap_binds) {                             // workgroup-uniform
            allow = cpre >= cap ? 0u : min(nconf, cap - cpre);
            uint32_t part = 0;                       // conflicts of the tiles up to this workgroup's next one
            for (uint32_t t = tile + threadIdx.x; t < min(tile + nwg, ntiles); t += 256u) part += tile_cnt[t * 3];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
            __syncthreads();
            if (lane == 0) s_c[wave] = part;
            __syncthreads();
            cpre += s_c[0] + s_c[1] + s_c[2] + s_c[3];
        }
        const bool touched = nconf != 0u || nkill != 0u;                                  // workgroup-uniform
        const bool nosplat = ((skipmask >> (iter & 63u)) & 1ull) != 0ull;                // box outside the index map's view
        if (nosplat) skipped += tn;
        if (!touched && nosplat) continue;                                                // the bulk of the map: not even read
        float4 pv[4];
        float pt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                // unconditional, clamped: all loads of the lane in flight together
            const uint32_t kc = min((tile * TILE_WORDS + r * 4 + wave) * 64u + lane, N - 1u);
            pv[r] = set.pos_conf[kc];
            pt[r] = nosplat ? 0.0f : set.time[kc];
        }
        uint32_t killed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int w = r * 4 + wave;
            const uint32_t word = tile * TILE_WORDS + (uint32_t)w;
            const uint32_t k = word * 64u + lane;
            const uint64_t base = (uint64_t)word * 64u;
            uint64_t range = 0ull;
            if (base < N) { const uint64_t rem = (uint64_t)N - base; range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull); }
            const uint64_t valid = range & ((tdead != 0u) ? alive[word] : ~0ull);
            uint64_t keep = valid, ce = 0ull;
            if (touched) {
                ce = cm[word];
                if (allow != nconf) {                // the conflict cap binds inside this tile: only its first `allow` conflicts count
                    uint32_t before = 0;
                    for (int x = 0; x < w; ++x) before += (uint32_t)__popcll(cm[tile * TILE_WORDS + x]);
                    ce = before >= allow ? 0ull : first_n_bits(ce, allow - before);
                }
                keep = ~(zm[word] | (ce & dm[word])) & valid;
                if (keep != valid && lane == 0) alive[word] = keep | ~range;            // the dead keep their slots
                killed += (uint32_t)__popcll(valid ^ keep);
            }
            const bool kp = (keep >> lane) & 1ull;
            if (kp && ((ce >> lane) & 1ull)) set.pos_conf[k].w = pv[r].w - 1.0f;       // conflict.vert:72
            if (!nosplat) {                                                              // workgroup-uniform
                bool drew = false;
                if (kp) drew = splat_one(fp, pv[r].x, pv[r].y, pv[r].z, pt[r], k, keyT);
                vis += (uint32_t)__popcll(__ballot(drew));
            }
        }
        if (killed && lane == 0) atomicAdd(&tile_dead[tile], killed);
        killed_wg += killed;
    }
    __syncthreads();
    if (lane == 0) { s_a[wave] = vis; s_b[wave] = killed_wg; }
    __syncthreads();
    if (threadIdx.x == 0)
        blk_part[wi] = make_uint4(s_a[0] + s_a[1] + s_a[2] + s_a[3], skipped, s_b[0] + s_b[1] + s_b[2] + s_b[3], 0u);
}

// ---------------------------------------------------------------------------------------------
// ONE pass over the surfels per frame (p2 + p3 + p4 + p6 of a frame whose cull only marks the dead):
// conflict test (conflict.vert:25-83, conflict.geom:13-24), confidence decrement (conflict.vert:72,
// update_conf.vert:11-27), cull (back_map.geom:15-28: the dead keep their slots) and index-map splat
// (index_map.vert:38-64) from ONE load of pos_conf and ONE world->camera transform per surfel.
//
// The "first W*H conflicts only" rule (conflictVbo holds W*H records, src/GlobalModel.cpp:54-57) needs the
// conflict total, which exists only after the pass: the pass therefore treats EVERY conflict as effective and
// leaves what k_pass_fixup needs to take the surplus back, exactly, should the cap bind:
//   cm[word]        conflicts of the 64 slots of `word` (valid, not the id-0 surfel)
//   km[word]        slots this pass killed BECAUSE of a conflict (alive and conf > 0 before, conf - 1 <= 0)
//   wave_cnt[tile]  conflicts per 256-slot quarter of the tile (uint4; one word per wave, no barrier)
//   undo[slot]      the confidence a surviving, decremented surfel had before (restoring by +1.0f would
//                   not be exact for every float)
// Each wave settles four consecutive 64-slot words on its own: no LDS, no barrier inside a tile.
// ---------------------------------------------------------------------------------------------
// data.vert:33-52,87-88: is pixel q a candidate (a valid measurement on the checkerboard)?  Exactly the tests local_surfel
// applies before it does any arithmetic (frame path, i.e. not the raw cloud of the frame after reset()).
__device__ __forceinline__ bool candidate_pixel(int q, const FrameParams &fp, const float *__restrict__ depthT,
                                                const float *__restrict__ xs, const float *__restrict__ ys)
{
    // branch-free, every load unconditional (q is in range): a caller's unrolled loop keeps all of them in flight
    const int H = fp.H, W = fp.W;
    const int i = q / H, j = q - i * H;
    const float z = depthT[q];
    const float zl = depthT[i > 0 ? q - H : q];
    const float zu = depthT[j > 0 ? q - 1 : q];
    const float zr = depthT[i < W - 1 ? q + H : q];
    const float zd = depthT[j < H - 1 ? q + 1 : q];
    const int par = ((int)xs[i] + (int)ys[j]) % 2;
    return (zl != 0.0f) & (zu != 0.0f) & (zr != 0.0f) & (zd != 0.0f) & (z > fp.min_depth) & (z < fp.max_depth) & (par == 1);
}

// Candidate pixels per association block (256 pixels) and per group of CAND_GROUP (4, 8 or 16) blocks, counted by the otherwise idle
// worker workgroups of k_pass_fixup (they only depend on the frame).  (Inside k_surfel_pass, as extra workgroups, the
// counting cost that kernel its register allocation: 194 v_readlane SGPR spills, 16.5 -> 19.5 us.)  With them every
// candidate pixel owns a model slot before the association runs: slot = offset + (candidates before it in pixel order).
// k_associate_direct writes new surfels straight there -- the order of the reference's append (src/GlobalModel.cpp:67-74,
// unstable.vert) with no count that depends on the association itself -- and marks the slots of pixels that fuse instead
// as dead, which the deferred compaction squeezes out like any other dead slot.

template <int CAND_GROUP>
__device__ __forceinline__ void cand_count_block(uint32_t cg, const FrameParams &fp, const float *__restrict__ depthT,
                                                 const float *__restrict__ xs, const float *__restrict__ ys, int nblocks,
                                                 uint32_t *__restrict__ blk_cand, uint32_t *__restrict__ grp_cand)
{
    __shared__ uint32_t s_w[CAND_GROUP][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool c[CAND_GROUP];
#pragma unroll
    for (int k = 0; k < CAND_GROUP; ++k) {
        const int q = ((int)cg * CAND_GROUP + k) * PIX_BLOCK + (int)threadIdx.x;
        c[k] = candidate_pixel(min(q, fp.P - 1), fp, depthT, xs, ys) & (q < fp.P);
    }
#pragma unroll
    for (int k = 0; k < CAND_GROUP; ++k) {
        const uint64_t m = __ballot(c[k]);
        if (lane == 0) s_w[k][wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t v = 0;
        const int b = (int)cg * CAND_GROUP + lane;
        if (lane < CAND_GROUP) v = s_w[lane][0] + s_w[lane][1] + s_w[lane][2] + s_w[lane][3];
        if (lane < CAND_GROUP && b < nblocks) blk_cand[b] = v;
        v = wave_sum_u32(v);
        if (lane == 0) grp_cand[cg] = v;
    }
}

struct PassAcc { uint32_t vis, killed, nconf; };

// ---------------------------------------------------------------------------------------------
// A tile is settled with its lanes COMPACTED first.  The exact tests cost ~350 IEEE-exact VALU instructions per surfel, and a
// wave pays them for a whole 64-slot word as soon as ONE of its lanes is in view -- on a KITTI frame 61 % of the lanes of such
// words are, on the 20 M-surfel stress model (uniformly scattered surfels: 97 % of the words hold a surfel in view, 3.5 of 64
// lanes on average) 5 %.  So the workgroup first runs a cheap test over a tile's 1 024 slots (one load, the 3x4 transform, one
// v_rcp_f32 and six compares per slot) that rejects only what BOTH exact view tests are certain to reject, collects the
// slots of the rest in an LDS list -- over SEVERAL tiles while they fit (round 3: with ~50 listed slots per tile on the
// scattered model the exact phase was one wave's dependent chain per tile; batched, 8 tiles share it) -- and then runs the
// exact per-surfel code over that dense list, one entry per thread and round (two entries: 91 VGPRs instead of 75 and a
// workgroup less per CU; measured slower at both sizes).  Bit-exact by construction: the pre-test is a strict superset (2-pixel margin against a
// <= 1e-3-pixel difference between x * rcp(z) and the correctly rounded quotient; every comparison is written so that a NaN
// does NOT reject; a surfel with conf <= 0, which dies wherever it is, is always kept), the masks are assembled with LDS
// atomicOr instead of ballots, and every global side effect (undo, confidence, alive, key map, counters) is per slot or a
// sum.  Workgroup-uniform control flow; one barrier per tile plus three per flush.
// ---------------------------------------------------------------------------------------------
constexpr int PASS_BATCH = 8;            // tiles whose compacted lanes may share one run of the exact tests

struct PassLds {
    float4 pos[TILE];                    // (x, y, z, confidence) of the listed slots, parked by phase A: phase B starts without a global round trip
    uint32_t list[TILE];                 // listed slots: (tile's index in the batch << 10) | slot within its tile
    uint32_t n;                          // entries
    uint32_t pend[3];                    // entries the tile at hand wants to add (rotating: a counter is reset two tiles after its use)
    uint32_t btile[PASS_BATCH], bflag[PASS_BATCH], drew[PASS_BATCH];          // tiles of the batch; their flags (1: outside the conflict volume, 2: cannot reach the index map, 4: holds dead slots)
    uint32_t cm[PASS_BATCH][2 * TILE_WORDS], km[PASS_BATCH][2 * TILE_WORDS], gone[PASS_BATCH][2 * TILE_WORDS];   // per word (lo, hi): conflicts; killed by a conflict; removed (dead | conflict & dies)
};

// ---- phase B + the tiles' bookkeeping for the `nb` tiles of the batch (workgroup-uniform; leaves the list and the masks empty)
__device__ __forceinline__ void pass_flush(const SurfelSet &set, DevState *__restrict__ st, const FrameParams &fp,
                                           const uint2 *__restrict__ dcT, uint64_t *__restrict__ cm, uint64_t *__restrict__ km,
                                           uint4 *__restrict__ wave_cnt, uint64_t *__restrict__ alive,
                                           uint32_t *__restrict__ tile_dead, uint64_t *__restrict__ keyT, float *__restrict__ undo,
                                           uint32_t N, uint32_t exempt, uint32_t nb, uint32_t wave, int lane, PassAcc &acc,
                                           uint32_t *__restrict__ tb, PassLds &L)
{
    float4 *__restrict__ pc = set.pos_conf;
    const uint32_t tid = threadIdx.x;
    __syncthreads();                                   // the list, the batch table
    const uint32_t n_act = L.n;
    // the exact tests (pass_words / splat_one, per lane) over the dense list, two entries per thread at a time
    // (a single round of four entries per thread, staged so that a full tile pays each round trip once, was measured: 79
    // VGPRs, 64 spilled scalars, and slower at every size but the smallest)
    uint32_t my_vis = 0;
    for (uint32_t b0 = 0; b0 + wave * 64u < n_act; b0 += 256u) {      // wave-uniform (no barrier inside): a wave without entries is through
        bool has[1], sk0[1], sk1[1];
        uint32_t sl[1], k[1], bi[1];
        float4 e[1];
        float pt[1];
#pragma unroll
        for (int r = 0; r < 1; ++r) {
            const uint32_t idx = b0 + (uint32_t)r * 256u + tid;
            has[r] = idx < n_act;
            const uint32_t ent = L.list[min(idx, n_act - 1u)];
            bi[r] = ent >> 10; sl[r] = ent & 1023u;
            const uint32_t fl = L.bflag[bi[r]];
            sk0[r] = (fl & 1u) != 0u; sk1[r] = (fl & 2u) != 0u;
            k[r] = L.btile[bi[r]] * (uint32_t)TILE + sl[r];
            e[r] = L.pos[min(idx, n_act - 1u)];
            pt[r] = sk1[r] ? 0.0f : set.time[k[r]];
        }
        bool conf[1], kp[1];
        float zc[1], lam[1];
        uint32_t qq[1];
        bool inview[1];
#pragma unroll
        for (int r = 0; r < 1; ++r) {
            inview[r] = false; zc[r] = 0.f; lam[r] = 0.f; qq[r] = 0u; conf[r] = false;
            if (has[r] && !sk0[r]) {
                const float3 ph = xform3(fp.t_inv, e[r].x, e[r].y, e[r].z);
                if (!(ph.z <= fp.min_depth || ph.z >= fp.max_depth)) {
                    const float xl = ph.x / ph.z;
                    const float yl = ph.y / ph.z;
                    const float u = fp.fx * xl + fp.cx;
                    const float vv = fp.fy * yl + fp.cy;
                    if (!(u < fp.stereo_border || u > fp.cols || vv < 0.0f || vv > fp.rows)) {
                        const int ti = tex_idx(u / fp.cols, fp.W), tj = tex_idx(vv / fp.rows, fp.H);
                        qq[r] = (uint32_t)(ti * fp.H + tj);
                        lam[r] = sqrtf((xl * xl + yl * yl) + 1.0f);
                        zc[r] = ph.z;
                        inview[r] = true;
                    }
                }
            }
        }
        uint2 g[1];
#pragma unroll
        for (int r = 0; r < 1; ++r) g[r] = dcT[qq[r]];          // unconditional (pixel 0 for the others)
#pragma unroll
        for (int r = 0; r < 1; ++r) {
            if (inview[r]) {
                float depth = __uint_as_float(g[r].x);
                if ((g[r].y >> 24) == 10u) depth = fp.max_depth + 1.0f;
                if (fp.is_clean == 0 && depth == 0.0f) depth = fp.max_depth + 20.0f;
                conf[r] = (depth * lam[r] - zc[r] * lam[r] > fp.conflict_thresh * zc[r]) && (k[r] != exempt);
            }
            // (a tile outside the conflict volume holds no dead surfel either: its listed slots are all kept)
            const bool dies = has[r] && !sk0[r] && !(e[r].w - 1.0f > 0.0f);
            const bool dead = has[r] && !sk0[r] && !(e[r].w > 0.0f);
            kp[r] = has[r] && !(dead || (conf[r] && dies));
            const uint32_t wi = (sl[r] >> 6) * 2u + ((sl[r] >> 5) & 1u), bit = 1u << (sl[r] & 31u);
            if (conf[r]) atomicOr(&L.cm[bi[r]][wi], bit);
            if (conf[r] && dies && !dead) atomicOr(&L.km[bi[r]][wi], bit);
            if (has[r] && !kp[r]) {
                atomicOr(&L.gone[bi[r]][wi], bit);
                if (k[r] == exempt) st->fl_dirty = 1u;               // "id 0" died: the fixup searches its successor
            }
        }
#pragma unroll
        for (int r = 0; r < 1; ++r) {
            if (kp[r] && conf[r]) {
                undo[k[r]] = e[r].w;
                pc[k[r]].w = e[r].w - 1.0f;                              // conflict.vert:72
            }
            if (!sk1[r] && kp[r] && splat_one(fp, e[r].x, e[r].y, e[r].z, pt[r], k[r], keyT)) { ++my_vis; L.drew[bi[r]] = 1u; }
        }
    }
    acc.vis += wave_sum_u32(my_vis);
    __syncthreads();
    // ---- per tile of the batch: masks, alive words, dead count, quarter-tile conflict counts (a wave per tile, one lane per word)
    for (uint32_t b = wave; b < nb; b += 4u) {
        const uint32_t tile = L.btile[b], fl = L.bflag[b];
        uint32_t nc = 0, ng = 0;
        if (lane < TILE_WORDS) {
            const uint32_t word = tile * TILE_WORDS + (uint32_t)lane;
            const uint64_t c = (uint64_t)L.cm[b][2 * lane] | ((uint64_t)L.cm[b][2 * lane + 1] << 32);
            const uint64_t kk = (uint64_t)L.km[b][2 * lane] | ((uint64_t)L.km[b][2 * lane + 1] << 32);
            const uint64_t gone = (uint64_t)L.gone[b][2 * lane] | ((uint64_t)L.gone[b][2 * lane + 1] << 32);
            L.cm[b][2 * lane] = 0u; L.cm[b][2 * lane + 1] = 0u; L.km[b][2 * lane] = 0u; L.km[b][2 * lane + 1] = 0u;
            L.gone[b][2 * lane] = 0u; L.gone[b][2 * lane + 1] = 0u;
            if (!(fl & 1u)) { cm[word] = c; km[word] = kk; }
            if (gone) {
                const uint64_t base = (uint64_t)word * 64u;
                const uint64_t rem = (uint64_t)N - base;                 // base < N: a slot of this word was valid
                const uint64_t range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
                const uint64_t was = range & ((fl & 4u) ? alive[word] : ~0ull);
                alive[word] = (was & ~gone) | ~range;                    // the dead keep their slots
            }
            nc = (uint32_t)__popcll(c);
            ng = (uint32_t)__popcll(gone);
        }
        // conflicts per quarter tile = sums over four consecutive lanes (words)
        uint32_t q = nc;
        q += __shfl_xor(q, 1);
        q += __shfl_xor(q, 2);
        const uint32_t q0 = lane_bcast(q, 0), q1 = lane_bcast(q, 4), q2 = lane_bcast(q, 8), q3 = lane_bcast(q, 12);
        const uint32_t killed = wave_sum_u32(ng);
        if (lane == 0) {
            wave_cnt[tile] = make_uint4(q0, q1, q2, q3);
            if (killed) atomicAdd(&tile_dead[tile], killed);
            // Something of this tile went into the index map, so it can be fused in this frame: stamp the tile's box with the frame's
            // time.  (k_associate_direct leaves the time word to this kernel; "drawn at t" is never older than the last update of any
            // surfel of the tile.  The stamp also tells the next frame's tile flags -- computed while this frame's association may
            // still be moving surfels, k_assoc_prep -- which tiles not to skip: a tile that is merely visited must not keep itself
            // alive that way.)
            if (L.drew[b]) { atomicMax(&tb[(size_t)tile * 8 + 7], f2ord((float)fp.time)); L.drew[b] = 0u; }
        }
        acc.killed += killed;
        acc.nconf += q0 + q1 + q2 + q3;
    }
    if (tid == 0) L.n = 0u;
    __syncthreads();
}

// ---- phase A of one tile: the cheap superset test over its 1 024 slots (wave <-> four consecutive words, loads in flight
// together); the slots that need the exact tests join the workgroup's list, which is flushed first if they would not fit.
// `it` counts the workgroup's visited tiles; `nb` the tiles in the current batch.
__device__ __forceinline__ void pass_tile_append(const SurfelSet &set, DevState *__restrict__ st, const FrameParams &fp,
                                                 const uint2 *__restrict__ dcT, uint64_t *__restrict__ cm, uint64_t *__restrict__ km,
                                                 uint4 *__restrict__ wave_cnt, uint64_t *__restrict__ alive,
                                                 uint32_t *__restrict__ tile_dead, uint64_t *__restrict__ keyT, float *__restrict__ undo,
                                                 uint32_t N, uint32_t exempt, uint32_t tile, uint32_t wave, bool sk0, bool sk1,
                                                 bool any_dead, int lane, PassAcc &acc, uint32_t *__restrict__ tb, PassLds &L, uint32_t it,
                                                 uint32_t &nb)
{
    const float4 *__restrict__ pc = set.pos_conf;
    const uint32_t tid = threadIdx.x;
    float4 v[4];
    uint64_t valid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t k = (tile * TILE_WORDS + wave * 4u + (uint32_t)r) * 64u + (uint32_t)lane;
        v[r] = pc[min(k, N - 1u)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t word = tile * TILE_WORDS + wave * 4u + (uint32_t)r;
        const uint64_t base = (uint64_t)word * 64u;
        uint64_t range = 0ull;
        if (base < N) { const uint64_t rem = (uint64_t)N - base; range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull); }
        valid[r] = range & (any_dead ? alive[word] : ~0ull);
    }
    if (tid == 0) L.pend[(it + 1u) % 3u] = 0u;         // (last read two tiles ago: every thread has passed a barrier since)
    const float zs_max = fp.depth_cutoff * 1.5f;       // splat_one's far limit
    uint64_t m[4];
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bool act = false;
        // (a wave-uniform early-out on the camera-frame depth alone -- a third of the transform, then a ballot -- was measured:
        //  no gain at KITTI size, -5 % on the scattered 20 M-surfel model where no word is behind the camera as a whole)
        if ((valid[r] >> lane) & 1ull) {
            const float3 ph = xform3(fp.t_inv, v[r].x, v[r].y, v[r].z);
            const float rz = __builtin_amdgcn_rcpf(ph.z);
            const float ua = (fp.fx * ph.x) * rz + fp.cx, va = (fp.fy * ph.y) * rz + fp.cy;
            const bool out_img = ua < -2.0f || ua > fp.cols + 2.0f || va < -2.0f || va > fp.rows + 2.0f;      // (false for NaN)
            const bool rej_c = sk0 || ph.z <= fp.min_depth || ph.z >= fp.max_depth || out_img;              // conflict.vert:25-49 cannot pass
            const bool rej_s = sk1 || ph.z >= zs_max || ph.z <= 0.0f || out_img;                             // index_map.vert:38-64 cannot pass
            act = !rej_c || !rej_s || (!sk0 && !(v[r].w > 0.0f));
        }
        m[r] = __ballot(act);
        cnt += (uint32_t)__popcll(m[r]);
    }
    if (lane == 0 && cnt) atomicAdd(&L.pend[it % 3u], cnt);
    __syncthreads();                                   // this tile's demand; the previous tile's entries
    if (L.n + L.pend[it % 3u] > (uint32_t)TILE || nb == (uint32_t)PASS_BATCH) {       // workgroup-uniform
        pass_flush(set, st, fp, dcT, cm, km, wave_cnt, alive, tile_dead, keyT, undo, N, exempt, nb, wave, lane, acc, tb, L);
        nb = 0u;
        // (the tile's 16 KB again, from the cache: keeping them in registers across the flush cost the kernel 29 VGPRs -- 99
        //  instead of 70 -- and with them two waves per SIMD)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t k = (tile * TILE_WORDS + wave * 4u + (uint32_t)r) * 64u + (uint32_t)lane;
            v[r] = pc[min(k, N - 1u)];
        }
    }
    if (tid == 0) { L.btile[nb] = tile; L.bflag[nb] = (sk0 ? 1u : 0u) | (sk1 ? 2u : 0u) | (any_dead ? 4u : 0u); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t base = 0;
        if (lane == 0 && m[r]) base = atomicAdd(&L.n, (uint32_t)__popcll(m[r]));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((m[r] >> lane) & 1ull) {
            const uint32_t at = base + (uint32_t)__popcll(m[r] & ((1ull << lane) - 1ull));
            L.list[at] = (nb << 10) | ((wave * 4u + (uint32_t)r) * 64u + (uint32_t)lane);
            L.pos[at] = v[r];
        }
    }
    ++nb;
}

// READY = true: k_prep evaluated the tile skip flags of the frame (one byte per tile, loaded together with DevState);
// READY = false: the kernel evaluates them itself (frames whose k_prep ran before the previous frame had finished: the
// depth filter chain on the second stream).  Workgroup <-> tile round-robin, wave <-> quarter tile.
template <bool READY>
__global__ __launch_bounds__(256) void k_surfel_pass(Model M, DevState *__restrict__ st, FrameParams fp,
                                                     const uint2 *__restrict__ dcT, uint64_t *__restrict__ cm,
                                                     uint64_t *__restrict__ km, uint4 *__restrict__ wave_cnt,
                                                     uint32_t *__restrict__ tb, uint8_t *__restrict__ tile_flags,
                                                     uint4 *__restrict__ part /* [grid] (visible, splat-skipped, killed, conflict-skipped) */,
                                                     uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead,
                                                     uint32_t *__restrict__ conf_sub, uint64_t *__restrict__ keyT,
                                                     float *__restrict__ undo,
                                                     uint32_t tile_bound /* host upper bound of the number of tiles (>= 1) */,
                                                     uint32_t *__restrict__ frame_sub /* sets 0, 1: visible, killed -- sub-counters like conf_sub */,
                                                     unsigned long long *__restrict__ trace = nullptr /* SM_PASS_TRACE: 8 words per workgroup */)
{
    // (stamps go straight to memory: kept in registers until the exit they cost the kernel 30 more spilled scalars)
    unsigned long long *const tr = trace ? trace + (size_t)blockIdx.x * 8 : nullptr;
    if (tr && threadIdx.x == 0) { tr[0] = wall_clock64(); tr[1] = 0ull; tr[2] = 0ull; tr[4] = ~0ull; tr[5] = 0ull; }
    bool tr_first = true;
    // Workgroups are dispatched in blockIdx order, ~2 800 per us: the last of 2 048 enters the chip ~3 us after the first.  The
    // newest tiles -- the surfels the camera is looking at, i.e. the tiles with all the work -- are the highest ones, so the
    // mapping is reversed: block 0 takes the highest tile of the grid, and a workgroup with several tiles starts with its
    // highest (the flags of its first 64 tiles sit one per lane whatever the order).
    const uint32_t tile_grid = gridDim.x, bid = gridDim.x - 1u - blockIdx.x;
    __shared__ uint8_t s_flags[64];
    __shared__ uint32_t s_a[4], s_b[4], s_c[4];
    __shared__ PassLds s_pass;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // flags and dead counts of this workgroup's first 64 tiles: addresses known without DevState, issued with it
    uint32_t m_flag = 0, m_dead = 0;
    if (READY) {
        const uint64_t tl = min((uint64_t)bid + (uint64_t)lane * tile_grid, (uint64_t)tile_bound - 1u);
        m_flag = tile_flags[tl];
        m_dead = tile_dead[tl];
    }
    const uint32_t N = st->count;
    const uint32_t exempt = st->first_live;            // the surfel the reference addresses as id 0
    const SurfelSet set = M.s[st->cur];
    PassAcc acc = {0u, 0u, 0u};
    uint32_t sskip = 0, cskip = 0;
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    uint64_t skip0 = 0, skip1 = 0;
    const uint32_t n_it = bid < ntiles ? (ntiles - 1u - bid) / tile_grid + 1u : 0u;       // this workgroup's tiles: bid + iter * grid
    const bool desc = n_it <= 64u;                      // (all of them fit the one-per-lane flags: highest first)
    uint32_t n_visited = 0, n_batch = 0;               // tiles this workgroup has read; tiles in the current batch
    {
        for (uint32_t i = threadIdx.x; i < (uint32_t)PASS_BATCH * 2u * TILE_WORDS; i += 256u) { (&s_pass.cm[0][0])[i] = 0u; (&s_pass.km[0][0])[i] = 0u; (&s_pass.gone[0][0])[i] = 0u; }
        if (threadIdx.x < (uint32_t)PASS_BATCH) s_pass.drew[threadIdx.x] = 0u;
        if (threadIdx.x < 3u) s_pass.pend[threadIdx.x] = 0u;
        if (threadIdx.x == 0) s_pass.n = 0u;
        __syncthreads();
    }
    for (uint32_t it = 0; it < n_it; ++it) {
        const uint32_t iter = desc ? n_it - 1u - it : it;
        const uint32_t tile = bid + iter * tile_grid;
        if (desc ? it == 0u : (iter & 63u) == 0u) {
            const uint32_t tile0 = desc ? bid : tile;   // the tile lane 0's flag belongs to
            const uint64_t tl = (uint64_t)tile0 + (uint64_t)lane * tile_grid;
            uint32_t f;
            if (READY) {
                if (tile0 != bid) { m_flag = tile_flags[min(tl, (uint64_t)ntiles - 1u)]; m_dead = tile_dead[min(tl, (uint64_t)ntiles - 1u)]; }
                f = tl < ntiles ? m_flag : 3u;
            } else {
                __syncthreads();
                tile_flags_batch(tile0, tile_grid, ntiles, fp, tb, s_flags);
                __syncthreads();
                f = s_flags[lane];
                if (wave == 0 && tl < ntiles) tile_flags[tl] = (uint8_t)f;      // bit 1 is read again by the fixup's repair
                m_dead = tl < ntiles ? tile_dead[tl] : 0u;
            }
            skip0 = __ballot((f & 1u) != 0u);
            skip1 = __ballot((f & 2u) != 0u);
        }
        const int sl = (int)(iter & 63u);
        const bool sk0 = (skip0 >> sl) & 1ull, sk1 = (skip1 >> sl) & 1ull;   // workgroup-uniform
        if (!READY) {
            const uint32_t tn = min((uint32_t)TILE, N - tile * TILE);
            if (sk0) cskip += tn;
            if (sk1) sskip += tn;
        }
        if (sk0 && sk1) {                       // the bulk of the map once the camera has passed: not even read
            if (!READY && threadIdx.x == 0) wave_cnt[tile] = make_uint4(0u, 0u, 0u, 0u);
            continue;
        }
        {
            const bool tr_now = tr && tr_first;
            tr_first = false;
            if (tr_now && threadIdx.x == 0) { tr[1] = wall_clock64(); tr[4] = tile; }
            pass_tile_append(set, st, fp, dcT, cm, km, wave_cnt, alive, tile_dead, keyT, undo, N, exempt, tile, wave, sk0, sk1,
                             lane_bcast(m_dead, sl) != 0u, lane, acc, tb, s_pass, n_visited, n_batch);
            ++n_visited;
            if (tr_now && threadIdx.x == 0) { tr[2] = wall_clock64(); tr[5] = s_pass.n; }
        }
    }
    if (n_batch)                                       // workgroup-uniform
        pass_flush(set, st, fp, dcT, cm, km, wave_cnt, alive, tile_dead, keyT, undo, N, exempt, n_batch, wave, lane, acc, tb, s_pass);
    __syncthreads();
    if (lane == 0) { s_a[wave] = acc.vis; s_b[wave] = acc.killed; s_c[wave] = acc.nconf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nv = s_a[0] + s_a[1] + s_a[2] + s_a[3], nk = s_b[0] + s_b[1] + s_b[2] + s_b[3];
        part[bid] = make_uint4(nv, sskip, nk, cskip);
        const uint32_t nc = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        if (nc) atomicAdd(&conf_sub[(bid & 63u) * SUB_STRIDE], nc);      // 64 counters, <= 32 adders each: one load per lane to read the total
        if (nv) atomicAdd(&frame_sub[(bid & 63u) * SUB_STRIDE], nv);
        if (nk) atomicAdd(&frame_sub[SUB_SET + (bid & 63u) * SUB_STRIDE], nk);
        if (tr) {
            uint32_t hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            tr[3] = wall_clock64(); tr[6] = ((unsigned long long)xcc << 32) | hw; tr[7] = ntiles;
        }
    }
}

// survivors of one 64-slot word after k_surfel_pass AND its fixup: alive now, or killed by a conflict that the cap
// makes ineffective (ordinal >= cap).  `before` = conflicts in all slots below this word.
__device__ __forceinline__ uint64_t ineffective_conflicts(uint64_t c, uint32_t before, uint32_t cap)
{
    if (before >= cap) return c;
    return c & ~first_n_bits(c, cap - before);
}

// Arguments of the direct-append frame form (k_associate_direct), handed to k_pass_fixup's publisher
struct DirectArgs {
    int on;                              // 1: this frame appends directly (k_associate_direct follows; no k_append_scan)
    uint32_t *blk_cand, *grp_cand;       // out: candidate pixels per association block / per group of CAND_GROUP blocks (this frame)
    uint32_t n_grp, cg;                  // groups; association blocks per group (4, 8 or 16)
    int n_pix_blocks;
    const float *depthT, *xs, *ys;
    uint32_t *frame_sub;                 // 4 x 64 sub-counters: visible, killed (this frame's pass); new, fused (the PREVIOUS frame's association)
    const uint2 *fix_prev;               // the previous frame's k_pass_fixup partials (read if its conflict cap bound)
    uint32_t n_fix_prev;
    FrameLog *log;
};

// DevState fields of the pending frame, loaded before the reductions so that completing it costs no further round trip
struct PendFields { uint32_t cull_n, garbage_prev, n_kill, visible, conflict, n_static, conf_skipped, splat_skipped, tick, frames_logged; };

__device__ __forceinline__ void finalize_write(DevState *__restrict__ st, FrameLog *__restrict__ log, const PendFields &pf, uint32_t U,
                                               uint32_t F, uint32_t vadd, uint32_t res);

// Completes the statistics of a direct-append frame once its association has finished: new / fused totals from the
// per-block counts, the fixup's corrections if the conflict cap bound, the dead-slot total (culled + fused candidates'
// empty slots), the frame-log entry.  Executed by one 256-thread workgroup; no-op unless DevState::pend is set.
__device__ __forceinline__ void finalize_frame(DevState *__restrict__ st, uint32_t *__restrict__ frame_sub,
                                               const uint2 *__restrict__ fix_prev, uint32_t n_fix_prev, FrameLog *__restrict__ log,
                                               uint32_t *s_red /* 16 words of LDS */)
{
    if (st->pend == 0u) return;                         // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t un = 0, fu = 0, va = 0, rs = 0;
    if (wave == 0) {
        uint32_t *a = frame_sub + 2 * SUB_SET + lane * SUB_STRIDE, *b = frame_sub + 3 * SUB_SET + lane * SUB_STRIDE;
        un = *a; fu = *b; *a = 0u; *b = 0u;
    }
    if (st->cap_binds)
        for (uint32_t b = threadIdx.x; b < n_fix_prev; b += 256u) { const uint2 c = fix_prev[b]; va += c.x; rs += c.y; }
    un = wave_sum_u32(un); fu = wave_sum_u32(fu); va = wave_sum_u32(va); rs = wave_sum_u32(rs);
    __syncthreads();
    if (lane == 0) { s_red[wave] = un; s_red[4 + wave] = fu; s_red[8 + wave] = va; s_red[12 + wave] = rs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t U = s_red[0] + s_red[1] + s_red[2] + s_red[3], F = s_red[4] + s_red[5] + s_red[6] + s_red[7];
        const uint32_t vadd = s_red[8] + s_red[9] + s_red[10] + s_red[11], res = s_red[12] + s_red[13] + s_red[14] + s_red[15];
        PendFields pf;
        pf.cull_n = st->cull_n; pf.garbage_prev = st->garbage_prev; pf.n_kill = st->n_kill; pf.visible = st->visible_count;
        pf.conflict = st->conflict_count; pf.n_static = st->n_static; pf.conf_skipped = st->n_conf_skipped;
        pf.splat_skipped = st->n_splat_skipped; pf.tick = st->pend_tick; pf.frames_logged = st->frames_logged;
        finalize_write(st, log, pf, U, F, vadd, res);
    }
    __syncthreads();
}

// (one thread) the pending frame's totals into DevState and the frame log
__device__ __forceinline__ void finalize_write(DevState *__restrict__ st, FrameLog *__restrict__ log, const PendFields &pf, uint32_t U,
                                               uint32_t F, uint32_t vadd, uint32_t res)
{
    const uint32_t n_slots = pf.cull_n, g_prev = pf.garbage_prev;
    const uint32_t n_kill = pf.n_kill - res, vis = pf.visible + vadd;
    const uint32_t g_cull = g_prev + n_kill;
    st->n_kill = n_kill;
    st->visible_count = vis;
    st->garbage = g_cull + F;                       // the slots of candidate pixels that fused stay empty
    st->holes_last = F;
    st->unstable_count = U;
    st->fused_count = F;
    st->data_count = U + F;
    st->append_n = U;
    if (log) {
        FrameLog e;
        e.tick = pf.tick; e.n_before = n_slots - g_prev; e.n_after_cull = n_slots - g_cull; e.n_kill = n_kill;
        e.conflict_count = pf.conflict; e.visible_count = vis; e.fused_count = F; e.unstable_count = U;
        e.n_static = pf.n_static; e.n_conf_skipped = pf.conf_skipped; e.n_splat_skipped = pf.splat_skipped; e.n_slots = n_slots;
        log[pf.frames_logged % FRAME_LOG_LEN] = e;
        st->frames_logged = pf.frames_logged + 1;
    }
    st->n_conf_skipped = 0;
    st->pend = 0u;
}

__global__ __launch_bounds__(256) void k_frame_finalize(DevState *__restrict__ st, uint32_t *__restrict__ frame_sub,
                                                        const uint2 *__restrict__ fix_prev, uint32_t n_fix_prev, FrameLog *__restrict__ log)
{
    __shared__ uint32_t s_red[16];
    finalize_frame(st, frame_sub, fix_prev, n_fix_prev, log, s_red);
}

// ---------------------------------------------------------------------------------------------
// After k_surfel_pass: workgroup 0 publishes DevState (as k_cull_lazy_frame's publisher does); the other workgroups
// return at once unless the conflict cap binds (total > W*H: src/GlobalModel.cpp:54-57).  Then they take
// back every conflict beyond the first `cap` in slot order: a surfel the pass killed because of such a conflict is
// resurrected (alive bit, dead count, splat), a surviving one gets its confidence back from the undo plane.
// Conflict ordinals come from prefix sums of the per-quarter-tile counts.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pass_fixup(Model M, DevState *__restrict__ st, FrameParams fp,
                                                    const uint64_t *__restrict__ cm, const uint64_t *__restrict__ km,
                                                    const uint4 *__restrict__ wave_cnt, const uint8_t *__restrict__ tile_flags,
                                                    const uint4 *__restrict__ part, uint32_t n_part,
                                                    uint2 *__restrict__ fix_part /* [workers] (visible added, resurrected) */,
                                                    uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead,
                                                    const uint32_t *__restrict__ conf_sub, uint64_t *__restrict__ keyT,
                                                    const float *__restrict__ undo, unsigned long long *__restrict__ host_stat,
                                                    const uint2 *__restrict__ prep_part, uint32_t n_prep /* k_prep's skip statistics (it evaluated the tile flags), or 0 */,
                                                    DirectArgs da, uint32_t *__restrict__ tb /* tile bounds: a tile drawn only through a resurrected surfel gets the frame's time stamp too */)
{
    __shared__ uint32_t s_a[4], s_b[4], s_c[4];
    __shared__ uint32_t s_fl;
    __shared__ uint32_t s_red9[9][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nwg = gridDim.x - 1u;               // workers; workgroup 0 (dispatched first) publishes
    const uint32_t ctotal = wave_sum_u32(conf_sub[lane * SUB_STRIDE]);
    const uint32_t cap = fp.conflict_cap;
    const bool cap_binds = ctotal > cap;
    const uint32_t N = st->count;                      // occupied slots: unchanged by a cull that only marks the dead
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    if (blockIdx.x == 0u) {
        // ---- every load the publisher needs, issued together (each dependent round trip costs ~1 us on this single workgroup)
        const uint32_t pend = st->pend, cap_prev = st->cap_binds, g_in = st->garbage, old_first = st->first_live;
        const bool dirty = st->fl_dirty != 0u;
        PendFields pf;
        pf.cull_n = st->cull_n; pf.garbage_prev = st->garbage_prev; pf.n_kill = st->n_kill; pf.visible = st->visible_count;
        pf.conflict = st->conflict_count; pf.n_static = st->n_static; pf.conf_skipped = st->n_conf_skipped;
        pf.splat_skipped = st->n_splat_skipped; pf.tick = st->pend_tick; pf.frames_logged = st->frames_logged;
        uint32_t red[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // new, fused, vis+, resurrected (previous frame) | conf-skip, splat-skip, visible, killed | candidates
        // the sums the pass / the previous association left in 64 sub-counters each (one load per lane; consumed: zeroed)
        if (wave == 0) {
            uint32_t *c = da.frame_sub + lane * SUB_STRIDE;
            red[6] = c[0]; red[7] = c[SUB_SET]; red[0] = c[2 * SUB_SET]; red[1] = c[3 * SUB_SET];
            c[0] = 0u; c[SUB_SET] = 0u; c[2 * SUB_SET] = 0u; c[3 * SUB_SET] = 0u;
        }
        // (the fixup partials of the previous frame are only meaningful if its conflict cap bound: masked after the
        //  reduction, so that no load waits for DevState)
        for (uint32_t b = threadIdx.x; b < da.n_fix_prev; b += 256u) { const uint2 c = da.fix_prev[b]; red[2] += c.x; red[3] += c.y; }
        if (n_prep) for (uint32_t b = threadIdx.x; b < n_prep; b += 256u) { const uint2 c = prep_part[b]; red[4] += c.x; red[5] += c.y; }
        else for (uint32_t b = threadIdx.x; b < n_part; b += 256u) { const uint4 c = part[b]; red[4] += c.w; red[5] += c.y; }
        // ---- one round of reductions
#pragma unroll
        for (int x = 0; x < 9; ++x) red[x] = wave_sum_u32(red[x]);
        if (lane == 0) {
#pragma unroll
            for (int x = 0; x < 9; ++x) s_red9[x][wave] = red[x];
        }
        if (threadIdx.x == 0) s_fl = 0xFFFFFFFFu;
        __syncthreads();
        uint32_t tot[9];
#pragma unroll
        for (int x = 0; x < 9; ++x) tot[x] = s_red9[x][0] + s_red9[x][1] + s_red9[x][2] + s_red9[x][3];
        if (!cap_prev || !pend) { tot[2] = 0u; tot[3] = 0u; }
        // the previous frame appended directly: its statistics (incl. the dead-slot total used below) are completed first
        uint32_t g0 = g_in;
        if (pend) {
            g0 = pf.garbage_prev + (pf.n_kill - tot[3]) + tot[1];
            if (threadIdx.x == 0) finalize_write(st, da.log, pf, tot[0], tot[1], tot[2], tot[3]);
        }
        const uint32_t cskip_tot = tot[4], sskip_tot = tot[5], vis_tot = tot[6], kill_tot = tot[7];
        uint32_t first_live = old_first;
        if (dirty) {
            // The surfel that was id 0 died in the pass (conf <= 0: only an uploaded model holds such surfels).  Its
            // successor is the first slot that is alive after the fixup: alive now, or killed by a conflict beyond the cap.
            first_live = N;
            const uint32_t t0 = min(old_first, N ? N - 1u : 0u) / TILE;
            uint32_t before = 0;                       // conflicts in the tiles below the one being searched
            if (cap_binds) {
                uint32_t p = 0;
                for (uint32_t t = threadIdx.x; t < t0; t += 256u) { const uint4 c = wave_cnt[t]; p += c.x + c.y + c.z + c.w; }
                p = wave_sum_u32(p);
                if (lane == 0) s_c[wave] = p;
                __syncthreads();
                before = s_c[0] + s_c[1] + s_c[2] + s_c[3];
                __syncthreads();
            }
            for (uint32_t t = t0; t < ntiles && N; ++t) {                 // one tile per round, 16 words on 16 threads
                const uint4 c4 = wave_cnt[t];
                if (threadIdx.x < TILE_WORDS) {
                    const uint32_t word = t * TILE_WORDS + threadIdx.x;
                    const uint64_t base = (uint64_t)word * 64u;
                    if (base < N) {
                        const uint64_t rem = (uint64_t)N - base;
                        const uint64_t range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
                        uint64_t live = alive[word] & range;
                        if (cap_binds && (c4.x | c4.y | c4.z | c4.w)) {
                            uint32_t pre = before;
                            for (uint32_t x = t * TILE_WORDS; x < word; ++x) pre += (uint32_t)__popcll(cm[x]);
                            live |= km[word] & ineffective_conflicts(cm[word], pre, cap) & range;
                        }
                        if (word == old_first / 64u) live &= ~((2ull << (old_first % 64u)) - 1ull);     // strictly after the old one
                        if (live) atomicMin(&s_fl, word * 64u + (uint32_t)(__ffsll((long long)live) - 1));
                    }
                }
                __syncthreads();
                const uint32_t found = s_fl;
                __syncthreads();
                if (found != 0xFFFFFFFFu) { first_live = found; break; }
                before += c4.x + c4.y + c4.z + c4.w;
            }
        }
        if (threadIdx.x == 0) {
            st->n_conf_skipped = cskip_tot;
            st->n_splat_skipped = sskip_tot;
            st->n_static = N;
            st->conflict_count = min(ctotal, cap);
            if (fp.splat_follows) st->visible_count = 0;
            st->cull_n = N;
            st->cull_src = st->cur;
            st->cull_dst = st->cur;
            st->garbage_prev = g0;
            st->cap_binds = cap_binds ? 1u : 0u;
            st->do_compact = 0u;
            st->first_live = first_live;
            st->fl_dirty = 0u;
            st->offset = N;                             // the dead keep their slots until the next compaction
            st->holes_last = 0u;
            if (da.on) {
                // provisional totals of the pass (k_associate_direct's first block publishes the new count; the statistics
                // are completed by finalize_frame / finalize_write once the association is through)
                st->visible_count = vis_tot;
                st->n_kill = kill_tot;
                st->pend = 1u;
                st->pend_tick = (uint32_t)fp.time;
            }
            if (host_stat)
                __hip_atomic_store(host_stat, ((unsigned long long)st->stat_frames << 32) | (unsigned long long)N, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const uint32_t wi = blockIdx.x - 1u;
    // ---- direct append: the candidate pixels of the frame, per association block and per group (workgroup-uniform loop)
    if (da.on)
        for (uint32_t g = wi; g < da.n_grp; g += nwg) {          // (da.cg is uniform)
            if (da.cg == 4u) cand_count_block<4>(g, fp, da.depthT, da.xs, da.ys, da.n_pix_blocks, da.blk_cand, da.grp_cand);
            else if (da.cg == 8u) cand_count_block<8>(g, fp, da.depthT, da.xs, da.ys, da.n_pix_blocks, da.blk_cand, da.grp_cand);
            else cand_count_block<16>(g, fp, da.depthT, da.xs, da.ys, da.n_pix_blocks, da.blk_cand, da.grp_cand);
        }
    if (!cap_binds) return;
    // ---- the cap binds: take the conflicts beyond the first `cap` back
    const SurfelSet set = M.s[st->cur];
    uint32_t cpre = 0;                                  // conflicts in all tiles below this workgroup's current one
    {
        uint32_t p = 0;
        for (uint32_t t = threadIdx.x; t < min(wi, ntiles); t += 256u) { const uint4 c = wave_cnt[t]; p += c.x + c.y + c.z + c.w; }
        p = wave_sum_u32(p);
        if (lane == 0) s_c[wave] = p;
        __syncthreads();
        cpre = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        __syncthreads();
    }
    uint32_t vis = 0, resurrected = 0;
    for (uint32_t tile = wi; tile < ntiles; tile += nwg) {
        const uint4 c4 = wave_cnt[tile];
        const uint32_t nconf = c4.x + c4.y + c4.z + c4.w;
        const uint32_t tile_pre = cpre;
        {   // advance the prefix to this workgroup's next tile
            uint32_t p = 0;
            for (uint32_t t = tile + threadIdx.x; t < min(tile + nwg, ntiles); t += 256u) { const uint4 c = wave_cnt[t]; p += c.x + c.y + c.z + c.w; }
            p = wave_sum_u32(p);
            __syncthreads();
            if (lane == 0) s_c[wave] = p;
            __syncthreads();
            cpre += s_c[0] + s_c[1] + s_c[2] + s_c[3];
        }
        if (nconf == 0u || tile_pre + nconf <= cap) continue;              // every conflict of the tile is effective
        const bool nosplat = (tile_flags[tile] & 2u) != 0u;
        uint32_t wpre = tile_pre + (wave > 0 ? c4.x : 0u) + (wave > 1 ? c4.y : 0u) + (wave > 2 ? c4.z : 0u);
        uint32_t res_wave = 0, vis_tile = vis;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t word = tile * TILE_WORDS + (uint32_t)wave * 4u + (uint32_t)r;
            if ((uint64_t)word * 64u >= N) break;                             // wave-uniform
            const uint64_t c = cm[word];
            const uint64_t ineff = ineffective_conflicts(c, wpre, cap);
            wpre += (uint32_t)__popcll(c);
            if (ineff == 0ull) continue;
            const uint64_t res = ineff & km[word];                            // killed by a conflict that does not count
            const uint64_t restore = ineff & ~res & alive[word];              // survived, decremented
            const uint32_t k = word * 64u + lane;
            if ((restore >> lane) & 1ull) set.pos_conf[k].w = undo[k];
            if (res) {
                if (lane == 0) alive[word] |= res;
                res_wave += (uint32_t)__popcll(res);
                if (!nosplat) {
                    bool drew = false;
                    if ((res >> lane) & 1ull) {
                        const float4 pv = set.pos_conf[k];
                        drew = splat_one(fp, pv.x, pv.y, pv.z, set.time[k], k, keyT);
                    }
                    vis += (uint32_t)__popcll(__ballot(drew));
                }
            }
        }
        if (res_wave && lane == 0) atomicSub(&tile_dead[tile], res_wave);
        // A resurrected surfel that went into the index map can be fused by this frame's association: the tile carries the
        // frame's time stamp like a tile k_surfel_pass drew itself (pass_quarter / pass_tile_compact: "drawn at t" bounds the
        // last update of every surfel of the tile, and tells the next frame's tile flags which boxes may still grow)
        if (vis != vis_tile && lane == 0) atomicMax(&tb[(size_t)tile * 8 + 7], f2ord((float)fp.time));
        resurrected += res_wave;
    }
    __syncthreads();
    if (lane == 0) { s_a[wave] = vis; s_b[wave] = resurrected; }
    __syncthreads();
    if (threadIdx.x == 0) fix_part[wi] = make_uint2(s_a[0] + s_a[1] + s_a[2] + s_a[3], s_b[0] + s_b[1] + s_b[2] + s_b[3]);
}

// standalone p6 (IndexMap::predictIndices) over the current model
__global__ __launch_bounds__(256) void k_splat(Model M, DevState *__restrict__ st, FrameParams fp,
                                               uint64_t *__restrict__ keyT, const uint32_t *__restrict__ seg_lstart,
                                               const uint32_t *__restrict__ seg_gbase)
{
    __shared__ uint32_t s_vis[4];
    const uint32_t N = st->count;
    const SurfelSet cur = M.s[st->cur];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t vis = 0;
    const uint32_t nchunks = (N + 255u) / 256u;
    for (uint32_t b = blockIdx.x; b < nchunks; b += gridDim.x) {
        const uint32_t k = b * 256u + threadIdx.x;
        bool drew = false;
        if (k < N) {
            const float4 v = cur.pos_conf[k];
            drew = splat_one(fp, v.x, v.y, v.z, cur.time[k], local_to_global(k, seg_lstart, seg_gbase, fp.nseg), keyT);
        }
        vis += (uint32_t)__popcll(__ballot(drew));
    }
    if (lane == 0) s_vis[wave] = vis;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3];
        if (t) atomicAdd(&st->visible_count, t);
    }
}

// ---------------------------------------------------------------------------------------------
// p8 data association (data.vert:59-234) for pixel q = i*H + j.
// ---------------------------------------------------------------------------------------------
struct LocalSurfel {
    float3 pos;       // vPosLocal
    float3 nrm;       // vNormLocal
    float radius;     // radii_n
    float cr, cg, cb; // color_n
    uint32_t sem;
    float xl, yl, lambda;
};

__device__ __forceinline__ float3 get_vertex(float z, float x, float y, const FrameParams &fp, float inv_fx, float inv_fy)
{
    // geometry.glsl:5-9
    float3 r;
    r.x = (x - fp.cx) * z * inv_fx;
    r.y = (y - fp.cy) * z * inv_fy;
    r.z = z;
    return r;
}

__device__ __forceinline__ bool local_surfel(int q, const FrameParams &fp, const float *__restrict__ depthT,
                                             const uint32_t *__restrict__ rgbsT, const float *__restrict__ xs,
                                             const float *__restrict__ ys, LocalSurfel &L, int qi = -1, int qj = 0)
{
    const int H = fp.H, W = fp.W;
    const int i = qi >= 0 ? qi : q / H, j = qi >= 0 ? qj : q - i * H;     // (qi, qj): the caller knows the column / row of q already
    // init_mode: xs/ys hold the FeedbackBuffer's own pixel coordinates (src/FeedbackBuffer.cpp:47-53); they
    // follow the association tables in the same arrays at offsets W and H
    const float x = fp.init_mode ? xs[W + i] : xs[i], y = fp.init_mode ? ys[H + j] : ys[j];
    const float inv_fx = fp.init_mode ? fp.inv_fx_fb : fp.inv_fx, inv_fy = fp.init_mode ? fp.inv_fy_fb : fp.inv_fy;
    const float z = depthT[q];
    // clamp-to-edge neighbours: at the border the neighbour depth is the pixel's own (A1)
    const float zl = depthT[i > 0 ? q - H : q];
    const float zu = depthT[j > 0 ? q - 1 : q];
    const float zr = depthT[i < W - 1 ? q + H : q];
    const float zd = depthT[j < H - 1 ? q + 1 : q];
    if (fp.init_mode) {
        // surfel_feedback.vert:80-92: 0 < z < maxDepth and the checkerboard; no neighbour test
        if (!(z > 0.0f && z < fp.max_depth)) return false;
    } else {
        // checkNeighbours data.vert:33-52 + range data.vert:87
        if (zl == 0.0f || zu == 0.0f || zr == 0.0f || zd == 0.0f) return false;
        if (!(z > fp.min_depth && z < fp.max_depth)) return false;
    }
    if ((((int)x + (int)y) % 2) != 1) return false;       // data.vert:88 / surfel_feedback.vert:81
    L.xl = (x - fp.cx) * inv_fx;
    L.yl = (y - fp.cy) * inv_fy;
    L.lambda = sqrtf((L.xl * L.xl + L.yl * L.yl) + 1.0f);
    L.pos = get_vertex(z, x, y, fp, inv_fx, inv_fy);
    // getNormal geometry.glsl:12-24
    const float3 xf = get_vertex(zr, x + 1.0f, y, fp, inv_fx, inv_fy);
    const float3 xb = get_vertex(zl, x - 1.0f, y, fp, inv_fx, inv_fy);
    const float3 yf = get_vertex(zd, x, y + 1.0f, fp, inv_fx, inv_fy);
    const float3 yb = get_vertex(zu, x, y - 1.0f, fp, inv_fx, inv_fy);
    const float3 del_x = make_float3(xb.x - xf.x, xb.y - xf.y, xb.z - xf.z);
    const float3 del_y = make_float3(yb.x - yf.x, yb.y - yf.y, yb.z - yf.z);
    L.nrm = normalize3(cross3(del_x, del_y));
    const uint32_t c = rgbsT[q];
    L.cr = (float)((c >> 16) & 0xFFu) / 255.0f;     // GL_RGB32F upload of u8 (A1)
    L.cg = (float)((c >> 8) & 0xFFu) / 255.0f;
    L.cb = (float)(c & 0xFFu) / 255.0f;
    L.sem = c >> 24;
    L.radius = get_radius(L.pos.z, L.nrm.z, inv_fx, inv_fy);
    return true;
}

// Where a fused surfel went (for the tile-bounds update)
struct FuseMove { uint32_t id; float x, y, z; };

// Tile boxes of the surfels a workgroup fused (k_associate_direct), grown through a small LDS table: every fused lane finds
// its tile's slot (hash + linear probing, LDS compare-and-swap on the tag) and applies six LDS atomicMax; after a barrier
// the used slots go out with one atomicMax per word that actually grows -- rare: a fused surfel seldom leaves its tile's
// box.  The box's time word is not touched here: k_surfel_pass, which always precedes this kernel, stamps every tile it
// visits for the index map with the frame's time (one atomic per tile from the workgroup that owns it), and only such
// tiles can hold a surfel that is fused in this frame.  Measured on a frame with 99 k fuses (this kernel, us): per fused
// lane eight loads of the box + compares + atomics 45; wave-level groups by tile with DPP reductions 30 (a wave's 64 pixels
// fuse into surfels of ~8 tiles, every group a serial round); this table with the time word in it 28 (every workgroup
// saw a stale time and sent the atomic: ~14 per tile line at ~0.2 us each); seven blind global atomics per lane 135; no
// update at all 17.6.  nfused_blk is workgroup-uniform; no-op (no barrier) when it is 0.
constexpr uint32_t FB_SLOTS = 64u, FB_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ void fuse_bounds_block(uint32_t *__restrict__ tb, bool is_fused, const FuseMove &mv,
                                                  uint32_t nfused_blk, uint32_t *s_tag /* [FB_SLOTS] */, uint32_t *s_box /* [FB_SLOTS * 8] */)
{
    if (nfused_blk == 0u) return;
    if (nfused_blk <= 2u) {
        // a fuse or two (the usual KITTI frame: depth noise defeats data.vert's match test for all but ~2 pixels): blind
        // atomics, nothing waits for them.  The table below costs the one workgroup that holds the frame's fuse two barriers,
        // the LDS fill and a global load per word -- tools/pass_trace.py showed that workgroup leaving k_assoc_prep 2.1 us
        // after every other one, frame after frame.
        if (is_fused) {
            uint32_t *b = tb + (size_t)(mv.id / (uint32_t)TILE) * 8;
            const uint32_t ox = f2ord(mv.x), oy = f2ord(mv.y), oz = f2ord(mv.z);
            atomicMax(&b[0], ~ox); atomicMax(&b[1], ~oy); atomicMax(&b[2], ~oz);
            atomicMax(&b[4], ox); atomicMax(&b[5], oy); atomicMax(&b[6], oz);
            if (mv.x != mv.x || mv.y != mv.y || mv.z != mv.z) atomicAdd(&b[3], 1u);
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < FB_SLOTS; i += blockDim.x) s_tag[i] = FB_EMPTY;
    for (uint32_t i = threadIdx.x; i < FB_SLOTS * 8u; i += blockDim.x) s_box[i] = 0u;
    __syncthreads();
    if (is_fused) {
        const uint32_t tile = mv.id / (uint32_t)TILE;
        const uint32_t ox = f2ord(mv.x), oy = f2ord(mv.y), oz = f2ord(mv.z);
        const bool bad = mv.x != mv.x || mv.y != mv.y || mv.z != mv.z;
        uint32_t h = (tile * 0x9E3779B1u) >> 26;
        int slot = -1;
#pragma unroll 1
        for (int probe = 0; probe < 8; ++probe) {
            const uint32_t sidx = (h + (uint32_t)probe) & (FB_SLOTS - 1u);
            const uint32_t old = atomicCAS(&s_tag[sidx], FB_EMPTY, tile);
            if (old == FB_EMPTY || old == tile) { slot = (int)sidx; break; }
        }
        if (slot >= 0) {
            uint32_t *b = s_box + (uint32_t)slot * 8u;
            atomicMax(&b[0], ~ox); atomicMax(&b[1], ~oy); atomicMax(&b[2], ~oz);
            atomicMax(&b[4], ox); atomicMax(&b[5], oy); atomicMax(&b[6], oz);
            if (bad) atomicAdd(&b[3], 1u);
        } else {                                     // more than a handful of colliding tiles: straight to memory (rare)
            uint32_t *b = tb + (size_t)tile * 8;
            if (~ox > b[0]) atomicMax(&b[0], ~ox);
            if (~oy > b[1]) atomicMax(&b[1], ~oy);
            if (~oz > b[2]) atomicMax(&b[2], ~oz);
            if (ox > b[4]) atomicMax(&b[4], ox);
            if (oy > b[5]) atomicMax(&b[5], oy);
            if (oz > b[6]) atomicMax(&b[6], oz);
            if (bad) atomicAdd(&b[3], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < FB_SLOTS * 8u; i += blockDim.x) {
        const uint32_t tile = s_tag[i >> 3], w = i & 7u, v = s_box[i];
        if (tile == FB_EMPTY || v == 0u) continue;
        uint32_t *g = tb + (size_t)tile * 8 + w;
        if (w == 3u) atomicAdd(g, v);
        else if (v > *g) atomicMax(g, v);
    }
}

// Association + in-place fuse (p8 + p9 + p10) of pixel q.  Every surfel id occupies at most one
// key-map pixel, so the read-modify-write of surfel `id` by this thread is race-free.
// Returns is_valid (candidate pixel) / is_fused (matched and fused into an existing surfel);
// a valid, not fused pixel is a new surfel described by L.
__device__ __forceinline__ void associate_pixel(int q, const SurfelSet &cur, const FrameParams &fp,
                                                const float *__restrict__ depthT, const uint32_t *__restrict__ rgbsT,
                                                const uint64_t *__restrict__ keyT, const float *__restrict__ xs,
                                                const float *__restrict__ ys, const uint32_t *__restrict__ gseg_base,
                                                const uint32_t *__restrict__ seg_lstart, LocalSurfel &L, bool &is_valid,
                                                bool &is_fused, uint32_t *__restrict__ tb, uint32_t first_live,
                                                const uint64_t *__restrict__ own_alive = nullptr /* slot-addressed sharding: this rank's alive bits */,
                                                FuseMove *mv = nullptr /* given: the caller grows the tile boxes (fuse_bounds_block) */,
                                                int qi = -1, int qj = 0 /* column / row of q, if the caller has them */)
{
    is_valid = false;
    is_fused = false;
    uint32_t f_id = 0;                        // the surfel this lane fused into, and where it moved
    float f_x = 0.f, f_y = 0.f, f_z = 0.f;
    // (the key does not depend on the pixel's own surfel: its load is issued with the stencil's, not after the arithmetic)
    const uint64_t key_q = fp.init_mode ? KEY_EMPTY : keyT[min(q, fp.P - 1)];
    if (q < fp.P && local_surfel(q, fp, depthT, rgbsT, xs, ys, L, qi, qj)) {
        is_valid = true;
        const uint64_t key = key_q;
        const int32_t gid = (int32_t)(uint32_t)(key & 0xFFFFFFFFull);
        uint32_t id = 0;
        // data.vert:142 "id > 0" on the GLOBAL id (single GPU: the slot of the first live surfel is id 0);
        // only the rank that owns the winner tries to fuse it.  Slot-addressed sharding (own_alive): ids are global slot
        // numbers on every rank and a rank owns exactly the slots whose alive bit it holds.
        bool mine = false;
        if (!fp.init_mode && key != KEY_EMPTY) {
            if (own_alive) {
                id = (uint32_t)gid;
                mine = id != first_live && ((own_alive[id >> 6] >> (id & 63u)) & 1ull) != 0ull;
            } else {
                mine = (fp.world > 1 ? gid > 0 : (uint32_t)gid != first_live) &&
                       global_to_local((uint32_t)gid, gseg_base, fp.n_gseg, seg_lstart, fp.rank, fp.world, &id);
            }
        }
        // The key carries the winner's depth: d24 is index_map.vert's z / depth_cutoff in 24 bits, computed in THIS frame from
        // the very transform data.vert:151 applies to the same position (an id sits in one pixel, so nothing has moved it
        // since), i.e. camera-frame z to within depth_cutoff / 2^23 plus a few ulp.  A pixel whose measured depth is further
        // from it than the threshold plus a millimetre-scale margin cannot pass that test, whatever the surfel's class: the
        // 16-byte gather of its position -- a 64-byte line per keyed pixel, the largest single item of this kernel's HBM
        // traffic -- is only issued for the others (depth noise of 15 mm: one keyed pixel in twenty).
        if (mine) {
            const float z_key = ((float)(uint32_t)(key >> 32) * (2.0f / 16777215.0f) - 1.0f) * fp.depth_cutoff;
            const float slack = (1.0e-3f + 1.0e-5f * fp.depth_cutoff) * L.lambda;
            if (fabsf(z_key - L.pos.z) * L.lambda > fp.fuse_thresh + slack) mine = false;            // (false for NaN: the exact test decides)
        }
        if (mine) {
            // what is left after the filter is a pixel in twenty at 15 mm depth noise (most of them with a positive threshold and
            // little noise): position, colour word and normal + radius go out TOGETHER -- the pixel or two per frame that
            // really fuse sit in the workgroup that leaves the launch last, and every dependent gather there is ~1.5 us of it
            const float4 pc = cur.pos_conf[id];
            const uint32_t col = cur.color[id];
            const float4 nr = cur.norm_rad[id];
            // index_map.vert:40,61 camera-frame attributes, recomputed from the model
            const float3 vo = xform3(fp.t_inv, pc.x, pc.y, pc.z);
            const bool near = fabsf(vo.z * L.lambda - L.pos.z * L.lambda) <= fp.fuse_thresh;
            const uint32_t sem_o = col >> 24;
            if (near && L.sem == sem_o) {                                                            // data.vert:151
                const float3 ray = make_float3(L.xl, L.yl, 1.0f);
                const float3 cr = cross3(ray, vo);
                const float dist = sqrtf(dot3(cr, cr)) / sqrtf(dot3(ray, ray));
                const float3 no = normalize3(rot3(fp.t_inv, nr.x, nr.y, nr.z));                       // index_map.vert:63
                const float ang = acos_spec(dot3(no, L.nrm) / (sqrtf(dot3(no, no)) * sqrtf(dot3(L.nrm, L.nrm))));
                if (dist < 1000.0f && fabsf(ang) < 0.5f) {                                           // data.vert:158
                    is_fused = true;
                    const float c_n = 0.9f, c_o = pc.w;
                    const float w = c_n + c_o;
                    float4 opc, onr;
                    uint32_t ocol;
                    if (L.radius < 1.5f * nr.w) {                                                     // data.vert:177-194
                        const float pnx = ((c_n * L.pos.x) + (c_o * vo.x)) / w;
                        const float pny = ((c_n * L.pos.y) + (c_o * vo.y)) / w;
                        const float pnz = ((c_n * L.pos.z) + (c_o * vo.z)) / w;
                        const float3 pw = xform3(fp.pose, pnx, pny, pnz);
                        opc = make_float4(pw.x, pw.y, pw.z, w);
                        const float ar = ((c_n * L.cr) + (c_o * L.cr)) / w;                          // sic data.vert:183
                        const float ag = ((c_n * L.cg) + (c_o * L.cg)) / w;
                        const float ab = ((c_n * L.cb) + (c_o * L.cb)) / w;
                        ocol = encode_color(ar, ag, ab, L.sem);
                        const float nx = ((c_n * L.nrm.x) + (c_o * no.x)) / w;
                        const float ny = ((c_n * L.nrm.y) + (c_o * no.y)) / w;
                        const float nz = ((c_n * L.nrm.z) + (c_o * no.z)) / w;
                        const float3 nw = normalize3(rot3(fp.pose, nx, ny, nz));
                        onr = make_float4(nw.x, nw.y, nw.z, (L.radius > nr.w) ? nr.w : L.radius);
                    } else {                                                                          // data.vert:195-208
                        const float3 pw = xform3(fp.pose, vo.x, vo.y, vo.z);
                        opc = make_float4(pw.x, pw.y, pw.z, w);
                        ocol = encode_color((float)((col >> 16) & 0xFFu) / 255.0f, (float)((col >> 8) & 0xFFu) / 255.0f,
                                            (float)(col & 0xFFu) / 255.0f, L.sem);
                        const float3 nw = normalize3(rot3(fp.pose, no.x, no.y, no.z));
                        onr = make_float4(nw.x, nw.y, nw.z, nr.w);
                    }
                    cur.pos_conf[id] = opc;          // fuse.vert:17-49 scatter, in place
                    cur.norm_rad[id] = onr;
                    cur.color[id] = ocol;
                    cur.time[id] = (float)fp.time;   // initTime kept (data.vert:187)
                    f_id = id; f_x = opc.x; f_y = opc.y; f_z = opc.z;
                }
            }
        }
    }
    // The fused surfels moved: their tiles' boxes must grow.  k_associate_direct does it per workgroup (fuse_bounds_block: it
    // passes mv); the other forms wave-level: lanes grouped by tile, each group reduces its box with DPP and publishes it
    // with ONE atomicMax wave instruction (lanes 0..7).
    if (mv) { mv->id = f_id; mv->x = f_x; mv->y = f_y; mv->z = f_z; }
    else bounds_expand_wave(tb, is_fused, f_id / (uint32_t)TILE, f_x, f_y, f_z, (float)fp.time, false);
}

// data.vert:210-225: the new surfel of a valid, unmatched pixel, written to model slot `slot`
__device__ __forceinline__ float3 write_new_surfel(const SurfelSet &cur, uint32_t slot, const LocalSurfel &L, const FrameParams &fp)
{
    const float3 pw = xform3(fp.pose, L.pos.x, L.pos.y, L.pos.z);
    const float3 nw = normalize3(rot3(fp.pose, L.nrm.x, L.nrm.y, L.nrm.z));
    cur.pos_conf[slot] = make_float4(pw.x, pw.y, pw.z, 0.9f);
    cur.norm_rad[slot] = make_float4(nw.x, nw.y, nw.z, L.radius);
    cur.color[slot] = encode_color(L.cr, L.cg, L.cb, L.sem);
    cur.init_time[slot] = (float)fp.time;
    cur.time[slot] = (float)fp.time;
    return pw;
}

// Three-kernel form (used when a reduction over ranks must happen between association and append):
// new surfels are only flagged here (two ballot words per wave).
__global__ __launch_bounds__(PIX_BLOCK) void k_associate(Model M, const DevState *__restrict__ st, FrameParams fp,
                                                         const float *__restrict__ depthT,
                                                         const uint32_t *__restrict__ rgbsT,
                                                         const uint64_t *__restrict__ keyT,
                                                         const float *__restrict__ xs, const float *__restrict__ ys,
                                                         uint64_t *__restrict__ validmask, uint64_t *__restrict__ fusedmask,
                                                         const uint32_t *__restrict__ gseg_base,
                                                         const uint32_t *__restrict__ seg_lstart,
                                                         uint2 *__restrict__ blk_cnt /* (new, fused) per block */,
                                                         uint32_t *__restrict__ tb)
{
    __shared__ uint32_t s_n[4], s_f[4];
    const SurfelSet cur = M.s[st->cur];
    const int q = blockIdx.x * PIX_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool is_valid, is_fused;
    LocalSurfel L;
    associate_pixel(q, cur, fp, depthT, rgbsT, keyT, xs, ys, gseg_base, seg_lstart, L, is_valid, is_fused, tb, st->first_live);
    // two ballot words per wave: candidate pixels, and pixels fused by THIS rank (disjoint across ranks,
    // so a sum-reduction of the words over the ranks is their union)
    const uint64_t vw = __ballot(is_valid), fw = __ballot(is_fused);
    if (lane == 0) {
        const int word = blockIdx.x * (PIX_BLOCK / 64) + wave;
        if (word * 64 < fp.P) { validmask[word] = vw; fusedmask[word] = fw; }
        s_n[wave] = (uint32_t)__popcll(vw & ~fw);
        s_f[wave] = (uint32_t)__popcll(fw);
    }
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = make_uint2(s_n[0] + s_n[1] + s_n[2] + s_n[3], s_f[0] + s_f[1] + s_f[2] + s_f[3]);
}

// ---------------------------------------------------------------------------------------------
// Direct-append form of p8..p11 (the default on frames whose cull only marks the dead): association + in-place fuse, and
// every NEW surfel written straight to its final slot = offset + (candidate pixels before it in pixel order) -- the
// candidate counts per block and per group come from k_pass_fixup's worker workgroups (cand_count_block), so nothing here
// waits for another block and no append kernel follows.  A candidate pixel that fuses
// leaves its slot empty: marked dead (alive bit, per-tile dead count) like a culled surfel.  Survivor order and new-
// surfel order are the reference's (stable cull; column-major append, src/GlobalModel.cpp:67-74), ids handed out by
// the API are positions among the live surfels as with any deferred compaction.
// ---------------------------------------------------------------------------------------------
// Slot-addressed sharding of ONE stream over several GPUs ("sharded mode, in-stream form"): every rank
// addresses surfels by the slot number the single-GPU run would use; it stores (and holds the alive bit of) only the
// slots of the segments it owns.  The association kernel then also leaves the two ballot planes the ranks exchange.
struct ShardArgs {
    uint64_t *validmask;      // candidate pixels (identical on every rank: the frame is replicated)
    uint64_t *ownmask;        // pixels fused by THIS rank (kept: k_shard_settle tells them from the pixels other ranks fused)
    uint64_t *gmask;          // the same words again, sum-reduced IN PLACE over the ranks afterwards (disjoint bit sets: sum == union);
                              //   4 more words follow: [nw + 0..2] this rank's conflicts / surfels drawn into the index map / surfels killed
    uint32_t nwords;          // ceil(P / 64)
    int owner;                // 1: this rank owns the frame's new surfels (frame's segment index % world == rank)
};

struct AssocArgs {
    Model M; DevState *st; FrameParams fp;
    const float *depthT; const uint32_t *rgbsT; const uint64_t *keyT; const float *xs, *ys;
    const uint32_t *blk_cand /* candidate pixels per block ... */, *grp_cand /* ... and per group of CAND_GROUP blocks */;
    uint32_t *frame_sub /* sets 2, 3: new, fused -- 64 sub-counters each */, *tb;
    uint64_t *alive; uint32_t *tile_dead; uint32_t n_grp, cg; unsigned long long *host_stat;
};

// bit i of x -> bit 2 i (Morton spread)
__device__ __forceinline__ uint64_t spread_bits32(uint32_t x)
{
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

// Sum of grp_cand[0 .. n) over the lanes of a wave (each lane returns its share; wave_sum_u32 completes it): up to eight loads
// per lane and round, all unconditional (clamped index) and in flight together.  (As `for (g = lane; g < n; g += 64) sum +=
// grp_cand[g]` hipcc emitted a loop with a wait per pair of loads: with ~450 groups an association workgroup near the
// end of the image spent four dependent round trips here before its first own load -- tools/pass_trace.py showed the
// workgroups' durations growing with their index, 6.9 -> 9.2 us.)
__device__ __forceinline__ uint32_t group_sum_lane(const uint32_t *__restrict__ grp_cand, uint32_t n, uint32_t n_alloc, int lane)
{
    uint32_t sum = 0;
    for (uint32_t base = 0; base < n; base += 512u) {               // one round for every image up to 512 groups
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = grp_cand[min(base + (uint32_t)lane + 64u * (uint32_t)i, n_alloc - 1u)];
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += (base + (uint32_t)lane + 64u * (uint32_t)i < n) ? x[i] : 0u;
    }
    return sum;
}

// one association workgroup (256 threads); wg = its index in pixel order.
// PAIR = false: one pixel per thread, the workgroup is association block `wg` (PIX_BLOCK pixels).
// PAIR = true: TWO consecutive pixels per thread, the workgroup covers blocks 2 wg and 2 wg + 1.  data.vert:88 keeps only
// the pixels with (int)x + (int)y odd -- half of every wave sat out the whole association with one pixel per lane -- and
// sm_create refuses image sizes where (int)xs[i] != i or (int)ys[j] != j, so the test is "(i + j) odd": of the pixels q0
// (even) and q0 + 1 in column-major order exactly one passes it, whatever H is (same column: j and j + 1; across the end
// of a column only if H is odd, (i, H - 1) and (i + 1, 0): i + H - 1 and i + 1 differ in parity; with H even an even q0
// never is the last pixel of a column).  The lane takes that one: every lane of the wave holds a pixel of the
// checkerboard, in pixel order, so ballots, ranks and slots are what they were -- with half the waves.
template <bool SHARD, bool PAIR>
__device__ __forceinline__ void associate_direct_block(const AssocArgs &a, const ShardArgs &sh, const uint32_t wg)
{
    const uint32_t blk = PAIR ? wg * 2u : wg;            // first association block of the workgroup (an even one: same group as the next)
    const Model &M = a.M; DevState *__restrict__ st = a.st; const FrameParams &fp = a.fp;
    const float *__restrict__ depthT = a.depthT; const uint32_t *__restrict__ rgbsT = a.rgbsT; const uint64_t *__restrict__ keyT = a.keyT;
    const float *__restrict__ xs = a.xs, *__restrict__ ys = a.ys;
    const uint32_t *__restrict__ blk_cand = a.blk_cand, *__restrict__ grp_cand = a.grp_cand;
    uint32_t *__restrict__ frame_sub = a.frame_sub, *__restrict__ tb = a.tb;
    uint64_t *__restrict__ alive = a.alive; uint32_t *__restrict__ tile_dead = a.tile_dead;
    const uint32_t n_grp = a.n_grp; unsigned long long *__restrict__ host_stat = a.host_stat;
    __shared__ uint32_t s_v[4], s_n[4], s_f[4];
    __shared__ uint32_t s_hole[12], s_dead[2];          // empty slots of this block: 6 alive words (lo, hi), 2 tiles
    __shared__ uint32_t s_tag[FB_SLOTS], s_box[FB_SLOTS * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 12) s_hole[threadIdx.x] = 0u;
    if (threadIdx.x < 2) s_dead[threadIdx.x] = 0u;
    // candidates before this block = the groups before its group + the blocks of its group before it: a few loads per lane,
    // issued together with DevState, one wave reduction (every wave computes it for itself)
    const uint32_t grp = blk / a.cg, in_grp = blk % a.cg;
    uint32_t pre = (lane < (int)in_grp) ? blk_cand[grp * a.cg + lane] : 0u;
    pre += group_sum_lane(grp_cand, grp, n_grp, lane);
    const SurfelSet cur = M.s[st->cur];
    const uint32_t offset = st->offset;
    int q = (int)blk * PIX_BLOCK + (int)threadIdx.x, qi = -1, qj = 0;
    if (PAIR) {
        const int q0 = (int)blk * PIX_BLOCK + 2 * (int)threadIdx.x;
        qi = q0 / fp.H; qj = q0 - qi * fp.H;
        if (((qi + qj) & 1) == 0) { q = q0 + 1; if (++qj == fp.H) { qj = 0; ++qi; } }     // q0 is off the checkerboard: its successor is on it
        else q = q0;
    }
    bool is_valid, is_fused;
    LocalSurfel L;
    FuseMove mv;
    associate_pixel(q, cur, fp, depthT, rgbsT, keyT, xs, ys, nullptr, nullptr, L, is_valid, is_fused, tb, st->first_live,
                    SHARD ? alive : nullptr, &mv, qi, qj);
    const uint64_t vw = __ballot(is_valid), fw = __ballot(is_fused);
    if (lane == 0) { s_v[wave] = (uint32_t)__popcll(vw); s_n[wave] = (uint32_t)__popcll(vw & ~fw); s_f[wave] = (uint32_t)__popcll(fw); }
    if (SHARD && !PAIR && lane == 0) {
        const uint32_t word = blk * (PIX_BLOCK / 64) + (uint32_t)wave;
        if (word < sh.nwords) { sh.validmask[word] = vw; sh.ownmask[word] = fw; sh.gmask[word] = fw; }
    }
    if (SHARD && PAIR) {
        // the mask planes are one bit per PIXEL: lane l holds pixel 2 l or 2 l + 1 of the wave's 128, so the ballots (one bit
        // per lane) are spread to the even bit positions and the lanes that took the odd pixel move up by one
        const uint64_t odd = __ballot((q & 1) != 0);
        if (lane == 0) {
            const uint32_t word = blk * (PIX_BLOCK / 64) + (uint32_t)wave * 2u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t v32 = (uint32_t)(vw >> (32 * h)), f32 = (uint32_t)(fw >> (32 * h)), o32 = (uint32_t)(odd >> (32 * h));
                const uint64_t vm = spread_bits32(v32 & ~o32) | (spread_bits32(v32 & o32) << 1);
                const uint64_t fm = spread_bits32(f32 & ~o32) | (spread_bits32(f32 & o32) << 1);
                if (word + (uint32_t)h < sh.nwords) { sh.validmask[word + h] = vm; sh.ownmask[word + h] = fm; sh.gmask[word + h] = fm; }
            }
        }
    }
    pre = wave_sum_u32(pre);
    __syncthreads();
    if (SHARD && wg == 0 && threadIdx.x == 0) {
        // this rank's share of the frame's counters travels with the mask (k_pass_fixup published them)
        sh.gmask[sh.nwords] = st->conflict_count; sh.gmask[sh.nwords + 1] = st->visible_count;
        sh.gmask[sh.nwords + 2] = st->n_kill; sh.gmask[sh.nwords + 3] = 0ull;
    }
    if (wg == 0 && wave == 0) {
        // every candidate pixel of the frame owns a slot: the new count (the host never lets a frame of this form start
        // without room for all of them), published for the next frame's kernels and for the host's capacity bound
        uint32_t d = 0;
        d = group_sum_lane(grp_cand, n_grp, n_grp, lane);
        d = wave_sum_u32(d);
        if (lane == 0) {
            st->count = offset + d;
            const uint32_t fr = st->stat_frames + 1u;
            st->stat_frames = fr;
            if (host_stat)
                __hip_atomic_store(host_stat, ((unsigned long long)fr << 32) | (unsigned long long)(offset + d), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (!SHARD && threadIdx.x == 0) {                    // (sharded: k_shard_settle counts, from the masks of all ranks)
        const uint32_t nn = s_n[0] + s_n[1] + s_n[2] + s_n[3], nf = s_f[0] + s_f[1] + s_f[2] + s_f[3];
        if (nn) atomicAdd(&frame_sub[2 * SUB_SET + (wg & 63u) * SUB_STRIDE], nn);
        if (nf) atomicAdd(&frame_sub[3 * SUB_SET + (wg & 63u) * SUB_STRIDE], nf);
    }
    fuse_bounds_block(tb, is_fused, mv, s_f[0] + s_f[1] + s_f[2] + s_f[3], s_tag, s_box);
    uint32_t rank = (uint32_t)__popcll(vw & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += s_v[w];
    const uint32_t slot = offset + pre + rank;
    const bool room = (uint64_t)slot < (uint64_t)fp.max_vertices;       // always, by the host's capacity rule for this frame form
    // sharded: the owner of the frame's segment writes every candidate this rank did not fuse (one that another rank fused
    // is emptied again by k_shard_settle); on the other ranks every candidate slot stays empty
    const bool wr = is_valid && !is_fused && room && (!SHARD || sh.owner != 0);
    const bool hole = is_valid && room && (is_fused || (SHARD && sh.owner == 0));
    float3 pw = make_float3(0.f, 0.f, 0.f);
    if (wr) pw = write_new_surfel(cur, slot, L, fp);
    bounds_expand_wave(tb, wr, slot / (uint32_t)TILE, pw.x, pw.y, pw.z, (float)fp.time, false);
    // The slot of a pixel that fused stays empty.  The block's <= 256 candidate slots are consecutive, i.e. they touch
    // <= 5 alive words and <= 2 tiles: collected in LDS, then one global atomic per word / tile (a global atomic per
    // fused pixel cost 300 us on a frame with 100 k fuses: memory-side atomics on one line serialise).
    const uint32_t blk_first = offset + pre, w_first = blk_first >> 6, t_first = blk_first / (uint32_t)TILE;
    const uint32_t nfused_blk = (SHARD && sh.owner == 0) ? s_v[0] + s_v[1] + s_v[2] + s_v[3] : s_f[0] + s_f[1] + s_f[2] + s_f[3];   // workgroup-uniform
    if (nfused_blk) {
        if (hole) {
            const uint32_t w = (slot >> 6) - w_first, bit = slot & 63u;
            atomicOr(&s_hole[w * 2u + (bit >> 5)], 1u << (bit & 31u));
            atomicAdd(&s_dead[slot / (uint32_t)TILE - t_first], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const uint64_t m = (uint64_t)s_hole[threadIdx.x * 2u] | ((uint64_t)s_hole[threadIdx.x * 2u + 1u] << 32);
            if (m) atomicAnd((unsigned long long *)&alive[w_first + threadIdx.x], ~m);
        } else if (threadIdx.x < 8) {
            const uint32_t d = s_dead[threadIdx.x - 6u];
            if (d) atomicAdd(&tile_dead[t_first + threadIdx.x - 6u], d);
        }
    }
    if (is_valid && !room) st->error = -2;
}

template <bool SHARD, bool PAIR>
__global__ __launch_bounds__(PIX_BLOCK) void k_associate_direct(AssocArgs a, ShardArgs sh)
{
    associate_direct_block<SHARD, PAIR>(a, sh, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// The previous frame's association and this frame's image preparation in ONE launch (plain asynchronous streams,
// DESIGN.md 4 "Three launches per frame"): the two are independent -- the association of frame f-1 reads that frame's
// planes and key map, the preparation of frame f writes the other set -- so the host holds the association back until
// the next frame's images arrive and saves a launch, and the small k_prep runs in the shadow of the association.
// Block ranges: the 32x32-pixel image tiles, the frame's tile flags (tile_prep_block with the "may still change" rule),
// the association blocks.
// ---------------------------------------------------------------------------------------------
template <bool PAIR, bool CHAIN>
__global__ __launch_bounds__(PIX_BLOCK) void k_assoc_prep(AssocArgs a, PrepArgs p, FrameParams fp_new, TilePrep tp, uint32_t n_assoc,
                                                          uint32_t n_img, ChainArgs ch, unsigned long long *__restrict__ trace = nullptr /* SM_PASS_TRACE */)
{
    struct Stamp {              // entry / exit time of every workgroup (thread 0), for tools/pass_trace.py
        unsigned long long *t; unsigned long long t0;
        __device__ Stamp(unsigned long long *tr) : t(tr), t0(tr ? wall_clock64() : 0ull) {}
        __device__ ~Stamp() { if (t && threadIdx.x == 0 && blockIdx.x < 65536u) { t[(size_t)blockIdx.x * 2] = t0; t[(size_t)blockIdx.x * 2 + 1] = wall_clock64(); } }   // (the buffer holds 65 536 records)
    } stamp(trace);
    // Dispatch order.  Without the chain: the association first -- all ~1 500 workgroups are in the chip within 0.3 us and their
    // loads are one burst served roughly in dispatch order; the association is the part with two or three DEPENDENT round trips,
    // the image tiles have one.  With the chain (CHAIN): the chain tiles first -- they are the long workgroups of the launch (169
    // taps per pixel), the association fills the chip around them.
    __shared__ __align__(16) unsigned char s_chain[CHAIN ? CHAIN_LDS_BYTES : 16];
    if (CHAIN) {
        if (blockIdx.x < n_img) { prep_chain_block(p, ch, fp_new, blockIdx.x, s_chain); return; }      // workgroup-uniform
        const uint32_t b = blockIdx.x - n_img;
        if (b >= n_assoc) { tile_prep_block(fp_new, tp, n_img + n_assoc); return; }
        ShardArgs none;
        none.validmask = nullptr; none.ownmask = nullptr; none.gmask = nullptr; none.nwords = 0u; none.owner = 1;
        associate_direct_block<false, PAIR>(a, none, b);
        return;
    }
    if (blockIdx.x >= n_assoc) {                                                                  // workgroup-uniform
        const uint32_t b = blockIdx.x - n_assoc;
        if (b < tp.nfb) { tile_prep_block(fp_new, tp, n_assoc); return; }
        prep_image_block<PIX_BLOCK>(p, fp_new, b - tp.nfb);
        return;
    }
    ShardArgs none;
    none.validmask = nullptr; none.ownmask = nullptr; none.gmask = nullptr; none.nwords = 0u; none.owner = 1;
    associate_direct_block<false, PAIR>(a, none, blockIdx.x);
}

// stand-alone form (when something reads the frame's counters before the next frame's k_prep has run)
__global__ __launch_bounds__(PIX_BLOCK) void k_shard_settle(ShardSettle a)
{
    shard_settle_body<1>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Slot-addressed sharding: the W*H conflict cap (src/GlobalModel.cpp:54-57), exactly.  Only the first W*H
// conflicts in surfel order take effect, and with the surfels spread over the ranks that order runs across ranks: a slot's
// conflict ordinal needs the conflicts of every rank in the slots below it.  k_surfel_pass has (speculatively) applied every
// conflict of this rank and left cm / km / undo / the quarter-tile counts, as on one GPU; the ranks then sum-reduce
//   x[0]                     conflicts of the frame
//   x[1 + 2 t .. 1 + 2 t + 1]   tile t's four quarter counts (two per word)
//   x[1 + 2 T + w]           conflict mask of word w            (T, 16 T words: the host's slot bound, equal on all ranks)
// -- slots have one owner, so the sum of the masks is their union -- and k_shard_cap_repair takes this rank's surplus
// back like k_pass_fixup's repair does, with ordinals from the reduced buffer: confidences restored from the undo plane,
// victims resurrected and drawn into the (still local) key map, counters corrected, BEFORE the key-map exchange and the
// association.  One all-reduce of 144 bytes per 1024 slots; the host skips it while the model has no more slots than pixels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_shard_cap_pack(const DevState *__restrict__ st, const uint4 *__restrict__ wave_cnt,
                                                        const uint64_t *__restrict__ cm, const uint32_t *__restrict__ conf_sub,
                                                        uint64_t *__restrict__ x, uint32_t tiles_bound)
{
    const uint32_t N = st->count;
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, gsz = gridDim.x * 256u;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const uint32_t tot = wave_sum_u32(conf_sub[threadIdx.x * SUB_STRIDE]);
        if (threadIdx.x == 0) x[0] = tot;
    }
    for (uint32_t w = gid; w < tiles_bound * (uint32_t)TILE_WORDS; w += gsz) {
        const uint32_t t = w / TILE_WORDS, qtr = (w % TILE_WORDS) / 4u;
        uint64_t m = 0ull;
        uint4 c = make_uint4(0u, 0u, 0u, 0u);
        if (t < ntiles) {
            c = wave_cnt[t];                      // (zero for the tiles the pass skipped: their cm words are stale)
            const uint32_t cq = qtr == 0 ? c.x : qtr == 1 ? c.y : qtr == 2 ? c.z : c.w;
            if (cq) m = cm[w];
        }
        x[1 + 2 * (size_t)tiles_bound + w] = m;
        if ((w % TILE_WORDS) == 0u) {
            x[1 + 2 * (size_t)t] = (uint64_t)c.x | ((uint64_t)c.y << 32);
            x[2 + 2 * (size_t)t] = (uint64_t)c.z | ((uint64_t)c.w << 32);
        }
    }
}

__global__ __launch_bounds__(256) void k_shard_cap_repair(Model M, DevState *__restrict__ st, FrameParams fp, const uint64_t *__restrict__ x,
                                                          uint32_t tiles_bound, uint32_t cap, const uint4 *__restrict__ wave_cnt,
                                                          const uint64_t *__restrict__ km, const uint8_t *__restrict__ tile_flags,
                                                          uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead,
                                                          uint64_t *__restrict__ keyT, const float *__restrict__ undo, uint32_t *__restrict__ tb)
{
    if (x[0] <= (uint64_t)cap) return;                   // the cap does not bind (nearly every frame): nothing to take back
    __shared__ uint32_t s_c[4], s_a[4], s_b[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t N = st->count;
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const uint32_t nwg = gridDim.x, wi = blockIdx.x;
    const SurfelSet set = M.s[st->cur];
    const uint64_t *__restrict__ xq = x + 1, *__restrict__ xm = x + 1 + 2 * (size_t)tiles_bound;
    auto tile_conf = [&](uint32_t t) { const uint64_t a = xq[2 * (size_t)t], b = xq[2 * (size_t)t + 1]; return (uint32_t)a + (uint32_t)(a >> 32) + (uint32_t)b + (uint32_t)(b >> 32); };
    uint32_t cpre = 0;                                   // conflicts (all ranks) in the tiles below this workgroup's current one
    {
        uint32_t p = 0;
        for (uint32_t t = threadIdx.x; t < min(wi, ntiles); t += 256u) p += tile_conf(t);
        p = wave_sum_u32(p);
        if (lane == 0) s_c[wave] = p;
        __syncthreads();
        cpre = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        __syncthreads();
    }
    uint32_t vis = 0, resurrected = 0;
    for (uint32_t tile = wi; tile < ntiles; tile += nwg) {
        const uint64_t qa = xq[2 * (size_t)tile], qb = xq[2 * (size_t)tile + 1];
        const uint32_t g0 = (uint32_t)qa, g1 = (uint32_t)(qa >> 32), g2 = (uint32_t)qb, g3 = (uint32_t)(qb >> 32);
        const uint32_t nconf = g0 + g1 + g2 + g3, tile_pre = cpre;
        {   // advance the prefix to this workgroup's next tile
            uint32_t p = 0;
            for (uint32_t t = tile + threadIdx.x; t < min(tile + nwg, ntiles); t += 256u) p += tile_conf(t);
            p = wave_sum_u32(p);
            __syncthreads();
            if (lane == 0) s_c[wave] = p;
            __syncthreads();
            cpre += s_c[0] + s_c[1] + s_c[2] + s_c[3];
        }
        if (nconf == 0u || tile_pre + nconf <= cap) continue;              // every conflict of the tile is effective
        const uint4 own4 = wave_cnt[tile];
        const uint32_t own_q = wave == 0 ? own4.x : wave == 1 ? own4.y : wave == 2 ? own4.z : own4.w;     // this rank's conflicts in the wave's quarter
        const bool nosplat = (tile_flags[tile] & 2u) != 0u;
        uint32_t wpre = tile_pre + (wave > 0 ? g0 : 0u) + (wave > 1 ? g1 : 0u) + (wave > 2 ? g2 : 0u);
        uint32_t res_wave = 0;
        const uint32_t vis_tile = vis;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t word = tile * TILE_WORDS + (uint32_t)wave * 4u + (uint32_t)r;
            if ((uint64_t)word * 64u >= N) break;                             // wave-uniform
            const uint64_t cg = xm[word];                                     // conflicts of this word over all ranks
            const uint64_t ineff = ineffective_conflicts(cg, wpre, cap);
            wpre += (uint32_t)__popcll(cg);
            if (ineff == 0ull || own_q == 0u) continue;                       // (own_q == 0: this rank's km word is stale, and it has nothing here)
            const uint64_t res = ineff & km[word];                            // killed HERE by a conflict that does not count
            const uint64_t restore = ineff & ~res & alive[word];              // survived here, decremented (alive bits are this rank's slots only)
            const uint32_t k = word * 64u + lane;
            if ((restore >> lane) & 1ull) set.pos_conf[k].w = undo[k];
            if (res) {
                if (lane == 0) alive[word] |= res;
                res_wave += (uint32_t)__popcll(res);
                if (!nosplat) {
                    bool drew = false;
                    if ((res >> lane) & 1ull) {
                        const float4 pv = set.pos_conf[k];
                        drew = splat_one(fp, pv.x, pv.y, pv.z, set.time[k], k, keyT);
                    }
                    vis += (uint32_t)__popcll(__ballot(drew));
                }
            }
        }
        if (res_wave && lane == 0) atomicSub(&tile_dead[tile], res_wave);
        if (vis != vis_tile && lane == 0) atomicMax(&tb[(size_t)tile * 8 + 7], f2ord((float)fp.time));      // as k_pass_fixup
        resurrected += res_wave;
    }
    __syncthreads();
    if (lane == 0) { s_a[wave] = vis; s_b[wave] = resurrected; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // this rank's share of the frame's counters (k_pass_fixup published them; the association's first block sends them round)
        const uint32_t v = s_a[0] + s_a[1] + s_a[2] + s_a[3], rs = s_b[0] + s_b[1] + s_b[2] + s_b[3];
        if (v) atomicAdd(&st->visible_count, v);
        if (rs) atomicSub(&st->n_kill, rs);
    }
}

// ---------------------------------------------------------------------------------------------
// Slot-addressed sharding: physical compaction BETWEEN frames.  The single-GPU run squeezes the dead slots out and a
// survivor's new slot is the number of live slots below it -- over ALL ranks.  Every rank contributes its alive bits
// (non-owned and dead slots are 0), the planes are sum-reduced (disjoint bit sets: sum == union) into `galive`, and
// each rank then moves only its own survivors:  stage (own survivors -> a second SoA set at their new slot, own bits
// of the new alive plane), unstage (copy back, rebuild the alive words, dead counts and bounds of every tile from the
// first moving one on).  Two plain passes instead of the in-place hand-off protocol of k_compact: this runs once per
// `compact_period` frames, outside the frame.  info = {first moving tile, new slot count, old slot count}.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_shard_alive_copy(const DevState *__restrict__ st, const uint64_t *__restrict__ alive,
                                                          uint64_t *__restrict__ out, uint64_t *__restrict__ new_alive,
                                                          uint32_t nw_bound /* words the ranks exchange: the host's bound, the same on every rank */)
{
    const uint32_t N = st->count;
    for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < nw_bound; w += gridDim.x * 256u) {
        const uint64_t base = (uint64_t)w * 64u;
        uint64_t range = 0ull;
        if (base < N) { const uint64_t rem = (uint64_t)N - base; range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull); }
        out[w] = alive[w] & range;
        new_alive[w] = 0ull;
    }
}

__global__ __launch_bounds__(256) void k_shard_tile_popc(const DevState *__restrict__ st, const uint64_t *__restrict__ galive,
                                                         uint32_t *__restrict__ tile_keep)
{
    const uint32_t N = st->count;
    const uint32_t ntiles = (N + TILE - 1) / TILE, nw = (N + 63u) / 64u;
    const int lane = threadIdx.x & 63;
    const uint32_t wave_g = (blockIdx.x * 256u + threadIdx.x) >> 6;
    for (uint32_t t0 = wave_g * 4u; t0 < ntiles; t0 += (gridDim.x * 4u) * 4u) {      // 4 tiles per wave: 16 lanes per tile
        const uint32_t t = t0 + (uint32_t)(lane >> 4), w = t * TILE_WORDS + (uint32_t)(lane & 15);
        uint32_t p = (t < ntiles && w < nw) ? (uint32_t)__popcll(galive[w]) : 0u;
        p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4); p += __shfl_xor(p, 8);
        if ((lane & 15) == 0 && t < ntiles) tile_keep[t] = p;
    }
}

__global__ __launch_bounds__(1024) void k_shard_scan(DevState *__restrict__ st, const uint32_t *__restrict__ tile_keep,
                                                     uint32_t *__restrict__ tile_base, uint32_t *__restrict__ info,
                                                     unsigned long long *__restrict__ host_stat)
{
    __shared__ uint32_t s_sum[1024], s_fm[1024];
    const uint32_t N = st->count;
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const uint32_t chunk = (ntiles + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * chunk, e = min(b + chunk, ntiles);
    uint32_t sum = 0, fm = 0xFFFFFFFFu;
    for (uint32_t t = b; t < e; ++t) {
        const uint32_t k = tile_keep[t];
        if (k != (uint32_t)TILE && fm == 0xFFFFFFFFu) fm = t;
        sum += k;
    }
    s_sum[threadIdx.x] = sum; s_fm[threadIdx.x] = fm;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {                   // inclusive Hillis-Steele scan of the chunk sums
        const uint32_t v = threadIdx.x >= d ? s_sum[threadIdx.x - d] : 0u;
        const uint32_t f = threadIdx.x >= d ? s_fm[threadIdx.x - d] : 0xFFFFFFFFu;
        __syncthreads();
        s_sum[threadIdx.x] += v; s_fm[threadIdx.x] = min(s_fm[threadIdx.x], f);
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? s_sum[threadIdx.x - 1] : 0u;
    for (uint32_t t = b; t < e; ++t) { tile_base[t] = run; run += tile_keep[t]; }
    if (threadIdx.x == 1023u) {
        info[0] = min(s_fm[1023], ntiles);       // first tile that loses or moves surfels (== ntiles: nothing to do)
        info[1] = s_sum[1023];                   // live surfels over all ranks = the new slot count
        info[2] = N;
        // publish the compacted state (k_shard_stage / k_shard_unstage take the old count from info[2], not from DevState)
        const uint32_t Nn = s_sum[1023];
        // `offset` is what the reference reports between frames: the surfels that were there before the last append
        const uint32_t live_before_append = st->offset - (st->garbage - st->holes_last);
        st->count = Nn; st->offset = live_before_append;
        st->garbage = 0u; st->garbage_prev = 0u; st->holes_last = 0u;
        st->first_live = 0u; st->fl_dirty = 0u;  // the first live surfel of the union moves to slot 0
        if (host_stat)
            __hip_atomic_store(host_stat, ((unsigned long long)st->stat_frames << 32) | (unsigned long long)Nn, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void k_shard_stage(Model M, const DevState *__restrict__ st, const uint64_t *__restrict__ alive,
                                                     const uint64_t *__restrict__ galive, const uint32_t *__restrict__ tile_base,
                                                     const uint32_t *__restrict__ info, uint64_t *__restrict__ new_alive)
{
    const SurfelSet src = M.s[st->cur], dst = M.s[st->cur ^ 1u];
    const uint32_t fm = info[0], N = info[2];
    const uint32_t ntiles = (N + TILE - 1) / TILE, nw = (N + 63u) / 64u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t t = fm + blockIdx.x; t < ntiles; t += gridDim.x) {
        // live slots (any rank) of the tile's words below each of this wave's words
        const uint32_t w16 = t * TILE_WORDS + (uint32_t)(lane & 15);
        const uint32_t pc = (lane < 16 && w16 < nw) ? (uint32_t)__popcll(galive[w16]) : 0u;
        uint32_t before = tile_base[t];
        for (int i = 0; i < wave * 4; ++i) before += lane_bcast(pc, i);
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const uint32_t w = t * TILE_WORDS + (uint32_t)wave * 4u + (uint32_t)r;
            if (w >= nw) break;                                   // wave-uniform
            const uint64_t g = galive[w];
            const uint64_t rem = (uint64_t)N - (uint64_t)w * 64u;
            const uint64_t own = alive[w] & g & (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull));
            const uint32_t rk = (uint32_t)__popcll(g & ((1ull << lane) - 1ull));       // rank among the word's live slots (any rank)
            const bool mine = (own >> lane) & 1ull;
            if (mine) {
                const uint32_t k = w * 64u + (uint32_t)lane;
                const uint32_t d = before + rk;
                dst.pos_conf[d] = src.pos_conf[k];
                dst.norm_rad[d] = src.norm_rad[k];
                dst.color[d] = src.color[k];
                dst.init_time[d] = src.init_time[k];
                dst.time[d] = src.time[k];
            }
            // the word's survivors land in the run [before, before + popc(g)): this rank's bits of it, gathered with two wave
            // sums (distinct bits: sum == or) and published with <= 2 atomics (an atomic per surfel serialises on the line)
            const uint32_t lo = wave_sum_u32((mine && rk < 32u) ? (1u << rk) : 0u), hi = wave_sum_u32((mine && rk >= 32u) ? (1u << (rk - 32u)) : 0u);
            const uint64_t run = (uint64_t)lo | ((uint64_t)hi << 32);
            if (run && lane == 0) {
                const uint32_t sft = before & 63u;
                atomicOr((unsigned long long *)&new_alive[before >> 6], run << sft);
                if (sft && (run >> (64u - sft))) atomicOr((unsigned long long *)&new_alive[(before >> 6) + 1u], run >> (64u - sft));
            }
            before += (uint32_t)__popcll(g);
        }
    }
}

__global__ __launch_bounds__(256) void k_shard_unstage(Model M, const DevState *__restrict__ st, const uint32_t *__restrict__ info,
                                                       const uint64_t *__restrict__ new_alive, uint64_t *__restrict__ alive,
                                                       uint32_t *__restrict__ tile_dead, uint32_t *__restrict__ tb)
{
    __shared__ uint32_t s_live[4];
    const SurfelSet dst = M.s[st->cur], src = M.s[st->cur ^ 1u];
    const uint32_t fm = info[0], Nn = info[1], No = info[2];
    const uint32_t ntiles_old = (No + TILE - 1) / TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t t = fm + blockIdx.x; t < ntiles_old; t += gridDim.x) {
        if (threadIdx.x < 8) atomicExch(&tb[(size_t)t * 8 + threadIdx.x], 0u);     // empty box (memory-side, before the atomicMax below)
        __syncthreads();
        uint32_t live = 0;
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const uint32_t w = t * TILE_WORDS + (uint32_t)wave * 4u + (uint32_t)r;
            const uint64_t base = (uint64_t)w * 64u;
            uint64_t range = 0ull;
            if (base < Nn) { const uint64_t rem = (uint64_t)Nn - base; range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull); }
            const uint64_t m = (base < No ? new_alive[w] : 0ull) & range;
            const uint32_t d = w * 64u + (uint32_t)lane;
            const bool mine = (m >> lane) & 1ull;
            float4 pv = make_float4(0.f, 0.f, 0.f, 1.f);
            float tl = 0.f;
            if (mine) {
                pv = src.pos_conf[d]; tl = src.time[d];
                dst.pos_conf[d] = pv;
                dst.norm_rad[d] = src.norm_rad[d];
                dst.color[d] = src.color[d];
                dst.init_time[d] = src.init_time[d];
                dst.time[d] = tl;
            }
            bounds_expand_wave(tb, mine, t, pv.x, pv.y, pv.z, tl, !(pv.w > 0.0f));
            if (lane == 0) alive[w] = m | ~range;                 // free slots (>= the new count) read 1
            live += (uint32_t)__popcll(m);
        }
        if (lane == 0) s_live[wave] = live;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint64_t tb0 = (uint64_t)t * TILE;
            const uint32_t occ = tb0 < Nn ? (uint32_t)min((uint64_t)TILE, (uint64_t)Nn - tb0) : 0u;
            tile_dead[t] = occ - (s_live[0] + s_live[1] + s_live[2] + s_live[3]);      // slots of other ranks' surfels count as dead here
        }
        __syncthreads();
    }
}

// AoS export of this rank's surfels of the (compacted) union, zeros in the slots of other ranks: the integer sum of
// the planes of all ranks is the single GlobalModel (GlobalModel::downloadMap layout, 12 floats per surfel)
__global__ void k_shard_export_aos(Model M, const DevState *__restrict__ st, const uint64_t *__restrict__ alive,
                                   float *__restrict__ dst, uint32_t n)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const SurfelSet cur = M.s[st->cur];
    float4 *o = reinterpret_cast<float4 *>(dst + (size_t)k * 12);
    if ((alive[k >> 6] >> (k & 63u)) & 1ull) {
        const float4 pc = cur.pos_conf[k], nr = cur.norm_rad[k];
        o[0] = pc;
        o[1] = make_float4(__uint_as_float(cur.color[k]), 0.0f, cur.init_time[k], cur.time[k]);
        o[2] = nr;
    } else {
        o[0] = make_float4(0.f, 0.f, 0.f, 0.f); o[1] = o[0]; o[2] = o[0];
    }
}

// ---------------------------------------------------------------------------------------------
// Single-kernel form: association + in-place fuse + ORDERED append of the new surfels (p8..p11).
// Persistent workgroups take pixel blocks round-robin in increasing order; the position of a
// block's new surfels in the model is offset + (new surfels of all lower blocks), obtained with a
// decoupled look-back over 8-byte granules {epoch:16 | status:2 | new:23 | fused:23}: a block
// publishes its AGGREGATE, sums the granules of its predecessors 64 at a time (one per lane) down
// to the first PREFIX, publishes its own PREFIX, then writes its surfels -- still in registers --
// straight to their final slots (pixel order = src/GlobalModel.cpp:67-74 order).  A granule is
// written by one sc1 8-byte store and polled with sc1 loads (MI355X_MICROARCH.md, hand-off
// granules); blocks only wait for lower-numbered blocks and the grid is fully co-resident.
// ---------------------------------------------------------------------------------------------
constexpr uint64_t G_AGG = 1, G_PREFIX = 2;

__device__ __forceinline__ uint64_t granule(uint32_t epoch, uint64_t status, uint32_t nnew, uint32_t nfused)
{
    return ((uint64_t)(epoch & 0xFFFFu) << 48) | (status << 46) | ((uint64_t)(nnew & 0x7FFFFFu) << 23) | (uint64_t)(nfused & 0x7FFFFFu);
}

__global__ __launch_bounds__(PIX_BLOCK) void k_associate_append(Model M, DevState *__restrict__ st, FrameParams fp,
                                                                const float *__restrict__ depthT,
                                                                const uint32_t *__restrict__ rgbsT,
                                                                const uint64_t *__restrict__ keyT,
                                                                const float *__restrict__ xs, const float *__restrict__ ys,
                                                                unsigned long long *__restrict__ desc, uint32_t epoch, int nblocks,
                                                                FrameLog *__restrict__ log, uint32_t *__restrict__ tb,
                                                                const uint2 *__restrict__ compact_part, uint32_t n_compact_part,
                                                                uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead)
{
    __shared__ uint64_t s_nw[4];
    __shared__ uint32_t s_f[4];
    __shared__ uint32_t s_excl[2];
    const SurfelSet cur = M.s[st->cur];
    const uint32_t offset = st->offset, first_live = st->first_live;
    const uint32_t garbage = st->garbage, garbage_prev = st->garbage_prev, n_slots = st->cull_n;
    post_compact_fill(st, alive, tile_dead, blockIdx.x * PIX_BLOCK + threadIdx.x, gridDim.x * PIX_BLOCK);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int q = b * PIX_BLOCK + threadIdx.x;
        bool is_valid, is_fused;
        LocalSurfel L;
        associate_pixel(q, cur, fp, depthT, rgbsT, keyT, xs, ys, nullptr, nullptr, L, is_valid, is_fused, tb, first_live);
        const bool is_new = is_valid && !is_fused;
        const uint64_t nw = __ballot(is_new), fw = __ballot(is_fused);
        if (lane == 0) { s_nw[wave] = nw; s_f[wave] = (uint32_t)__popcll(fw); }
        __syncthreads();
        const uint32_t own_new = (uint32_t)(__popcll(s_nw[0]) + __popcll(s_nw[1]) + __popcll(s_nw[2]) + __popcll(s_nw[3]));
        const uint32_t own_f = s_f[0] + s_f[1] + s_f[2] + s_f[3];
        if (wave == 0) {
            if (lane == 0)
                __hip_atomic_store(&desc[b], granule(epoch, G_AGG, own_new, own_f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t excl_n = 0, excl_f = 0;
            int base = b - 1;
            uint32_t spins = 0;
            while (base >= 0) {
                const int j = base - lane;
                uint64_t g = granule(epoch, G_PREFIX, 0, 0);                  // virtual predecessor of block 0
                if (j >= 0) g = __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ready = ((g >> 48) == (uint64_t)(epoch & 0xFFFFu)) && (((g >> 46) & 3ull) != 0ull);
                const bool ispre = ready && (((g >> 46) & 3ull) == G_PREFIX);
                const uint64_t pm = __ballot(ispre), rm = __ballot(ready);
                const int first = pm ? __ffsll((long long)pm) - 1 : 64;
                const uint64_t need = first < 63 ? ((1ull << (first + 1)) - 1ull) : ~0ull;
                if ((rm & need) != need) {                                    // a granule in the window is not there yet
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { if (lane == 0) st->error = -6; break; }
                    continue;
                }
                uint32_t vn = (lane <= first) ? (uint32_t)((g >> 23) & 0x7FFFFFull) : 0u;
                uint32_t vf = (lane <= first) ? (uint32_t)(g & 0x7FFFFFull) : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { vn += __shfl_xor(vn, o); vf += __shfl_xor(vf, o); }
                excl_n += vn; excl_f += vf;
                if (first < 64) break;
                base -= 64;
            }
            if (lane == 0) {
                __hip_atomic_store(&desc[b], granule(epoch, G_PREFIX, excl_n + own_new, excl_f + own_f), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                s_excl[0] = excl_n; s_excl[1] = excl_f;
            }
        }
        __syncthreads();
        const uint32_t excl_n = s_excl[0];
        const bool fits = (uint64_t)offset + excl_n + own_new <= (uint64_t)fp.max_vertices;
        {
            uint32_t rank = (uint32_t)__popcll(nw & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) rank += (uint32_t)__popcll(s_nw[w]);
            const uint32_t slot = offset + excl_n + rank;
            const bool wr = is_new && fits;
            float3 pw = make_float3(0.f, 0.f, 0.f);
            if (wr) pw = write_new_surfel(cur, slot, L, fp);
            bounds_expand_wave(tb, wr, slot / (uint32_t)TILE, pw.x, pw.y, pw.z, (float)fp.time, false);
        }
        if (b == nblocks - 1 && threadIdx.x == 0) {
            // this block's inclusive prefix is the frame total (GlobalModel::concatenate src/GlobalModel.cpp:629)
            const uint32_t ntot = excl_n + own_new, ftot = s_excl[1] + own_f;
            if (n_compact_part) {
                uint32_t cv = 0, cs = 0;
                for (uint32_t i = 0; i < n_compact_part; ++i) { cv += compact_part[i].x; cs += compact_part[i].y; }
                st->visible_count = cv; st->n_splat_skipped = cs;
            }
            st->unstable_count = ntot;
            st->fused_count = ftot;
            st->data_count = ntot + ftot;
            if ((uint64_t)offset + ntot > (uint64_t)fp.max_vertices) {
                st->error = -2;          // SM_E_CAPACITY: the frame's new surfels are dropped
                st->append_n = 0;
                st->count = offset;
            } else {
                st->append_n = ntot;
                st->count = offset + ntot;
            }
            if (fp.log_frame && log) {
                FrameLog e;
                e.tick = (uint32_t)fp.time; e.n_before = n_slots - garbage_prev; e.n_after_cull = offset - garbage; e.n_kill = st->n_kill;
                e.conflict_count = st->conflict_count; e.visible_count = st->visible_count;
                e.fused_count = ftot; e.unstable_count = ntot; e.n_static = st->n_static;
            e.n_conf_skipped = st->n_conf_skipped; e.n_splat_skipped = st->n_splat_skipped; e.n_slots = n_slots;
            st->n_conf_skipped = 0;
                log[st->frames_logged % FRAME_LOG_LEN] = e;
                st->frames_logged = st->frames_logged + 1;
            }
        }
        __syncthreads();
    }
}

// scan of the per-block new-surfel counts; publishes data/unstable/fused counts and the new
// model count (GlobalModel::concatenate src/GlobalModel.cpp:629) with the capacity check the
// reference lacks.
__global__ __launch_bounds__(1024) void k_scan_new(DevState *__restrict__ st, FrameParams fp, int nblocks,
                                                   const uint64_t *__restrict__ validmask,
                                                   const uint64_t *__restrict__ fusedmask,
                                                   uint32_t *__restrict__ blk_prefix, FrameLog *__restrict__ log,
                                                   const uint2 *__restrict__ compact_part, uint32_t n_compact_part)
{
    __shared__ uint32_t s_scan[17];
    uint32_t pv = 0, ps = 0, vis_tot, skip_tot;
    for (uint32_t b = threadIdx.x; b < n_compact_part; b += 1024u) { const uint2 c = compact_part[b]; pv += c.x; ps += c.y; }
    block_scan_1024(pv, &vis_tot, s_scan);
    block_scan_1024(ps, &skip_tot, s_scan);
    if (threadIdx.x == 0 && n_compact_part) { st->visible_count = vis_tot; st->n_splat_skipped = skip_tot; }
    const uint32_t nb = (uint32_t)nblocks;
    const uint32_t nwords = ((uint32_t)fp.P + 63u) >> 6;
    uint32_t ncarry = 0, fcarry = 0;
    // rounds of 1024 pixel blocks, thread <-> block: the 4+4 mask words of a block are one 32-byte run
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024u) {
        const uint32_t b = b0 + threadIdx.x;
        uint32_t ns = 0, fs = 0;
        if (b < nb) {
#pragma unroll
            for (uint32_t w = 0; w < PIX_BLOCK / 64; ++w) {
                const uint32_t word = b * (PIX_BLOCK / 64) + w;
                if (word < nwords) {
                    const uint64_t v = validmask[word], f = fusedmask[word];
                    ns += (uint32_t)__popcll(v & ~f);
                    fs += (uint32_t)__popcll(f);
                }
            }
        }
        uint32_t nt;
        const uint32_t npre = block_scan_1024(ns, &nt, s_scan);
        if (b < nb) blk_prefix[b] = ncarry + npre;
        ncarry += nt;
        fcarry += fs;                              // per-thread partial, reduced once below
    }
    uint32_t ftot;
    block_scan_1024(fcarry, &ftot, s_scan);
    const uint32_t ntot = ncarry;
    if (threadIdx.x == 0) {
        st->unstable_count = ntot;
        st->fused_count = ftot;
        st->data_count = ntot + ftot;
        if (!fp.append_here) {                   // another rank owns this frame's new surfels
            st->append_n = 0;
            st->count = st->offset;
        } else if ((uint64_t)st->offset + ntot > (uint64_t)fp.max_vertices) {
            st->error = -2;          // SM_E_CAPACITY: append nothing instead of corrupting state
            st->append_n = 0;
            st->count = st->offset;
        } else {
            st->append_n = ntot;
            st->count = st->offset + ntot;
        }
        if (fp.log_frame && log) {
            FrameLog e;
            e.tick = (uint32_t)fp.time; e.n_before = st->cull_n; e.n_after_cull = st->offset; e.n_kill = st->n_kill;
            e.conflict_count = st->conflict_count; e.visible_count = st->visible_count;
            e.fused_count = ftot; e.unstable_count = ntot; e.n_static = st->n_static;
            e.n_conf_skipped = st->n_conf_skipped; e.n_splat_skipped = st->n_splat_skipped; e.n_slots = st->cull_n;
            st->n_conf_skipped = 0;
            log[st->frames_logged % FRAME_LOG_LEN] = e;
            st->frames_logged = st->frames_logged + 1;
        }
    }
}

// p11 concatenate (unstable.vert:13-34 + glCopyBufferSubData src/GlobalModel.cpp:627): the new
// surfels are (re)computed here and written straight to their final slot, in pixel order.
__global__ __launch_bounds__(PIX_BLOCK) void k_append(Model M, const DevState *__restrict__ st, FrameParams fp,
                                                      const float *__restrict__ depthT,
                                                      const uint32_t *__restrict__ rgbsT,
                                                      const float *__restrict__ xs, const float *__restrict__ ys,
                                                      const uint64_t *__restrict__ validmask,
                                                      const uint64_t *__restrict__ fusedmask,
                                                      const uint32_t *__restrict__ blk_prefix, uint32_t *__restrict__ tb)
{
    if (st->append_n == 0) return;
    const SurfelSet cur = M.s[st->cur];
    const int q = blockIdx.x * PIX_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int word0 = blockIdx.x * (PIX_BLOCK / 64);
    const int nwords = (fp.P + 63) >> 6;
    if (word0 + wave >= nwords) return;                         // wave-uniform
    const uint64_t mw = validmask[word0 + wave] & ~fusedmask[word0 + wave];
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += (uint32_t)__popcll(validmask[word0 + w] & ~fusedmask[word0 + w]);
    const uint32_t slot = st->offset + blk_prefix[blockIdx.x] + before + (uint32_t)__popcll(mw & ((1ull << lane) - 1ull));
    LocalSurfel L;
    const bool wr = ((mw >> lane) & 1ull) && local_surfel(q, fp, depthT, rgbsT, xs, ys, L);   // flagged pixels are valid
    float3 pw = make_float3(0.f, 0.f, 0.f);
    if (wr) pw = write_new_surfel(cur, slot, L, fp);
    bounds_expand_wave(tb, wr, slot / (uint32_t)TILE, pw.x, pw.y, pw.z, (float)fp.time, false);
}

// Single-GPU form of p11 without the separate scan kernel: every block sums the (new, fused) counts of
// the blocks before it (a few KB, L2-resident) and the last block publishes the frame totals.
__global__ __launch_bounds__(PIX_BLOCK) void k_append_scan(Model M, DevState *__restrict__ st, FrameParams fp,
                                                           const float *__restrict__ depthT,
                                                           const uint32_t *__restrict__ rgbsT,
                                                           const float *__restrict__ xs, const float *__restrict__ ys,
                                                           const uint64_t *__restrict__ validmask,
                                                           const uint64_t *__restrict__ fusedmask,
                                                           const uint2 *__restrict__ blk_cnt, FrameLog *__restrict__ log,
                                                           uint32_t *__restrict__ tb, const uint2 *__restrict__ compact_part,
                                                           uint32_t n_compact_part, uint64_t *__restrict__ alive,
                                                           uint32_t *__restrict__ tile_dead,
                                                           unsigned long long *__restrict__ host_stat,
                                                           const uint4 *__restrict__ lazy_part /* k_cull_lazy_frame's / k_surfel_pass's partials (then compact_part is unused) */,
                                                           const uint2 *__restrict__ fix_part /* k_pass_fixup's (visible added, resurrected), read when the cap bound; or null */,
                                                           uint32_t n_fix_part)
{
    __shared__ uint32_t s_red[2][4];
    __shared__ uint32_t s_cp[3][4];
    const SurfelSet cur = M.s[st->cur];
    const uint32_t offset = st->offset;
    const uint32_t garbage = st->garbage, garbage_prev = st->garbage_prev, n_slots = st->cull_n;
    post_compact_fill(st, alive, tile_dead, blockIdx.x * PIX_BLOCK + threadIdx.x, gridDim.x * PIX_BLOCK);
    const int q = blockIdx.x * PIX_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // block 0 also publishes the frame totals: it is dispatched first, so its longer chain (all counts, the cull's
    // partials, DevState, the frame log) overlaps with the other blocks instead of trailing them
    const bool last = blockIdx.x == 0;
    if (!last && blk_cnt[blockIdx.x].x == 0u) return;             // no new surfel in this block's pixels (sky, border)
    const int upto = last ? (int)gridDim.x : (int)blockIdx.x;     // the totals block needs every count
    uint32_t pn = 0, pf = 0, tn = 0;                              // prefix of new; totals (last block only)
    for (int b = threadIdx.x; b < upto; b += PIX_BLOCK) {
        const uint2 c = blk_cnt[b];
        if (b < (int)blockIdx.x) pn += c.x;
        if (last) { pf += c.y; tn += c.x; }
    }
    uint32_t cv = 0, cs = 0, ck = 0;                              // visible / splat-skipped (/ killed) partials of the cull kernel
    if (last) {
        if (lazy_part) {
            for (uint32_t b = threadIdx.x; b < n_compact_part; b += PIX_BLOCK) { const uint4 c = lazy_part[b]; cv += c.x; cs += c.y; ck += c.z; }
            if (fix_part && st->cap_binds)         // the conflict cap bound: the fixup resurrected surfels (and drew them)
                for (uint32_t b = threadIdx.x; b < n_fix_part; b += PIX_BLOCK) { const uint2 c = fix_part[b]; cv += c.x; ck -= c.y; }
        } else
            for (uint32_t b = threadIdx.x; b < n_compact_part; b += PIX_BLOCK) { const uint2 c = compact_part[b]; cv += c.x; cs += c.y; }
        cv = wave_sum_u32(cv); cs = wave_sum_u32(cs); ck = wave_sum_u32(ck);
        if (lane == 0) { s_cp[0][wave] = cv; s_cp[1][wave] = cs; s_cp[2][wave] = ck; }
    }
    pn = wave_sum_u32(pn); pf = wave_sum_u32(pf); tn = wave_sum_u32(tn);
    if (lane == 0) { s_red[0][wave] = pn; s_red[1][wave] = pf; }
    __shared__ uint32_t s_tn[4];
    if (lane == 0) s_tn[wave] = tn;
    __syncthreads();
    const uint32_t prefix = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    if (last && threadIdx.x == 0) {
        const uint32_t ntot = s_tn[0] + s_tn[1] + s_tn[2] + s_tn[3];
        const uint32_t ftot = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        if (n_compact_part) {
            st->visible_count = s_cp[0][0] + s_cp[0][1] + s_cp[0][2] + s_cp[0][3];
            if (!fix_part) st->n_splat_skipped = s_cp[1][0] + s_cp[1][1] + s_cp[1][2] + s_cp[1][3];   // (k_pass_fixup published it already)
        }
        uint32_t garbage_now = garbage;
        if (lazy_part) {                          // the cull folded its finalize step in: complete the kill bookkeeping
            const uint32_t killed = s_cp[2][0] + s_cp[2][1] + s_cp[2][2] + s_cp[2][3];
            garbage_now = garbage_prev + killed;
            st->garbage = garbage_now;
            st->n_kill = killed;
        }
        st->unstable_count = ntot;
        st->fused_count = ftot;
        st->data_count = ntot + ftot;
        if ((uint64_t)offset + ntot > (uint64_t)fp.max_vertices) {
            st->error = -2;          // SM_E_CAPACITY: the frame's new surfels are dropped (no block writes, see below)
            st->append_n = 0;
            st->count = offset;
        } else {
            st->append_n = ntot;
            st->count = offset + ntot;
        }
        {
            const uint32_t fr = st->stat_frames + 1u;
            st->stat_frames = fr;
            if (host_stat)
                __hip_atomic_store(host_stat, ((unsigned long long)fr << 32) | (unsigned long long)st->count, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (fp.log_frame && log) {
            FrameLog e;
            e.tick = (uint32_t)fp.time; e.n_before = n_slots - garbage_prev; e.n_after_cull = offset - garbage_now; e.n_kill = st->n_kill;
            e.conflict_count = st->conflict_count; e.visible_count = st->visible_count;
            e.fused_count = ftot; e.unstable_count = ntot; e.n_static = st->n_static;
            e.n_conf_skipped = st->n_conf_skipped; e.n_splat_skipped = st->n_splat_skipped; e.n_slots = n_slots;
            st->n_conf_skipped = 0;
            log[st->frames_logged % FRAME_LOG_LEN] = e;
            st->frames_logged = st->frames_logged + 1;
        }
    }
    const int word0 = blockIdx.x * (PIX_BLOCK / 64);
    const int nwords = (fp.P + 63) >> 6;
    if (word0 + wave >= nwords) return;                         // wave-uniform
    const uint64_t mw = validmask[word0 + wave] & ~fusedmask[word0 + wave];
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += (uint32_t)__popcll(validmask[word0 + w] & ~fusedmask[word0 + w]);
    const uint32_t slot = offset + prefix + before + (uint32_t)__popcll(mw & ((1ull << lane) - 1ull));
    LocalSurfel L;
    // beyond capacity the frame is dropped anyway (see the totals above)
    const bool wr = ((mw >> lane) & 1ull) && (uint64_t)slot < (uint64_t)fp.max_vertices &&
                    local_surfel(q, fp, depthT, rgbsT, xs, ys, L);
    float3 pw = make_float3(0.f, 0.f, 0.f);
    if (wr) pw = write_new_surfel(cur, slot, L, fp);
    bounds_expand_wave(tb, wr, slot / (uint32_t)TILE, pw.x, pw.y, pw.z, (float)fp.time, false);
}

// The raw per-frame surfel cloud of FeedbackBuffer::compute (src/FeedbackBuffer.cpp:85-145, surfel_feedback.vert:25-63,
// surfel_feedback.geom:17-26): every checkerboard pixel with 0 < z < maxDepth as a CAMERA-frame surfel
// (pos, 0.9 | colour, 0, time, time | normal, radius), no neighbour test.  One record slot per pixel + a flag; the host
// keeps the flagged ones in vertex order (x-outer / y-inner, src/FeedbackBuffer.cpp:47-54).  Not on the hot path: the
// reference fills this buffer every frame for the GUI's "Draw raw" view only (src/SurfelMapping.cpp:172).
__global__ __launch_bounds__(256) void k_raw_cloud(FrameParams fp, const float *__restrict__ depthT, const uint32_t *__restrict__ rgbsT,
                                                   const float *__restrict__ xs, const float *__restrict__ ys,
                                                   float4 *__restrict__ rec /* [P][3] */, uint8_t *__restrict__ flag)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= fp.P) return;
    LocalSurfel L;
    const bool ok = local_surfel(q, fp, depthT, rgbsT, xs, ys, L);       // fp.init_mode = 1: the feedback buffer's rules
    flag[q] = ok ? 1 : 0;
    if (!ok) return;
    rec[(size_t)q * 3 + 0] = make_float4(L.pos.x, L.pos.y, L.pos.z, 0.9f);                             // surfel_feedback.vert:96
    rec[(size_t)q * 3 + 1] = make_float4(__uint_as_float(encode_color(L.cr, L.cg, L.cb, L.sem)), 0.0f, (float)fp.time, (float)fp.time);
    rec[(size_t)q * 3 + 2] = make_float4(L.nrm.x, L.nrm.y, L.nrm.z, L.radius);
}

// rebuild of the tile bounds from the stored model (upload / import / device append)
__global__ void k_tile_bounds_reset(uint32_t *__restrict__ tb, uint32_t first, uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint4 *b = reinterpret_cast<uint4 *>(tb + (size_t)(first + t) * 8);
    b[0] = make_uint4(0u, 0u, 0u, 0u);
    b[1] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(256) void k_tile_bounds_build(Model M, const DevState *__restrict__ st, uint32_t *__restrict__ tb,
                                                           uint32_t first_surfel)
{
    const SurfelSet cur = M.s[st->cur];
    const uint32_t N = st->count;
    const uint32_t k = first_surfel + blockIdx.x * 256u + threadIdx.x;
    const bool a = k < N;
    float4 v = make_float4(0.f, 0.f, 0.f, 1.f);
    float t = 0.f;
    if (a) { v = cur.pos_conf[k]; t = cur.time[k]; }
    bounds_expand_wave(tb, a, k / (uint32_t)TILE, v.x, v.y, v.z, t, !(v.w > 0.0f));
}

// survivors per creation-frame segment after the pending cull (multi-GPU bookkeeping; the conflict cap
// is off in sharded runs, so keep = ~(zm | cm & dm))
__global__ void k_seg_counts(const DevState *__restrict__ st, const uint64_t *__restrict__ cm,
                             const uint64_t *__restrict__ dm, const uint64_t *__restrict__ zm,
                             const uint32_t *__restrict__ tile_keep_prefix, const uint32_t *__restrict__ group_keep_base,
                             const uint32_t *__restrict__ seg_lstart, int nseg, uint32_t *__restrict__ out)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nseg) return;
    const uint32_t N = st->cull_n;
    auto kept_before = [&](uint32_t x) -> uint32_t {
        if (x >= N) return st->count;                       // k_scan_cull already published the survivor total
        const uint32_t tile = x / TILE, within = x % TILE;
        uint32_t sum = tile_keep_prefix[tile] + group_keep_base[tile / GROUP];
        const uint32_t w0 = tile * TILE_WORDS;
        for (uint32_t w = 0; w <= within / 64; ++w) {
            const uint32_t word = w0 + w;
            const uint64_t base = (uint64_t)word * 64u;
            if (base >= N) break;
            const uint64_t rem = (uint64_t)N - base;
            const uint64_t valid = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
            uint64_t keep = ~(zm[word] | (cm[word] & dm[word])) & valid;
            if (w == within / 64) keep &= ((1ull << (within % 64)) - 1ull);
            sum += (uint32_t)__popcll(keep);
        }
        return sum;
    };
    out[sidx] = kept_before(seg_lstart[sidx + 1]) - kept_before(seg_lstart[sidx]);
}

// ---------------------------------------------------------------------------------------------
// export helpers (not on the hot path)
// ---------------------------------------------------------------------------------------------
__global__ void k_export_aos(Model M, const DevState *__restrict__ st, float *__restrict__ dst, uint32_t first, uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const SurfelSet cur = M.s[st->cur];
    const uint32_t k = first + t;
    const float4 pc = cur.pos_conf[k], nr = cur.norm_rad[k];
    float *o = dst + (size_t)t * 12;
    o[0] = pc.x; o[1] = pc.y; o[2] = pc.z; o[3] = pc.w;
    o[4] = __uint_as_float(cur.color[k]); o[5] = 0.0f; o[6] = cur.init_time[k]; o[7] = cur.time[k];
    o[8] = nr.x; o[9] = nr.y; o[10] = nr.z; o[11] = nr.w;
}

__global__ void k_import_aos(Model M, const DevState *__restrict__ st, const float *__restrict__ src, uint32_t first, uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const SurfelSet cur = M.s[st->cur];
    const uint32_t k = first + t;
    const float *o = src + (size_t)t * 12;
    cur.pos_conf[k] = make_float4(o[0], o[1], o[2], o[3]);
    cur.color[k] = __float_as_uint(o[4]);
    cur.init_time[k] = o[6];
    cur.time[k] = o[7];
    cur.norm_rad[k] = make_float4(o[8], o[9], o[10], o[11]);
}

// index-map textures (index_map.vert:61-63) materialised from the key map, row-major output
__global__ void k_export_index(Model M, const DevState *__restrict__ st, FrameParams fp,
                               const uint64_t *__restrict__ keyT, int32_t *__restrict__ id_out,
                               float4 *__restrict__ vc, float4 *__restrict__ ct, float4 *__restrict__ nr)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= fp.P) return;
    const int j = p / fp.W, i = p - j * fp.W;
    const uint64_t key = keyT[(size_t)i * fp.H + j];
    const SurfelSet cur = M.s[st->cur];
    int32_t id = 0;
    float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
    if (key != KEY_EMPTY) {
        id = (int32_t)(uint32_t)(key & 0xFFFFFFFFull);
        const float4 pc = cur.pos_conf[id];
        const float3 ph = xform3(fp.t_inv, pc.x, pc.y, pc.z);
        a = make_float4(ph.x, ph.y, ph.z, pc.w);
        b = make_float4(__uint_as_float(cur.color[id]), 0.0f, cur.init_time[id], cur.time[id]);
        const float4 n = cur.norm_rad[id];
        const float3 nn = normalize3(rot3(fp.t_inv, n.x, n.y, n.z));
        c = make_float4(nn.x, nn.y, nn.z, n.w);
    }
    if (id_out) id_out[p] = id;
    if (vc) vc[p] = a;
    if (ct) ct[p] = b;
    if (nr) nr[p] = c;
}


// ---------------------------------------------------------------------------------------------
// Novel-view renderer: GlobalModel::renderImage (src/GlobalModel.cpp:772-833),
// draw_image.vert:18-28, draw_image_adaptive.geom:38-83, draw_image.frag:11-19.  Every surfel is a
// screen-space quad (two triangles) with a per-fragment circle test, z-buffered with GL_LESS.
// Rasterisation (DESIGN.md "Renderer"): 24.8 fixed-point vertices, 64-bit edge functions, top-left fill
// rule, barycentrics in double -> float, the same 64-bit atomicMin key (d24 << 32 | id) as the index map.
// ---------------------------------------------------------------------------------------------
struct RVert { long long X, Y; float zw, tx, ty; };

struct RenderParams {
    float t_inv[16];
    float fx, fy, cx, cy, cols, rows;
    int w, h;
};

__device__ __forceinline__ long long edge64(const RVert &a, const RVert &b, long long px, long long py)
{
    return (b.X - a.X) * (py - a.Y) - (b.Y - a.Y) * (px - a.X);
}

__device__ __forceinline__ bool top_left(const RVert &a, const RVert &b)
{
    const long long dx = b.X - a.X, dy = b.Y - a.Y;
    return (dy == 0 && dx > 0) || (dy < 0);
}

__device__ __forceinline__ void raster_tri(RVert v0, RVert v1, RVert v2, int w, int h, uint32_t id, uint64_t *__restrict__ key)
{
    long long area = edge64(v0, v1, v2.X, v2.Y);
    if (area == 0) return;
    if (area < 0) { const RVert t = v1; v1 = v2; v2 = t; area = -area; }
    long long minX = min(v0.X, min(v1.X, v2.X)), maxX = max(v0.X, max(v1.X, v2.X));
    long long minY = min(v0.Y, min(v1.Y, v2.Y)), maxY = max(v0.Y, max(v1.Y, v2.Y));
    long long x0 = (minX - 128) >> 8, x1 = (maxX - 128) >> 8, y0 = (minY - 128) >> 8, y1 = (maxY - 128) >> 8;
    x0 = max(x0, 0ll); y0 = max(y0, 0ll);
    x1 = min(x1, (long long)w - 1); y1 = min(y1, (long long)h - 1);
    const int b0 = top_left(v1, v2) ? 0 : -1, b1 = top_left(v2, v0) ? 0 : -1, b2 = top_left(v0, v1) ? 0 : -1;
    for (long long py = y0; py <= y1; ++py)
        for (long long px = x0; px <= x1; ++px) {
            const long long cx = px * 256 + 128, cy = py * 256 + 128;
            const long long e0 = edge64(v1, v2, cx, cy), e1 = edge64(v2, v0, cx, cy), e2 = edge64(v0, v1, cx, cy);
            if (e0 + b0 < 0 || e1 + b1 < 0 || e2 + b2 < 0) continue;
            const float l0 = (float)((double)e0 / (double)area), l1 = (float)((double)e1 / (double)area),
                        l2 = (float)((double)e2 / (double)area);
            const float tx = (l0 * v0.tx + l1 * v1.tx) + l2 * v2.tx;
            const float ty = (l0 * v0.ty + l1 * v1.ty) + l2 * v2.ty;
            if (tx * tx + ty * ty > 1.0f) continue;                         // draw_image.frag:13-14
            const float zw = (l0 * v0.zw + l1 * v1.zw) + l2 * v2.zw;
            if (!(zw >= 0.0f && zw <= 1.0f)) continue;
            const uint32_t d24 = (uint32_t)floor((double)zw * 16777215.0 + 0.5);
            if (d24 >= 16777215u) continue;
            atomicMin((unsigned long long *)&key[(size_t)py * w + px], (unsigned long long)(((uint64_t)d24 << 32) | id));
        }
}

__global__ __launch_bounds__(256) void k_render_splat(Model M, const DevState *__restrict__ st, RenderParams rp,
                                                      uint64_t *__restrict__ key)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= st->count) return;
    const SurfelSet cur = M.s[st->cur];
    const float maxDepth = 200.0f;                                          // src/GlobalModel.cpp:797
    const float4 pc = cur.pos_conf[k];
    const float3 ph = xform3(rp.t_inv, pc.x, pc.y, pc.z);                  // draw_image.vert:20
    if (ph.z >= maxDepth || ph.z <= 1.0f) return;                           // draw_image_adaptive.geom:41
    const float4 nr = cur.norm_rad[k];
    const float3 n = normalize3(rot3(rp.t_inv, nr.x, nr.y, nr.z));
    const float r = nr.w;
    float3 x, y;
    if (ph.z > 5.0f) {                                                      // :47-52
        const float3 tn = make_float3(0.0f, 0.0f, 1.0f);
        const float3 a = normalize3(make_float3(tn.y - tn.z, -tn.x, tn.x));
        x = make_float3(a.x * r * 1.41421356f, a.y * r * 1.41421356f, a.z * r * 1.41421356f);
        y = cross3(tn, x);
    } else {                                                                // :53-63
        const float cosAngle = dot3(ph, n) / (sqrtf(dot3(ph, ph)) * sqrtf(dot3(n, n)));
        const float radius = r / (1.0f + 0.5f * fabsf(cosAngle));
        const float3 a = normalize3(make_float3(n.y - n.z, -n.x, n.x));
        x = make_float3(a.x * radius * 1.41421356f, a.y * radius * 1.41421356f, a.z * radius * 1.41421356f);
        y = cross3(n, x);
    }
    const float sx[4] = {x.x, y.x, -y.x, -x.x}, sy[4] = {x.y, y.y, -y.y, -x.y}, sz[4] = {x.z, y.z, -y.z, -x.z};
    const float tcx[4] = {-1.0f, 1.0f, -1.0f, 1.0f}, tcy[4] = {-1.0f, -1.0f, 1.0f, 1.0f};
    RVert rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float X = ph.x + sx[q], Y = ph.y + sy[q], Z = ph.z + sz[q];
        if (!(Z > 0.0f)) return;                                            // would need polygon clipping: not drawn
        const float xn = ((((rp.fx * X) / Z) + rp.cx) - (rp.cols * 0.5f)) / (rp.cols * 0.5f);   // projectPoint :31-36
        const float yn = ((((rp.fy * Y) / Z) + rp.cy) - (rp.rows * 0.5f)) / (rp.rows * 0.5f);
        const float zn = (2.0f * Z / maxDepth) - 1.0f;
        const float xw = (rp.cols * 0.5f) * xn + (rp.cols * 0.5f), yw = (rp.rows * 0.5f) * yn + (rp.rows * 0.5f);
        if (!(fabsf(xw) < 1.0e6f && fabsf(yw) < 1.0e6f)) return;
        rv[q].X = (long long)floor((double)xw * 256.0 + 0.5);
        rv[q].Y = (long long)floor((double)yw * 256.0 + 0.5);
        rv[q].zw = 0.5f * zn + 0.5f;
        rv[q].tx = tcx[q]; rv[q].ty = tcy[q];
    }
    raster_tri(rv[0], rv[1], rv[2], rp.w, rp.h, k, key);                    // triangle strip
    raster_tri(rv[2], rv[1], rv[3], rp.w, rp.h, k, key);
}

__global__ void k_render_resolve(Model M, const DevState *__restrict__ st, const uint64_t *__restrict__ key, int npix,
                                 uint8_t *__restrict__ bgr, uint8_t *__restrict__ sem)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const uint64_t kk = key[p];
    uint8_t b = 0, g = 0, r = 0, s = 0;
    if (kk != KEY_EMPTY) {
        const uint32_t sc = M.s[st->cur].color[(uint32_t)(kk & 0xFFFFFFFFull)];
        b = (uint8_t)(sc & 0xFFu); g = (uint8_t)((sc >> 8) & 0xFFu); r = (uint8_t)((sc >> 16) & 0xFFu);   // vBGR = srgb.wzy
        s = (uint8_t)(((sc >> 24) & 0xFFu) + 1u);                                                              // class + 1
    }
    bgr[(size_t)p * 3] = b; bgr[(size_t)p * 3 + 1] = g; bgr[(size_t)p * 3 + 2] = r;
    sem[p] = s;
}

}  // namespace sm
