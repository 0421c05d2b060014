// sm_k_prep.h -- the frame's preparation: tile skip flags, image planes (p0a + pack + transpose), the depth pre-processing chain p0a..p0e as one LDS-tiled stage, the settle step of a sharded frame.
// Part of sm_kernels.h (included there, in order, inside namespace sm); shader citations: /root/reference/src/Shaders/<file>:<line>.
#pragma once

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
// Per-tile skip flags of the frame (bit 0: outside the conflict view volume, conflict.vert:35; bit 1: cannot reach the index
// map, index_map.vert:45-55 incl. the timeDelta gate), from the tile bounds as they stand at frame start.
// ---------------------------------------------------------------------------------------------
// The flags of the next (up to) 64 tiles of a workgroup (k_conflict, on the frames that compact), corner-parallel: the 8 box corners of a tile go to 8
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
    // ... and, in the two-launch frame, so do that frame's publisher and repair crew: on the rare frames where they change anything
    // (slow_frame) the flag workgroups wait for them first -- a resurrected surfel may stamp a tile
    const uint32_t *slow_conf_sub;   // that frame's conflict sub-counters, or null: nothing to wait for
    uint32_t slow_cap, slow_need;
    int slow_par;
};

__device__ __forceinline__ void tile_prep_block(const FrameParams &fp, const TilePrep &tp, uint32_t first_block = 0u)
{
    __shared__ uint32_t s_sk[2][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Concurrent with the previous frame's association (k_assoc_prep): the slot count that association will publish is
    // offset + (its candidate pixels); tiles it can still change must not be skipped on stale bounds.  Those are the tiles
    // from the old end on (appends) and the tiles that frame drew into the index map (a fuse moves a surfel: its box may
    // grow) -- k_surfel_pass stamped exactly those with the frame's time, so "stamped last frame" means "do not skip".
    if (tp.slow_conf_sub && slow_frame(tp.st, tp.slow_conf_sub, tp.slow_cap, tp.slow_par, lane))
        wait_slow_frame(const_cast<DevState *>(tp.st), tp.slow_par, tp.slow_need);
    uint32_t N = tp.st->count, first_new_tile = 0xFFFFFFFFu;
    if (tp.grp_cand) {
        uint32_t d = 0;
        for (uint32_t g = lane; g < tp.n_grp; g += 64u) d += tp.grp_cand[g];
        const uint32_t off = tp.st->offset;
        N = off + wave_sum_u32(d);
        first_new_tile = off / (uint32_t)TILE;
    }
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    // ONE tile per thread, its eight box corners one after the other.  (Round 2 spread the corners over eight lanes -- an eighth of
    // the arithmetic per lane -- which at 7 000 tiles made this 256 workgroups of k_assoc_prep's grid: together with the
    // association and the image tiles more than the chip holds at once, and the last image tiles started 7 us into an 11 us
    // launch.  The arithmetic is ~200 instructions per tile either way; this form is 29 workgroups.)
    const uint32_t per_wg = blockDim.x, blk = blockIdx.x - first_block;
    uint32_t cskip = 0, sskip = 0;
    for (uint32_t t = blk * per_wg + threadIdx.x; t < ntiles; t += tp.nfb * per_wg) {
        const uint4 lo = ((const uint4 *)tp.tb)[(size_t)t * 2], hi = ((const uint4 *)tp.tb)[(size_t)t * 2 + 1];
        const uint32_t b0 = lo.x, b1 = lo.y, b2 = lo.z, b3 = lo.w, b4 = hi.x, b5 = hi.y, b6 = hi.z, b7 = hi.w;
        const float x0 = ord2f(~b0), y0 = ord2f(~b1), z0 = ord2f(~b2), x1 = ord2f(b4), y1 = ord2f(b5), z1 = ord2f(b6);
        const float gd = plane_guard(fp, x0, y0, z0, x1, y1, z1);
        bool finite = true, right = true, left_c = true, left_s = true, below = true, above = true;
        float zmin = 0.f, zmax = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float3 p = xform3(fp.t_inv, (c & 1) ? x1 : x0, (c & 2) ? y1 : y0, (c & 4) ? z1 : z0);
            finite = finite && (p.x - p.x == 0.0f) && (p.y - p.y == 0.0f) && (p.z - p.z == 0.0f);
            right = right && (fp.fx * p.x + (fp.cx - fp.cols - 2.0f) * p.z > gd);
            left_c = left_c && (fp.fx * p.x + (fp.cx - fp.stereo_border + 2.0f) * p.z < -gd);
            left_s = left_s && (fp.fx * p.x + (fp.cx + 2.0f) * p.z < -gd);
            below = below && (fp.fy * p.y + (fp.cy - fp.rows - 2.0f) * p.z > gd);
            above = above && (fp.fy * p.y + (fp.cy + 2.0f) * p.z < -gd);
            zmin = c ? fminf(zmin, p.z) : p.z;
            zmax = c ? fmaxf(zmax, p.z) : p.z;
        }
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
        const uint32_t tn = min((uint32_t)TILE, N - t * TILE);
        tp.tile_flags[t] = (uint8_t)f;
        if (f & 1u) { tp.wave_cnt[t] = make_uint4(0u, 0u, 0u, 0u); cskip += tn; }
        if (f & 2u) sskip += tn;
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
    uint32_t *nf;                     // new / fused sub-counter sets
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
        if (nn) atomicAdd(&a.nf[(word & 63u) * SUB_STRIDE], nn);
        if (nf) atomicAdd(&a.nf[SUB_SET + (word & 63u) * SUB_STRIDE], nf);
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
