// sm_k_cull.h -- the cull of the frames that compact: conflict test (p2), scans, finalize, in-place stable compaction + splat (p3..p6); the index-map splat of one surfel.
// Part of sm_kernels.h (included there, in order, inside namespace sm); shader citations: /root/reference/src/Shaders/<file>:<line>.
#pragma once

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
    const uint32_t exempt = fp.no_exempt ? 0xFFFFFFFFu : st->first_live;   // the surfel the reference addresses as id 0
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
        st->fl_dirty2[0] = 0u; st->fl_dirty2[1] = 0u;    // (a two-launch frame's publisher, merged into this frame's preparation launch, left them to the next pass -- there is none here)
        st->slow_done[0] = 0u; st->slow_done[1] = 0u;
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
                        drew = splat_one(fp, pv[r].x, pv[r].y, pv[r].z, pt[r], k, keyT);
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
                    drew = splat_one(fp, v[r].x, v[r].y, v[r].z, tl[r], nid[r], keyT);
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
// A cull outside the frame path that does not compact (cleanPoints under deferred compaction): survivors keep their slots,
// so nothing depends on other tiles or even on the other words of a tile.  Each wave settles four 64-surfel words on its own
// -- apply the confidence decrement, clear the dead from the alive mask -- with no LDS, no barriers, no hand-off.  (Frames
// use k_surfel_pass, which tests, culls and splats in one read; this kernel applies the masks k_conflict left.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cull_lazy(Model M, const DevState *__restrict__ st, FrameParams fp,
                                                   const uint64_t *__restrict__ cm, const uint64_t *__restrict__ dm,
                                                   const uint64_t *__restrict__ zm, const uint32_t *__restrict__ tile_cnt,
                                                   const uint32_t *__restrict__ tile_allow,
                                                   uint64_t *__restrict__ alive, uint32_t *__restrict__ tile_dead)
{
    const uint32_t N = st->cull_n;
    const SurfelSet set = M.s[st->cull_src];
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool had_dead = st->garbage_prev != 0u, cap_binds = st->cap_binds != 0u;
    uint32_t iter = 0;
    uint32_t m_nconf = 0, m_nkill = 0, m_dead = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++iter) {
        if ((iter & 63u) == 0u) {                    // metadata of this workgroup's next 64 tiles, one per lane
            const uint64_t tl = (uint64_t)tile + (uint64_t)lane * gridDim.x;
            const bool in = tl < ntiles;
            const uint32_t tt = in ? (uint32_t)tl : 0u;
            m_nconf = tile_cnt[tt * 3]; m_nkill = tile_cnt[tt * 3 + 1];
            m_dead = had_dead ? tile_dead[tt] : 0u;
        }
        const int sl = (int)(iter & 63u);
        const uint32_t nconf = lane_bcast(m_nconf, sl), nkill = lane_bcast(m_nkill, sl);
        const uint32_t tdead = lane_bcast(m_dead, sl);
        const bool touched = nconf != 0u || nkill != 0u;                                  // workgroup-uniform
        if (!touched) continue;                                                           // the bulk of the map: not even read
        float4 pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                // unconditional, clamped: all loads of the lane in flight together
            const uint32_t kc = min((tile * TILE_WORDS + r * 4 + wave) * 64u + lane, N - 1u);
            pv[r] = set.pos_conf[kc];
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
        }
        if (killed && lane == 0) atomicAdd(&tile_dead[tile], killed);
    }
}
