// sm_k_shard.h -- ONE stream over several GPUs, slot-addressed: the W*H cap across ranks, the end of a sharded frame, the compaction between frames.
// Part of sm_kernels.h (included there, in order, inside namespace sm); shader citations: /root/reference/src/Shaders/<file>:<line>.
#pragma once

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
        st->first_live = 0u; st->fl_dirty2[0] = 0u; st->fl_dirty2[1] = 0u;  // the first live surfel of the union moves to slot 0
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
