// sm_k_pass.h -- the frames that only mark the dead: k_surfel_pass (p2 + p3 + p4 + p6 in one read of the model; candidate counts behind it) and the fixup step (frame state, the W*H cap's repair: k_pass_fixup, or the first workgroups of the next k_assoc_prep).
// Part of sm_kernels.h (included there, in order, inside namespace sm); shader citations: /root/reference/src/Shaders/<file>:<line>.
#pragma once

// ---------------------------------------------------------------------------------------------
// ONE pass over the surfels per frame (p2 + p3 + p4 + p6 of a frame whose cull only marks the dead):
// conflict test (conflict.vert:25-83, conflict.geom:13-24), confidence decrement (conflict.vert:72,
// update_conf.vert:11-27), cull (back_map.geom:15-28: the dead keep their slots) and index-map splat
// (index_map.vert:38-64) from ONE load of pos_conf and ONE world->camera transform per surfel.
//
// The "first W*H conflicts only" rule (conflictVbo holds W*H records, src/GlobalModel.cpp:54-57) needs the
// conflict total, which exists only after the pass: the pass therefore treats EVERY conflict as effective and
// leaves what the fixup step (fixup_repair) needs to take the surplus back, exactly, should the cap bind:
//   cm[word]        conflicts of the 64 slots of `word` (valid, not the id-0 surfel)
//   km[word]        slots this pass killed BECAUSE of a conflict (alive and conf > 0 before, conf - 1 <= 0)
//   wave_cnt[tile]  conflicts per 256-slot quarter of the tile (uint4)
//   undo[slot]      the confidence a surviving, decremented surfel had before (restoring by +1.0f would
//                   not be exact for every float)
// A tile (or a quarter of one) is settled in two phases through LDS: pass_tile_append (cheap superset test, lane compaction)
// and pass_flush (the exact tests over up to PASS_BATCH tiles' listed slots, then the tiles' bookkeeping).
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

// The same counts from inside k_surfel_pass's launch (two-launch frame): a few extra workgroups behind the ones
// that own tiles -- they start as soon as the first of those leave the chip.  Four blocks per round whatever the group size, so
// that the pass's register allocation is not touched (cand_count_block<16> holds 80 loads in flight: 149 VGPRs).
struct CandArgs {
    uint32_t n_pass;                     // workgroups of the launch that own tiles; the ones behind them count candidates (0 groups: none)
    uint32_t n_grp, cg;                  // candidate groups; association blocks per group (4, 8 or 16)
    int n_pix_blocks;
    const float *depthT, *xs, *ys;
    uint32_t *blk_cand, *grp_cand;
};

__device__ __forceinline__ void cand_count_group_lean(uint32_t g, const CandArgs &ca, const FrameParams &fp, uint32_t *s_w /* 16 words of LDS */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t gsum = 0;
#pragma unroll 1
    for (uint32_t r = 0; r < ca.cg; r += 4u) {                           // workgroup-uniform
        bool c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = (int)((g * ca.cg + r + (uint32_t)k) * (uint32_t)PIX_BLOCK + threadIdx.x);
            c[k] = candidate_pixel(min(q, fp.P - 1), fp, ca.depthT, ca.xs, ca.ys) & (q < fp.P);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t m = __ballot(c[k]);
            if (lane == 0) s_w[k * 4 + wave] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (wave == 0) {
            uint32_t v = 0;
            const int b = (int)(g * ca.cg + r) + lane;
            if (lane < 4) v = s_w[lane * 4] + s_w[lane * 4 + 1] + s_w[lane * 4 + 2] + s_w[lane * 4 + 3];
            if (lane < 4 && b < ca.n_pix_blocks) ca.blk_cand[b] = v;
            gsum += wave_sum_u32(v);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) ca.grp_cand[g] = gsum;
}

// A workgroup barrier that orders LDS traffic only: no fence on global memory, so a load issued before it (the next unit's
// prefetch) stays in flight across it.  Every barrier of the pass separates LDS producers from LDS consumers.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
// SPLIT: a workgroup's unit of work is one SPLIT-th of a tile (TILE_WORDS / SPLIT consecutive words; its index within the tile
// travels in bits 8.. of the batch entry's flags): the bookkeeping touches only the unit's words.
template <int SPLIT>
__device__ __forceinline__ void pass_flush(const SurfelSet &set, DevState *__restrict__ st, const FrameParams &fp,
                                           const uint2 *__restrict__ dcT, uint64_t *__restrict__ cm, uint64_t *__restrict__ km,
                                           uint4 *__restrict__ wave_cnt, uint64_t *__restrict__ alive,
                                           uint32_t *__restrict__ tile_dead, uint64_t *__restrict__ keyT, float *__restrict__ undo,
                                           uint32_t N, uint32_t exempt, uint32_t nb, uint32_t wave, int lane, PassAcc &acc,
                                           uint32_t *__restrict__ tb, PassLds &L)
{
    float4 *__restrict__ pc = set.pos_conf;
    const uint32_t tid = threadIdx.x;
    lds_barrier();                                     // the list, the batch table
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
                if (k[r] == exempt) st->fl_dirty2[fp.par] = 1u;      // "id 0" died: the publisher searches its successor
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
    lds_barrier();
    // ---- per tile of the batch: masks, alive words, dead count, quarter-tile conflict counts (a wave per tile, one lane per word)
    for (uint32_t b = wave; b < nb; b += 4u) {
        const uint32_t tile = L.btile[b], fl = L.bflag[b], upart = fl >> 8;
        uint32_t nc = 0, ng = 0;
        if (lane < TILE_WORDS && (SPLIT == 1 || (uint32_t)lane / (uint32_t)(TILE_WORDS / SPLIT) == upart)) {
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
            if (SPLIT == 1) wave_cnt[tile] = make_uint4(q0, q1, q2, q3);
            else if (SPLIT == 4) ((uint32_t *)&wave_cnt[tile])[upart] = upart == 0u ? q0 : upart == 1u ? q1 : upart == 2u ? q2 : q3;
            else ((uint2 *)&wave_cnt[tile])[upart] = upart == 0u ? make_uint2(q0, q1) : make_uint2(q2, q3);
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
    lds_barrier();
}

// ---- phase A of one tile: the cheap superset test over its 1 024 slots (wave <-> four consecutive words, loads in flight
// together); the slots that need the exact tests join the workgroup's list, which is flushed first if they would not fit.
// `it` counts the workgroup's visited tiles; `nb` the tiles in the current batch.  SPLIT / upart: the workgroup takes the
// upart-th SPLIT-th of the tile (RW = 4 / SPLIT words per wave).
template <int SPLIT>
__device__ __forceinline__ void pass_tile_append(const SurfelSet &set, DevState *__restrict__ st, const FrameParams &fp,
                                                 const uint2 *__restrict__ dcT, uint64_t *__restrict__ cm, uint64_t *__restrict__ km,
                                                 uint4 *__restrict__ wave_cnt, uint64_t *__restrict__ alive,
                                                 uint32_t *__restrict__ tile_dead, uint64_t *__restrict__ keyT, float *__restrict__ undo,
                                                 uint32_t N, uint32_t exempt, uint32_t tile, uint32_t wave, bool sk0, bool sk1,
                                                 bool any_dead, int lane, PassAcc &acc, uint32_t *__restrict__ tb, PassLds &L, uint32_t it,
                                                 uint32_t &nb, uint32_t &n_list, uint32_t upart,
                                                 const float4 *pre = nullptr /* SPLIT == 4: the wave's word, loaded by the caller while the previous unit was at work */)
{
    constexpr int RW = 4 / SPLIT;                      // words per wave
    const float4 *__restrict__ pc = set.pos_conf;
    const uint32_t tid = threadIdx.x;
    const uint32_t w0 = upart * (uint32_t)(TILE_WORDS / SPLIT) + wave * (uint32_t)RW;      // the wave's first word within the tile
    float4 v[RW];
    uint64_t valid[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const uint32_t k = (tile * TILE_WORDS + w0 + (uint32_t)r) * 64u + (uint32_t)lane;
        v[r] = (RW == 1 && pre) ? *pre : pc[min(k, N - 1u)];
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const uint32_t word = tile * TILE_WORDS + w0 + (uint32_t)r;
        const uint64_t base = (uint64_t)word * 64u;
        uint64_t range = 0ull;
        if (base < N) { const uint64_t rem = (uint64_t)N - base; range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull); }
        valid[r] = range & (any_dead ? alive[word] : ~0ull);
    }
    if (tid == 0) L.pend[(it + 1u) % 3u] = 0u;         // (last read two tiles ago: every thread has passed a barrier since)
    const float zs_max = fp.depth_cutoff * 1.5f;       // splat_one's far limit
    uint64_t m[RW];
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
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
    lds_barrier();                                     // this tile's demand; the previous tile's entries
    // The list length the decision uses is a REGISTER every thread advances identically (n_list), not L.n: the waves that are
    // through with the test start appending (atomicAdd on L.n) while others still evaluate it -- read from LDS, two waves could
    // see different lengths, disagree on flushing and part ways at the barriers inside (seen as a one-in-ten-runs surplus of
    // ~126 surfels on an 8-context run and a one-off abort in a compaction).  L.pend[it % 3] is stable until two tiles on.
    const uint32_t need = L.pend[it % 3u];
    if (n_list + need > (uint32_t)TILE || nb == (uint32_t)PASS_BATCH) {               // workgroup-uniform
        pass_flush<SPLIT>(set, st, fp, dcT, cm, km, wave_cnt, alive, tile_dead, keyT, undo, N, exempt, nb, wave, lane, acc, tb, L);
        nb = 0u;
        n_list = 0u;
        // (the tile's 16 KB again, from the cache: keeping them in registers across the flush cost the kernel 29 VGPRs -- 99
        //  instead of 70 -- and with them two waves per SIMD)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const uint32_t k = (tile * TILE_WORDS + w0 + (uint32_t)r) * 64u + (uint32_t)lane;
            v[r] = pc[min(k, N - 1u)];
        }
    }
    if (tid == 0) { L.btile[nb] = tile; L.bflag[nb] = (sk0 ? 1u : 0u) | (sk1 ? 2u : 0u) | (any_dead ? 4u : 0u) | (upart << 8); }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        uint32_t base = 0;
        if (lane == 0 && m[r]) base = atomicAdd(&L.n, (uint32_t)__popcll(m[r]));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((m[r] >> lane) & 1ull) {
            const uint32_t at = base + (uint32_t)__popcll(m[r] & ((1ull << lane) - 1ull));
            L.list[at] = (nb << 10) | ((w0 + (uint32_t)r) * 64u + (uint32_t)lane);
            L.pos[at] = v[r];
        }
    }
    ++nb;
    n_list += need;
}

// ---- phase A of up to FOUR quarter-tile units at once (k_surfel_pass<4>): the same quarter `upart` of four DIFFERENT tiles of the
// workgroup's sequence, one word per wave and unit -- the four 16-byte loads of a thread go out together.  (One unit after the
// other, even with the next unit's word prefetched, every unit was a dependent load -> test -> LDS append -> barrier: a
// workgroup with three visited units spent 4.5 us in phase A.)  tiles[r], and bit r of sk0m / sk1m / deadm, describe unit r < nun.
__device__ __forceinline__ void pass_units4_append(const SurfelSet &set, DevState *__restrict__ st, const FrameParams &fp,
                                                   const uint2 *__restrict__ dcT, uint64_t *__restrict__ cm, uint64_t *__restrict__ km,
                                                   uint4 *__restrict__ wave_cnt, uint64_t *__restrict__ alive,
                                                   uint32_t *__restrict__ tile_dead, uint64_t *__restrict__ keyT, float *__restrict__ undo,
                                                   uint32_t N, uint32_t exempt, const uint32_t (&tiles)[4], uint32_t nun, uint32_t sk0m,
                                                   uint32_t sk1m, uint32_t deadm, uint32_t wave, int lane, PassAcc &acc,
                                                   uint32_t *__restrict__ tb, PassLds &L, uint32_t it, uint32_t &nb, uint32_t &n_list, uint32_t upart)
{
    const float4 *__restrict__ pc = set.pos_conf;
    const uint32_t tid = threadIdx.x;
    const uint32_t wofs = upart * (uint32_t)(TILE_WORDS / 4) + wave;       // the wave's word within a tile
    float4 v[4];
    uint64_t valid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t k = (tiles[r] * TILE_WORDS + wofs) * 64u + (uint32_t)lane;       // (tiles[r >= nun] repeats a valid tile)
        v[r] = pc[min(k, N - 1u)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t word = tiles[r] * TILE_WORDS + wofs;
        const uint64_t base = (uint64_t)word * 64u;
        uint64_t range = 0ull;
        if ((uint32_t)r < nun && base < N) { const uint64_t rem = (uint64_t)N - base; range = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull); }
        valid[r] = range & (((deadm >> r) & 1u) ? alive[word] : ~0ull);
    }
    if (tid == 0) L.pend[(it + 1u) % 3u] = 0u;
    const float zs_max = fp.depth_cutoff * 1.5f;       // splat_one's far limit
    uint64_t m[4];
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool sk0 = (sk0m >> r) & 1u, sk1 = (sk1m >> r) & 1u;
        bool act = false;
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
    lds_barrier();                                     // the group's demand; the previous group's entries
    const uint32_t need = L.pend[it % 3u];             // (<= 4 x 256 = the list's capacity)
    if (n_list + need > (uint32_t)TILE || nb + nun > (uint32_t)PASS_BATCH) {          // workgroup-uniform (registers only: see pass_tile_append)
        pass_flush<4>(set, st, fp, dcT, cm, km, wave_cnt, alive, tile_dead, keyT, undo, N, exempt, nb, wave, lane, acc, tb, L);
        nb = 0u;
        n_list = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t k = (tiles[r] * TILE_WORDS + wofs) * 64u + (uint32_t)lane;
            v[r] = pc[min(k, N - 1u)];
        }
    }
    if (tid < nun) {
        L.btile[nb + tid] = tid == 0u ? tiles[0] : tid == 1u ? tiles[1] : tid == 2u ? tiles[2] : tiles[3];
        L.bflag[nb + tid] = ((sk0m >> tid) & 1u) | (((sk1m >> tid) & 1u) << 1) | (((deadm >> tid) & 1u) << 2) | (upart << 8);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t base = 0;
        if (lane == 0 && m[r]) base = atomicAdd(&L.n, (uint32_t)__popcll(m[r]));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((m[r] >> lane) & 1ull) {
            const uint32_t at = base + (uint32_t)__popcll(m[r] & ((1ull << lane) - 1ull));
            L.list[at] = ((nb + (uint32_t)r) << 10) | (wofs * 64u + (uint32_t)lane);
            L.pos[at] = v[r];
        }
    }
    nb += nun;
    n_list += need;
}

// The frame's preparation launch evaluated the tile skip flags (one byte per tile, loaded together with DevState).
// Workgroup <-> tile round-robin.  SPLIT = 4: four consecutive workgroups share a sequence of tiles, a quarter each (ca.n_pass
// is a multiple of SPLIT).  The newest ~150 tiles of a KITTI model hold ~900 surfels in view each; as ONE workgroup's list that
// is four rounds of the exact tests on four waves while most of the chip has nothing left to do (tools/pass_trace.py: such a
// workgroup left 9 us after its cheap test, the launch with it) -- as four workgroups' lists it is one round each.
template <int SPLIT>
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
                                                     CandArgs ca /* two-launch frame: the workgroups from ca.n_pass on count the frame's candidate pixels */,
                                                     unsigned long long *__restrict__ trace = nullptr /* SM_PASS_TRACE: 8 words per workgroup */)
{
    if (blockIdx.x >= ca.n_pass) {                      // workgroup-uniform
        __shared__ uint32_t s_cw[16];
        const uint32_t g = blockIdx.x - ca.n_pass;
        if (g == 0u && threadIdx.x == 0u) {
            // nothing of this launch moves the slot count: the association that follows appends from here.  (k_pass_fixup's
            // publisher wrote this word; in the two-launch frame it runs NEXT to that association and leaves it alone.)  The
            // doubled words of the other parity are the previous frame's: everybody who read them is through.
            st->offset = st->count;
            st->fl_dirty2[fp.par ^ 1] = 0u;
            st->slow_done[fp.par ^ 1] = 0u;
        }
        cand_count_group_lean(g, ca, fp, s_cw);
        return;
    }
    // (stamps go straight to memory: kept in registers until the exit they cost the kernel 30 more spilled scalars)
    unsigned long long *const tr = trace ? trace + (size_t)blockIdx.x * 8 : nullptr;
    if (tr && threadIdx.x == 0) { tr[0] = wall_clock64(); tr[1] = 0ull; tr[2] = 0ull; tr[4] = ~0ull; tr[5] = 0ull; }
    bool tr_first = true;
    // Workgroups are dispatched in blockIdx order, ~2 800 per us: the last of 2 048 enters the chip ~3 us after the first.  The
    // newest tiles -- the surfels the camera is looking at, i.e. the tiles with all the work -- are the highest ones, so the
    // mapping is reversed: block 0 takes the highest tile of the grid, and a workgroup with several tiles starts with its
    // highest (the flags of its first 64 tiles sit one per lane whatever the order).
    const uint32_t wid = ca.n_pass - 1u - blockIdx.x;                    // (its partial sums and sub-counters go by this)
    const uint32_t tile_grid = ca.n_pass / (uint32_t)SPLIT, bid = wid / (uint32_t)SPLIT, upart = wid % (uint32_t)SPLIT;
    __shared__ uint32_t s_a[4], s_b[4], s_c[4];
    __shared__ PassLds s_pass;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // flags and dead counts of this workgroup's first 64 tiles: addresses known without DevState, issued with it
    uint32_t m_flag = 0, m_dead = 0;
    {
        const uint64_t tl = min((uint64_t)bid + (uint64_t)lane * tile_grid, (uint64_t)tile_bound - 1u);
        m_flag = tile_flags[tl];
        m_dead = tile_dead[tl];
    }
    const uint32_t N = st->count;
    const uint32_t exempt = st->first_live;            // the surfel the reference addresses as id 0
    const SurfelSet set = M.s[st->cur];
    PassAcc acc = {0u, 0u, 0u};
    const uint32_t sskip = 0, cskip = 0;               // (the skip statistics come from the preparation launch's flag workgroups)
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    uint64_t skip0 = 0, skip1 = 0;
    const uint32_t n_it = bid < ntiles ? (ntiles - 1u - bid) / tile_grid + 1u : 0u;       // this workgroup's tiles: bid + iter * grid
    const bool desc = n_it <= 64u;                      // (all of them fit the one-per-lane flags: highest first)
    uint32_t n_visited = 0, n_batch = 0, n_list = 0;   // tiles this workgroup has read; tiles in the current batch; entries on its list
    {
        for (uint32_t i = threadIdx.x; i < (uint32_t)PASS_BATCH * 2u * TILE_WORDS; i += 256u) { (&s_pass.cm[0][0])[i] = 0u; (&s_pass.km[0][0])[i] = 0u; (&s_pass.gone[0][0])[i] = 0u; }
        if (threadIdx.x < (uint32_t)PASS_BATCH) s_pass.drew[threadIdx.x] = 0u;
        if (threadIdx.x < 3u) s_pass.pend[threadIdx.x] = 0u;
        if (threadIdx.x == 0) s_pass.n = 0u;
        __syncthreads();
    }
    if (SPLIT == 4 && desc && n_it) {
        // Quarter-tile units, four of the workgroup's visited tiles at a time (pass_units4_append)
        {
            const uint64_t tl = (uint64_t)bid + (uint64_t)lane * tile_grid;
            const uint32_t f = tl < ntiles ? m_flag : 3u;
            skip0 = __ballot((f & 1u) != 0u);
            skip1 = __ballot((f & 2u) != 0u);
        }
        uint64_t vis = ~(skip0 & skip1) & (n_it >= 64u ? ~0ull : ((1ull << n_it) - 1ull));
        while (vis) {                                   // workgroup-uniform: the next (up to) four visited tiles, newest first
            uint32_t tiles[4], nun = 0, sk0m = 0, sk1m = 0, deadm = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (vis) {
                    const int sl = 63 - __clzll((long long)vis);
                    vis &= ~(1ull << sl);
                    tiles[r] = bid + (uint32_t)sl * tile_grid;
                    sk0m |= (uint32_t)((skip0 >> sl) & 1ull) << r;
                    sk1m |= (uint32_t)((skip1 >> sl) & 1ull) << r;
                    deadm |= (lane_bcast(m_dead, sl) != 0u ? 1u : 0u) << r;
                    nun = (uint32_t)r + 1u;
                } else
                    tiles[r] = tiles[0];
            }
            const bool tr_now = tr && tr_first;
            tr_first = false;
            if (tr_now && threadIdx.x == 0) { tr[1] = wall_clock64(); tr[4] = tiles[0]; }
            pass_units4_append(set, st, fp, dcT, cm, km, wave_cnt, alive, tile_dead, keyT, undo, N, exempt, tiles, nun, sk0m, sk1m, deadm,
                               wave, lane, acc, tb, s_pass, n_visited, n_batch, n_list, upart);
            ++n_visited;
            if (tr_now && threadIdx.x == 0) { tr[2] = wall_clock64(); tr[5] = s_pass.n; }
        }
    } else
    for (uint32_t it = 0; it < n_it; ++it) {
        const uint32_t iter = desc ? n_it - 1u - it : it;
        const uint32_t tile = bid + iter * tile_grid;
        if (desc ? it == 0u : (iter & 63u) == 0u) {
            const uint32_t tile0 = desc ? bid : tile;   // the tile lane 0's flag belongs to
            const uint64_t tl = (uint64_t)tile0 + (uint64_t)lane * tile_grid;
            if (tile0 != bid) { m_flag = tile_flags[min(tl, (uint64_t)ntiles - 1u)]; m_dead = tile_dead[min(tl, (uint64_t)ntiles - 1u)]; }
            const uint32_t f = tl < ntiles ? m_flag : 3u;
            skip0 = __ballot((f & 1u) != 0u);
            skip1 = __ballot((f & 2u) != 0u);
        }
        const int sl = (int)(iter & 63u);
        const bool sk0 = (skip0 >> sl) & 1ull, sk1 = (skip1 >> sl) & 1ull;   // workgroup-uniform
        if (sk0 && sk1) continue;               // the bulk of the map once the camera has passed: not even read
        {
            const bool tr_now = tr && tr_first;
            tr_first = false;
            if (tr_now && threadIdx.x == 0) { tr[1] = wall_clock64(); tr[4] = tile; }
            pass_tile_append<SPLIT>(set, st, fp, dcT, cm, km, wave_cnt, alive, tile_dead, keyT, undo, N, exempt, tile, wave, sk0, sk1,
                                    lane_bcast(m_dead, sl) != 0u, lane, acc, tb, s_pass, n_visited, n_batch, n_list, upart);
            ++n_visited;
            if (tr_now && threadIdx.x == 0) { tr[2] = wall_clock64(); tr[5] = s_pass.n; }
        }
    }
    if (n_batch)                                       // workgroup-uniform
        pass_flush<SPLIT>(set, st, fp, dcT, cm, km, wave_cnt, alive, tile_dead, keyT, undo, N, exempt, n_batch, wave, lane, acc, tb, s_pass);
    __syncthreads();
    if (lane == 0) { s_a[wave] = acc.vis; s_b[wave] = acc.killed; s_c[wave] = acc.nconf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nv = s_a[0] + s_a[1] + s_a[2] + s_a[3], nk = s_b[0] + s_b[1] + s_b[2] + s_b[3];
        part[wid] = make_uint4(nv, sskip, nk, cskip);
        const uint32_t nc = s_c[0] + s_c[1] + s_c[2] + s_c[3];
        if (nc) atomicAdd(&conf_sub[(wid & 63u) * SUB_STRIDE], nc);      // 64 counters, <= 32 adders each: one load per lane to read the total
        if (nv) atomicAdd(&frame_sub[(wid & 63u) * SUB_STRIDE], nv);
        if (nk) atomicAdd(&frame_sub[SUB_SET + (wid & 63u) * SUB_STRIDE], nk);
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
    int on;                              // 1: this frame appends directly (k_associate_direct follows; no k_append_scan); 2: ... and its candidate pixels were counted by the pass's launch
    uint32_t *blk_cand, *grp_cand;       // out: candidate pixels per association block / per group of CAND_GROUP blocks (this frame)
    uint32_t n_grp, cg;                  // groups; association blocks per group (4, 8 or 16)
    int n_pix_blocks;
    const float *depthT, *xs, *ys;
    uint32_t *frame_sub;                 // 2 x 64 sub-counters: visible, killed (this frame's pass)
    uint32_t *nf_prev;                   // 2 x 64 sub-counters: new, fused of the PREVIOUS frame's association (the sets alternate where that association runs next to this publisher)
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
__device__ __forceinline__ void finalize_frame(DevState *__restrict__ st, uint32_t *__restrict__ nf /* that frame's new / fused sub-counter sets */,
                                               const uint2 *__restrict__ fix_prev, uint32_t n_fix_prev, FrameLog *__restrict__ log,
                                               uint32_t *s_red /* 16 words of LDS */)
{
    if (st->pend == 0u) return;                         // workgroup-uniform
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t un = 0, fu = 0, va = 0, rs = 0;
    if (wave == 0) {
        uint32_t *a = nf + lane * SUB_STRIDE, *b = nf + SUB_SET + lane * SUB_STRIDE;
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

__global__ __launch_bounds__(256) void k_frame_finalize(DevState *__restrict__ st, uint32_t *__restrict__ nf,
                                                        const uint2 *__restrict__ fix_prev, uint32_t n_fix_prev, FrameLog *__restrict__ log)
{
    __shared__ uint32_t s_red[16];
    finalize_frame(st, nf, fix_prev, n_fix_prev, log, s_red);
}

// ---------------------------------------------------------------------------------------------
// After k_surfel_pass: one workgroup publishes DevState (fixup_publisher); the others return at once unless the conflict cap
// binds (total > W*H: src/GlobalModel.cpp:54-57).  Then they take back every conflict beyond the first `cap`
// in slot order (fixup_repair): a surfel the pass killed because of such a conflict is resurrected (alive bit, dead count,
// splat), a surviving one gets its confidence back from the undo plane.  Conflict ordinals come from prefix sums of the
// per-quarter-tile counts.
//
// Two homes.  k_pass_fixup: a launch of its own between the pass and the association (frames whose caller waits, sharded
// streams, a held-back association that is flushed).  MERGED: the first workgroups of the NEXT frame's preparation launch
// (k_assoc_prep), next to the association they used to precede -- the two-launch frame.  There the publisher only
// completes statistics unless the cap binds or the "id 0" surfel died; every association / tile-flag workgroup of that launch
// tells the two cases apart for itself (slow_frame) and waits for the publisher and the repair crew only then (wait_slow_frame).
// What a MERGED publisher must not touch, because workgroups of its own launch read or write it: count (the association's first
// workgroup publishes the new one; N is taken from `offset`, which the pass's launch set), offset, first_live unless it changes,
// the dirty word (cleared one launch later), the host statistic.
// ---------------------------------------------------------------------------------------------
struct FixArgs {
    const uint64_t *cm, *km;
    const uint4 *wave_cnt;
    const uint8_t *tile_flags;
    const uint4 *part; uint32_t n_part;
    uint2 *fix_part;                     // [workers] (visible added, resurrected)
    uint64_t *alive; uint32_t *tile_dead;
    const uint32_t *conf_sub;            // the frame's 64 conflict sub-counters
    uint64_t *keyT;
    const float *undo;
    unsigned long long *host_stat;
    const uint2 *prep_part; uint32_t n_prep;     // the preparation launch's skip statistics (it evaluated the tile flags), or 0
    DirectArgs da;
    uint32_t *tb;                        // tile bounds: a tile drawn only through a resurrected surfel gets the frame's time stamp too
    uint32_t n_crew;                     // MERGED: workgroups behind the publisher that repair (0: the launch carries no fixup)
};

template <bool MERGED>
__device__ __forceinline__ void fixup_publisher(DevState *__restrict__ st, const FrameParams &fp, const FixArgs &x, uint32_t ctotal)
{
    __shared__ uint32_t s_c[4];
    __shared__ uint32_t s_fl;
    __shared__ uint32_t s_red9[9][4];
    const DirectArgs &da = x.da;
    const uint64_t *__restrict__ cm = x.cm, *__restrict__ km = x.km;
    const uint4 *__restrict__ wave_cnt = x.wave_cnt;
    const uint64_t *__restrict__ alive = x.alive;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cap = fp.conflict_cap;
    const bool cap_binds = ctotal > cap;
    const uint32_t N = MERGED ? st->offset : st->count;   // occupied slots: unchanged by a cull that only marks the dead
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    // ---- every load the publisher needs, issued together (each dependent round trip costs ~1 us on this single workgroup)
    const uint32_t pend = st->pend, cap_prev = st->cap_binds, g_in = st->garbage, old_first = st->first_live;
    const bool dirty = st->fl_dirty2[fp.par] != 0u;
    PendFields pf;
    pf.cull_n = st->cull_n; pf.garbage_prev = st->garbage_prev; pf.n_kill = st->n_kill; pf.visible = st->visible_count;
    pf.conflict = st->conflict_count; pf.n_static = st->n_static; pf.conf_skipped = st->n_conf_skipped;
    pf.splat_skipped = st->n_splat_skipped; pf.tick = st->pend_tick; pf.frames_logged = st->frames_logged;
    uint32_t red[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // new, fused, vis+, resurrected (previous frame) | conf-skip, splat-skip, visible, killed | candidates
    // the sums the pass / the previous association left in 64 sub-counters each (one load per lane; consumed: zeroed)
    if (wave == 0) {
        uint32_t *c = da.frame_sub + lane * SUB_STRIDE, *n = da.nf_prev + lane * SUB_STRIDE;
        red[6] = c[0]; red[7] = c[SUB_SET]; red[0] = n[0]; red[1] = n[SUB_SET];
        c[0] = 0u; c[SUB_SET] = 0u; n[0] = 0u; n[SUB_SET] = 0u;
    }
    // (the fixup partials of the previous frame are only meaningful if its conflict cap bound: masked after the
    //  reduction, so that no load waits for DevState)
    for (uint32_t b = threadIdx.x; b < da.n_fix_prev; b += 256u) { const uint2 c = da.fix_prev[b]; red[2] += c.x; red[3] += c.y; }
    if (x.n_prep) for (uint32_t b = threadIdx.x; b < x.n_prep; b += 256u) { const uint2 c = x.prep_part[b]; red[4] += c.x; red[5] += c.y; }
    else for (uint32_t b = threadIdx.x; b < x.n_part; b += 256u) { const uint4 c = x.part[b]; red[4] += c.w; red[5] += c.y; }
    // ---- one round of reductions
#pragma unroll
    for (int i = 0; i < 9; ++i) red[i] = wave_sum_u32(red[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) s_red9[i][wave] = red[i];
    }
    if (threadIdx.x == 0) s_fl = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t tot[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) tot[i] = s_red9[i][0] + s_red9[i][1] + s_red9[i][2] + s_red9[i][3];
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
                        for (uint32_t w = t * TILE_WORDS; w < word; ++w) pre += (uint32_t)__popcll(cm[w]);
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
        if (!MERGED || dirty) st->first_live = first_live;
        if (!MERGED) {
            st->fl_dirty2[fp.par] = 0u;
            st->offset = N;                         // the dead keep their slots until the next compaction
        }
        st->holes_last = 0u;
        if (da.on) {
            // provisional totals of the pass (k_associate_direct's first block publishes the new count; the statistics
            // are completed by finalize_frame / finalize_write once the association is through)
            st->visible_count = vis_tot;
            st->n_kill = kill_tot;
            st->pend = 1u;
            st->pend_tick = (uint32_t)fp.time;
        }
        if (!MERGED && x.host_stat)
            __hip_atomic_store(x.host_stat, ((unsigned long long)st->stat_frames << 32) | (unsigned long long)N, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        if (MERGED && (cap_binds || dirty)) {       // somebody waits for this
            st->slow_frames = st->slow_frames + 1u;
            __hip_atomic_fetch_add(&st->slow_done[fp.par], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the cap binds: worker `wi` of `nwg` takes the conflicts beyond the first `cap` back in its tiles (wi, wi + nwg, ...)
__device__ __forceinline__ void fixup_repair(const Model &M, DevState *__restrict__ st, const FrameParams &fp, const FixArgs &x,
                                             uint32_t wi, uint32_t nwg, uint32_t N)
{
    __shared__ uint32_t s_a[4], s_b[4], s_c[4];
    const uint64_t *__restrict__ cm = x.cm, *__restrict__ km = x.km;
    const uint4 *__restrict__ wave_cnt = x.wave_cnt;
    uint64_t *__restrict__ alive = x.alive; uint32_t *__restrict__ tile_dead = x.tile_dead;
    uint64_t *__restrict__ keyT = x.keyT; const float *__restrict__ undo = x.undo; uint32_t *__restrict__ tb = x.tb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cap = fp.conflict_cap;
    const uint32_t ntiles = (N + TILE - 1) / TILE;
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
        const bool nosplat = (x.tile_flags[tile] & 2u) != 0u;
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
        // frame's time stamp like a tile k_surfel_pass drew itself (pass_flush: "drawn at t" bounds the
        // last update of every surfel of the tile, and tells the next frame's tile flags which boxes may still grow)
        if (vis != vis_tile && lane == 0) atomicMax(&tb[(size_t)tile * 8 + 7], f2ord((float)fp.time));
        resurrected += res_wave;
    }
    __syncthreads();
    if (lane == 0) { s_a[wave] = vis; s_b[wave] = resurrected; }
    __syncthreads();
    if (threadIdx.x == 0) x.fix_part[wi] = make_uint2(s_a[0] + s_a[1] + s_a[2] + s_a[3], s_b[0] + s_b[1] + s_b[2] + s_b[3]);
}

// MERGED form: block b of the 1 + n_crew fixup workgroups that open k_assoc_prep's grid
__device__ __forceinline__ void fixup_merged_block(const Model &M, DevState *__restrict__ st, const FrameParams &fp, const FixArgs &x, uint32_t b)
{
    const int lane = threadIdx.x & 63;
    const uint32_t ctotal = wave_sum_u32(x.conf_sub[lane * SUB_STRIDE]);
    if (b == 0u) { fixup_publisher<true>(st, fp, x, ctotal); return; }
    const bool cap_binds = ctotal > fp.conflict_cap;
    if (!cap_binds && st->fl_dirty2[fp.par] == 0u) return;                  // (workgroup-uniform) the usual frame
    if (cap_binds) fixup_repair(M, st, fp, x, b - 1u, x.n_crew, st->offset);
    else if (threadIdx.x == 0) x.fix_part[b - 1u] = make_uint2(0u, 0u);
    __threadfence();                                   // every thread's repairs, before the one count the waiters acquire
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&st->slow_done[fp.par], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_pass_fixup(Model M, DevState *__restrict__ st, FrameParams fp, FixArgs x)
{
    const int lane = threadIdx.x & 63;
    const uint32_t nwg = gridDim.x - 1u;               // workers; workgroup 0 (dispatched first) publishes
    const uint32_t ctotal = wave_sum_u32(x.conf_sub[lane * SUB_STRIDE]);
    if (blockIdx.x == 0u) { fixup_publisher<false>(st, fp, x, ctotal); return; }
    const uint32_t wi = blockIdx.x - 1u;
    const DirectArgs &da = x.da;
    // ---- direct append: the candidate pixels of the frame, per association block and per group (workgroup-uniform loop)
    if (da.on == 1)                                    // (2: the pass's launch counted them already)
        for (uint32_t g = wi; g < da.n_grp; g += nwg) {          // (da.cg is uniform)
            if (da.cg == 4u) cand_count_block<4>(g, fp, da.depthT, da.xs, da.ys, da.n_pix_blocks, da.blk_cand, da.grp_cand);
            else if (da.cg == 8u) cand_count_block<8>(g, fp, da.depthT, da.xs, da.ys, da.n_pix_blocks, da.blk_cand, da.grp_cand);
            else cand_count_block<16>(g, fp, da.depthT, da.xs, da.ys, da.n_pix_blocks, da.blk_cand, da.grp_cand);
        }
    if (ctotal <= fp.conflict_cap) return;
    // ---- the cap binds: take the conflicts beyond the first `cap` back
    fixup_repair(M, st, fp, x, wi, nwg, st->count);
}

// standalone p6 (IndexMap::predictIndices) over the current model
__global__ __launch_bounds__(256) void k_splat(Model M, DevState *__restrict__ st, FrameParams fp,
                                               uint64_t *__restrict__ keyT)
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
            drew = splat_one(fp, v.x, v.y, v.z, cur.time[k], k, keyT);
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
