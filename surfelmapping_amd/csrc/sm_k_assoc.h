// sm_k_assoc.h -- data association + fuse + append (p8..p11): k_associate_direct / k_assoc_prep (direct append), k_associate + k_append_scan (dense append on compacting frames), the raw feedback cloud.
// Part of sm_kernels.h (included there, in order, inside namespace sm); shader citations: /root/reference/src/Shaders/<file>:<line>.
#pragma once

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

// What the association of pixel q reads from the frame's planes and the key map, loaded by the caller ahead of a decision it
// has to take first (two-launch frame: associate_direct_block)
struct PixelLoads { float z, zl, zu, zr, zd; uint32_t c; uint64_t key; };

__device__ __forceinline__ PixelLoads load_pixel(int q, int i, int j, const FrameParams &fp, const float *__restrict__ depthT,
                                                 const uint32_t *__restrict__ rgbsT, const uint64_t *__restrict__ keyT)
{
    const int H = fp.H, W = fp.W;
    PixelLoads r;
    q = min(q, fp.P - 1);
    r.key = keyT[q];
    r.z = depthT[q];
    r.zl = depthT[i > 0 ? q - H : q];
    r.zu = depthT[j > 0 ? q - 1 : q];
    r.zr = depthT[i < W - 1 && q + H < fp.P ? q + H : q];
    r.zd = depthT[j < H - 1 && q + 1 < fp.P ? q + 1 : q];
    r.c = rgbsT[q];
    return r;
}

__device__ __forceinline__ bool local_surfel(int q, const FrameParams &fp, const float *__restrict__ depthT,
                                             const uint32_t *__restrict__ rgbsT, const float *__restrict__ xs,
                                             const float *__restrict__ ys, LocalSurfel &L, int qi = -1, int qj = 0,
                                             const PixelLoads *pre = nullptr)
{
    const int H = fp.H, W = fp.W;
    const int i = qi >= 0 ? qi : q / H, j = qi >= 0 ? qj : q - i * H;     // (qi, qj): the caller knows the column / row of q already
    // init_mode: xs/ys hold the FeedbackBuffer's own pixel coordinates (src/FeedbackBuffer.cpp:47-53); they
    // follow the association tables in the same arrays at offsets W and H
    const float x = fp.init_mode ? xs[W + i] : xs[i], y = fp.init_mode ? ys[H + j] : ys[j];
    const float inv_fx = fp.init_mode ? fp.inv_fx_fb : fp.inv_fx, inv_fy = fp.init_mode ? fp.inv_fy_fb : fp.inv_fy;
    const float z = pre ? pre->z : depthT[q];
    // clamp-to-edge neighbours: at the border the neighbour depth is the pixel's own (A1)
    const float zl = pre ? pre->zl : depthT[i > 0 ? q - H : q];
    const float zu = pre ? pre->zu : depthT[j > 0 ? q - 1 : q];
    const float zr = pre ? pre->zr : depthT[i < W - 1 ? q + H : q];
    const float zd = pre ? pre->zd : depthT[j < H - 1 ? q + 1 : q];
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
    const uint32_t c = pre ? pre->c : rgbsT[q];
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
                                                const float *__restrict__ ys, LocalSurfel &L, bool &is_valid,
                                                bool &is_fused, uint32_t *__restrict__ tb, uint32_t first_live,
                                                const uint64_t *__restrict__ own_alive = nullptr /* slot-addressed sharding: this rank's alive bits */,
                                                FuseMove *mv = nullptr /* given: the caller grows the tile boxes (fuse_bounds_block) */,
                                                int qi = -1, int qj = 0 /* column / row of q, if the caller has them */,
                                                const PixelLoads *pre = nullptr /* the pixel's plane / key-map words, if the caller loaded them */)
{
    is_valid = false;
    is_fused = false;
    uint32_t f_id = 0;                        // the surfel this lane fused into, and where it moved
    float f_x = 0.f, f_y = 0.f, f_z = 0.f;
    // (the key does not depend on the pixel's own surfel: its load is issued with the stencil's, not after the arithmetic)
    const uint64_t key_q = fp.init_mode ? KEY_EMPTY : pre ? pre->key : keyT[min(q, fp.P - 1)];
    if (q < fp.P && local_surfel(q, fp, depthT, rgbsT, xs, ys, L, qi, qj, pre)) {
        is_valid = true;
        const uint64_t key = key_q;
        const int32_t gid = (int32_t)(uint32_t)(key & 0xFFFFFFFFull);
        uint32_t id = 0;
        // data.vert:142 "id > 0": the slot of the first live surfel is the reference's id 0.  Slot-addressed sharding
        // (own_alive): ids are the single-GPU slot numbers on every rank, a rank owns exactly the slots whose alive bit it
        // holds, and only the owner of the winner tries to fuse it.
        bool mine = false;
        if (!fp.init_mode && key != KEY_EMPTY) {
            id = (uint32_t)gid;
            mine = id != first_live && (!own_alive || ((own_alive[id >> 6] >> (id & 63u)) & 1ull) != 0ull);
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

// Association + in-place fuse of the frames that compact (and of the frame after reset()): new surfels are only flagged here
// (two ballot words per wave); k_append_scan writes them densely behind the compacted model.
__global__ __launch_bounds__(PIX_BLOCK) void k_associate(Model M, const DevState *__restrict__ st, FrameParams fp,
                                                         const float *__restrict__ depthT,
                                                         const uint32_t *__restrict__ rgbsT,
                                                         const uint64_t *__restrict__ keyT,
                                                         const float *__restrict__ xs, const float *__restrict__ ys,
                                                         uint64_t *__restrict__ validmask, uint64_t *__restrict__ fusedmask,
                                                         uint2 *__restrict__ blk_cnt /* (new, fused) per block */,
                                                         uint32_t *__restrict__ tb)
{
    __shared__ uint32_t s_n[4], s_f[4];
    const SurfelSet cur = M.s[st->cur];
    const int q = blockIdx.x * PIX_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool is_valid, is_fused;
    LocalSurfel L;
    associate_pixel(q, cur, fp, depthT, rgbsT, keyT, xs, ys, L, is_valid, is_fused, tb, st->first_live);
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
    uint32_t *nf /* new, fused: 64 sub-counters each */, *tb;
    uint64_t *alive; uint32_t *tile_dead; uint32_t n_grp, cg; unsigned long long *host_stat;
    // two-launch frame: the frame's publisher / repair crew run in the SAME launch (fixup_merged_block); on the rare frames where
    // they change anything the association waits for them first
    const uint32_t *slow_conf_sub;   // the frame's conflict sub-counters, or null: the fixup ran in a launch of its own
    uint32_t slow_need;              // workgroups to wait for (publisher + crew)
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
// TWO consecutive pixels per thread, the workgroup covers blocks 2 wg and 2 wg + 1.  data.vert:88 keeps only
// the pixels with (int)x + (int)y odd -- half of every wave sat out the whole association with one pixel per lane -- and
// sm_create refuses image sizes where (int)xs[i] != i or (int)ys[j] != j, so the test is "(i + j) odd": of the pixels q0
// (even) and q0 + 1 in column-major order exactly one passes it, whatever H is (same column: j and j + 1; across the end
// of a column only if H is odd, (i, H - 1) and (i + 1, 0): i + H - 1 and i + 1 differ in parity; with H even an even q0
// never is the last pixel of a column).  The lane takes that one: every lane of the wave holds a pixel of the
// checkerboard, in pixel order, so ballots, ranks and slots are what they were -- with half the waves.
template <bool SHARD>
__device__ __forceinline__ void associate_direct_block(const AssocArgs &a, const ShardArgs &sh, const uint32_t wg)
{
    const uint32_t blk = wg * 2u;                        // first association block of the workgroup (an even one: same group as the next)
    const Model &M = a.M; DevState *__restrict__ st = a.st; const FrameParams &fp = a.fp;
    const float *__restrict__ depthT = a.depthT; const uint32_t *__restrict__ rgbsT = a.rgbsT; const uint64_t *__restrict__ keyT = a.keyT;
    const float *__restrict__ xs = a.xs, *__restrict__ ys = a.ys;
    const uint32_t *__restrict__ blk_cand = a.blk_cand, *__restrict__ grp_cand = a.grp_cand;
    uint32_t *__restrict__ nf_sub = a.nf, *__restrict__ tb = a.tb;
    uint64_t *__restrict__ alive = a.alive; uint32_t *__restrict__ tile_dead = a.tile_dead;
    const uint32_t n_grp = a.n_grp; unsigned long long *__restrict__ host_stat = a.host_stat;
    __shared__ uint32_t s_v[4], s_n[4], s_f[4];
    __shared__ uint32_t s_hole[12], s_dead[2];          // empty slots of this block: 6 alive words (lo, hi), 2 tiles
    __shared__ uint32_t s_tag[FB_SLOTS], s_box[FB_SLOTS * 8];
    __shared__ uint32_t s_slow;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // two-launch frame: the frame's conflict total, for the one decision below (wave 0 only; first load of the workgroup)
    uint32_t gate_cs = 0, gate_dirty = 0;
    if (!SHARD && a.slow_conf_sub && wave == 0) { gate_cs = a.slow_conf_sub[lane * SUB_STRIDE]; gate_dirty = st->fl_dirty2[fp.par]; }
    if (threadIdx.x < 12) s_hole[threadIdx.x] = 0u;
    if (threadIdx.x < 2) s_dead[threadIdx.x] = 0u;
    // candidates before this block = the groups before its group + the blocks of its group before it: a few loads per lane,
    // issued together with DevState, one wave reduction (every wave computes it for itself)
    const uint32_t grp = blk / a.cg, in_grp = blk % a.cg;
    uint32_t pre = (lane < (int)in_grp) ? blk_cand[grp * a.cg + lane] : 0u;
    pre += group_sum_lane(grp_cand, grp, n_grp, lane);
    const SurfelSet cur = M.s[st->cur];
    const uint32_t offset = st->offset;
    const int q0 = (int)blk * PIX_BLOCK + 2 * (int)threadIdx.x;
    int q = q0, qi = q0 / fp.H, qj = q0 - (q0 / fp.H) * fp.H;
    if (((qi + qj) & 1) == 0) { q = q0 + 1; if (++qj == fp.H) { qj = 0; ++qi; } }     // q0 is off the checkerboard: its successor is on it
    bool is_valid, is_fused;
    LocalSurfel L;
    FuseMove mv;
    // The pixel's loads go out with the ones above, BEFORE the one decision of the two-launch frame: did the pass leave work for
    // the publisher / the repair crew of this very launch (the conflict cap binds, "id 0" died)?  One more load per lane and a
    // wave reduction; on such a frame the wave waits for them and takes the words they may have changed again.
    PixelLoads pl = load_pixel(q, qi, qj, fp, depthT, rgbsT, keyT);
    uint32_t first_live = st->first_live;
    if (!SHARD && a.slow_conf_sub) {                    // workgroup-uniform
        // Wave 0 takes the decision for the workgroup (its sub-counter load went out first, see the top); the others meet it at a
        // bare s_barrier -- no fence, so that nobody's loads above have to land first -- and read the verdict from LDS.  (Every
        // wave reading the 64 sub-counter lines for itself cost the launch 1 us at 1242x375 and 5 us at 1920x1080.)
        if (wave == 0) {
            const bool slow = (wave_sum_u32(gate_cs) > fp.conflict_cap) | (gate_dirty != 0u);
            if (lane == 0) s_slow = slow ? 1u : 0u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s_slow) {
            wait_slow_frame(st, fp.par, a.slow_need);
            pl.key = __hip_atomic_load(&keyT[min(q, fp.P - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            first_live = __hip_atomic_load(&st->first_live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    associate_pixel(q, cur, fp, depthT, rgbsT, keyT, xs, ys, L, is_valid, is_fused, tb, first_live,
                    SHARD ? alive : nullptr, &mv, qi, qj, &pl);
    const uint64_t vw = __ballot(is_valid), fw = __ballot(is_fused);
    if (lane == 0) { s_v[wave] = (uint32_t)__popcll(vw); s_n[wave] = (uint32_t)__popcll(vw & ~fw); s_f[wave] = (uint32_t)__popcll(fw); }
    if (SHARD) {
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
        if (nn) atomicAdd(&nf_sub[(wg & 63u) * SUB_STRIDE], nn);
        if (nf) atomicAdd(&nf_sub[SUB_SET + (wg & 63u) * SUB_STRIDE], nf);
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

template <bool SHARD>
__global__ __launch_bounds__(PIX_BLOCK) void k_associate_direct(AssocArgs a, ShardArgs sh)
{
    associate_direct_block<SHARD>(a, sh, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// The previous frame's association and this frame's image preparation in ONE launch (plain asynchronous streams,
// DESIGN.md 4 "Three launches per frame"): the two are independent -- the association of frame f-1 reads that frame's
// planes and key map, the preparation of frame f writes the other set -- so the host holds the association back until
// the next frame's images arrive and saves a launch, and the small k_prep runs in the shadow of the association.
// Block ranges: the 32x32-pixel image tiles, the frame's tile flags (tile_prep_block with the "may still change" rule),
// the association blocks.
// ---------------------------------------------------------------------------------------------
template <bool CHAIN>
__global__ __launch_bounds__(PIX_BLOCK) void k_assoc_prep(AssocArgs a, PrepArgs p, FrameParams fp_new, TilePrep tp, uint32_t n_assoc,
                                                          uint32_t n_img, ChainArgs ch, FixArgs fx, uint32_t n_fix /* 0, or 1 + fx.n_crew */,
                                                          unsigned long long *__restrict__ trace = nullptr /* SM_PASS_TRACE */)
{
    struct Stamp {              // entry / exit time of every workgroup (thread 0), for tools/pass_trace.py
        unsigned long long *t; unsigned long long t0;
        __device__ Stamp(unsigned long long *tr) : t(tr), t0(tr ? wall_clock64() : 0ull) {}
        __device__ ~Stamp() { if (t && threadIdx.x == 0 && blockIdx.x < 65536u) { t[(size_t)blockIdx.x * 2] = t0; t[(size_t)blockIdx.x * 2 + 1] = wall_clock64(); } }   // (the buffer holds 65 536 records)
    } stamp(trace);
    // Two-launch frame: the publisher and the repair crew of the frame whose association this launch carries come FIRST in
    // dispatch order -- on the rare frames where the others wait for them they are in the chip whatever else is resident.
    if (blockIdx.x < n_fix) { fixup_merged_block(a.M, a.st, a.fp, fx, blockIdx.x); return; }      // workgroup-uniform
    const uint32_t bx = blockIdx.x - n_fix;
    // Dispatch order.  Without the chain: the association first -- all ~1 500 workgroups are in the chip within 0.3 us and their
    // loads are one burst served roughly in dispatch order; the association is the part with two or three DEPENDENT round trips,
    // the image tiles have one.  With the chain (CHAIN): the chain tiles first -- they are the long workgroups of the launch (169
    // taps per pixel), the association fills the chip around them.
    __shared__ __align__(16) unsigned char s_chain[CHAIN ? CHAIN_LDS_BYTES : 16];
    if (CHAIN) {
        if (bx < n_img) { prep_chain_block(p, ch, fp_new, bx, s_chain); return; }      // workgroup-uniform
        const uint32_t b = bx - n_img;
        if (b >= n_assoc) { tile_prep_block(fp_new, tp, n_fix + n_img + n_assoc); return; }
        ShardArgs none;
        none.validmask = nullptr; none.ownmask = nullptr; none.gmask = nullptr; none.nwords = 0u; none.owner = 1;
        associate_direct_block<false>(a, none, b);
        return;
    }
    if (bx >= n_assoc) {                                                                  // workgroup-uniform
        const uint32_t b = bx - n_assoc;
        if (b < tp.nfb) { tile_prep_block(fp_new, tp, n_fix + n_assoc); return; }
        prep_image_block<PIX_BLOCK>(p, fp_new, b - tp.nfb);
        return;
    }
    ShardArgs none;
    none.validmask = nullptr; none.ownmask = nullptr; none.gmask = nullptr; none.nwords = 0u; none.owner = 1;
    associate_direct_block<false>(a, none, bx);
}

// p11 concatenate (unstable.vert:13-34 + glCopyBufferSubData src/GlobalModel.cpp:627) on the frames that compact, after k_associate:
// the new surfels are (re)computed here and written straight to their final slot, in pixel order.  Every block sums the
// (new, fused) counts of the blocks before it (a few KB, L2-resident) instead of a scan kernel; block 0 publishes the frame totals.
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
                                                           const uint4 *__restrict__ lazy_part /* k_surfel_pass's partials (then compact_part is unused) */,
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
