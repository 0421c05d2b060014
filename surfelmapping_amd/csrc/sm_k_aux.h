// sm_k_aux.h -- off the hot path: tile-bounds rebuild, AoS export / import, index-map textures, the novel-view renderer.
// Part of sm_kernels.h (included there, in order, inside namespace sm); shader citations: /root/reference/src/Shaders/<file>:<line>.
#pragma once

// first surfel (position in the compacted model) created after time stamp t0: the model is kept in creation order, so the
// surfels a rig rank has not yet contributed to the single GlobalModel are the suffix from there on (sm_rig_consolidate_step)
__global__ __launch_bounds__(256) void k_first_newer(Model M, const DevState *__restrict__ st, float t0, uint32_t *__restrict__ out)
{
    const uint32_t N = st->count;
    const float *__restrict__ it = M.s[st->cur].init_time;
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < N; k += gridDim.x * 256u)
        if (it[k] > t0) { best = k; break; }               // (k ascending per thread: its first hit is its smallest)
    best = 0xFFFFFFFFu - wave_max_u32(0xFFFFFFFFu - best);
    if ((threadIdx.x & 63) == 0 && best != 0xFFFFFFFFu) atomicMin(out, best);
}

// SM_CHECK_ALIVE=1 (diagnostic): the invariant every compaction relies on -- per tile, occupied slots - dead count == live bits --
// checked after a stage; out[0] counts the tiles that violate it, out[1..4] describe the first one seen
__global__ __launch_bounds__(256) void k_check_alive(const DevState *__restrict__ st, const uint64_t *__restrict__ alive,
                                                     const uint32_t *__restrict__ tile_dead, uint32_t *__restrict__ out, uint32_t stage)
{
    const uint32_t N = st->count;
    const uint32_t ntiles = (N + TILE - 1) / TILE;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < ntiles; t += gridDim.x * 256u) {
        uint32_t live = 0;
        for (int w = 0; w < TILE_WORDS; ++w) {
            const uint64_t base = ((uint64_t)t * TILE_WORDS + w) * 64u;
            if (base >= N) break;
            const uint64_t rem = (uint64_t)N - base;
            live += (uint32_t)__popcll(alive[(size_t)t * TILE_WORDS + w] & (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)));
        }
        const uint32_t occ = min((uint32_t)TILE, N - t * (uint32_t)TILE);
        if (occ - tile_dead[t] != live && atomicAdd(&out[0], 1u) == 0u) { out[1] = stage; out[2] = t; out[3] = live; out[4] = occ - tile_dead[t]; out[5] = N; }
    }
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
