// sm_api.hip -- C-ABI (include/sm_c_api.h) of the gfx950 surfel-fusion core: context, buffers,
// frame sequencing (SurfelMapping::processFrame, /root/reference/src/SurfelMapping.cpp:115-251)
// and launches of the kernels in sm_kernels.h.  No CPU fallback: without a HIP device
// sm_create() fails with SM_E_NO_DEVICE.
#include "../../include/sm_c_api.h"
#include "sm_kernels.h"

#include <hip/hip_runtime.h>

#include <dirent.h>
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>      // types and enums only: the library is bound at run time (sm_shard_rccl_*), never linked

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace sm;

namespace {

thread_local std::string g_err;

void set_err(const char *what, hipError_t e, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err = buf;
}

#define HIPCK(expr)                                              \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) {                                  \
            set_err(#expr, e_, __FILE__, __LINE__);              \
            return SM_E_HIP;                                     \
        }                                                        \
    } while (0)

constexpr int EV_RING = 256;
constexpr int N_EV = 9;           // start, prep, conflict, scan_cull, compact, associate, scan_new, append, + calibration
constexpr int MAX_GRID = 2048;   // 256 CUs x 8 workgroups
constexpr int COMPACT_GRID = 1024;  // k_compact: 256 CUs x 4 workgroups, must be fully co-resident (in-place hand-off)

// general 4x4 inverse, column-major, cofactor expansion, inv = adj * (1/det), fp32
// (the role of Eigen::Matrix4f::inverse() at src/GlobalModel.cpp:419, src/IndexMap.cpp:157)
void invert4(const float *m, float *out)
{
    float a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
           m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
           m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
           m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
            m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
           m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
           m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
           m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
            m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
           m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
           m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
            m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
            m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
           m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
           m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
            m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
            m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    const float rdet = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = a[i] * rdet;
}


// column-major 4x4 product, c_ij = ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j) + a_i3 b_3j
void mul4(const float *a, const float *b, float *out)
{
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r[j * 4 + i] = ((a[i] * b[j * 4] + a[4 + i] * b[j * 4 + 1]) + a[8 + i] * b[j * 4 + 2]) + a[12 + i] * b[j * 4 + 3];
    memcpy(out, r, sizeof r);
}

// exp of DESIGN.md "Arithmetic": k = rint(x*log2e); r = (x - k*ln2hi) - k*ln2lo; degree-6 Horner; ldexp
float exp_spec(float x)
{
    const float LOG2E = 1.44269502162933349609375f;
    const float LN2HI = 0.693145751953125f, LN2LO = 1.428606765330187045037746429443359375e-06f;
    const float k = rintf(x * LOG2E);
    const float r = (x - k * LN2HI) - k * LN2LO;
    float p = 1.0f / 720.0f;
    p = 1.0f / 120.0f + r * p;
    p = 1.0f / 24.0f + r * p;
    p = 1.0f / 6.0f + r * p;
    p = 0.5f + r * p;
    p = 1.0f + r * p;
    p = 1.0f + r * p;
    return ldexpf(p, (int)k);
}

}  // namespace

// Two HIP runtimes in one process.  PyTorch's ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7) and
// request it by the name "libamdhip64.so"; this library requests "libamdhip64.so.7".  The dynamic loader matches a
// request against the names / sonames of what is loaded already: torch first -> its copy satisfies our request (one
// runtime, fine); this library first -> ROCm's copy is loaded, torch's later request for "libamdhip64.so" does not match its
// soname, so torch/lib/libamdhip64.so is loaded TOO, and torch / its RCCL then run on another runtime than the one that
// owns this library's device memory (seen as a process exit inside RCCL: gpurun_out/pytest_gpu2.log, round 1).
// surfelmapping_amd.capi.load() avoids it by pre-loading torch's copy when torch is installed; any other host gets a
// refusal with the two paths instead of undefined behaviour.
namespace {
int collect_hip_runtime(struct dl_phdr_info *info, size_t, void *data)
{
    auto *v = static_cast<std::vector<std::string> *>(data);
    if (info->dlpi_name && std::strstr(info->dlpi_name, "libamdhip64")) v->push_back(info->dlpi_name);
    return 0;
}
// true (and g_err set) if more than one libamdhip64 is mapped into this process
bool hip_runtime_conflict(const char *where)
{
    std::vector<std::string> libs;
    dl_iterate_phdr(collect_hip_runtime, &libs);
    std::sort(libs.begin(), libs.end());
    libs.erase(std::unique(libs.begin(), libs.end()), libs.end());
    if (libs.size() <= 1) return false;
    g_err = std::string(where) + ": two HIP runtimes are loaded in this process (" + libs[0] + ", " + libs[1] +
            "); device memory of one is not valid in the other.  Load the other user's runtime first (Python: import torch, or "
            "surfelmapping_amd.capi, before anything that loads ROCm's libamdhip64; C++: link RCCL and this library against the same ROCm)";
    return true;
}
}  // namespace

// Contexts of one process that share a GPU: the in-place compaction kernel waits on tile hand-off flags and, in its
// default form, needs its whole grid resident -- two of them running at the same time can starve each other
// (SM_E_STALL).  As soon as a second context exists on a device, compactions there use the ticket-ordered form of the
// kernel, which makes no residency assumption (SM_COMPACT_TICKETS=1 forces it, e.g. when several PROCESSES share a GPU;
// =0 keeps the round-robin form for a process whose contexts never run at the same time).
namespace {
constexpr int MAX_DEV = 64;
std::mutex g_compact_mu;
int g_ctx_on_dev[MAX_DEV] = {};

// Other PROCESSES on the same GPU are invisible to the counter above.  The KFD driver lists every process with its
// queues under /sys/class/kfd/kfd/proc/<pid>/queues/<n>/gpuid (host pids: inside a container our own pid does not match,
// so processes are only counted).  Device -> gpuid goes through the PCI address (topology/nodes/<n>/properties location_id).
// Returns the number of processes that hold queues on this device's GPU (>= 1 once this process has a context there),
// or -1 if the tables cannot be read.
int kfd_processes_on_gpu(int dev)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, dev) != hipSuccess) return -1;
    unsigned dom = 0, b = 0, d = 0, f = 0;
    if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) return -1;
    const unsigned long want_loc = ((unsigned long)b << 8) | ((unsigned long)d << 3) | f;
    unsigned long gpuid = 0;
    bool found = false;
    for (int n = 0; n < 64 && !found; ++n) {
        char path[128];
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", n);
        FILE *fp = fopen(path, "r");
        if (!fp) { if (n > 16) break; continue; }
        char key[64]; unsigned long val, loc = ~0ul, domain = 0;
        while (fscanf(fp, "%63s %lu", key, &val) == 2) {
            if (!strcmp(key, "location_id")) loc = val;
            else if (!strcmp(key, "domain")) domain = val;
        }
        fclose(fp);
        if (loc == want_loc && domain == dom) {
            snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/gpu_id", n);
            FILE *fg = fopen(path, "r");
            if (fg) { found = fscanf(fg, "%lu", &gpuid) == 1 && gpuid != 0; fclose(fg); }
        }
    }
    if (!found) return -1;
    DIR *pd = opendir("/sys/class/kfd/kfd/proc");
    if (!pd) return -1;
    int procs = 0;
    while (struct dirent *pe = readdir(pd)) {
        if (pe->d_name[0] < '0' || pe->d_name[0] > '9') continue;
        char qdir[256];
        snprintf(qdir, sizeof qdir, "/sys/class/kfd/kfd/proc/%s/queues", pe->d_name);
        DIR *qd = opendir(qdir);
        if (!qd) continue;
        bool here = false;
        while (struct dirent *qe = readdir(qd)) {
            if (qe->d_name[0] < '0' || qe->d_name[0] > '9') continue;
            char gp[400];
            snprintf(gp, sizeof gp, "%s/%s/gpuid", qdir, qe->d_name);
            FILE *fg = fopen(gp, "r");
            unsigned long g = 0;
            if (fg) { if (fscanf(fg, "%lu", &g) == 1 && g == gpuid) here = true; fclose(fg); }
            if (here) break;
        }
        closedir(qd);
        if (here) ++procs;
    }
    closedir(pd);
    return procs;
}

// cached per device, refreshed at most once per second (a few sysfs reads: ~0.2 ms)
bool gpu_shared_with_other_process(int dev)
{
    static std::mutex mu;
    static std::chrono::steady_clock::time_point last[MAX_DEV];
    static int cached[MAX_DEV];
    static bool valid[MAX_DEV] = {};
    if (dev < 0 || dev >= MAX_DEV) return false;
    std::lock_guard<std::mutex> lk(mu);
    const auto now = std::chrono::steady_clock::now();
    if (!valid[dev] || std::chrono::duration_cast<std::chrono::milliseconds>(now - last[dev]).count() > 1000) {
        cached[dev] = kfd_processes_on_gpu(dev);
        last[dev] = now; valid[dev] = true;
    }
    return cached[dev] > 1;
}

bool compaction_needs_tickets(int dev)
{
    static const char *env = std::getenv("SM_COMPACT_TICKETS");      // "1": always, "0": never (contexts known not to overlap)
    if (env) return env[0] != '0';
    if (dev < 0 || dev >= MAX_DEV) return true;
    {
        std::lock_guard<std::mutex> lk(g_compact_mu);
        if (g_ctx_on_dev[dev] > 1) return true;
    }
    return gpu_shared_with_other_process(dev);
}
}  // namespace

struct sm_ctx {
    sm_config cfg{};
    int W = 0, H = 0, P = 0;
    uint32_t cap = 0;                 // MAX_VERTICES
    hipStream_t stream = nullptr;
    // The four frame planes exist twice (the *_nx pointers are the set of the other frame): a frame's association is held back
    // and runs in the NEXT frame's preparation launch, which writes the other set.  (Rounds 1-2 ran the depth filter chain of
    // frame f+1 on a second stream instead; since round 3 the chain is a stage of the preparation launch itself.)
    int plane_set = 0;                 // which plane set the current frame uses
    Model M{};
    DevState *d_state = nullptr;
    DevState *h_state = nullptr;      // pinned mirror
    // column-major frame images
    float *d_depthT = nullptr, *d_filteredT = nullptr, *d_lastT = nullptr;
    uint32_t *d_rgbsT = nullptr;
    uint2 *d_dcT = nullptr;            // (depth bits, rgbs) of the frame the conflict test sees
    float *d_depthT_nx = nullptr; uint32_t *d_rgbsT_nx = nullptr; uint64_t *d_keyT_nx = nullptr; uint2 *d_dcT_nx = nullptr;
    uint64_t *d_keyT = nullptr;
    // row-major staging of the caller's inputs
    uint8_t *d_rgb = nullptr, *d_sem = nullptr;
    uint16_t *d_depth_raw = nullptr;
    // sm_process_frame_async: a ring of device input sets filled on a copy stream, so that the H2D copy of frame f+1 runs while
    // frame f computes; images in buffers of sm_host_alloc are copied from in place, others through pinned staging
    static constexpr int IN_RING = 3;
    struct InSlot { uint8_t *rgb = nullptr, *sem = nullptr; uint16_t *depth = nullptr; unsigned char *h_stage = nullptr;
                    hipEvent_t ev_in = nullptr, ev_free = nullptr; bool used = false; };
    InSlot in[IN_RING];
    hipStream_t stream_in = nullptr;   // ONE copy stream.  (Two -- colour on one engine, depth + class on another -- were 80 instead of 89 us per frame on
                                       // one box of the pool and stalled for 10-16 ms every few dozen frames on others; tools/h2d_probe.hip: per frame, three
                                       // copies on two streams 68 us + stalls, on one stream 87, ONE copy of the whole frame 58 = the PCIe rate.)
    size_t in_off_depth = 0, in_off_sem = 0, in_bytes = 0;   // a frame's images as ONE block: colour | depth | class, 16-byte aligned (sm_host_alloc_frame)
    uint32_t in_next = 0;
    const uint16_t *in_last_depth = nullptr; const uint8_t *in_last_sem = nullptr;     // device copies of the last depth / semantic image given
    int in_depth_slot = -1, in_sem_slot = -1;                                          // ... and the input sets that hold them
    // Pinned host buffers handed out by sm_host_alloc (their ranges in `pinned`): the sources sm_process_frame_async copies from in
    // place.  Caller memory is never registered: hipHostRegister / hipHostUnregister of heap ranges left the runtime treating
    // later, unrelated host arrays at the same addresses as pinned -- a GPU memory fault in whatever copied to or from them next.
    std::vector<std::pair<const unsigned char *, size_t>> pinned;
    std::vector<void *> host_allocs;
    float *d_depth_f32 = nullptr;
    float *d_xs = nullptr, *d_ys = nullptr;
    float h_wtab[169];                 // depth_smooth.frag's 13 x 13 weights (host-computed, handed to the chain stage as kernel arguments)
    // cull scratch
    uint64_t *d_cm = nullptr, *d_dm = nullptr, *d_zm = nullptr;
    uint32_t *d_tile_cnt = nullptr, *d_tile_allow = nullptr, *d_tile_keep = nullptr, *d_tile_flag = nullptr;
    uint32_t *d_group_tot = nullptr, *d_group_base = nullptr;
    uint64_t *d_alive = nullptr;       // 1 bit per slot: 0 = killed since the last physical compaction (free slots are 1)
    uint32_t *d_tile_dead = nullptr;   // dead slots per tile
    size_t alive_words = 0, dead_tiles = 0;
    bool maybe_garbage = false;        // a deferred-compaction cull ran since the last physical compaction
    bool keys_are_slots = false;       // the key map was drawn by a cull that did not compact: its ids are slot numbers
    int culls_since_compact = 0;       // deferred-compaction schedule (host side: it picks the kernels)
    uint32_t frames_enq = 0;           // appends enqueued so far (compared with the tag of *h_stat)
    unsigned long long *h_stat = nullptr, *d_stat = nullptr;   // pinned, device-written: frames<<32 | occupied slots
    uint32_t *d_tb = nullptr;          // per-tile bounds (8 words per tile)
    uint8_t *d_tile_flags = nullptr;   // per-tile skip flags of the current frame
    uint8_t *d_tile_flags_nx = nullptr; uint4 *d_wave_cnt_nx = nullptr; uint2 *d_prep_part_nx = nullptr;   // the other frame's (two-launch frame: its publisher runs next to this frame's flag workgroups)
    uint32_t *d_conf_part = nullptr;   // per-workgroup partial counters (instead of same-address atomics)
    uint2 *d_compact_part = nullptr;
    uint4 *d_lazy_part = nullptr;      // partials of k_surfel_pass (visible, splat-skipped, killed, conflict-skipped)
    bool lazy_part_live = false;       // the next append folds d_lazy_part (not d_compact_part) into the counters
    // one pass over the surfels per frame (k_surfel_pass + k_pass_fixup) on the frames whose cull only marks the dead
    uint4 *d_wave_cnt = nullptr;       // conflicts per quarter tile (one word per wave)
    float *d_undo = nullptr;           // confidence before this frame's decrement, per slot (read only if the conflict cap binds)
    uint2 *d_fix_part = nullptr;       // partials of k_pass_fixup (visible added, resurrected)
    bool fix_part_live = false;        // the next append also folds d_fix_part in (when the cap bound)
    uint32_t n_fix_part = 0;           // worker workgroups of the last k_pass_fixup
    bool ev_one_pass[EV_RING] = {};    // which frames of the event ring ran the one-pass kernels
    bool ev_direct[EV_RING] = {};      // ... and appended directly
    bool ev_merged[EV_RING] = {};      // the frame's preparation launch was k_assoc_prep (it carried the previous frame's association)
    bool ev_deferred[EV_RING] = {};    // the frame's own association was held back (no kernel between its marks 4 and 5)
    // tile skip flags of the frame, evaluated by extra workgroups of the preparation launch
    uint2 *d_prep_part = nullptr;
    uint32_t n_prep_blocks = 0;        // flag workgroups the frame's k_prep ran (0: the pass kernel evaluates the flags itself)
    bool want_list = false;            // set by enqueue_frame before begin_frame launches k_prep
    int fix_grid = 128;
    // direct append (k_associate_direct): candidate counts per association block / per group, group prefixes
    uint32_t *d_blk_cand = nullptr, *d_grp_cand = nullptr;
    uint32_t *d_frame_sub = nullptr;   // 2 x 64 sub-counters: visible, killed (k_surfel_pass)
    uint32_t *d_nf_sub = nullptr, *d_nf_sub_nx = nullptr;   // 2 x 64 each: new, fused (k_associate_direct) of this / the other frame
    uint32_t *nf_last = nullptr;       // the set the last direct association counted into (its statistics may still be pending)
    uint32_t n_grp = 0, cand_group = 16;
    bool pend_finalize = false;        // the last frame's statistics are completed by the next k_pass_fixup or by k_frame_finalize
    int fix_set = 0;                   // k_pass_fixup's partials alternate between two sets (the previous frame's are read one frame later)
    unsigned long long *d_pass_trace = nullptr;   // SM_PASS_TRACE=<file prefix>: per-workgroup time stamps of the last k_surfel_pass launch, dumped by sm_destroy
    int pass_trace_grid = 0;
    unsigned long long *d_ap_trace = nullptr;     // the same for the last k_assoc_prep launch: (entry, exit) per workgroup
    int ap_trace_n[4] = {0, 0, 0, 0};             // its association / tile-flag / image workgroups (dispatch order); fixup workgroups ahead of them
    uint32_t *d_conf_sub = nullptr;    // 2 x 64 conflict sub-counters (one set per frame parity: zeroed by that frame's k_prep)
    int conf_sub_set = 0;
    uint32_t n_conf_part = 0, n_compact_part = 0;
    uint32_t tb_tiles = 0;
    uint32_t cull_epoch = 0;
    int compact_grid = COMPACT_GRID;
    int pass_grid = MAX_GRID;          // workgroups of k_surfel_pass that are resident at once (a larger grid runs its tail as a second, thin wave)
    // association scratch
    uint64_t *d_validmask = nullptr, *d_fusedmask = nullptr;
    uint2 *d_blk_cnt = nullptr;
    // slot-addressed sharding of one stream, in-stream form (sm_shard_stream_*)
    bool ss_on = false;
    bool rig_on = false;               // sm_rig_configure: rank / world / collective are used by sm_rig_consolidate only
    float rig_last_time = -1.0e30f;    // creation time stamp up to which this rank's surfels are in the incremental GlobalModel (sm_rig_consolidate_step)
    int ss_rank = 0, ss_world = 1;
    uint32_t ss_frames = 0;            // fusing frames so far = index of the next segment (its owner: index % world)
    sm_collective_fn ss_coll = nullptr;
    void *ss_user = nullptr;
    void *ss_comm = nullptr;           // ncclComm_t when the built-in RCCL binding is used
    uint64_t *d_galive = nullptr, *d_new_alive = nullptr, *d_gmask = nullptr;
    uint32_t *d_chk = nullptr;         // SM_CHECK_ALIVE=1: result words of k_check_alive
    uint64_t *d_capx = nullptr;        // the conflict-cap exchange of a sharded frame: total | quarter-tile counts | conflict masks (k_shard_cap_pack)
    uint32_t *d_ss_info = nullptr;
    // deferred association (k_assoc_prep): the association of an asynchronous frame is held back until the next frame's images
    // arrive and then shares that frame's k_prep launch (three launches per frame instead of four)
    bool defer_ok = false;             // this context may defer (plain stream, no depth filter chain, no per-kernel timing)
    bool assoc_pending = false;
    AssocArgs assoc_args{};            // the held-back association (its FrameParams and that frame's planes)
    bool merge_assoc = false;          // set by enqueue_frame: the k_prep launch of this call carries assoc_args
    // two-launch frame: the fixup step (publisher, cap repair) of a frame whose association is held back rides on the same
    // launch as that association; the candidate count moved into the pass's launch
    bool two_launch = false;           // this context uses it (defer_ok, SM_TWO_LAUNCH != 0)
    uint32_t est_fr0 = 0, est_slots0 = 0, est_rate = 0xFFFFFFFFu;   // launch_surfel_pass's estimate of the slots per frame (from the pinned statistic)
    bool fix_pending = false;          // the last frame's fixup has not run yet
    FixArgs fix_args{};
    static constexpr uint32_t N_CREW = 32;
    bool ss_settle_pending = false;    // the last sharded frame's k_shard_settle work rides on the next k_prep (or runs stand-alone first)
    ShardSettle ss_settle{};
    int n_pix_blocks = 0;
    uint32_t n_odd_pixels = 0;
    // export staging
    void *d_export = nullptr;
    size_t export_bytes = 0;
    // host frame state (src/SurfelMapping.h:100-103)
    int tick = 0;
    bool ref_set = false;
    bool raw_valid = false;            // a frame that computes the raw feedback cloud has run (every call but the reference frame)
    int raw_tick = 0;                  // its time stamp
    float curr_pose[16], last_pose[16];
    uint32_t count_bound = 0;         // host upper bound of the device-side count (grid sizing)
    bool pending_cull = false;
    uint32_t count_before_cull = 0, offset_before_cull = 0;
    sm_counts counts{};
    std::vector<void *> user_allocs;
    // timing
    hipEvent_t ev[N_EV][EV_RING];     // per-frame timeline: before prep, then after each kernel
    bool ev_compacted[EV_RING] = {};  // which cull kernel the frame of that slot ran
    FrameLog *d_log = nullptr;
    bool ev_ok = false;
    uint64_t ev_frames = 0, ev_read = 0;
};

namespace {

template <typename T>
int dalloc(T **p, size_t n)
{
    HIPCK(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return SM_OK;
}

int alloc_set(SurfelSet &s, size_t cap)
{
    int rc;
    if ((rc = dalloc(&s.pos_conf, cap))) return rc;
    if ((rc = dalloc(&s.norm_rad, cap))) return rc;
    if ((rc = dalloc(&s.color, cap))) return rc;
    if ((rc = dalloc(&s.init_time, cap))) return rc;
    if ((rc = dalloc(&s.time, cap))) return rc;
    return SM_OK;
}

void free_set(SurfelSet &s)
{
    (void)hipFree(s.pos_conf); (void)hipFree(s.norm_rad); (void)hipFree(s.color);
    (void)hipFree(s.init_time); (void)hipFree(s.time);
}

FrameParams make_params(const sm_ctx *s, const float *pose)
{
    FrameParams fp;
    memset(&fp, 0, sizeof fp);
    memcpy(fp.pose, pose, 64);
    invert4(pose, fp.t_inv);
    const sm_config &c = s->cfg;
    fp.fx = c.fx; fp.fy = c.fy; fp.cx = c.cx; fp.cy = c.cy;
    fp.inv_fx = (float)(1.0 / (double)c.fx);
    fp.inv_fy = (float)(1.0 / (double)c.fy);
    fp.cols = (float)c.width; fp.rows = (float)c.height;
    fp.W = c.width; fp.H = c.height; fp.P = s->P;
    fp.min_depth = c.near_clip; fp.max_depth = c.far_clip;
    fp.conflict_thresh = c.fuse_thresh;
    fp.fuse_thresh = c.fuse_thresh;
    fp.stereo_border = c.stereo_border;
    fp.is_clean = 0;
    fp.time = s->tick;
    fp.time_delta = c.time_delta;
    fp.depth_cutoff = c.far_clip;
    fp.conflict_cap = c.conflict_cap ? (uint32_t)s->P : 0xFFFFFFFFu;
    fp.max_vertices = s->cap;
    fp.init_mode = 0;
    fp.inv_fx_fb = 1.0f / c.fx;
    fp.inv_fy_fb = 1.0f / c.fy;
    fp.use_bounds = c.disable_tile_bounds ? 0 : 1;
    fp.compact_now = 1;                     // the per-pass entry points compact at every cull
    fp.maintenance = 0;
    fp.par = s->plane_set;
    return fp;
}

int grid_surfels(const sm_ctx *s)
{
    const uint64_t tiles = ((uint64_t)s->count_bound + TILE - 1) / TILE;
    return (int)std::min<uint64_t>(std::max<uint64_t>(tiles, 1), MAX_GRID);
}

void publish_stat(sm_ctx *s);

int push_state(sm_ctx *s)
{
    HIPCK(hipMemcpyAsync(s->d_state, s->h_state, sizeof(DevState), hipMemcpyHostToDevice, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    publish_stat(s);
    return SM_OK;
}

// a direct-append frame leaves its new / fused totals, the dead-slot total and its log entry to be completed by the next
// frame's k_pass_fixup; everything else that reads them asks for the completion first
int flush_assoc(sm_ctx *s);

// SM_CHECK_ALIVE=1 (diagnostic): check the alive-bits / dead-count invariant after a stage; reported by sm_sync
void check_alive(sm_ctx *s, uint32_t stage)
{
    if (!s->d_chk) return;
    hipLaunchKernelGGL(k_check_alive, dim3(64), dim3(256), 0, s->stream, s->d_state, s->d_alive, s->d_tile_dead, s->d_chk, stage);
}

int finalize_if_pending(sm_ctx *s)
{
    if (flush_assoc(s)) return SM_E_HIP;      // a held-back association comes first: everything below reads its results
    if (s->ss_settle_pending) {          // a sharded frame whose settle step has not run yet: stand-alone, before anything reads its results
        s->ss_settle_pending = false;
        hipLaunchKernelGGL(k_shard_settle, dim3(s->ss_settle.n), dim3(PIX_BLOCK), 0, s->stream, s->ss_settle);
        HIPCK(hipGetLastError());
    }
    if (!s->pend_finalize) return SM_OK;
    s->pend_finalize = false;
    hipLaunchKernelGGL(k_frame_finalize, dim3(1), dim3(256), 0, s->stream, s->d_state, s->nf_last ? s->nf_last : s->d_nf_sub,
                       s->d_fix_part + (size_t)s->fix_set * MAX_GRID, s->n_fix_part, s->d_log);
    HIPCK(hipGetLastError());
    return SM_OK;
}

int pull_state(sm_ctx *s)
{
    int rcf = finalize_if_pending(s);
    if (rcf) return rcf;
    HIPCK(hipMemcpyAsync(s->h_state, s->d_state, sizeof(DevState), hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    const DevState &d = *s->h_state;
    s->counts.count = s->pending_cull ? s->count_before_cull : d.count - d.garbage;   // dead slots are not surfels
    s->counts.offset = d.offset - (d.garbage - d.holes_last);   // (empty slots of fused candidates lie above `offset`)
    s->counts.data_count = d.data_count;
    s->counts.conflict_count = d.conflict_count;
    s->counts.unstable_count = d.unstable_count;
    s->counts.fused_count = d.fused_count;
    s->counts.visible_count = d.visible_count;
    s->counts.tick = s->tick;
    s->count_bound = std::max(d.count, d.cull_n * (s->pending_cull ? 1u : 0u));
    return SM_OK;
}

int take_error(sm_ctx *s)
{
    if (s->h_state->error != 0) {
        const int e = s->h_state->error;
        s->h_state->error = 0;
        HIPCK(hipMemcpyAsync(&s->d_state->error, &s->h_state->error, sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
        HIPCK(hipStreamSynchronize(s->stream));
        g_err = e == SM_E_CAPACITY ? "model capacity (MAX_VERTICES) exceeded; frame's new surfels dropped"
              : e == SM_E_UNSUPPORTED ? "unsupported operation flagged on the device"
              : "device-side error";
        return e;
    }
    return SM_OK;
}

// ---- launches ----

// workgroups of the direct association (k_associate_direct / k_assoc_prep): one per two association blocks (every thread
// takes two consecutive pixels)
static inline uint32_t assoc_wgs(const sm_ctx *s)
{
    return (uint32_t)(s->n_pix_blocks + 1) / 2u;
}

// `chain`: the frame runs the depth pre-processing chain p0a..p0e (preprocess = 1): the launch is k_assoc_prep<., true>, whose
// image workgroups are chain tiles (prep_chain_block) -- with or without a held-back association to carry
int launch_prep(sm_ctx *s, const uint8_t *rgb, const uint16_t *raw, const uint8_t *sem, const float *dm,
                const FrameParams &fp, bool clear_keys, hipStream_t st = nullptr, const ChainArgs *chain = nullptr)
{
    const int tiles = ((s->W + 31) / 32) * ((s->H + 31) / 32);
    // the frame's tile skip flags for the one-pass surfel kernel: a few extra workgroups (128 tiles each per round)
    TilePrep tp;
    memset(&tp, 0, sizeof tp);
    s->n_prep_blocks = 0;
    if (clear_keys && s->want_list) {
        const uint64_t ntl = ((uint64_t)s->count_bound + TILE - 1) / TILE;
        tp.nfb = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((ntl + 1023) / 1024, 1), 64);     // one tile per thread (k_prep: 1 024 threads)
        tp.st = s->d_state; tp.tb = s->d_tb; tp.tile_flags = s->d_tile_flags; tp.wave_cnt = s->d_wave_cnt; tp.prep_part = s->d_prep_part;
        s->n_prep_blocks = tp.nfb;
    }
    if (s->ev_ok) s->ev_merged[s->ev_frames % EV_RING] = s->merge_assoc || chain != nullptr;
    if (s->merge_assoc || chain) {
        // the held-back association of the previous frame (if any) + this frame's tile flags + its image / chain tiles in one launch
        const bool carry = s->merge_assoc;
        s->merge_assoc = false;
        if (carry) s->assoc_pending = false;
        if (chain && s->ss_settle_pending) {          // (a sharded stream's settle step rides on k_prep only: stand-alone here)
            s->ss_settle_pending = false;
            hipLaunchKernelGGL(k_shard_settle, dim3(s->ss_settle.n), dim3(PIX_BLOCK), 0, s->stream, s->ss_settle);
            HIPCK(hipGetLastError());
        }
        if (tp.nfb) {
            const uint64_t ntl = ((uint64_t)s->count_bound + TILE - 1) / TILE;
            tp.nfb = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((ntl + 255) / 256, 1), 128);     // one tile per thread
            if (carry) { tp.grp_cand = s->assoc_args.grp_cand; tp.n_grp = s->assoc_args.n_grp; tp.prev_time = s->assoc_args.fp.time; }
            s->n_prep_blocks = tp.nfb;
        }
        // two-launch frame: the held-back association's frame has not had its fixup step yet -- its publisher and repair crew
        // open this launch, the association and the flag workgroups check for themselves whether they have to wait for them
        FixArgs fx;
        memset(&fx, 0, sizeof fx);
        uint32_t n_fix = 0;
        s->assoc_args.slow_conf_sub = nullptr; s->assoc_args.slow_need = 0u;
        if (carry && s->fix_pending) {
            s->fix_pending = false;
            fx = s->fix_args;
            n_fix = 1u + fx.n_crew;
            s->assoc_args.slow_conf_sub = fx.conf_sub; s->assoc_args.slow_need = n_fix;
            if (tp.nfb) { tp.slow_conf_sub = fx.conf_sub; tp.slow_cap = s->assoc_args.fp.conflict_cap; tp.slow_need = n_fix; tp.slow_par = s->assoc_args.fp.par; }
        }
        PrepArgs pa;
        pa.rgb = rgb; pa.depth_raw = raw; pa.sem = sem; pa.depth_f32 = dm; pa.depthT = s->d_depthT; pa.rgbsT = s->d_rgbsT;
        pa.keyT = clear_keys ? s->d_keyT : nullptr; pa.dcT = s->d_dcT;
        pa.conf_sub = clear_keys ? s->d_conf_sub + SUB_SET * s->conf_sub_set : nullptr;
        const uint32_t n_assoc = carry ? assoc_wgs(s) : 0u;
        ChainArgs ca;
        memset(&ca, 0, sizeof ca);
        uint32_t n_img = (uint32_t)tiles;
        if (chain) { ca = *chain; n_img = (uint32_t)(((s->W + CH_TX - 1) / CH_TX) * ((s->H + CH_TY - 1) / CH_TY)); }
        if (s->d_ap_trace) { s->ap_trace_n[0] = (int)n_assoc; s->ap_trace_n[1] = (int)tp.nfb; s->ap_trace_n[2] = (int)n_img; s->ap_trace_n[3] = (int)n_fix; }    // (chain: dispatched image | association | flags; the fixup workgroups before them)
        const dim3 grid(n_fix + tp.nfb + n_assoc + n_img);
        if (chain) hipLaunchKernelGGL((k_assoc_prep<true>), grid, dim3(PIX_BLOCK), 0, s->stream, s->assoc_args, pa, fp, tp, n_assoc, n_img, ca, fx, n_fix, s->d_ap_trace);
        else hipLaunchKernelGGL((k_assoc_prep<false>), grid, dim3(PIX_BLOCK), 0, s->stream, s->assoc_args, pa, fp, tp, n_assoc, n_img, ca, fx, n_fix, s->d_ap_trace);
        HIPCK(hipGetLastError());
        return SM_OK;
    }
    // the previous frame of a sharded stream is finished by extra workgroups of this launch (on the main stream only)
    ShardSettle ss;
    memset(&ss, 0, sizeof ss);
    if (s->ss_settle_pending && (!st || st == s->stream)) { ss = s->ss_settle; s->ss_settle_pending = false; }
    // a frame's k_prep (clear_keys) also zeroes the conflict sub-counters of that frame (set chosen by begin_frame)
    hipLaunchKernelGGL(k_prep, dim3(tiles + tp.nfb + (ss.n + 3u) / 4u), dim3(1024), 0, st ? st : s->stream, rgb, raw, sem, dm, s->d_depthT, s->d_rgbsT,
                       clear_keys ? s->d_keyT : nullptr, fp, s->d_dcT, clear_keys ? s->d_conf_sub + SUB_SET * s->conf_sub_set : nullptr, tp, ss);
    HIPCK(hipGetLastError());
    return SM_OK;
}

int mark(sm_ctx *s, int which, bool timed)
{
    if (timed && s->ev_ok) HIPCK(hipEventRecord(s->ev[which][s->ev_frames % EV_RING], s->stream));
    return SM_OK;
}

// p2: the conflict test alone (masks, per-tile counts, per-workgroup partial sums); nothing of the model changes
int launch_conflict_test(sm_ctx *s, const FrameParams &fp, bool timed = false)
{
    if (finalize_if_pending(s)) return SM_E_HIP;
    s->n_conf_part = (uint32_t)grid_surfels(s);
    hipLaunchKernelGGL(k_conflict, dim3(s->n_conf_part), dim3(256), 0, s->stream, s->M, s->d_state, fp, s->d_dcT,
                       s->d_cm, s->d_dm, s->d_zm, s->d_tile_cnt, s->d_tb, s->d_tile_flags, s->d_conf_part, s->d_alive,
                       s->d_conf_sub + SUB_SET * s->conf_sub_set);
    HIPCK(hipGetLastError());
    if (mark(s, 2, timed)) return SM_E_HIP;
    return SM_OK;
}

// the scan / finalize step of a cull that does not fold it into the cull kernel: applies the conflict cap of `fp`
int launch_conflict_finalize(sm_ctx *s, const FrameParams &fp, bool timed = false)
{
    if (fp.compact_now) {
        // this cull compacts: the survivor prefixes are needed, scan them with one workgroup per 1024 tiles.
        // A cull that only marks the dead gets its totals from k_conflict's partial sums in the finalize kernel.
        const int ngroups = std::max<int>(1, (int)((((uint64_t)s->count_bound + TILE - 1) / TILE + GROUP - 1) / GROUP));
        hipLaunchKernelGGL(k_scan_cull, dim3(ngroups), dim3(1024), 0, s->stream, s->d_state, s->d_tile_cnt, s->d_tile_allow,
                           s->d_tile_keep, s->d_group_tot, s->d_tile_dead);
        HIPCK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_cull_finalize, dim3(1), dim3(1024), 0, s->stream, s->d_state, fp, s->d_cm, s->d_dm, s->d_zm,
                       s->d_tile_cnt, s->d_tile_allow, s->d_tile_keep, s->d_group_tot, s->d_group_base, s->d_conf_part, s->n_conf_part,
                       s->d_alive, s->d_tile_dead, s->d_stat);
    HIPCK(hipGetLastError());
    if (mark(s, 3, timed)) return SM_E_HIP;
    return SM_OK;
}

int launch_conflict(sm_ctx *s, const FrameParams &fp, bool timed = false)
{
    int rc = launch_conflict_test(s, fp, timed);
    if (rc) return rc;
    return launch_conflict_finalize(s, fp, timed);
}

// conflict test + cull (marks only) + splat in ONE pass over the surfels, then the publisher / cap fixup kernel.
// direct: the frame appends directly (k_associate_direct follows): the pass also counts the candidate pixels, the fixup
// publishes their group prefixes and the new count.
int launch_surfel_pass(sm_ctx *s, const FrameParams &fp, bool timed, bool direct)
{
    if (s->n_prep_blocks == 0) { g_err = "internal: one-pass frame without tile flags from the preparation launch"; return SM_E_ARG; }
    // two-launch frame: the association will be held back, and the fixup step with it (launch_prep carries both); the candidate
    // pixels are counted by extra workgroups of the pass's own launch
    const bool two = s->two_launch && s->defer_ok && timed && direct;
    // Grid: up to 2 048 workgroups while the model is small (most tiles are skipped by their flags; a wide grid spreads the few
    // hundred tiles with work), but no more than are RESIDENT once every workgroup has many tiles with work (>= 4 per
    // workgroup: beyond ~8 M slots) -- the surplus would start when the first ones finish and run a second, thin wave
    // (20 M scattered surfels: 160 us with 2 048 workgroups, 140 with 1 536 = 6 per CU, 152 with 5, 172 with 7)
    // (the regime is picked from an ESTIMATE of the occupied slots -- the pinned statistic plus a frame's worth of candidates for
    //  every append enqueued since -- not from count_bound, which after a hundred unsynchronised frames is the capacity)
    uint64_t slots_est = s->count_bound;
    {
        const unsigned long long v = __atomic_load_n(s->h_stat, __ATOMIC_RELAXED);
        const uint32_t fr = (uint32_t)(v >> 32), slots = (uint32_t)v;
        // growth per frame as the device has reported it (between two reports at least 8 frames apart), at most a frame's candidates
        if (fr < s->est_fr0 || slots < s->est_slots0) { s->est_fr0 = fr; s->est_slots0 = slots; }      // (a compaction, a reset: the rate stands)
        else if (fr >= s->est_fr0 + 8u) {
            s->est_rate = std::min<uint32_t>((slots - s->est_slots0) / (fr - s->est_fr0) + 1u, s->n_odd_pixels);
            s->est_fr0 = fr; s->est_slots0 = slots;
        }
        // (a compaction comes at least every `compact_period` frames: the slots do not grow for longer than that)
        const uint32_t ahead = std::min<uint32_t>(s->frames_enq - fr, (uint32_t)std::max(s->cfg.compact_period, 1));
        if (s->frames_enq >= fr) slots_est = std::min<uint64_t>(slots_est, (uint64_t)slots + (uint64_t)ahead * s->est_rate);
    }
    const uint64_t tiles_b = (slots_est + TILE - 1) / TILE;
    static const int pers_env = std::getenv("SM_PASS_PERSIST_TILES") ? std::atoi(std::getenv("SM_PASS_PERSIST_TILES")) : 0;
    // (8 192 / 16 384 / never on 100 and 200 KITTI frames: 38.7 / 38.4 / 38.5 us per frame -- the two forms are level there, and a
    //  scattered model pays 1.5x for quarter tiles at 20 M surfels: the lower threshold stays)
    const bool persistent = tiles_b > (uint64_t)(pers_env > 0 ? pers_env : 4 * MAX_GRID);
    // Quarter-tile units (k_surfel_pass<4>: four workgroups per tile sequence) while tiles are few and some of them dense; whole
    // tiles once every workgroup owns many (the scattered 20 M-surfel model: ~50 listed slots per tile, batches of 8 tiles)
    static const int split_env = std::getenv("SM_PASS_SPLIT") ? std::atoi(std::getenv("SM_PASS_SPLIT")) : 0;
    const int split = split_env == 1 || split_env == 4 ? split_env : persistent ? 1 : 4;
    int grid = persistent ? std::min(grid_surfels(s), s->pass_grid) : grid_surfels(s);
    if (split == 4) {
        static const int seq_env = std::getenv("SM_PASS_SEQ") ? std::atoi(std::getenv("SM_PASS_SEQ")) : 0;
        // (all of them resident: 2 048 workgroups were 17.5 us where 1 536 are 14.1 -- the last quarter started when the first left)
        const int max_seq = seq_env > 0 ? std::min(seq_env, MAX_GRID / 4) : 3 * MAX_GRID / 16;       // 384 sequences = 1 536 workgroups, six per CU
        const uint64_t tiles_all = ((uint64_t)s->count_bound + TILE - 1) / TILE;
        grid = 4 * (int)std::min<uint64_t>(std::max<uint64_t>(tiles_all, 1), (uint64_t)max_seq);
    }
    // fixup workers: the cap repair strides over the tiles; with direct append they first count the frame's candidate pixels, one group each
    const int fgrid = two ? (int)sm_ctx::N_CREW
                    : direct ? std::max(std::min(grid, s->fix_grid), (int)std::min<uint32_t>(s->n_grp, MAX_GRID)) : std::min(grid, s->fix_grid);
    const uint32_t n_fix_prev = s->n_fix_part;
    const uint2 *fix_prev = s->d_fix_part + (size_t)s->fix_set * MAX_GRID;
    s->fix_set ^= 1;
    uint2 *fix_cur = s->d_fix_part + (size_t)s->fix_set * MAX_GRID;
    s->n_conf_part = (uint32_t)grid;
    s->n_compact_part = (uint32_t)grid;
    s->lazy_part_live = true;
    s->fix_part_live = true;
    s->n_fix_part = (uint32_t)fgrid;
    uint32_t *sub = s->d_conf_sub + SUB_SET * s->conf_sub_set;
    const uint32_t tile_bound = (uint32_t)std::max<uint64_t>(((uint64_t)s->count_bound + TILE - 1) / TILE, 1);
    if (s->d_pass_trace) s->pass_trace_grid = grid;
    CandArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.n_pass = (uint32_t)grid;
    if (two) {
        ca.n_grp = s->n_grp; ca.cg = s->cand_group; ca.n_pix_blocks = s->n_pix_blocks;
        ca.depthT = s->d_depthT; ca.xs = s->d_xs; ca.ys = s->d_ys; ca.blk_cand = s->d_blk_cand; ca.grp_cand = s->d_grp_cand;
    }
    if (split == 4)
        hipLaunchKernelGGL(k_surfel_pass<4>, dim3(grid + (two ? (int)s->n_grp : 0)), dim3(256), 0, s->stream, s->M, s->d_state, fp, s->d_dcT, s->d_cm, s->d_dm /* km */,
                           s->d_wave_cnt, s->d_tb, s->d_tile_flags, s->d_lazy_part, s->d_alive, s->d_tile_dead, sub, s->d_keyT, s->d_undo, tile_bound,
                           s->d_frame_sub, ca, s->d_pass_trace);
    else
        hipLaunchKernelGGL(k_surfel_pass<1>, dim3(grid + (two ? (int)s->n_grp : 0)), dim3(256), 0, s->stream, s->M, s->d_state, fp, s->d_dcT, s->d_cm, s->d_dm /* km */,
                           s->d_wave_cnt, s->d_tb, s->d_tile_flags, s->d_lazy_part, s->d_alive, s->d_tile_dead, sub, s->d_keyT, s->d_undo, tile_bound,
                           s->d_frame_sub, ca, s->d_pass_trace);
    HIPCK(hipGetLastError());
    if (mark(s, 2, timed) || mark(s, 3, timed)) return SM_E_HIP;
    FixArgs x;
    memset(&x, 0, sizeof x);
    DirectArgs &da = x.da;
    da.on = direct ? (two ? 2 : 1) : 0;
    da.blk_cand = s->d_blk_cand; da.grp_cand = s->d_grp_cand; da.n_grp = s->n_grp; da.cg = s->cand_group; da.n_pix_blocks = s->n_pix_blocks;
    da.depthT = s->d_depthT; da.xs = s->d_xs; da.ys = s->d_ys;
    da.frame_sub = s->d_frame_sub;
    // the previous frame's new / fused counters: the other set where the sets alternate (its association may run next to this publisher)
    da.nf_prev = s->defer_ok ? s->d_nf_sub_nx : s->d_nf_sub;
    // (the previous frame's fixup partials: only if it appended directly and nothing has completed its statistics since)
    da.fix_prev = fix_prev; da.n_fix_prev = s->pend_finalize ? n_fix_prev : 0u;
    da.log = s->d_log;
    s->pend_finalize = false;            // the fixup's publisher completes the previous frame's statistics first
    x.cm = s->d_cm; x.km = s->d_dm; x.wave_cnt = s->d_wave_cnt; x.tile_flags = s->d_tile_flags; x.part = s->d_lazy_part; x.n_part = (uint32_t)grid;
    x.fix_part = fix_cur; x.alive = s->d_alive; x.tile_dead = s->d_tile_dead; x.conf_sub = sub; x.keyT = s->d_keyT; x.undo = s->d_undo;
    x.host_stat = s->d_stat; x.prep_part = s->d_prep_part; x.n_prep = s->n_prep_blocks; x.tb = s->d_tb; x.n_crew = (uint32_t)fgrid;
    if (two) {
        s->fix_args = x;
        s->fix_pending = true;
    } else {
        hipLaunchKernelGGL(k_pass_fixup, dim3(fgrid + 1), dim3(256), 0, s->stream, s->M, s->d_state, fp, x);
        HIPCK(hipGetLastError());
    }
    if (mark(s, 4, timed)) return SM_E_HIP;
    check_alive(s, 1u + 16u * (uint32_t)(s->tick & 0xFFFF));
    return SM_OK;
}

void fill_assoc_args(const sm_ctx *s, const FrameParams &fp, AssocArgs &a)
{
    a.M = s->M; a.st = s->d_state; a.fp = fp;
    a.depthT = s->d_depthT; a.rgbsT = s->d_rgbsT; a.keyT = s->d_keyT; a.xs = s->d_xs; a.ys = s->d_ys;
    a.blk_cand = s->d_blk_cand; a.grp_cand = s->d_grp_cand; a.nf = s->d_nf_sub; a.tb = s->d_tb;
    a.alive = s->d_alive; a.tile_dead = s->d_tile_dead; a.n_grp = s->n_grp; a.cg = s->cand_group; a.host_stat = s->d_stat;
    a.slow_conf_sub = nullptr; a.slow_need = 0u;
}

// association + in-place fuse + direct append (the frame's last kernel; its statistics are completed later)
int launch_associate_direct(sm_ctx *s, const FrameParams &fp, bool timed)
{
    ShardArgs sh;
    memset(&sh, 0, sizeof sh);
    AssocArgs a;
    fill_assoc_args(s, fp, a);
    if (s->ev_ok && timed) s->ev_deferred[s->ev_frames % EV_RING] = s->defer_ok;
    s->nf_last = s->d_nf_sub;
    if (s->defer_ok && timed) {
        // asynchronous plain stream: hold the association back; the next frame's k_prep launch carries it (k_assoc_prep),
        // anything else that needs its results launches it first (flush_assoc, reached through finalize_if_pending)
        s->assoc_args = a;
        s->assoc_pending = true;
    } else {
        hipLaunchKernelGGL((k_associate_direct<false>), dim3(assoc_wgs(s)), dim3(PIX_BLOCK), 0, s->stream, a, sh);
        HIPCK(hipGetLastError());
        check_alive(s, 2u + 16u * (uint32_t)(s->tick & 0xFFFF));
    }
    s->lazy_part_live = false;
    s->fix_part_live = false;
    s->pend_finalize = true;
    s->frames_enq++;
    if (mark(s, 5, timed) || mark(s, 6, timed) || mark(s, 7, timed)) return SM_E_HIP;
    return SM_OK;
}

int flush_assoc(sm_ctx *s)
{
    if (!s->assoc_pending) return SM_OK;
    s->assoc_pending = false;
    if (s->fix_pending) {                 // two-launch frame: that frame's fixup step has not run either -- in a launch of its own, first
        s->fix_pending = false;
        hipLaunchKernelGGL(k_pass_fixup, dim3(s->fix_args.n_crew + 1), dim3(256), 0, s->stream, s->M, s->d_state, s->assoc_args.fp, s->fix_args);
        HIPCK(hipGetLastError());
    }
    s->assoc_args.slow_conf_sub = nullptr; s->assoc_args.slow_need = 0u;
    ShardArgs sh;
    memset(&sh, 0, sizeof sh);
    hipLaunchKernelGGL((k_associate_direct<false>), dim3(assoc_wgs(s)), dim3(PIX_BLOCK), 0, s->stream, s->assoc_args, sh);
    HIPCK(hipGetLastError());
    check_alive(s, 3u + 16u * (uint32_t)(s->tick & 0xFFFF));
    return SM_OK;
}

int launch_compact(sm_ctx *s, const FrameParams &fp, bool splat, bool timed)
{
    s->lazy_part_live = false;
    s->fix_part_live = false;
    if (!fp.compact_now) {
        // deferred compaction: the cull only marks the dead -- lean kernel, no co-residency requirement
        if (splat) { g_err = "internal: a frame's cull that only marks the dead is k_surfel_pass"; return SM_E_ARG; }
        const int grid = grid_surfels(s);
        s->n_compact_part = 0u;
        hipLaunchKernelGGL(k_cull_lazy, dim3(grid), dim3(256), 0, s->stream, s->M, s->d_state, fp, s->d_cm, s->d_dm, s->d_zm,
                           s->d_tile_cnt, s->d_tile_allow, s->d_alive, s->d_tile_dead);
        HIPCK(hipGetLastError());
        if (mark(s, 4, timed)) return SM_E_HIP;
        return SM_OK;
    }
    const int grid = std::min(grid_surfels(s), s->compact_grid);
    const uint32_t epoch = ++s->cull_epoch;
    s->n_compact_part = splat ? (uint32_t)grid : 0u;
    FrameParams fpc = fp;
    fpc.compact_tickets = compaction_needs_tickets(s->cfg.device) ? 1 : 0;
    if (splat)
        hipLaunchKernelGGL(k_compact<true>, dim3(grid), dim3(256), 0, s->stream, s->M, s->d_state, fpc, s->d_cm,
                           s->d_dm, s->d_zm, s->d_tile_cnt, s->d_tile_allow, s->d_tile_keep, s->d_keyT, s->d_tile_flag, epoch,
                           s->d_group_base, s->d_tb, s->d_tile_flags, s->d_compact_part, s->d_alive,
                           s->d_tile_dead);
    else
        hipLaunchKernelGGL(k_compact<false>, dim3(grid), dim3(256), 0, s->stream, s->M, s->d_state, fpc, s->d_cm,
                           s->d_dm, s->d_zm, s->d_tile_cnt, s->d_tile_allow, s->d_tile_keep, s->d_keyT, s->d_tile_flag, epoch,
                           s->d_group_base, s->d_tb, s->d_tile_flags, s->d_compact_part, s->d_alive,
                           s->d_tile_dead);
    HIPCK(hipGetLastError());
    if (mark(s, 4, timed)) return SM_E_HIP;
    return SM_OK;
}

// Deferred-compaction schedule.  The HOST decides whether a cull compacts (it must launch the matching kernels, and it
// must do so without waiting for the device): every `compact_period`-th cull, and whenever dead slots could make the
// frame overflow the capacity (then the result would differ from the reference's).  The bound on the occupied slots
// comes from a pinned word the device updates after every cull and append: slots then + one frame's worth of new
// surfels for every append enqueued since.
bool decide_compact(sm_ctx *s)
{
    if (s->cfg.compact_period <= 1) return true;
    // Capacity: a cull that only marks the dead must not be able to make the frame overflow because of them.
    // bound = slots at the last device update + one frame's worth of new surfels for every append enqueued since.
    // When the host has run far ahead of the device the bound is loose; rather than compacting for nothing it then
    // lets the device catch up (the queue still holds every frame in between, so the GPU stays busy).
    // (This is the one place where an "enqueue only" call may wait, and only within one frame's worth of the capacity:
    //  at most SM_CAPACITY_WAIT_US, default 2000 us, then it compacts instead.)
    static const long wait_us = std::getenv("SM_CAPACITY_WAIT_US") ? std::atol(std::getenv("SM_CAPACITY_WAIT_US")) : 2000;
    const auto t_start = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        const unsigned long long v = __atomic_load_n(s->h_stat, __ATOMIC_RELAXED);
        const uint32_t fr = (uint32_t)(v >> 32), slots = (uint32_t)v;
        uint64_t bound = s->count_bound;
        const uint32_t ahead = s->frames_enq >= fr ? s->frames_enq - fr : 0u;
        if (s->frames_enq >= fr) bound = std::min<uint64_t>(bound, (uint64_t)slots + (uint64_t)ahead * s->n_odd_pixels);
        if (bound + s->n_odd_pixels <= s->cap) break;                  // fits even if every candidate pixel is new
        if (ahead <= 1u) return true;                                  // the bound is (nearly) exact: compact
        if ((uint64_t)slots + 2ull * s->n_odd_pixels > s->cap) return true;   // would not fit with the device caught up either
        if ((spins & 63u) == 63u &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_start).count() > wait_us)
            return true;                                               // the device is further behind than we are willing to wait for
        std::this_thread::yield();
    }
    return s->culls_since_compact + 1 >= s->cfg.compact_period;
}

void note_cull(sm_ctx *s, bool compacted)
{
    if (compacted) { s->culls_since_compact = 0; }
    else { s->culls_since_compact++; s->maybe_garbage = true; }
}

// refresh the pinned slot statistic after the host changed the model (device idle)
void publish_stat(sm_ctx *s)
{
    __atomic_store_n(s->h_stat, ((unsigned long long)s->h_state->stat_frames << 32) | (unsigned long long)s->h_state->count, __ATOMIC_RELAXED);
    s->frames_enq = s->h_state->stat_frames;
}

// after a cull that is not followed by the append kernel (which does this itself): restore the alive mask
int launch_post_fill(sm_ctx *s)
{
    hipLaunchKernelGGL(k_post_fill, dim3(1024), dim3(256), 0, s->stream, s->d_state, s->d_alive, s->d_tile_dead);
    HIPCK(hipGetLastError());
    check_alive(s, 4u + 16u * (uint32_t)(s->tick & 0xFFFF));
    return SM_OK;
}

// Physical compaction outside a frame: every entry point that exposes slots as surfel ids (downloads, the per-pass
// API, rendering, sharding, uploads) first squeezes out the slots that deferred culls left dead.  Nothing is killed:
// empty conflict masks, then the regular scan + in-place compaction, with the key map's ids translated on the way.
int ss_compact(sm_ctx *s);

int ensure_compact(sm_ctx *s)
{
    if (finalize_if_pending(s)) return SM_E_HIP;
    if (s->ss_on) {
        // slot-addressed sharding: a rank's arrays always hold the (dead) slots of the other ranks' surfels; the compaction is
        // a collective step, so every rank must be making this same call
        int rc = ss_compact(s);
        if (rc) return rc;
        HIPCK(hipStreamSynchronize(s->stream));
        return SM_OK;
    }
    if (!s->maybe_garbage) return SM_OK;
    if (s->pending_cull) { g_err = "internal: deferred compaction with a pending per-pass cull"; return SM_E_ARG; }
    FrameParams fp = make_params(s, s->curr_pose);
    fp.maintenance = 1;
    fp.compact_now = 1;
    fp.conflict_cap = 0xFFFFFFFFu;
    const uint64_t tiles = ((uint64_t)s->count_bound + TILE - 1) / TILE + 1;
    const size_t words = std::min<size_t>(tiles * TILE_WORDS, s->alive_words);
    HIPCK(hipMemsetAsync(s->d_cm, 0, words * 8, s->stream));
    HIPCK(hipMemsetAsync(s->d_dm, 0, words * 8, s->stream));
    HIPCK(hipMemsetAsync(s->d_zm, 0, words * 8, s->stream));
    HIPCK(hipMemsetAsync(s->d_tile_cnt, 0, std::min<size_t>(tiles, s->dead_tiles) * 12, s->stream));
    const int ngroups = std::max<int>(1, (int)((tiles + GROUP - 1) / GROUP));
    hipLaunchKernelGGL(k_scan_cull, dim3(ngroups), dim3(1024), 0, s->stream, s->d_state, s->d_tile_cnt, s->d_tile_allow,
                       s->d_tile_keep, s->d_group_tot, s->d_tile_dead);
    hipLaunchKernelGGL(k_cull_finalize, dim3(1), dim3(1024), 0, s->stream, s->d_state, fp, s->d_cm, s->d_dm, s->d_zm,
                       s->d_tile_cnt, s->d_tile_allow, s->d_tile_keep, s->d_group_tot, s->d_group_base, s->d_conf_part, 0u,
                       s->d_alive, s->d_tile_dead, s->d_stat);
    if (s->keys_are_slots)     // ids of the index map: slot -> position among the live surfels, as the API hands them out
        hipLaunchKernelGGL(k_remap_keys, dim3((s->P + 255) / 256), dim3(256), 0, s->stream, s->d_state, s->d_keyT, s->P, s->d_alive,
                           s->d_tile_keep, s->d_group_base);
    s->keys_are_slots = false;
    HIPCK(hipGetLastError());
    const uint32_t keep_part = s->n_compact_part;
    int rc = launch_compact(s, fp, false, false);
    s->n_compact_part = keep_part;
    if (rc) return rc;
    if ((rc = launch_post_fill(s))) return rc;
    s->maybe_garbage = false;
    s->culls_since_compact = 0;
    HIPCK(hipStreamSynchronize(s->stream));
    return SM_OK;
}

int launch_associate_only(sm_ctx *s, const FrameParams &fp)
{
    hipLaunchKernelGGL(k_associate, dim3(s->n_pix_blocks), dim3(PIX_BLOCK), 0, s->stream, s->M, s->d_state, fp,
                       s->d_depthT, s->d_rgbsT, s->d_keyT, s->d_xs, s->d_ys, s->d_validmask, s->d_fusedmask, s->d_blk_cnt, s->d_tb);
    HIPCK(hipGetLastError());
    return SM_OK;
}

// association + in-place fuse, then the dense ordered append (frames that compact; the frame after reset(); the per-pass API)
int launch_associate(sm_ctx *s, const FrameParams &fp, bool timed)
{
    int rc = launch_associate_only(s, fp);
    if (rc) return rc;
    if (mark(s, 5, timed) || mark(s, 6, timed)) return SM_E_HIP;
    // the append derives its own prefix from the per-block counts (no scan kernel)
    hipLaunchKernelGGL(k_append_scan, dim3(s->n_pix_blocks), dim3(PIX_BLOCK), 0, s->stream, s->M, s->d_state, fp, s->d_depthT,
                       s->d_rgbsT, s->d_xs, s->d_ys, s->d_validmask, s->d_fusedmask, s->d_blk_cnt, s->d_log, s->d_tb, s->d_compact_part,
                       s->n_compact_part, s->d_alive, s->d_tile_dead, s->d_stat, s->lazy_part_live ? s->d_lazy_part : nullptr,
                       (s->lazy_part_live && s->fix_part_live) ? s->d_fix_part + (size_t)s->fix_set * MAX_GRID : nullptr, s->n_fix_part);
    s->lazy_part_live = false;
    s->fix_part_live = false;
    s->frames_enq++;
    HIPCK(hipGetLastError());
    if (mark(s, 7, timed)) return SM_E_HIP;
    check_alive(s, 5u + 16u * (uint32_t)(s->tick & 0xFFFF));
    return SM_OK;
}

int rebuild_bounds(sm_ctx *s, uint32_t first_surfel, uint32_t count);

// empty the model on the device (reset path; synchronises)
int discard_model(sm_ctx *s)
{
    int rc = pull_state(s);
    if (rc) return rc;
    if (s->h_state->count == 0 && !s->maybe_garbage && s->h_state->conflict_count == 0 && s->h_state->visible_count == 0) return SM_OK;
    if (s->maybe_garbage) {
        HIPCK(hipMemsetAsync(s->d_alive, 0xFF, s->alive_words * 8, s->stream));
        HIPCK(hipMemsetAsync(s->d_tile_dead, 0, s->dead_tiles * 4, s->stream));
        s->maybe_garbage = false;
    }
    s->culls_since_compact = 0;
    DevState &d = *s->h_state;
    d.count = 0; d.offset = 0; d.garbage = 0; d.garbage_prev = 0; d.first_live = 0; d.do_compact = 0;
    d.conflict_count = 0; d.visible_count = 0;      // no conflict pass, no index map in the initialising frame
    if ((rc = push_state(s))) return rc;
    if ((rc = rebuild_bounds(s, 0, 0))) return rc;
    return pull_state(s);
}

void bump_bound(sm_ctx *s)
{
    s->count_bound = (uint32_t)std::min<uint64_t>((uint64_t)s->count_bound + s->n_odd_pixels, s->cap);
}

void end_frame(sm_ctx *s, bool timed = true);

// First half of SurfelMapping::processFrame once the textures are on the device
// (src/SurfelMapping.cpp:130-169): pre-processing and the reference-frame early-out.
// Returns 1 when the fusing passes must follow, 0 when the call ends here, <0 on error.
int begin_frame(sm_ctx *s, const uint8_t *d_rgb, const uint16_t *d_raw, const uint8_t *d_sem, const float *pose,
                FrameParams *fp_out)
{
    if (s->pending_cull) { g_err = "sm_stage_conflict without sm_stage_cull"; return SM_E_ARG; }
    memcpy(s->curr_pose, pose, 64);
    FrameParams fp = make_params(s, pose);
    const bool fusing = s->ref_set && s->tick != 0;
    int rc;
    if ((rc = mark(s, 8, fusing))) return rc;    // back-to-back pair 8 -> 0: the cost of an event record itself
    if ((rc = mark(s, 0, fusing))) return rc;
    // The reference frame and the frame after reset() do not draw the index map: its textures keep what the last
    // predictIndices left (src/SurfelMapping.cpp:142-169), so the key map is neither cleared nor exchanged then.
    const bool will_splat = fusing;
    // frame parity: the conflict sub-counters, the frame planes and the per-frame scratch of the fixup step alternate between two
    // sets, so that the pre-processing of frame f+1 never touches what frame f still reads
    s->plane_set ^= 1;
    s->conf_sub_set = s->plane_set;
    fp.par = s->plane_set;
    if (s->defer_ok) {
        std::swap(s->d_depthT, s->d_depthT_nx); std::swap(s->d_rgbsT, s->d_rgbsT_nx);
        std::swap(s->d_dcT, s->d_dcT_nx);
        if (will_splat) std::swap(s->d_keyT, s->d_keyT_nx);
        // per-frame scratch the previous frame's publisher / repair crew may still read while this frame's flag workgroups and
        // association write theirs (two-launch frame)
        std::swap(s->d_tile_flags, s->d_tile_flags_nx); std::swap(s->d_wave_cnt, s->d_wave_cnt_nx);
        std::swap(s->d_prep_part, s->d_prep_part_nx); std::swap(s->d_nf_sub, s->d_nf_sub_nx);
    }
    // metriciseDepth + filterDepth + removeMovings (src/SurfelMapping.cpp:136-139,156,254-365): with preprocess = 1 the whole
    // chain is one stage of the preparation launch (prep_chain_block); the reference frame stops before removeMovings
    ChainArgs ca;
    memset(&ca, 0, sizeof ca);
    if (s->cfg.preprocess) {
        ca.lastT = s->d_lastT; ca.filteredT = s->d_filteredT; memcpy(ca.w, s->h_wtab, sizeof ca.w);
        ca.border = (int)std::ceil(s->cfg.stereo_border - 0.5f);
        ca.do_movings = s->ref_set ? 1 : 0;
        if (s->ref_set) {                                 // src/SurfelMapping.cpp:345-349
            float linv[16];
            invert4(s->last_pose, linv);
            mul4(linv, s->curr_pose, ca.t_c2l.m);
        }
    }
    if ((rc = launch_prep(s, d_rgb, d_raw, d_sem, nullptr, fp, will_splat, nullptr, s->cfg.preprocess ? &ca : nullptr))) return rc;
    // preprocess == 0: DEPTH_FILTERED and LAST are the metric depth itself (nothing reads them on the
    // hot path); they alias d_depthT in sm_download_depth instead of being copied every frame.
    if (!s->ref_set) {                                    // src/SurfelMapping.cpp:142-154
        if (s->cfg.preprocess) std::swap(s->d_lastT, s->d_filteredT);   // LAST <- DEPTH_FILTERED without a copy
        memcpy(s->last_pose, s->curr_pose, 64);
        s->ref_set = true;
        s->tick++;
        return 0;
    }
    if ((rc = mark(s, 1, fusing))) return rc;
    s->raw_valid = true;                                  // computeFeedbackBuffers (src/SurfelMapping.cpp:164,172): on demand here
    s->raw_tick = s->tick;
    if (s->tick == 0) {
        // after reset(): computeFeedbackBuffers + GlobalModel::initialize + buildModelMap
        // (src/SurfelMapping.cpp:161-169): the raw cloud of this frame becomes the model
        fp.init_mode = 1;
        fp.log_frame = 0;
        // GlobalModel::initialize writes the raw cloud from the first slot of modelVbo on and sets count to the number written
        // (src/GlobalModel.cpp:211-228): a map that was
        // uploaded after reset() is discarded, not extended
        if ((rc = discard_model(s))) return rc;
        s->n_compact_part = 0;                            // no cull / splat ran: nothing to fold into visible_count
        s->lazy_part_live = false;
        if ((rc = launch_associate(s, fp, false))) return rc;
        bump_bound(s);
        end_frame(s, false);
        return 0;
    }
    fp.splat_follows = 1;
    fp.log_frame = 1;
    *fp_out = fp;
    return 1;
}

// tail of processFrame (src/SurfelMapping.cpp:244-248)
void end_frame(sm_ctx *s, bool timed)
{
    if (s->cfg.preprocess) std::swap(s->d_lastT, s->d_filteredT);   // :244 LAST <- DEPTH_FILTERED without a copy
    memcpy(s->last_pose, s->curr_pose, 64);               // :245 (LAST aliases the metric depth when preprocess == 0)
    if (s->ev_ok && timed) s->ev_frames++;
    s->tick++;
}

// SurfelMapping::processFrame body (src/SurfelMapping.cpp:130-251); enqueue only.
int enqueue_frame(sm_ctx *s, const uint8_t *d_rgb, const uint16_t *d_raw, const uint8_t *d_sem, const float *pose)
{
    if (s->ss_on) { g_err = "context is configured for sharding: use the sm_shard_* entry points"; return SM_E_ARG; }
    FrameParams fp;
    // the cull's kind is decided first: a frame whose cull only marks the dead lets k_prep evaluate the tile skip flags for
    // the one-pass surfel kernel
    const bool fusing = s->ref_set && s->tick != 0 && !s->pending_cull;
    const bool compact_now = fusing ? decide_compact(s) : true;
    s->want_list = fusing && !compact_now;
    // a held-back association rides on this frame's k_prep launch if this is again a fusing frame; anything else (the frame
    // after reset, ...) needs its results first
    // (a compacting frame too: its k_prep launch has no tile flags to make; the doubled words of DevState a merged publisher
    //  leaves set are cleared by k_cull_finalize there, by the next pass's launch otherwise)
    s->merge_assoc = s->assoc_pending && fusing && s->defer_ok;
    int rc = SM_OK;
    if (s->assoc_pending && !s->merge_assoc && (rc = flush_assoc(s))) return rc;
    rc = begin_frame(s, d_rgb, d_raw, d_sem, pose, &fp);
    s->want_list = false;
    s->merge_assoc = false;
    if (rc <= 0) return rc;
    fp.compact_now = compact_now ? 1u : 0u;
    note_cull(s, fp.compact_now != 0u);
    s->keys_are_slots = fp.compact_now == 0u;      // this frame's splat writes slot numbers iff nothing moves
    if (s->ev_ok) s->ev_compacted[s->ev_frames % EV_RING] = fp.compact_now != 0u;
    // a cull that only marks the dead is ONE pass over the surfels (k_surfel_pass + k_pass_fixup: conflict test, decrement, cull,
    // splat), and the association appends the new surfels directly (no append kernel)
    const bool one_pass = !fp.compact_now;
    if (s->ev_ok) { s->ev_one_pass[s->ev_frames % EV_RING] = one_pass; s->ev_direct[s->ev_frames % EV_RING] = one_pass; }
    if (one_pass) {
        if ((rc = launch_surfel_pass(s, fp, true, true))) return rc;        // :178-197
        if ((rc = launch_associate_direct(s, fp, true))) return rc;         // :212-239
        bump_bound(s);
        end_frame(s);
        return SM_OK;
    }
    if ((rc = launch_conflict(s, fp, true))) return rc;           // :178-187
    if ((rc = launch_compact(s, fp, true, true))) return rc;      // :189-197 (cull + mirror + index map)
    if ((rc = launch_associate(s, fp, true))) return rc;   // :212-239
    bump_bound(s);
    end_frame(s);
    return SM_OK;
}

int upload_inputs(sm_ctx *s, const uint8_t *rgb, const uint16_t *depth, const uint8_t *sem)
{
    const size_t P = (size_t)s->P;
    hipStream_t st = s->stream;
    if (rgb) HIPCK(hipMemcpyAsync(s->d_rgb, rgb, P * 3, hipMemcpyHostToDevice, st));
    if (depth) HIPCK(hipMemcpyAsync(s->d_depth_raw, depth, P * 2, hipMemcpyHostToDevice, st));
    if (sem) HIPCK(hipMemcpyAsync(s->d_sem, sem, P, hipMemcpyHostToDevice, st));
    return SM_OK;
}

// rebuild the bounds of every tile that holds a surfel with index >= first_surfel (after the model was written
// from outside the frame pipeline); the state on the device must already carry the new count
int rebuild_bounds(sm_ctx *s, uint32_t first_surfel, uint32_t count)
{
    const uint32_t t0 = first_surfel / TILE;
    if (t0 < s->tb_tiles) {
        const uint32_t n = s->tb_tiles - t0;
        hipLaunchKernelGGL(k_tile_bounds_reset, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_tb, t0, n);
        HIPCK(hipGetLastError());
    }
    const uint32_t k0 = t0 * TILE;
    if (count > k0) {
        hipLaunchKernelGGL(k_tile_bounds_build, dim3((count - k0 + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, s->d_tb, k0);
        HIPCK(hipGetLastError());
    }
    HIPCK(hipStreamSynchronize(s->stream));
    return SM_OK;
}

int ensure_export(sm_ctx *s, size_t bytes)
{
    if (bytes <= s->export_bytes) return SM_OK;
    if (s->d_export) (void)hipFree(s->d_export);
    s->d_export = nullptr; s->export_bytes = 0;
    HIPCK(hipMalloc(&s->d_export, bytes));
    s->export_bytes = bytes;
    return SM_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int sm_api_version(void) { return SM_API_VERSION; }

const char *sm_last_error(void) { return g_err.c_str(); }

int sm_default_config(sm_config *c, int width, int height, float fx, float fy, float cx, float cy)
{
    if (!c) return SM_E_ARG;
    memset(c, 0, sizeof *c);
    c->width = width; c->height = height;
    c->fx = fx; c->fy = fy; c->cx = cx; c->cy = cy;
    c->near_clip = 1.0f;
    c->far_clip = 30.0f;
    c->fuse_thresh = 0.0f;
    c->max_sqrt_vertices = 5000;
    c->time_delta = 200;
    c->stereo_border = 80.0f;
    c->preprocess = 1;
    c->conflict_cap = 1;
    c->device = 0;
    c->enable_timing = 0;
    c->compact_period = 24;
    return SM_OK;
}

sm_ctx *sm_create(const sm_config *c)
{
    if (!c || c->width <= 0 || c->height <= 0 || c->max_sqrt_vertices <= 0 || c->compact_period < 0 ||
        (uint64_t)c->width * c->height > (1u << 30) || (uint64_t)c->max_sqrt_vertices * c->max_sqrt_vertices > 0x7FFFFFFFull) {
        g_err = "sm_create: bad config";
        return nullptr;
    }
    if (hip_runtime_conflict("sm_create")) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || c->device >= ndev) {
        g_err = "sm_create: no HIP device visible (this library has no CPU fallback)";
        return nullptr;
    }
    if (hipSetDevice(c->device) != hipSuccess) { g_err = "hipSetDevice failed"; return nullptr; }
    sm_ctx *s = new sm_ctx();
    s->cfg = *c;
    if (c->device >= 0 && c->device < MAX_DEV) { std::lock_guard<std::mutex> lk(g_compact_mu); g_ctx_on_dev[c->device]++; }
    s->W = c->width; s->H = c->height; s->P = c->width * c->height;
    s->in_off_depth = ((size_t)s->P * 3 + 15) & ~(size_t)15;
    s->in_off_sem = s->in_off_depth + (((size_t)s->P * 2 + 15) & ~(size_t)15);
    s->in_bytes = s->in_off_sem + (size_t)s->P;
    s->cap = (uint32_t)c->max_sqrt_vertices * (uint32_t)c->max_sqrt_vertices;
    for (int i = 0; i < 16; ++i) s->curr_pose[i] = s->last_pose[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    const size_t P = (size_t)s->P, cap = s->cap;
    const size_t nwords = (cap + 63) / 64 + TILE_WORDS, ntiles = (cap + TILE - 1) / TILE + 1;
    s->n_pix_blocks = (s->P + PIX_BLOCK - 1) / PIX_BLOCK;
    bool ok = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && alloc_set(s->M.s[0], cap) == SM_OK;      // one SoA set: the compaction is in place
    ok = ok && dalloc(&s->d_state, 1) == SM_OK && dalloc(&s->d_log, FRAME_LOG_LEN) == SM_OK;
    ok = ok && hipHostMalloc((void **)&s->h_state, sizeof(DevState), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&s->h_stat, 8, hipHostMallocMapped) == hipSuccess &&
         hipHostGetDevicePointer((void **)&s->d_stat, s->h_stat, 0) == hipSuccess;
    if (ok) *s->h_stat = 0ull;
    ok = ok && dalloc(&s->d_depthT, P) == SM_OK && dalloc(&s->d_filteredT, P) == SM_OK && dalloc(&s->d_lastT, P) == SM_OK;
    ok = ok && dalloc(&s->d_rgbsT, P) == SM_OK && dalloc(&s->d_keyT, P) == SM_OK && dalloc(&s->d_dcT, P) == SM_OK &&
         hipMemset(s->d_dcT, 0, P * 8) == hipSuccess;
    // deferred association (three launches per frame; with or without the depth filter chain)
    {
        const char *e = std::getenv("SM_DEFER_ASSOC");                    // "0": every frame launches its own association
        s->defer_ok = !(e && e[0] == '0');
    }
    if (s->defer_ok)
        ok = ok && dalloc(&s->d_depthT_nx, P) == SM_OK && dalloc(&s->d_rgbsT_nx, P) == SM_OK && dalloc(&s->d_keyT_nx, P) == SM_OK &&
             dalloc(&s->d_dcT_nx, P) == SM_OK && hipMemset(s->d_depthT_nx, 0, P * 4) == hipSuccess &&
             hipMemset(s->d_rgbsT_nx, 0, P * 4) == hipSuccess && hipMemset(s->d_dcT_nx, 0, P * 8) == hipSuccess;
    ok = ok && dalloc(&s->d_rgb, P * 3) == SM_OK && dalloc(&s->d_sem, P) == SM_OK && dalloc(&s->d_depth_raw, P) == SM_OK;
    ok = ok && dalloc(&s->d_depth_f32, P) == SM_OK;
    ok = ok && dalloc(&s->d_xs, (size_t)s->W * 2) == SM_OK && dalloc(&s->d_ys, (size_t)s->H * 2) == SM_OK;
    ok = ok && dalloc(&s->d_cm, nwords) == SM_OK && dalloc(&s->d_dm, nwords) == SM_OK && dalloc(&s->d_zm, nwords) == SM_OK;
    s->alive_words = nwords; s->dead_tiles = ntiles;
    ok = ok && dalloc(&s->d_alive, nwords) == SM_OK && hipMemset(s->d_alive, 0xFF, nwords * 8) == hipSuccess &&
         dalloc(&s->d_tile_dead, ntiles) == SM_OK && hipMemset(s->d_tile_dead, 0, ntiles * 4) == hipSuccess;
    ok = ok && dalloc(&s->d_tile_cnt, ntiles * 3) == SM_OK && dalloc(&s->d_tile_allow, ntiles) == SM_OK &&
         dalloc(&s->d_tile_keep, ntiles) == SM_OK && dalloc(&s->d_tile_flag, ntiles) == SM_OK &&
         hipMemset(s->d_tile_flag, 0, ntiles * 4) == hipSuccess &&
         dalloc(&s->d_group_tot, (ntiles / GROUP + 2) * 4) == SM_OK && dalloc(&s->d_group_base, ntiles / GROUP + 2) == SM_OK;
    s->tb_tiles = (uint32_t)(ntiles + P / 2 / TILE + 8);
    ok = ok && dalloc(&s->d_conf_part, (size_t)MAX_GRID * 4) == SM_OK && dalloc(&s->d_compact_part, (size_t)MAX_GRID) == SM_OK &&
         dalloc(&s->d_lazy_part, (size_t)MAX_GRID) == SM_OK && dalloc(&s->d_fix_part, (size_t)MAX_GRID * 2 + 2) == SM_OK &&
         dalloc(&s->d_wave_cnt, ntiles) == SM_OK && dalloc(&s->d_undo, cap + TILE) == SM_OK && dalloc(&s->d_conf_sub, (size_t)2 * SUB_SET) == SM_OK &&
         dalloc(&s->d_prep_part, (size_t)256) == SM_OK &&
         hipMemset(s->d_conf_sub, 0, (size_t)2 * SUB_SET * 4) == hipSuccess;
    ok = ok && dalloc(&s->d_tb, (size_t)s->tb_tiles * 8) == SM_OK && dalloc(&s->d_tile_flags, (size_t)s->tb_tiles) == SM_OK &&
         hipMemset(s->d_tile_flags, 0, s->tb_tiles) == hipSuccess;
    ok = ok && dalloc(&s->d_validmask, (P + 63) / 64 + 4) == SM_OK && dalloc(&s->d_fusedmask, (P + 63) / 64 + 4) == SM_OK;
    ok = ok && dalloc(&s->d_blk_cnt, (size_t)s->n_pix_blocks) == SM_OK;
    // candidate groups: small groups make the counting workgroups short (k_pass_fixup 3.5 -> 2.5 us at 1242x375 with 4
    // instead of 16 blocks per group) but every association wave sums all groups before its own: keep ~250-500 groups
    s->cand_group = s->n_pix_blocks <= 2048 ? 4u : s->n_pix_blocks <= 4096 ? 8u : 16u;
    if (const char *e = std::getenv("SM_CAND_GROUP")) { const int v = std::atoi(e); if (v == 4 || v == 8 || v == 16) s->cand_group = (uint32_t)v; }
    s->n_grp = (uint32_t)((s->n_pix_blocks + s->cand_group - 1) / s->cand_group);
    ok = ok && dalloc(&s->d_blk_cand, (size_t)s->n_grp * CAND_GROUP_MAX) == SM_OK && dalloc(&s->d_grp_cand, (size_t)s->n_grp) == SM_OK &&
         dalloc(&s->d_frame_sub, (size_t)6 * SUB_SET) == SM_OK && hipMemset(s->d_frame_sub, 0, (size_t)6 * SUB_SET * 4) == hipSuccess;
    s->d_nf_sub = s->d_frame_sub + 2 * SUB_SET; s->d_nf_sub_nx = s->d_frame_sub + 4 * SUB_SET;
    if (s->defer_ok)
        ok = ok && dalloc(&s->d_tile_flags_nx, (size_t)s->tb_tiles) == SM_OK && hipMemset(s->d_tile_flags_nx, 0, s->tb_tiles) == hipSuccess &&
             dalloc(&s->d_wave_cnt_nx, ntiles) == SM_OK && dalloc(&s->d_prep_part_nx, (size_t)256) == SM_OK;
    {
        const char *e = std::getenv("SM_TWO_LAUNCH");                     // "0": the fixup step keeps its own launch (three launches per frame)
        s->two_launch = s->defer_ok && !(e && e[0] == '0');
    }
    if (!ok) { if (g_err.empty()) g_err = "sm_create: allocation failed"; sm_destroy(s); return nullptr; }

    // pixel-centre coordinates exactly as data.vert sees them:
    // texcoord = float((i+0.5)/(double)(float)W) (src/GlobalModel.cpp:71-72), x = texcoord*cols (data.vert:62-63)
    std::vector<float> xs(s->W * 2), ys(s->H * 2);   // [0,W): data.vert coordinates; [W,2W): FeedbackBuffer's (src/FeedbackBuffer.cpp:47-53)
    const float cols = (float)s->W, rows = (float)s->H;
    const float px = 1.0f / cols, py = 1.0f / rows;
    bool clamp_ok = true;
    auto tex = [](float t, int n) { float f = std::floor(t * (float)n); if (!(f >= 0.0f)) return 0; if (f > (float)(n - 1)) return n - 1; return (int)f; };
    uint32_t odd = 0;
    for (int i = 0; i < s->W; ++i) {
        const float tc = (float)((i + 0.5) / (double)cols);
        xs[i] = tc * cols;
        xs[s->W + i] = (float)((double)((float)i / cols) + 1.0 / (double)(2.0f * cols)) * cols;
        clamp_ok = clamp_ok && (int)xs[s->W + i] == i;
        clamp_ok = clamp_ok && tex(tc, s->W) == i && tex(tc - px, s->W) == std::max(i - 1, 0) &&
                   tex(tc + px, s->W) == std::min(i + 1, s->W - 1) && (int)xs[i] == i;
    }
    for (int j = 0; j < s->H; ++j) {
        const float tc = (float)((j + 0.5) / (double)rows);
        ys[j] = tc * rows;
        ys[s->H + j] = (float)((double)((float)j / rows) + 1.0 / (double)(2.0f * rows)) * rows;
        clamp_ok = clamp_ok && (int)ys[s->H + j] == j;
        clamp_ok = clamp_ok && tex(tc, s->H) == j && tex(tc - py, s->H) == std::max(j - 1, 0) &&
                   tex(tc + py, s->H) == std::min(j + 1, s->H - 1) && (int)ys[j] == j;
    }
    if (!clamp_ok) {   // the kernels index neighbours as i+-1 / j+-1; refuse sizes where fp32 texcoords disagree
        g_err = "sm_create: texel addressing for this image size is not the simple clamp form";
        sm_destroy(s);
        return nullptr;
    }
    for (int i = 0; i < s->W; ++i) odd += (uint32_t)((s->H + ((i & 1) ? 1 : 0)) / 2);
    s->n_odd_pixels = odd;
    // depth_smooth.frag weights: the host passes 0.5/30^2 as "sigPix" (src/SurfelMapping.cpp:292-309)
    float *wtab = s->h_wtab;
    {
        const float sigma_intensity = 30.0f;
        const float sigPix = 0.5f / (sigma_intensity * sigma_intensity);
        for (int iy = -6; iy <= 6; ++iy)
            for (int ix = -6; ix <= 6; ++ix)
                wtab[(iy + 6) * 13 + (ix + 6)] = exp_spec(-((float)(ix * ix + iy * iy) * sigPix));
    }
    memset(s->h_state, 0, sizeof(DevState));
    ok = hipMemcpy(s->d_xs, xs.data(), xs.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(s->d_ys, ys.data(), ys.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(s->d_state, s->h_state, sizeof(DevState), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(s->d_depthT, 0, P * 4) == hipSuccess && hipMemset(s->d_filteredT, 0, P * 4) == hipSuccess &&
         hipMemset(s->d_lastT, 0, P * 4) == hipSuccess && hipMemset(s->d_rgbsT, 0, P * 4) == hipSuccess &&
         hipMemset(s->d_rgb, 0, P * 3) == hipSuccess && hipMemset(s->d_sem, 0, P) == hipSuccess &&
         hipMemset(s->d_depth_raw, 0, P * 2) == hipSuccess && hipMemset(s->d_depth_f32, 0, P * 4) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_tile_bounds_reset, dim3((s->tb_tiles + 255) / 256), dim3(256), 0, s->stream, s->d_tb, 0u, s->tb_tiles);
        hipLaunchKernelGGL(k_fill_keys, dim3((s->P + 255) / 256), dim3(256), 0, s->stream, s->d_keyT, s->P);
        ok = hipStreamSynchronize(s->stream) == hipSuccess;
    }
    if (!ok) { g_err = "sm_create: device initialisation failed"; sm_destroy(s); return nullptr; }
    {
        // the in-place compaction needs every workgroup of k_compact resident at once
        int cus = 0, per_cu = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
        if (std::getenv("SM_CHECK_ALIVE") && (hipMalloc((void **)&s->d_chk, 32) != hipSuccess || hipMemset(s->d_chk, 0, 32) != hipSuccess)) s->d_chk = nullptr;
        if (std::getenv("SM_PASS_TRACE") && hipMalloc((void **)&s->d_pass_trace, (size_t)MAX_GRID * 64) != hipSuccess) s->d_pass_trace = nullptr;
        if (std::getenv("SM_PASS_TRACE") && hipMalloc((void **)&s->d_ap_trace, (size_t)65536 * 16) != hipSuccess) s->d_ap_trace = nullptr;
        {
            // k_surfel_pass: with more workgroups than the chip holds at once the surplus starts when the first ones are done --
            // on a model where every tile has work (20 M scattered surfels: ~10 tiles per workgroup) that is a second pass at an
            // eighth of the occupancy.  Grid = what is resident; tiles go round-robin.  (SM_PASS_WG_PER_CU overrides.)
            int pc = 0;
            if (cus > 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, k_surfel_pass<1>, 256, 0) == hipSuccess && pc > 0) {
                int want = std::max(1, pc - 1);      // the occupancy API over-reports by one block per CU here (measured; MI355X_MICROARCH.md)
                if (const char *e = std::getenv("SM_PASS_WG_PER_CU")) want = std::max(1, std::atoi(e));
                s->pass_grid = std::max(256, std::min(cus * want, MAX_GRID));
            }
        }
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_compact<true>, 256, 0) == hipSuccess && per_cu > 0) {
            // the occupancy API can over-report by one block per CU (MI355X_MICROARCH.md): stay at <= 4 and below it
            // (SM_COMPACT_WG_PER_CU overrides the margin for experiments)
            // <= 4 per CU: in that range the limit is VGPR/LDS-bound and the API is exact; above it keep a margin
            int want = per_cu <= 4 ? per_cu : 4;
            if (const char *e = std::getenv("SM_COMPACT_WG_PER_CU")) want = std::max(1, std::min(per_cu, std::atoi(e)));
            s->compact_grid = std::max(1, cus * want);
        }
    }
    if (c->enable_timing) {
        s->ev_ok = true;
        for (auto &row : s->ev)
            for (auto &e : row)
                if (hipEventCreate(&e) != hipSuccess) s->ev_ok = false;
    }
    return s;
}

void sm_destroy(sm_ctx *s)
{
    if (!s) return;
    if (s->cfg.device >= 0 && s->cfg.device < MAX_DEV) { std::lock_guard<std::mutex> lk(g_compact_mu); g_ctx_on_dev[s->cfg.device]--; }
    (void)hipSetDevice(s->cfg.device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->ss_comm) (void)sm_shard_rccl_finalize(s);         // a communicator the caller did not finalize
    if (s->stream_in) (void)hipStreamSynchronize(s->stream_in);
    for (auto &sl : s->in) {
        (void)hipFree(sl.rgb);          // (one block: depth and class follow the colour image)
        if (sl.h_stage) (void)hipHostFree(sl.h_stage);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_free) (void)hipEventDestroy(sl.ev_free);
    }
    if (s->stream_in) (void)hipStreamDestroy(s->stream_in);
    for (void *hp : s->host_allocs) (void)hipHostFree(hp);
    if (s->d_pass_trace) {
        // SM_PASS_TRACE=<prefix>: the last k_surfel_pass launch's per-workgroup record (wall_clock64 at entry / first tile /
        // after it / exit, that tile, its compacted entries, XCC | HW_ID, tiles) -> <prefix>.<n>.bin (tools/pass_trace.py)
        static std::atomic<int> n_dump{0};
        std::vector<unsigned long long> h((size_t)std::max(s->pass_trace_grid, 0) * 8);
        if (!h.empty() && hipMemcpy(h.data(), s->d_pass_trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            char path[512];
            snprintf(path, sizeof path, "%s.%d.bin", std::getenv("SM_PASS_TRACE") ? std::getenv("SM_PASS_TRACE") : "pass_trace", n_dump++);
            if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        }
        (void)hipFree(s->d_pass_trace);
    }
    if (s->d_ap_trace) {
        const int n = s->ap_trace_n[0] + s->ap_trace_n[1] + s->ap_trace_n[2] + s->ap_trace_n[3];
        std::vector<unsigned long long> h((size_t)std::max(std::min(n, 65536), 0) * 2 + 3);
        if (n > 0 && hipMemcpy(h.data() + 3, s->d_ap_trace, (h.size() - 3) * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            h[0] = (unsigned long long)s->ap_trace_n[0] | ((unsigned long long)s->ap_trace_n[3] << 32); h[1] = (unsigned long long)s->ap_trace_n[1]; h[2] = (unsigned long long)s->ap_trace_n[2];
            char path[512];
            snprintf(path, sizeof path, "%s.assoc_prep.bin", std::getenv("SM_PASS_TRACE") ? std::getenv("SM_PASS_TRACE") : "pass_trace");
            if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        }
        (void)hipFree(s->d_ap_trace);
    }
    (void)hipFree(s->d_depthT_nx); (void)hipFree(s->d_rgbsT_nx); (void)hipFree(s->d_keyT_nx); (void)hipFree(s->d_dcT_nx);
    free_set(s->M.s[0]); free_set(s->M.s[1]);
    (void)hipFree(s->d_state); (void)hipFree(s->d_log);
    if (s->h_state) (void)hipHostFree(s->h_state);
    if (s->h_stat) (void)hipHostFree(s->h_stat);
    (void)hipFree(s->d_depthT); (void)hipFree(s->d_filteredT); (void)hipFree(s->d_lastT);
    (void)hipFree(s->d_rgbsT); (void)hipFree(s->d_keyT); (void)hipFree(s->d_dcT);
    (void)hipFree(s->d_rgb); (void)hipFree(s->d_sem); (void)hipFree(s->d_depth_raw); (void)hipFree(s->d_depth_f32);
    (void)hipFree(s->d_xs); (void)hipFree(s->d_ys);
    (void)hipFree(s->d_cm); (void)hipFree(s->d_dm); (void)hipFree(s->d_zm); (void)hipFree(s->d_alive); (void)hipFree(s->d_tile_dead);
    (void)hipFree(s->d_tile_flags_nx); (void)hipFree(s->d_wave_cnt_nx); (void)hipFree(s->d_prep_part_nx);
    (void)hipFree(s->d_tile_cnt); (void)hipFree(s->d_tile_allow); (void)hipFree(s->d_tile_keep); (void)hipFree(s->d_tile_flag); (void)hipFree(s->d_group_tot); (void)hipFree(s->d_group_base); (void)hipFree(s->d_tb); (void)hipFree(s->d_tile_flags); (void)hipFree(s->d_conf_part); (void)hipFree(s->d_compact_part); (void)hipFree(s->d_lazy_part); (void)hipFree(s->d_conf_sub); (void)hipFree(s->d_fix_part); (void)hipFree(s->d_blk_cand); (void)hipFree(s->d_grp_cand); (void)hipFree(s->d_frame_sub); (void)hipFree(s->d_wave_cnt); (void)hipFree(s->d_undo); (void)hipFree(s->d_prep_part);
    (void)hipFree(s->d_validmask); (void)hipFree(s->d_fusedmask); (void)hipFree(s->d_blk_cnt);
    (void)hipFree(s->d_chk); (void)hipFree(s->d_galive); (void)hipFree(s->d_new_alive); (void)hipFree(s->d_gmask); (void)hipFree(s->d_ss_info); (void)hipFree(s->d_capx);
    if (s->d_export) (void)hipFree(s->d_export);
    for (void *p : s->user_allocs) (void)hipFree(p);
    if (s->ev_ok)
        for (auto &row : s->ev)
            for (auto &e : row) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int sm_sync(sm_ctx *s)
{
    if (!s) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = pull_state(s);
    if (rc) return rc;
    if (s->d_chk) {
        uint32_t h[8] = {0};
        HIPCK(hipMemcpy(h, s->d_chk, sizeof h, hipMemcpyDeviceToHost));
        if (h[0]) {
            fprintf(stderr, "SM_CHECK_ALIVE: %u tiles violate occupied - dead == live bits; first seen: stage %u (frame tick %u), tile %u, live bits %u, occupied - dead %u, slots %u\n",
                    h[0], h[1] & 15u, h[1] >> 4, h[2], h[3], h[4], h[5]);
            HIPCK(hipMemset(s->d_chk, 0, sizeof h));
        }
    }
    return take_error(s);
}

int sm_process_frame_device(sm_ctx *s, const uint8_t *d_rgb, const uint16_t *d_depth_mm, const uint8_t *d_semantic,
                            const float *pose16)
{
    if (!s || !d_rgb || !pose16) { g_err = "sm_process_frame_device: null argument"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    // a null depth / semantic keeps the previous texture (src/SurfelMapping.cpp:124-128)
    return enqueue_frame(s, d_rgb, d_depth_mm ? d_depth_mm : s->d_depth_raw, d_semantic ? d_semantic : s->d_sem, pose16);
}

int sm_process_frame(sm_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16)
{
    if (!s || !rgb || !pose16) { g_err = "sm_process_frame: null argument (rgb and pose are required)"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = upload_inputs(s, rgb, depth_mm, semantic);
    if (rc) return rc;
    rc = enqueue_frame(s, s->d_rgb, s->d_depth_raw, s->d_sem, pose16);
    if (rc) return rc;
    return sm_sync(s);
}

void *sm_host_alloc(sm_ctx *s, size_t bytes)
{
    if (!s || !bytes) return nullptr;
    if (hipSetDevice(s->cfg.device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { g_err = "sm_host_alloc: hipHostMalloc failed"; return nullptr; }
    s->pinned.emplace_back(static_cast<const unsigned char *>(p), bytes);
    s->host_allocs.push_back(p);
    return p;
}

int sm_host_alloc_frame(sm_ctx *s, uint8_t **rgb, uint16_t **depth_mm, uint8_t **semantic)
{
    if (!s || !rgb || !depth_mm || !semantic) { g_err = "sm_host_alloc_frame: null argument"; return SM_E_ARG; }
    unsigned char *blk = static_cast<unsigned char *>(sm_host_alloc(s, s->in_bytes));
    if (!blk) return SM_E_HIP;
    *rgb = blk; *depth_mm = reinterpret_cast<uint16_t *>(blk + s->in_off_depth); *semantic = blk + s->in_off_sem;
    return SM_OK;
}

int sm_host_free(sm_ctx *s, void *p)
{
    if (!s || !p) return SM_E_ARG;
    auto it = std::find(s->host_allocs.begin(), s->host_allocs.end(), p);
    if (it == s->host_allocs.end()) { g_err = "sm_host_free: not a buffer of sm_host_alloc"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->stream_in) HIPCK(hipStreamSynchronize(s->stream_in));
    s->host_allocs.erase(it);
    for (size_t i = 0; i < s->pinned.size(); ++i)
        if (s->pinned[i].first == p) { s->pinned.erase(s->pinned.begin() + (long)i); break; }
    HIPCK(hipHostFree(p));
    return SM_OK;
}

int sm_process_frame_async(sm_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16)
{
    if (!s || !rgb || !pose16) { g_err = "sm_process_frame_async: null argument (rgb and pose are required)"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    const size_t P = (size_t)s->P;
    if (!s->stream_in) {
        HIPCK(hipStreamCreateWithFlags(&s->stream_in, hipStreamNonBlocking));
        for (auto &sl : s->in) {
            unsigned char *blk = nullptr;
            HIPCK(hipMalloc((void **)&blk, s->in_bytes));
            sl.rgb = blk; sl.depth = reinterpret_cast<uint16_t *>(blk + s->in_off_depth); sl.sem = blk + s->in_off_sem;
            HIPCK(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
            HIPCK(hipEventCreateWithFlags(&sl.ev_free, hipEventDisableTiming));
        }
    }
    const int slot = (int)(s->in_next++ % sm_ctx::IN_RING);
    sm_ctx::InSlot &sl = s->in[slot];
    // the set is free again when the frame that used it last has run its preparation launch (the only reader of the images).
    // The HOST waits for that (three frames back: normally long past) rather than only the copy stream: it bounds the copies and
    // frames in flight.  With the host free to run ahead, hipMemcpyAsync stalled for 7-12 ms every few dozen frames (2 k instead
    // of 16 k frames/s) -- measured with registered and with hipHostMalloc'ed sources alike.
    if (sl.used) HIPCK(hipEventSynchronize(sl.ev_free));
    auto is_pinned = [&](const void *ptr, size_t n) {
        const unsigned char *q = static_cast<const unsigned char *>(ptr);
        for (auto &pr : s->pinned) if (q >= pr.first && q + n <= pr.first + pr.second) return true;
        return false;
    };
    const unsigned char *rgb_b = rgb, *dep_b = reinterpret_cast<const unsigned char *>(depth_mm);
    int rc = SM_OK;
    if (depth_mm && semantic && dep_b == rgb_b + s->in_off_depth && semantic == rgb_b + s->in_off_sem && is_pinned(rgb, s->in_bytes)) {
        // a frame block of sm_host_alloc_frame: ONE copy
        HIPCK(hipMemcpyAsync(sl.rgb, rgb, s->in_bytes, hipMemcpyHostToDevice, s->stream_in));
    } else if (!is_pinned(rgb, P * 3) || (depth_mm && !is_pinned(depth_mm, P * 2)) || (semantic && !is_pinned(semantic, P))) {
        // pageable caller memory: through this set's pinned staging in the frame-block layout (host memcpys; the previous copy out
        // of it -- three frames ago -- must have completed), then ONE copy of what was given
        if (!sl.h_stage) HIPCK(hipHostMalloc((void **)&sl.h_stage, s->in_bytes, hipHostMallocDefault));
        if (sl.used) HIPCK(hipEventSynchronize(sl.ev_in));
        memcpy(sl.h_stage, rgb, P * 3);
        if (depth_mm) memcpy(sl.h_stage + s->in_off_depth, depth_mm, P * 2);
        if (semantic) memcpy(sl.h_stage + s->in_off_sem, semantic, P);
        if (depth_mm && semantic) HIPCK(hipMemcpyAsync(sl.rgb, sl.h_stage, s->in_bytes, hipMemcpyHostToDevice, s->stream_in));
        else {
            HIPCK(hipMemcpyAsync(sl.rgb, sl.h_stage, P * 3, hipMemcpyHostToDevice, s->stream_in));
            if (depth_mm) HIPCK(hipMemcpyAsync(sl.depth, sl.h_stage + s->in_off_depth, P * 2, hipMemcpyHostToDevice, s->stream_in));
            if (semantic) HIPCK(hipMemcpyAsync(sl.sem, sl.h_stage + s->in_off_sem, P, hipMemcpyHostToDevice, s->stream_in));
        }
    } else {
        // separate pinned buffers (sm_host_alloc): copied from in place, one after the other
        HIPCK(hipMemcpyAsync(sl.rgb, rgb, P * 3, hipMemcpyHostToDevice, s->stream_in));
        if (depth_mm) HIPCK(hipMemcpyAsync(sl.depth, depth_mm, P * 2, hipMemcpyHostToDevice, s->stream_in));
        if (semantic) HIPCK(hipMemcpyAsync(sl.sem, semantic, P, hipMemcpyHostToDevice, s->stream_in));
    }
    // a null depth / semantic keeps the previous texture (src/SurfelMapping.cpp:124-128)
    if (depth_mm) { s->in_last_depth = sl.depth; s->in_depth_slot = slot; }
    if (semantic) { s->in_last_sem = sl.sem; s->in_sem_slot = slot; }
    HIPCK(hipEventRecord(sl.ev_in, s->stream_in));
    HIPCK(hipStreamWaitEvent(s->stream, sl.ev_in, 0));
    rc = enqueue_frame(s, sl.rgb, s->in_last_depth ? s->in_last_depth : s->d_depth_raw, s->in_last_sem ? s->in_last_sem : s->d_sem, pose16);
    HIPCK(hipEventRecord(sl.ev_free, s->stream));
    sl.used = true;
    // a frame without its own depth / semantic image read another set's: that set is busy until this frame has prepared too
    for (int o : {s->in_depth_slot, s->in_sem_slot})
        if (o >= 0 && o != slot) { HIPCK(hipEventRecord(s->in[o].ev_free, s->stream)); s->in[o].used = true; }
    return rc;
}

int sm_inputs_consumed(sm_ctx *s)
{
    if (!s) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->stream_in) HIPCK(hipStreamSynchronize(s->stream_in));
    return SM_OK;
}

int sm_clean_points(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16)
{
    return sm_clean_points_ex(s, depth_mm, semantic, pose16, 1);
}

}  // extern "C"

namespace {
// SurfelMapping::cleanPoints with the view already in device memory (sm_clean_points_ex uploads it; sm_rig_consolidate
// takes it from the gathered views of the rig).  `cap_hook`, if given, runs between the conflict test (which changes nothing)
// and the cull: it receives this model's conflict count and returns the number of them that may take effect, in surfel order
// (src/GlobalModel.cpp:54-57: conflictVbo holds W*H records) -- a rig slice learns its share of the union's W*H there -- or a
// negative error code, which abandons the cull with the model untouched.
int clean_points_device(sm_ctx *s, const uint16_t *d_depth_mm, const uint8_t *d_semantic, const float *pose16, int exempt_first,
                        const std::function<long long(uint32_t)> *cap_hook = nullptr)
{
    if (s->pending_cull) { g_err = "sm_stage_conflict without sm_stage_cull"; return SM_E_ARG; }
    // cleanPoints culls without redrawing the index map (src/SurfelMapping.cpp:496-532): the map keeps ids of the model
    // as it was, so they are settled (slot -> position) before this cull changes the positions
    int rc = ensure_compact(s);
    if (rc) return rc;
    memcpy(s->curr_pose, pose16, 64);
    FrameParams fp = make_params(s, pose16);
    if ((rc = launch_prep(s, s->d_rgb, d_depth_mm, d_semantic, nullptr, fp, false))) return rc;   // metriciseDepth only
    fp.max_depth = s->cfg.far_clip - 15.0f;     // src/SurfelMapping.cpp:515
    fp.conflict_thresh = 0.1f;                  // :516
    fp.is_clean = 1;                            // :517
    fp.no_exempt = exempt_first ? 0 : 1;
    fp.compact_now = decide_compact(s) ? 1u : 0u;
    if ((rc = launch_conflict_test(s, fp))) return rc;
    if (cap_hook) {
        std::vector<uint32_t> part((size_t)s->n_conf_part * 4);
        HIPCK(hipMemcpyAsync(part.data(), s->d_conf_part, part.size() * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCK(hipStreamSynchronize(s->stream));
        uint64_t local = 0;
        for (uint32_t b = 0; b < s->n_conf_part; ++b) local += part[(size_t)b * 4 + 1];
        const long long allow = (*cap_hook)((uint32_t)local);
        if (allow < 0) return (int)allow;
        fp.conflict_cap = (uint32_t)std::min<long long>(allow, 0xFFFFFFFFll);
    }
    note_cull(s, fp.compact_now != 0u);
    if ((rc = launch_conflict_finalize(s, fp))) return rc;
    if ((rc = launch_compact(s, fp, false, false))) return rc;
    if ((rc = launch_post_fill(s))) return rc;
    return sm_sync(s);
}
}  // namespace

extern "C" {

int sm_clean_points_ex(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16, int exempt_first)
{
    if (!s || !depth_mm || !semantic || !pose16) { g_err = "sm_clean_points: null argument"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->pending_cull) { g_err = "sm_stage_conflict without sm_stage_cull"; return SM_E_ARG; }
    int rc = finalize_if_pending(s);             // (a held-back association still reads the staging buffers' frame planes)
    if (rc) return rc;
    if ((rc = upload_inputs(s, nullptr, depth_mm, semantic))) return rc;
    return clean_points_device(s, s->d_depth_raw, s->d_sem, pose16, exempt_first);
}

int sm_clean_points_cb(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16, int exempt_first,
                       sm_cap_fn fn, void *user)
{
    if (!s || !depth_mm || !semantic || !pose16) { g_err = "sm_clean_points: null argument"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->pending_cull) { g_err = "sm_stage_conflict without sm_stage_cull"; return SM_E_ARG; }
    int rc = finalize_if_pending(s);
    if (rc) return rc;
    if ((rc = upload_inputs(s, nullptr, depth_mm, semantic))) return rc;
    if (!fn) return clean_points_device(s, s->d_depth_raw, s->d_sem, pose16, exempt_first);
    const std::function<long long(uint32_t)> hook = [&](uint32_t local) { return fn(user, local); };
    return clean_points_device(s, s->d_depth_raw, s->d_sem, pose16, exempt_first, &hook);
}

int sm_reset(sm_ctx *s)
{
    if (!s) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    // the index map survives reset() (the reference only resets the model buffer, src/SurfelMapping.cpp:436-441): bring
    // its ids to the form the API hands out (positions among the live surfels) while the slots can still be translated
    int rc = s->pending_cull ? SM_OK : ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    const uint32_t cur = s->h_state->cur;
    memset(s->h_state, 0, sizeof(DevState));
    s->h_state->cur = cur;
    s->tick = 0;                                 // refFrameIsSet stays (src/SurfelMapping.cpp:436-441)
    s->pending_cull = false;
    if ((rc = push_state(s))) return rc;
    if ((rc = rebuild_bounds(s, 0, 0))) return rc;
    return pull_state(s);
}

int sm_get_counts(sm_ctx *s, sm_counts *out)
{
    if (!s || !out) return SM_E_ARG;
    *out = s->counts;
    out->tick = s->tick;
    return SM_OK;
}

int sm_download_model_aos(sm_ctx *s, float *dst12, uint32_t cap, uint32_t *n)
{
    if (!s || !n) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    const uint32_t cnt = s->pending_cull ? s->count_before_cull : s->h_state->count;
    *n = cnt;
    if (!dst12) return SM_OK;
    if (cap < cnt) { g_err = "sm_download_model_aos: destination too small"; return SM_E_CAPACITY; }
    if (s->pending_cull) { g_err = "sm_download_model_aos between sm_stage_conflict and sm_stage_cull"; return SM_E_ARG; }
    const uint32_t CH = 1u << 22;                // 4 Mi surfels (192 MiB) per staging chunk
    if ((rc = ensure_export(s, (size_t)std::min(cnt, CH) * 48))) return rc;
    for (uint32_t first = 0; first < cnt; first += CH) {
        const uint32_t m = std::min(CH, cnt - first);
        hipLaunchKernelGGL(k_export_aos, dim3((m + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, (float *)s->d_export, first, m);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(dst12 + (size_t)first * 12, s->d_export, (size_t)m * 48, hipMemcpyDeviceToHost, s->stream));
        HIPCK(hipStreamSynchronize(s->stream));
    }
    return SM_OK;
}

int sm_upload_model_aos(sm_ctx *s, const float *src12, uint32_t n)
{
    if (!s || (!src12 && n)) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    if (n > s->cap) { g_err = "sm_upload_model_aos: exceeds MAX_VERTICES"; return SM_E_CAPACITY; }
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    const uint32_t CH = 1u << 22;
    if (n && (rc = ensure_export(s, (size_t)std::min(n, CH) * 48))) return rc;
    for (uint32_t first = 0; first < n; first += CH) {
        const uint32_t m = std::min(CH, n - first);
        HIPCK(hipMemcpyAsync(s->d_export, src12 + (size_t)first * 12, (size_t)m * 48, hipMemcpyHostToDevice, s->stream));
        hipLaunchKernelGGL(k_import_aos, dim3((m + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, (const float *)s->d_export, first, m);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(s->stream));
    }
    s->h_state->count = n;                       // src/GlobalModel.cpp:995
    s->h_state->offset = n;
    s->h_state->garbage = 0; s->h_state->garbage_prev = 0; s->h_state->first_live = 0; s->h_state->do_compact = 0;
    s->pending_cull = false;
    if ((rc = push_state(s))) return rc;
    if ((rc = rebuild_bounds(s, 0, n))) return rc;
    return pull_state(s);
}

int sm_save_map(sm_ctx *s, const char *path, int32_t start_id, int32_t end_id)
{
    if (!s || !path) return SM_E_ARG;
    uint32_t n = 0;
    int rc = sm_download_model_aos(s, nullptr, 0, &n);
    if (rc) return rc;
    std::vector<float> buf((size_t)n * 12);
    if ((rc = sm_download_model_aos(s, buf.data(), n, &n))) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) { g_err = std::string(path) + " is not open!"; return SM_E_ARG; }
    // u32 count | i32 startId | i32 endId | count*12 f32   (src/GlobalModel.cpp:927-932)
    bool ok = fwrite(&n, 4, 1, f) == 1 && fwrite(&start_id, 4, 1, f) == 1 && fwrite(&end_id, 4, 1, f) == 1;
    ok = ok && (n == 0 || fwrite(buf.data(), 48, n, f) == n);
    ok = (fclose(f) == 0) && ok;
    if (!ok) { g_err = std::string(path) + " saved err!!"; return SM_E_ARG; }
    return SM_OK;
}

int sm_load_map(sm_ctx *s, const char *path, int32_t *start_id, int32_t *end_id)
{
    if (!s || !path) return SM_E_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) { g_err = std::string(path) + " is not open!"; return SM_E_ARG; }
    uint32_t n = 0; int32_t a = 0, b = 0;
    bool ok = fread(&n, 4, 1, f) == 1 && fread(&a, 4, 1, f) == 1 && fread(&b, 4, 1, f) == 1;
    std::vector<float> buf;
    if (ok && n <= s->cap) { buf.resize((size_t)n * 12); ok = n == 0 || fread(buf.data(), 48, n, f) == n; }
    fclose(f);
    if (!ok) { g_err = std::string(path) + " read err!!"; return SM_E_ARG; }
    if (n > s->cap) { g_err = "map larger than MAX_VERTICES"; return SM_E_CAPACITY; }
    if (start_id) *start_id = a;
    if (end_id) *end_id = b;
    return sm_upload_model_aos(s, buf.data(), n);
}

int sm_download_index_map(sm_ctx *s, int32_t *id, float *vert_conf4, float *color_time4, float *norm_rad4)
{
    if (!s) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    const size_t P = (size_t)s->P;
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = ensure_export(s, P * 52))) return rc;
    char *base = (char *)s->d_export;
    int32_t *d_id = (int32_t *)(base + P * 48);
    float4 *d_vc = (float4 *)base, *d_ct = (float4 *)(base + P * 16), *d_nr = (float4 *)(base + P * 32);
    FrameParams fp = make_params(s, s->curr_pose);
    hipLaunchKernelGGL(k_export_index, dim3((s->P + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, fp, s->d_keyT, d_id, d_vc, d_ct, d_nr);
    HIPCK(hipGetLastError());
    if (id) HIPCK(hipMemcpyAsync(id, d_id, P * 4, hipMemcpyDeviceToHost, s->stream));
    if (vert_conf4) HIPCK(hipMemcpyAsync(vert_conf4, d_vc, P * 16, hipMemcpyDeviceToHost, s->stream));
    if (color_time4) HIPCK(hipMemcpyAsync(color_time4, d_ct, P * 16, hipMemcpyDeviceToHost, s->stream));
    if (norm_rad4) HIPCK(hipMemcpyAsync(norm_rad4, d_nr, P * 16, hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return SM_OK;
}

int sm_download_raw_cloud(sm_ctx *s, float *dst12, uint32_t cap, uint32_t *n)
{
    if (!s || !n) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    *n = 0;
    if (!s->raw_valid) return SM_OK;                     // nothing computed yet (the reference's buffer is empty before the 2nd frame)
    const size_t P = (size_t)s->P;
    int rc = ensure_export(s, P * 49);
    if (rc) return rc;
    float4 *d_rec = (float4 *)s->d_export;
    uint8_t *d_flag = (uint8_t *)s->d_export + P * 48;
    FrameParams fp = make_params(s, s->curr_pose);
    fp.init_mode = 1;
    fp.time = s->raw_tick;
    hipLaunchKernelGGL(k_raw_cloud, dim3((s->P + 255) / 256), dim3(256), 0, s->stream, fp, s->d_depthT, s->d_rgbsT, s->d_xs, s->d_ys, d_rec, d_flag);
    HIPCK(hipGetLastError());
    std::vector<uint8_t> flag(P);
    HIPCK(hipMemcpyAsync(flag.data(), d_flag, P, hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    uint32_t cnt = 0;
    for (size_t q = 0; q < P; ++q) cnt += flag[q];
    *n = cnt;
    if (!dst12) return SM_OK;
    if (cap < cnt) { g_err = "sm_download_raw_cloud: destination too small"; return SM_E_CAPACITY; }
    std::vector<float> rec(P * 12);
    HIPCK(hipMemcpyAsync(rec.data(), d_rec, P * 48, hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    uint32_t w = 0;
    for (size_t q = 0; q < P; ++q)                       // q = i * H + j: the feedback buffer's vertex order
        if (flag[q]) { memcpy(dst12 + (size_t)w * 12, rec.data() + q * 12, 48); ++w; }
    return SM_OK;
}

int sm_download_depth(sm_ctx *s, int which, float *dst)
{
    if (!s || !dst) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    const bool alias = s->cfg.preprocess == 0;
    // preprocess == 1: after every processFrame LAST == DEPTH_FILTERED (src/SurfelMapping.cpp:244); the two
    // buffers are swapped instead of copied, so both names read d_lastT.
    const float *src = which == SM_TEX_DEPTH_METRIC ? s->d_depthT : which == SM_TEX_DEPTH_FILTERED ? (alias ? s->d_depthT : s->d_lastT)
                     : which == SM_TEX_LAST ? (alias ? s->d_depthT : s->d_lastT) : nullptr;
    if (!src) return SM_E_ARG;
    int rc = ensure_export(s, (size_t)s->P * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(k_untranspose_f32, dim3((s->P + 255) / 256), dim3(256), 0, s->stream, src, (float *)s->d_export, s->W, s->H);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(dst, s->d_export, (size_t)s->P * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return SM_OK;
}

int sm_render_image(sm_ctx *s, const float *view16, int w, int h, float fx, float fy, float cx, float cy, uint8_t *bgr_out,
                    uint8_t *sem_out)
{
    if (!s || !view16 || w <= 0 || h <= 0 || (uint64_t)w * h > (1u << 28) || !bgr_out || !sem_out) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->pending_cull) { g_err = "sm_render_image between sm_stage_conflict and sm_stage_cull"; return SM_E_ARG; }
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    const size_t npix = (size_t)w * h;
    if ((rc = ensure_export(s, npix * 12))) return rc;            // [keys u64 | bgr | sem]
    uint64_t *d_key = (uint64_t *)s->d_export;
    uint8_t *d_bgr = (uint8_t *)s->d_export + npix * 8, *d_sem = d_bgr + npix * 3;
    RenderParams rp;
    invert4(view16, rp.t_inv);
    rp.fx = fx; rp.fy = fy; rp.cx = cx; rp.cy = cy; rp.cols = (float)w; rp.rows = (float)h; rp.w = w; rp.h = h;
    hipLaunchKernelGGL(k_fill_keys, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s->stream, d_key, (int)npix);
    const uint32_t cnt = s->h_state->count;
    if (cnt) hipLaunchKernelGGL(k_render_splat, dim3((cnt + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, rp, d_key);
    hipLaunchKernelGGL(k_render_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s->stream, s->M, s->d_state, d_key,
                       (int)npix, d_bgr, d_sem);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(bgr_out, d_bgr, npix * 3, hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipMemcpyAsync(sem_out, d_sem, npix, hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return SM_OK;
}

// ---- per-pass entry points ----

int sm_set_frame(sm_ctx *s, const uint8_t *rgb, const float *depth_metric, const uint8_t *semantic)
{
    if (!s) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = finalize_if_pending(s);             // (a held-back association still reads the planes this call rewrites)
    if (rc) return rc;
    if ((rc = upload_inputs(s, rgb, nullptr, semantic))) return rc;
    if (depth_metric) HIPCK(hipMemcpyAsync(s->d_depth_f32, depth_metric, (size_t)s->P * 4, hipMemcpyHostToDevice, s->stream));
    FrameParams fp = make_params(s, s->curr_pose);
    // re-pack every plane from the staged inputs; depth only when given (else keep depthT)
    const int tiles = ((s->W + 31) / 32) * ((s->H + 31) / 32);
    TilePrep tp;
    memset(&tp, 0, sizeof tp);
    ShardSettle ss;
    memset(&ss, 0, sizeof ss);
    hipLaunchKernelGGL(k_prep, dim3(tiles), dim3(1024), 0, s->stream, s->d_rgb, (const uint16_t *)nullptr, s->d_sem,
                       depth_metric ? s->d_depth_f32 : nullptr, depth_metric ? s->d_depthT : nullptr, s->d_rgbsT,
                       (uint64_t *)nullptr, fp, s->d_dcT, (uint32_t *)nullptr, tp, ss);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(s->stream));
    return SM_OK;
}

int sm_set_tick(sm_ctx *s, int32_t tick)
{
    if (!s) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = finalize_if_pending(s);
    if (rc) return rc;
    s->tick = tick;
    s->ref_set = true;
    return SM_OK;
}

int sm_stage_conflict(sm_ctx *s, const float *pose16, float min_depth, float max_depth, float fuse_thresh, int is_clean)
{
    if (!s || !pose16) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = s->pending_cull ? SM_OK : ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    if (s->pending_cull) {
        // processConflict may be called again before backMapping (it only rewrites conflictVbo,
        // src/GlobalModel.cpp:449-454): re-arm the not yet applied cull.
        s->h_state->count = s->h_state->cull_n;
        s->h_state->cur = s->h_state->cull_src;
        s->h_state->offset = s->offset_before_cull;
        s->pending_cull = false;
        if ((rc = push_state(s))) return rc;
        if ((rc = pull_state(s))) return rc;
    }
    memcpy(s->curr_pose, pose16, 64);
    FrameParams fp = make_params(s, pose16);
    fp.min_depth = min_depth; fp.max_depth = max_depth; fp.conflict_thresh = fuse_thresh; fp.is_clean = is_clean;
    s->count_before_cull = s->h_state->count;
    s->offset_before_cull = s->h_state->offset;
    if ((rc = launch_conflict(s, fp))) return rc;
    s->pending_cull = true;
    return sm_sync(s);
}

int sm_stage_cull(sm_ctx *s)
{
    if (!s) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    if (!s->pending_cull) { g_err = "sm_stage_cull without sm_stage_conflict"; return SM_E_ARG; }
    FrameParams fp = make_params(s, s->curr_pose);
    int rc = launch_compact(s, fp, false, false);
    if (rc) return rc;
    s->pending_cull = false;
    return sm_sync(s);
}

int sm_stage_splat(sm_ctx *s, const float *pose16, int32_t time, float depth_cutoff, int32_t time_delta)
{
    if (!s || !pose16) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->pending_cull) { g_err = "sm_stage_splat between sm_stage_conflict and sm_stage_cull"; return SM_E_ARG; }
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    memcpy(s->curr_pose, pose16, 64);
    FrameParams fp = make_params(s, pose16);
    fp.time = time; fp.depth_cutoff = depth_cutoff; fp.time_delta = time_delta;
    HIPCK(hipMemsetAsync(&s->d_state->visible_count, 0, 4, s->stream));
    s->n_compact_part = 0;                       // k_splat counts with an atomic; no k_compact partials to fold in
    s->lazy_part_live = false;
    hipLaunchKernelGGL(k_fill_keys, dim3((s->P + 255) / 256), dim3(256), 0, s->stream, s->d_keyT, s->P);
    HIPCK(hipGetLastError());
    const int grid = (int)std::min<uint64_t>(std::max<uint64_t>(((uint64_t)s->h_state->count + 255) / 256, 1), MAX_GRID);
    hipLaunchKernelGGL(k_splat, dim3(grid), dim3(256), 0, s->stream, s->M, s->d_state, fp, s->d_keyT);
    HIPCK(hipGetLastError());
    return sm_sync(s);
}

int sm_stage_associate_fuse(sm_ctx *s, const float *pose16, int32_t time, float depth_min, float depth_max)
{
    if (!s || !pose16) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->pending_cull) { g_err = "sm_stage_associate_fuse between sm_stage_conflict and sm_stage_cull"; return SM_E_ARG; }
    memcpy(s->curr_pose, pose16, 64);
    FrameParams fp = make_params(s, pose16);
    fp.time = time; fp.min_depth = depth_min; fp.max_depth = depth_max;
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = launch_associate(s, fp, false))) return rc;
    return sm_sync(s);
}

int sm_stage_timings(sm_ctx *s, sm_timings *out)
{
    if (!s || !out) return SM_E_ARG;
    memset(out, 0, sizeof *out);
    if (!s->ev_ok) { g_err = "sm_stage_timings: create the context with enable_timing=1"; return SM_E_UNSUPPORTED; }
    HIPCK(hipSetDevice(s->cfg.device));
    HIPCK(hipStreamSynchronize(s->stream));
    uint64_t first = s->ev_read;
    if (s->ev_frames - first > EV_RING) first = s->ev_frames - EV_RING;
    double seg[7] = {0}, run = 0, ovh = 0, cull[2] = {0, 0};
    double own[6] = {0};          // pass, fixup (one-pass frames) | conflict (others) | associate (direct) | associate, append (others)
    double prep2[2] = {0, 0};     // k_prep alone | k_assoc_prep
    double scan_own = 0;          // k_scan_cull + k_cull_finalize on the frames that ran k_conflict
    uint32_t nfr = 0, ncls[2] = {0, 0}, n_op = 0, n_dir = 0, n_merged = 0, n_alone = 0;
    for (uint64_t f = first; f < s->ev_frames; ++f) {
        const int slot = (int)(f % EV_RING);
        float ms = 0;
        bool ok = true;
        double loc[7];
        for (int k = 0; k < 7 && ok; ++k) {
            ok = hipEventElapsedTime(&ms, s->ev[k][slot], s->ev[k + 1][slot]) == hipSuccess;
            loc[k] = ms;
        }
        ok = ok && hipEventElapsedTime(&ms, s->ev[0][slot], s->ev[7][slot]) == hipSuccess;
        float o = 0;
        ok = ok && hipEventElapsedTime(&o, s->ev[8][slot], s->ev[0][slot]) == hipSuccess;
        if (!ok) continue;
        for (int k = 0; k < 7; ++k) seg[k] += loc[k];
        cull[s->ev_compacted[slot] ? 1 : 0] += loc[3];
        ncls[s->ev_compacted[slot] ? 1 : 0]++;
        if (s->ev_one_pass[slot]) { own[0] += loc[1]; own[1] += loc[3]; n_op++; } else { own[2] += loc[1]; scan_own += loc[2]; }
        if (s->ev_direct[slot]) { n_dir++; if (!s->ev_deferred[slot]) { own[3] += loc[4]; n_alone++; } } else { own[4] += loc[4]; own[5] += loc[6]; }
        prep2[s->ev_merged[slot] ? 1 : 0] += loc[0];
        if (s->ev_merged[slot]) n_merged++;
        run += ms;
        ovh += o;
        nfr++;
    }
    s->ev_read = s->ev_frames;
    out->frames = nfr;
    if (nfr) {
        const double inv = 1.0 / nfr;
        // every segment contains one event record; subtract its measured cost (the back-to-back pair) so that the
        // per-kernel figures are launch durations, comparable with rocprofv3's
        const double oh = ovh * inv;
        out->event_overhead = (float)oh;
        for (double &x : seg) x = std::max(0.0, x - oh * nfr);
        out->k_prep = (float)(seg[0] * inv); out->k_conflict = (float)(seg[1] * inv); out->k_scan_cull = (float)(seg[2] * inv);
        out->k_compact = (float)(seg[3] * inv); out->k_associate = (float)(seg[4] * inv); out->k_scan_new = (float)(seg[5] * inv);
        out->k_append = (float)(seg[6] * inv);
        out->k_cull_lazy = ncls[0] ? (float)std::max(0.0, cull[0] / ncls[0] - oh) : 0.0f;
        out->k_compact_own = ncls[1] ? (float)std::max(0.0, cull[1] / ncls[1] - oh) : 0.0f;
        out->frames_compact = ncls[1];
        auto avg = [&](double sum, uint32_t n) { return n ? (float)std::max(0.0, sum / n - oh) : 0.0f; };
        out->k_surfel_pass = avg(own[0], n_op); out->k_pass_fixup = avg(own[1], n_op); out->k_conflict_own = avg(own[2], nfr - n_op);
        out->k_scan_own = avg(scan_own, nfr - n_op);
        out->k_associate_direct = avg(own[3], n_alone); out->k_associate_own = avg(own[4], nfr - n_dir); out->k_append_own = avg(own[5], nfr - n_dir);
        out->frames_one_pass = n_op; out->frames_direct = n_dir;
        out->k_assoc_prep = avg(prep2[1], n_merged); out->k_prep_own = avg(prep2[0], nfr - n_merged);
        out->frames_merged = n_merged; out->frames_assoc_alone = n_alone;
        out->preprocess = out->k_prep;
        out->conflict = out->k_conflict + out->k_scan_cull + out->k_compact;
        out->index_map = 0.0f;
        out->data_association = out->k_associate;
        out->concatenate = out->k_scan_new + out->k_append;
        out->run = (float)(run * inv);
    }
    return SM_OK;
}

int sm_read_frame_log(sm_ctx *s, sm_frame_log *out, uint32_t n, uint32_t *written)
{
    if (!s || !out || !written) return SM_E_ARG;
    static_assert(sizeof(sm_frame_log) == sizeof(FrameLog), "frame log layout");
    static_assert(SM_FRAME_LOG_LEN == FRAME_LOG_LEN, "frame log length");
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = pull_state(s);
    if (rc) return rc;
    const uint32_t total = s->h_state->frames_logged;
    uint32_t m = std::min(std::min(n, total), (uint32_t)FRAME_LOG_LEN);
    std::vector<FrameLog> ring(FRAME_LOG_LEN);
    HIPCK(hipMemcpy(ring.data(), s->d_log, sizeof(FrameLog) * FRAME_LOG_LEN, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < m; ++k) {
        const uint32_t idx = (total - m + k) % FRAME_LOG_LEN;
        memcpy(&out[k], &ring[idx], sizeof(FrameLog));
    }
    *written = m;
    return SM_OK;
}

void *sm_device_alloc(sm_ctx *s, size_t bytes)
{
    if (!s) return nullptr;
    if (hipSetDevice(s->cfg.device) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(bytes, 1)) != hipSuccess) { g_err = "sm_device_alloc: hipMalloc failed"; return nullptr; }
    s->user_allocs.push_back(p);
    return p;
}

int sm_device_free(sm_ctx *s, void *p)
{
    if (!s || !p) return SM_E_ARG;
    auto it = std::find(s->user_allocs.begin(), s->user_allocs.end(), p);
    if (it == s->user_allocs.end()) return SM_E_ARG;
    s->user_allocs.erase(it);
    HIPCK(hipSetDevice(s->cfg.device));
    HIPCK(hipStreamSynchronize(s->stream));
    HIPCK(hipFree(p));
    return SM_OK;
}

int sm_device_upload(sm_ctx *s, void *dst_device, const void *src_host, size_t bytes)
{
    if (!s || !dst_device || !src_host) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    HIPCK(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return SM_OK;
}

int sm_export_model_device(sm_ctx *s, void **d_aos, uint32_t *n)
{
    if (!s || !d_aos || !n) return SM_E_ARG;
    if (hip_runtime_conflict("sm_export_model_device")) return SM_E_HIP;     // the pointer goes to foreign code (RCCL)
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->pending_cull) { g_err = "sm_export_model_device between sm_stage_conflict and sm_stage_cull"; return SM_E_ARG; }
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    const uint32_t cnt = s->h_state->count;
    if ((rc = ensure_export(s, (size_t)std::max(cnt, 1u) * 48))) return rc;
    if (cnt) {
        hipLaunchKernelGGL(k_export_aos, dim3((cnt + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, (float *)s->d_export, 0u, cnt);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(s->stream));
    }
    *d_aos = s->d_export;
    *n = cnt;
    return SM_OK;
}

int sm_append_model_aos_device(sm_ctx *s, const float *d_src12, uint32_t n)
{
    if (!s || (!d_src12 && n)) return SM_E_ARG;
    if (hip_runtime_conflict("sm_append_model_aos_device")) return SM_E_HIP;
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->pending_cull) { g_err = "sm_append_model_aos_device between sm_stage_conflict and sm_stage_cull"; return SM_E_ARG; }
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    const uint32_t cnt = s->h_state->count;
    if ((uint64_t)cnt + n > s->cap) { g_err = "sm_append_model_aos_device: exceeds MAX_VERTICES"; return SM_E_CAPACITY; }
    if (n) {
        hipLaunchKernelGGL(k_import_aos, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, d_src12, cnt, n);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(s->stream));
    }
    s->h_state->count = cnt + n;
    s->h_state->offset = cnt;
    if ((rc = push_state(s))) return rc;
    if ((rc = rebuild_bounds(s, cnt, cnt + n))) return rc;
    return pull_state(s);
}

int sm_device_download(sm_ctx *s, void *dst_host, const void *src_device, size_t bytes)
{
    if (!s || !dst_host || !src_device) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    HIPCK(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCK(hipStreamSynchronize(s->stream));
    return SM_OK;
}

}  // extern "C"

// ---- ONE stream sharded over `world` GPUs, in-stream form: slot-addressed, no host in the frame ----------------------
//
// Every rank addresses surfels by the slot number the single-GPU run uses (k_associate_direct: slot = offset + candidate
// pixels before the pixel -- computable on every rank, the frame is replicated) and stores only the segments it owns
// (segment = one frame's new surfels, owner = frame index % world); everywhere else its alive bits are 0, its tile
// bounds empty, so the one-pass surfel kernel runs unchanged and skips what it does not own.  DevState is replicated:
// all ranks publish the same counts.  A frame is  k_prep | k_surfel_pass | k_pass_fixup | all-reduce(min) key map |
// k_associate_direct<shard> | all-reduce(sum) fused mask + 3 counters | k_shard_settle, all on the context's stream.

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int find_rccl(struct dl_phdr_info *info, size_t, void *data)
{
    auto *v = static_cast<std::string *>(data);
    if (v->empty() && info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) *v = info->dlpi_name;
    return 0;
}

// RCCL is bound at run time: the copy already mapped into the process if there is one (a PyTorch process has its own
// bundled librccl; two RCCLs would work but the one that is there already shares the HIP runtime for certain), else ROCm's.
int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return SM_OK;
    std::string loaded;
    dl_iterate_phdr(find_rccl, &loaded);
    void *h = nullptr;
    // SM_RCCL_LIB: an explicit copy (surfelmapping_amd.capi names PyTorch's bundled one when it pre-loaded PyTorch's HIP
    // runtime: RCCL and the runtime then come from the same build)
    if (const char *e = std::getenv("SM_RCCL_LIB")) { if (e[0]) { h = dlopen(e, RTLD_NOW | RTLD_LOCAL); if (h) loaded = e; } }
    if (!h && !loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { g_err = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return SM_E_UNSUPPORTED; }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(h, "ncclAllGather"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommCount || !g_rccl.CommDestroy) {
        g_err = "RCCL: missing symbols in " + (loaded.empty() ? std::string("librccl.so") : loaded);
        return SM_E_UNSUPPORTED;
    }
    g_rccl.lib = h;
    return SM_OK;
}

int rccl_collective(void *user, const void *send, void *recv, size_t count, int op, void *stream)
{
    sm_ctx *s = static_cast<sm_ctx *>(user);
    const ncclResult_t r = op == SM_COLL_GATHER
        ? g_rccl.AllGather(send, recv, count, ncclUint64, static_cast<ncclComm_t>(s->ss_comm), static_cast<hipStream_t>(stream))
        : g_rccl.AllReduce(send, recv, count, ncclUint64, op == SM_COLL_MIN ? ncclMin : ncclSum,
                           static_cast<ncclComm_t>(s->ss_comm), static_cast<hipStream_t>(stream));
    if (r != ncclSuccess) {
        g_err = std::string(op == SM_COLL_GATHER ? "ncclAllGather: " : "ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed");
        return SM_E_HIP;
    }
    return SM_OK;
}

int ss_collective(sm_ctx *s, const void *send, void *recv, size_t count, int op)
{
    if (!s->ss_coll) {
        if (s->ss_world == 1) {          // one rank and no communicator: reduction and gather are the identity
            if (send != recv) HIPCK(hipMemcpyAsync(recv, send, count * 8, hipMemcpyDeviceToDevice, s->stream));
            return SM_OK;
        }
        g_err = "sharded stream: no collective installed (sm_shard_rccl_init or sm_shard_set_collective)";
        return SM_E_ARG;
    }
    const int rc = s->ss_coll(s->ss_user, send, recv, count, op, s->stream);
    if (rc && g_err.empty()) g_err = "sharded stream: the collective callback failed";
    return rc;
}

// Physical compaction between two frames of a sharded stream (k_shard_* in sm_kernels.h); enqueue only.
int ss_compact(sm_ctx *s)
{
    if (finalize_if_pending(s)) return SM_E_HIP;
    const uint64_t nw = ((uint64_t)s->count_bound + 63) / 64;
    const uint64_t tiles = ((uint64_t)s->count_bound + TILE - 1) / TILE;
    const int g1 = (int)std::min<uint64_t>(std::max<uint64_t>((nw + 255) / 256, 1), 1024);
    const int gt = (int)std::min<uint64_t>(std::max<uint64_t>(tiles, 1), MAX_GRID);
    hipLaunchKernelGGL(k_shard_alive_copy, dim3(g1), dim3(256), 0, s->stream, s->d_state, s->d_alive, s->d_galive, s->d_new_alive, (uint32_t)nw);
    HIPCK(hipGetLastError());
    int rc = ss_collective(s, s->d_galive, s->d_galive, (size_t)nw, SM_COLL_SUM);
    if (rc) return rc;
    hipLaunchKernelGGL(k_shard_tile_popc, dim3(std::max(1, std::min(gt / 16 + 1, 256))), dim3(256), 0, s->stream, s->d_state, s->d_galive, s->d_tile_keep);
    hipLaunchKernelGGL(k_shard_scan, dim3(1), dim3(1024), 0, s->stream, s->d_state, s->d_tile_keep, s->d_tile_allow, s->d_ss_info, s->d_stat);
    hipLaunchKernelGGL(k_shard_stage, dim3(gt), dim3(256), 0, s->stream, s->M, s->d_state, s->d_alive, s->d_galive, s->d_tile_allow, s->d_ss_info,
                       s->d_new_alive);
    hipLaunchKernelGGL(k_shard_unstage, dim3(gt), dim3(256), 0, s->stream, s->M, s->d_state, s->d_ss_info, s->d_new_alive, s->d_alive,
                       s->d_tile_dead, s->d_tb);
    HIPCK(hipGetLastError());
    s->culls_since_compact = 0;
    s->keys_are_slots = false;
    s->lazy_part_live = false;
    s->fix_part_live = false;
    return SM_OK;
}

}  // namespace

extern "C" {

int sm_debug_slow_frames(sm_ctx *s, uint32_t *n)
{
    if (!s || !n) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = pull_state(s);
    if (rc) return rc;
    *n = s->h_state->slow_frames;
    return SM_OK;
}

int sm_gpu_process_count(sm_ctx *s)
{
    if (!s) return SM_E_ARG;
    return kfd_processes_on_gpu(s->cfg.device);
}

int sm_shard_stream_configure(sm_ctx *s, int rank, int world)
{
    if (!s || world < 1 || rank < 0 || rank >= world) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->ss_on) { g_err = "sm_shard_stream_configure: already configured"; return SM_E_ARG; }
    int rc = pull_state(s);
    if (rc) return rc;
    if (s->h_state->count != 0 || s->maybe_garbage || s->tick != 0 || s->ref_set) {
        g_err = "sm_shard_stream_configure: the context must be new (no frame, no model)";
        return SM_E_ARG;
    }
    if ((rc = alloc_set(s->M.s[1], s->cap))) return rc;                        // staging set of the sharded compaction
    if ((rc = dalloc(&s->d_galive, s->alive_words)) || (rc = dalloc(&s->d_new_alive, s->alive_words)) ||
        (rc = dalloc(&s->d_gmask, (size_t)(s->P + 63) / 64 + 4)) || (rc = dalloc(&s->d_ss_info, 4)) ||
        (rc = dalloc(&s->d_capx, (size_t)1 + (size_t)(2 + TILE_WORDS) * s->dead_tiles)))
        return rc;
    HIPCK(hipMemset(s->d_ss_info, 0, 16));
    s->ss_on = true; s->ss_rank = rank; s->ss_world = world; s->ss_frames = 0;
    s->defer_ok = false;                       // the association of a sharded frame sits between two collectives
    return SM_OK;
}

int sm_shard_set_collective(sm_ctx *s, sm_collective_fn fn, void *user)
{
    if (!s || !(s->ss_on || s->rig_on)) { g_err = "sm_shard_set_collective: call sm_shard_stream_configure or sm_rig_configure first"; return SM_E_ARG; }
    s->ss_coll = fn; s->ss_user = user;
    return SM_OK;
}

int sm_shard_rccl_unique_id(void *out128)
{
    if (!out128) return SM_E_ARG;
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_err = "ncclGetUniqueId failed"; return SM_E_HIP; }
    memcpy(out128, &id, 128);
    return SM_OK;
}

int sm_shard_rccl_init(sm_ctx *s, const void *id128)
{
    if (!s || !id128 || !(s->ss_on || s->rig_on)) { g_err = "sm_shard_rccl_init: call sm_shard_stream_configure or sm_rig_configure first"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    if (hip_runtime_conflict("sm_shard_rccl_init")) return SM_E_HIP;
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, s->ss_world, id, s->ss_rank);
    if (r != ncclSuccess) { g_err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "failed"); return SM_E_HIP; }
    s->ss_comm = comm;
    s->ss_coll = rccl_collective; s->ss_user = s;
    return SM_OK;
}

int sm_shard_rccl_nranks(sm_ctx *s)
{
    if (!s) return SM_E_ARG;
    if (!s->ss_comm || !g_rccl.CommCount) { g_err = "sm_shard_rccl_nranks: no RCCL communicator on this context"; return SM_E_ARG; }
    int n = 0;
    const ncclResult_t r = g_rccl.CommCount(static_cast<ncclComm_t>(s->ss_comm), &n);
    if (r != ncclSuccess) { g_err = "ncclCommCount failed"; return SM_E_HIP; }
    return n;
}

int sm_shard_rccl_finalize(sm_ctx *s)
{
    if (!s) return SM_E_ARG;
    if (s->ss_comm && g_rccl.CommDestroy) {
        HIPCK(hipSetDevice(s->cfg.device));
        HIPCK(hipStreamSynchronize(s->stream));
        (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(s->ss_comm));
    }
    s->ss_comm = nullptr;
    if (s->ss_coll == rccl_collective) { s->ss_coll = nullptr; s->ss_user = nullptr; }
    return SM_OK;
}

int sm_shard_compact(sm_ctx *s)
{
    if (!s || !s->ss_on) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    return ss_compact(s);
}

// SurfelMapping::processFrame for one rank of a sharded stream; images already on the device; enqueue only
int sm_shard_frame_device(sm_ctx *s, const uint8_t *d_rgb, const uint16_t *d_depth_mm, const uint8_t *d_semantic, const float *pose16)
{
    if (!s || !d_rgb || !pose16) { g_err = "sm_shard_frame_device: null argument"; return SM_E_ARG; }
    if (!s->ss_on) { g_err = "sm_shard_frame_device: call sm_shard_stream_configure first"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    if (s->pending_cull) { g_err = "sm_stage_conflict without sm_stage_cull"; return SM_E_ARG; }
    if (s->ref_set && s->tick == 0) { g_err = "reset() is not supported in sharded mode"; return SM_E_UNSUPPORTED; }
    const bool fusing = s->ref_set && s->tick != 0;
    int rc;
    if (fusing) {
        // The compaction schedule must be the same on every rank: the period counter, and a capacity bound that only uses
        // what all ranks know (after a synchronisation the host's bound is the device's count, identical everywhere).
        bool compact = s->cfg.compact_period <= 1 || s->culls_since_compact + 1 >= s->cfg.compact_period;
        if (!compact && (uint64_t)s->count_bound + s->n_odd_pixels > s->cap) {
            if ((rc = pull_state(s))) return rc;
            compact = (uint64_t)s->count_bound + s->n_odd_pixels > s->cap;
        }
        if (compact && s->culls_since_compact > 0) {
            if ((rc = ss_compact(s))) return rc;
            if ((uint64_t)s->count_bound + s->n_odd_pixels > s->cap && (rc = pull_state(s))) return rc;
        }
    }
    s->want_list = fusing;
    FrameParams fp;
    rc = begin_frame(s, d_rgb, d_depth_mm ? d_depth_mm : s->d_depth_raw, d_semantic ? d_semantic : s->d_sem, pose16, &fp);
    s->want_list = false;
    if (rc <= 0) return rc;
    fp.compact_now = 0u;
    fp.conflict_cap = 0xFFFFFFFFu;            // applied over all ranks below (k_shard_cap_pack / k_shard_cap_repair), not per shard
    fp.shard_slots = 1;
    note_cull(s, false);
    s->keys_are_slots = true;
    if (s->ev_ok) { s->ev_compacted[s->ev_frames % EV_RING] = false; s->ev_one_pass[s->ev_frames % EV_RING] = true; s->ev_direct[s->ev_frames % EV_RING] = true; }
    if (s->n_prep_blocks == 0) { g_err = "internal: sharded frame without tile flags from k_prep"; return SM_E_ARG; }
    if ((rc = launch_surfel_pass(s, fp, true, true))) return rc;
    // The W*H conflict cap acts in surfel order over ALL ranks: exchange the conflict masks and take this rank's surplus back
    // before anything reads the key map (k_shard_cap_pack / k_shard_cap_repair).  Conflicts <= surfels, so a model with no
    // more slots than pixels cannot reach the cap; the bound is the host's, the same on every rank.
    if (s->cfg.conflict_cap && (uint64_t)s->count_bound > (uint64_t)s->P) {
        const uint32_t tbnd = (uint32_t)std::min<uint64_t>(((uint64_t)s->count_bound + TILE - 1) / TILE, s->dead_tiles);
        const int gp = (int)std::min<uint32_t>(std::max<uint32_t>((tbnd * (uint32_t)TILE_WORDS + 255u) / 256u, 1u), 1024u);
        hipLaunchKernelGGL(k_shard_cap_pack, dim3(gp), dim3(256), 0, s->stream, s->d_state, s->d_wave_cnt, s->d_cm,
                           s->d_conf_sub + SUB_SET * s->conf_sub_set, s->d_capx, tbnd);
        HIPCK(hipGetLastError());
        if ((rc = ss_collective(s, s->d_capx, s->d_capx, (size_t)1 + (size_t)(2 + TILE_WORDS) * tbnd, SM_COLL_SUM))) return rc;
        hipLaunchKernelGGL(k_shard_cap_repair, dim3(std::min<uint32_t>(std::max<uint32_t>(tbnd, 1u), (uint32_t)MAX_GRID)), dim3(256), 0, s->stream, s->M,
                           s->d_state, fp, s->d_capx, tbnd, (uint32_t)s->P, s->d_wave_cnt, s->d_dm /* km */, s->d_tile_flags, s->d_alive,
                           s->d_tile_dead, s->d_keyT, s->d_undo, s->d_tb);
        HIPCK(hipGetLastError());
    }
    if ((rc = ss_collective(s, s->d_keyT, s->d_keyT, (size_t)s->P, SM_COLL_MIN))) return rc;
    ShardArgs sh;
    sh.validmask = s->d_validmask; sh.ownmask = s->d_fusedmask; sh.gmask = s->d_gmask; sh.nwords = (uint32_t)((s->P + 63) / 64);
    sh.owner = (int)(s->ss_frames % (uint32_t)s->ss_world) == s->ss_rank ? 1 : 0;
    AssocArgs aa;
    fill_assoc_args(s, fp, aa);
    hipLaunchKernelGGL((k_associate_direct<true>), dim3(assoc_wgs(s)), dim3(PIX_BLOCK), 0, s->stream, aa, sh);
    HIPCK(hipGetLastError());
    if ((rc = mark(s, 5, true))) return rc;
    if ((rc = ss_collective(s, s->d_gmask, s->d_gmask, (size_t)sh.nwords + 4, SM_COLL_SUM))) return rc;   // in place, like the key map
    // the frame's last step (counts from the reduced mask, the owner's foreign-fused slots, the totals over the ranks) is
    // only needed by the next frame's surfel pass: it rides on that frame's k_prep (finalize_if_pending runs it earlier if asked)
    ShardSettle &ss = s->ss_settle;
    ss.n = (uint32_t)s->n_pix_blocks; ss.st = s->d_state; ss.validmask = s->d_validmask; ss.ownmask = s->d_fusedmask; ss.gmask = s->d_gmask;
    ss.nwords = sh.nwords; ss.blk_cand = s->d_blk_cand; ss.grp_cand = s->d_grp_cand; ss.nf = s->d_nf_sub; ss.alive = s->d_alive;
    ss.tile_dead = s->d_tile_dead; ss.owner = sh.owner; ss.cap_pixels = s->cfg.conflict_cap ? (uint32_t)s->P : 0xFFFFFFFFu; ss.max_vertices = s->cap; ss.cg = s->cand_group;
    s->ss_settle_pending = true;
    if ((rc = mark(s, 6, true)) || (rc = mark(s, 7, true))) return rc;
    s->lazy_part_live = false;
    s->fix_part_live = false;
    s->pend_finalize = true;
    s->frames_enq++;
    s->ss_frames++;
    bump_bound(s);
    end_frame(s);
    return SM_OK;
}

int sm_shard_frame(sm_ctx *s, const uint8_t *rgb, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16)
{
    if (!s || !rgb || !pose16) { g_err = "sm_shard_frame: null argument"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    int rc = upload_inputs(s, rgb, depth_mm, semantic);
    if (rc) return rc;
    if ((rc = sm_shard_frame_device(s, s->d_rgb, s->d_depth_raw, s->d_sem, pose16))) return rc;
    return sm_sync(s);
}

// This rank's part of the (compacted) union as a dense AoS plane of `*count` surfels with zeros where other ranks own the
// slot: the integer sum of the planes over the ranks is the single GlobalModel.  Collective (it compacts first).
int sm_shard_export_dense_device(sm_ctx *s, const float **d_out12, uint32_t *count)
{
    if (!s || !d_out12 || !count || !s->ss_on) return SM_E_ARG;
    HIPCK(hipSetDevice(s->cfg.device));
    if (hip_runtime_conflict("sm_shard_export_dense_device")) return SM_E_HIP;
    int rc = ensure_compact(s);
    if (rc) return rc;
    if ((rc = pull_state(s))) return rc;
    const uint32_t n = s->h_state->count;
    if ((rc = ensure_export(s, (size_t)std::max<uint32_t>(n, 1) * 48))) return rc;
    if (n) hipLaunchKernelGGL(k_shard_export_aos, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, s->d_alive, (float *)s->d_export, n);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(s->stream));
    *d_out12 = (const float *)s->d_export;
    *count = n;
    return SM_OK;
}


// ---- BASELINE configs[4]: a rig of `world` cameras, one per rank, consolidated into a single GlobalModel ----
// Frames go through the ordinary entry points (no collective).  sm_rig_consolidate is the definition of DESIGN.md 6 --
// union in rank order, cleanPoints against every camera's latest view in rank order -- entirely on the device: the views,
// the slice sizes, the per-view conflict totals and the cleaned slices cross the ranks through the installed collective
// (RCCL's all-reduce, or a callback); an all-gather is the sum of buffers that are zero outside the sender's part.

int sm_rig_configure(sm_ctx *s, int rank, int world)
{
    if (!s || world < 1 || rank < 0 || rank >= world) return SM_E_ARG;
    if (s->ss_on) { g_err = "sm_rig_configure: the context is configured for sharding"; return SM_E_ARG; }
    s->rig_on = true; s->ss_rank = rank; s->ss_world = world;
    return SM_OK;
}

namespace {
// The exchanges of a rig consolidation.  Every rank contributes a row of four words -- live surfels of its slice, conflicts of
// the view at hand, a status word, a spare -- through an all-gather, so that (a) all ranks see all counts and (b) a rank whose
// LOCAL step failed says so in the very exchange the others are waiting in: everybody then leaves together with an error
// instead of one rank returning early and the rest blocking inside RCCL.
struct RigXchg {
    sm_ctx *s; int W, r;
    unsigned long long *d_cnt = nullptr;            // [W][4]
    std::vector<unsigned long long> h;
    RigXchg(sm_ctx *s_, int W_, int r_) : s(s_), W(W_), r(r_), h((size_t)W_ * 4) {}
    // returns 0, this rank's own failure code, or SM_E_HIP when another rank failed
    int run(unsigned long long count, unsigned long long conflicts, int status)
    {
        unsigned long long row[4] = {count, conflicts, (unsigned long long)(long long)status, 0ull};
        if (hipMemcpyAsync(d_cnt + (size_t)r * 4, row, 32, hipMemcpyHostToDevice, s->stream) != hipSuccess) return SM_E_HIP;
        int rc = ss_collective(s, d_cnt + (size_t)r * 4, d_cnt, 4, SM_COLL_GATHER);
        if (rc) return rc;
        if (hipMemcpyAsync(h.data(), d_cnt, 32 * (size_t)W, hipMemcpyDeviceToHost, s->stream) != hipSuccess) return SM_E_HIP;
        if (hipStreamSynchronize(s->stream) != hipSuccess) return SM_E_HIP;
        if (status) return status;
        for (int q = 0; q < W; ++q)
            if (h[(size_t)q * 4 + 2]) { g_err = "sm_rig_consolidate: rank " + std::to_string(q) + " failed (code " + std::to_string((long long)h[(size_t)q * 4 + 2]) + "); all ranks abandon the consolidation"; return SM_E_HIP; }
        return SM_OK;
    }
    unsigned long long count(int q) const { return h[(size_t)q * 4]; }
    unsigned long long conflicts(int q) const { return h[(size_t)q * 4 + 1]; }
};
}  // namespace

// The single GlobalModel DURING a run ("all-gather of per-GPU new-surfel lists into the single GlobalModel, followed by
// one conflict pass of every camera's depth against the union").  One step, collective:
//   1. every rank's NEW surfels -- created since its previous step, still alive, in creation order (the model is kept in creation
//      order, so they are a suffix of the compacted model) -- are all-gathered and appended to `global` in rank order
//      (GlobalModel::concatenate's order for W append lists), on every rank;
//   2. the cameras' latest views are all-gathered and `global` is cleaned against each of them in rank order with
//      SurfelMapping::cleanPoints (src/SurfelMapping.cpp:496-532) -- replicated: every rank holds the same GlobalModel, so the W*H
//      conflict cap and the id-0 rule need no exchange, and the work runs on `global`'s own stream, next to the camera's frames.
// The camera's own slice is not touched (its fusion goes on as if alone); what an older surfel of it becomes later -- fused
// updates, its own culls -- reaches `global` only through the views' conflict tests.  sm_rig_consolidate above is the exact
// end-of-run union; this is the incremental model, defined by the same reference operations (tests/test_rig.py states it on
// oracles).
int sm_rig_consolidate_step(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16, sm_ctx *global,
                            uint32_t *new_surfels, uint32_t *global_count)
{
    if (!s || !depth_mm || !semantic || !pose16 || !global || !s->rig_on) { g_err = "sm_rig_consolidate_step: bad argument (sm_rig_configure first)"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    if (hip_runtime_conflict("sm_rig_consolidate_step")) return SM_E_HIP;
    const int W = s->ss_world, r = s->ss_rank;
    const size_t P = (size_t)s->P;
    const size_t off_sem = 2 * P, off_pose = (3 * P + 7) / 8 * 8, row = off_pose + 64;
    uint8_t *d_views = nullptr;
    float *d_lists = nullptr;
    uint32_t *d_first = nullptr;
    RigXchg x(s, W, r);
    auto done = [&](int code) { (void)hipFree(d_views); (void)hipFree(d_lists); (void)hipFree(d_first); (void)hipFree(x.d_cnt); return code; };
    HIPCK(hipMalloc((void **)&x.d_cnt, 32 * (size_t)W));
    // ---- local: compact (slots become positions), find the first surfel newer than the previous step, stage the view
    int st = ensure_compact(s);
    if (!st) st = pull_state(s);
    uint32_t cnt = 0, first = 0;
    if (!st) {
        cnt = s->h_state->count;
        first = cnt;
        if (hipMalloc((void **)&d_first, 4) != hipSuccess || hipMemsetAsync(d_first, 0xFF, 4, s->stream) != hipSuccess) st = SM_E_HIP;
        if (!st && cnt) {
            hipLaunchKernelGGL(k_first_newer, dim3(std::min<uint32_t>((cnt + 255u) / 256u, 1024u)), dim3(256), 0, s->stream, s->M, s->d_state, s->rig_last_time, d_first);
            uint32_t f = 0xFFFFFFFFu;
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&f, d_first, 4, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
                hipStreamSynchronize(s->stream) != hipSuccess) st = SM_E_HIP;
            else first = std::min(f, cnt);
        }
    }
    const uint32_t n_new = st ? 0u : cnt - first;
    if (!st && hipMalloc((void **)&d_views, row * (size_t)W) != hipSuccess) { g_err = "sm_rig_consolidate_step: out of device memory for the views"; st = SM_E_HIP; }
    if (!st && (hipMemsetAsync(d_views + row * r, 0, row, s->stream) != hipSuccess ||
                hipMemcpyAsync(d_views + row * r, depth_mm, 2 * P, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
                hipMemcpyAsync(d_views + row * r + off_sem, semantic, P, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
                hipMemcpyAsync(d_views + row * r + off_pose, pose16, 64, hipMemcpyHostToDevice, s->stream) != hipSuccess)) st = SM_E_HIP;
    int rc = x.run(n_new, 0ull, st);
    if (rc) return done(rc);
    // ---- 1. the new-surfel lists, all-gathered (padded to the longest) and appended to `global` in rank order
    unsigned long long T = 0, maxn = 0;
    std::vector<unsigned long long> nn((size_t)W);
    for (int q = 0; q < W; ++q) { nn[(size_t)q] = x.count(q); T += nn[(size_t)q]; maxn = std::max(maxn, nn[(size_t)q]); }
    if (new_surfels) *new_surfels = (uint32_t)T;
    st = SM_OK;
    if (T && hipMalloc((void **)&d_lists, (size_t)maxn * 48 * (size_t)W) != hipSuccess) { g_err = "sm_rig_consolidate_step: out of device memory for the lists"; st = SM_E_HIP; }
    if ((rc = x.run(n_new, 0ull, st))) return done(rc);
    if (T) {
        if (n_new) hipLaunchKernelGGL(k_export_aos, dim3((n_new + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, d_lists + (size_t)r * maxn * 12, first, n_new);
        if (hipGetLastError() != hipSuccess) return done(SM_E_HIP);
        if ((rc = ss_collective(s, d_lists + (size_t)r * maxn * 12, d_lists, (size_t)maxn * 6, SM_COLL_GATHER))) return done(rc);
    }
    if ((rc = ss_collective(s, d_views + row * r, d_views, row / 8, SM_COLL_GATHER))) return done(rc);
    std::vector<float> poses((size_t)W * 16);
    for (int v = 0; v < W; ++v)
        if (hipMemcpyAsync(&poses[(size_t)v * 16], d_views + row * v + off_pose, 64, hipMemcpyDeviceToHost, s->stream) != hipSuccess) return done(SM_E_HIP);
    if (hipStreamSynchronize(s->stream) != hipSuccess) return done(SM_E_HIP);
    for (int q = 0; q < W; ++q)
        if (nn[(size_t)q] && (rc = sm_append_model_aos_device(global, d_lists + (size_t)q * maxn * 12, (uint32_t)nn[(size_t)q]))) return done(rc);
    // ---- 2. the union cleaned against every camera's latest view, in rank order (the same work on every rank)
    for (int v = 0; v < W; ++v)
        if ((rc = clean_points_device(global, reinterpret_cast<const uint16_t *>(d_views + row * v), d_views + row * v + off_sem,
                                      &poses[(size_t)v * 16], 1))) return done(rc);
    if (global_count) *global_count = global->counts.count;
    s->rig_last_time = (float)(s->tick - 1);           // every surfel created so far carries a time stamp <= tick - 1
    return done(SM_OK);
}

int sm_rig_consolidate(sm_ctx *s, const uint16_t *depth_mm, const uint8_t *semantic, const float *pose16, sm_ctx *global,
                       uint32_t *view_conflicts, uint32_t *total_out)
{
    if (!s || !depth_mm || !semantic || !pose16 || !global || !s->rig_on) { g_err = "sm_rig_consolidate: bad argument (sm_rig_configure first)"; return SM_E_ARG; }
    HIPCK(hipSetDevice(s->cfg.device));
    if (hip_runtime_conflict("sm_rig_consolidate")) return SM_E_HIP;
    const int W = s->ss_world, r = s->ss_rank;
    const size_t P = (size_t)s->P;
    const size_t off_sem = 2 * P, off_pose = (3 * P + 7) / 8 * 8, row = off_pose + 64;        // bytes of one view (a multiple of 8)
    uint8_t *d_views = nullptr;
    float *d_union = nullptr;
    RigXchg x(s, W, r);
    auto done = [&](int code) { (void)hipFree(d_views); (void)hipFree(d_union); (void)hipFree(x.d_cnt); return code; };
    // the exchange buffer first: without it this rank cannot even tell the others that it failed
    HIPCK(hipMalloc((void **)&x.d_cnt, 32 * (size_t)W));
    // ---- local, fallible: settle the stream's pending work, stage this camera's latest view
    int st = finalize_if_pending(s);
    if (!st) st = pull_state(s);
    if (!st && hipMalloc((void **)&d_views, row * (size_t)W) != hipSuccess) { g_err = "sm_rig_consolidate: out of device memory for the views"; st = SM_E_HIP; }
    if (!st && (hipMemsetAsync(d_views + row * r, 0, row, s->stream) != hipSuccess ||
                hipMemcpyAsync(d_views + row * r, depth_mm, 2 * P, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
                hipMemcpyAsync(d_views + row * r + off_sem, semantic, P, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
                hipMemcpyAsync(d_views + row * r + off_pose, pose16, 64, hipMemcpyHostToDevice, s->stream) != hipSuccess)) {
        g_err = "sm_rig_consolidate: staging the view failed"; st = SM_E_HIP;
    }
    int rc = x.run(st ? 0ull : s->counts.count, 0ull, st);
    if (rc) return done(rc);
    // ---- 1. every rank learns every camera's latest view: all-gather, in place (3 bytes per pixel and camera)
    if ((rc = ss_collective(s, d_views + row * r, d_views, row / 8, SM_COLL_GATHER))) return done(rc);
    std::vector<float> poses((size_t)W * 16);
    st = SM_OK;
    for (int v = 0; v < W && !st; ++v)
        if (hipMemcpyAsync(&poses[(size_t)v * 16], d_views + row * v + off_pose, 64, hipMemcpyDeviceToHost, s->stream) != hipSuccess) st = SM_E_HIP;
    if (!st && hipStreamSynchronize(s->stream) != hipSuccess) st = SM_E_HIP;
    // ---- 2. the union cleaned against every view, in rank order: each rank cleans ITS slice (the test is per surfel and view)
    for (int v = 0; v < W; ++v) {
        int first = -1;
        for (int q = 0; q < W && first < 0; ++q) if (x.count(q) > 0) first = q;
        unsigned long long view_total = 0;
        bool hook_ran = false;
        // Between the conflict test and the cull the ranks exchange their conflict counts: at most W*H conflicts take effect per
        // view, in the surfel order of the UNION (src/GlobalModel.cpp:54-57) -- slices are concatenated in rank order, so this
        // rank's share is what the lower ranks left of the W*H, and "the first `share` conflicts of my slice" is exactly the rule
        // the single-model cull applies with that cap.
        const std::function<long long(uint32_t)> hook = [&](uint32_t local) -> long long {
            hook_ran = true;
            const int e = x.run(s->counts.count, local, SM_OK);
            if (e) return e;
            unsigned long long before = 0;
            for (int q = 0; q < W; ++q) { if (q < r) before += x.conflicts(q); view_total += x.conflicts(q); }
            if (!s->cfg.conflict_cap) return 0xFFFFFFFFll;
            return before >= (unsigned long long)s->P ? 0ll : (long long)std::min<unsigned long long>(local, (unsigned long long)s->P - before);
        };
        // surfel id 0 never conflicts (conflict.geom:15): the exemption belongs to the rank that holds the union's first surfel
        const int cl = st ? st : clean_points_device(s, reinterpret_cast<const uint16_t *>(d_views + row * v), d_views + row * v + off_sem,
                                                     &poses[(size_t)v * 16], first == r ? 1 : 0, &hook);
        // every rank makes both exchanges of a view whatever happened to it locally: a failure travels in the status word
        if (!hook_ran) (void)x.run(0ull, 0ull, cl ? cl : SM_E_HIP);
        if ((rc = x.run(s->counts.count, 0ull, cl))) return done(rc);              // the slices' sizes after this view
        if (view_conflicts) view_conflicts[v] = (uint32_t)(s->cfg.conflict_cap ? std::min<unsigned long long>(view_total, (unsigned long long)s->P) : view_total);
    }
    // ---- 3. the cleaned slices, all-gathered (padded to the largest) and appended in rank order to `global` on every rank
    unsigned long long T = 0, maxcnt = 0;
    std::vector<unsigned long long> cnt((size_t)W);
    for (int q = 0; q < W; ++q) { cnt[(size_t)q] = x.count(q); T += cnt[(size_t)q]; maxcnt = std::max(maxcnt, cnt[(size_t)q]); }
    if (total_out) *total_out = (uint32_t)T;
    (void)hipFree(d_views); d_views = nullptr;
    st = SM_OK;
    if (T && hipMalloc((void **)&d_union, (size_t)maxcnt * 48 * (size_t)W) != hipSuccess) { g_err = "sm_rig_consolidate: out of device memory for the union"; st = SM_E_HIP; }
    if (!st) st = ensure_compact(s);
    if (!st) st = pull_state(s);
    if ((rc = x.run(cnt[(size_t)r], 0ull, st))) return done(rc);
    if (T == 0) return done(SM_OK);
    const uint32_t own = s->h_state->count;
    if (own) hipLaunchKernelGGL(k_export_aos, dim3((own + 255) / 256), dim3(256), 0, s->stream, s->M, s->d_state, d_union + (size_t)r * maxcnt * 12, 0u, own);
    if (hipGetLastError() != hipSuccess) { g_err = "sm_rig_consolidate: export kernel launch failed"; return done(SM_E_HIP); }   // (the others' all-gather then fails or stalls: a launch failure is not recoverable)
    if ((rc = ss_collective(s, d_union + (size_t)r * maxcnt * 12, d_union, (size_t)maxcnt * 6, SM_COLL_GATHER))) return done(rc);
    if (hipStreamSynchronize(s->stream) != hipSuccess) return done(SM_E_HIP);
    for (int q = 0; q < W; ++q)
        if (cnt[(size_t)q] && (rc = sm_append_model_aos_device(global, d_union + (size_t)q * maxcnt * 12, (uint32_t)cnt[(size_t)q]))) return done(rc);
    return done(SM_OK);
}

}  // extern "C"
