// rtw_hip.hip — Monte-Carlo path tracer for MI355X (gfx950) behind the C ABI of include/rtw.h.
//
// What the reference does in ONE OptiX megakernel launch (Director.cpp:982-984: raygen -> traverse ->
// closest-hit/miss -> callables, one thread per pixel, one sample) is done here by one of two pipelines:
//
//   k_path (scenes walked with the candidate lists: <= 24 primitives)
//     one persistent launch per pass over the samples: whole paths in registers, a lane owns a (pixel, 64-sample block)
//     unit, regenerates camera rays as its paths end and stores one 16-byte sum per block; k_resolve_blocks adds the
//     block sums per pixel in order
//
//   wavefront kernels (tree scenes), batches of S samples per pixel, P = pixels*S paths in flight in HBM:
//     k_first                     primary rays, their closest hit and closest-hit program (light sample queued)
//     repeat for the wide bounces (1..19):
//       k_trace_bvh               radiance ray + queued shadow probe of every surviving path, surfaces only
//       k_shade                   (volume pass,) material scatter, light sample (probe queued), roulette, compaction
//     k_bounce x few              thin tail: several fused bounces per launch, in registers
//     two batches are in flight on two streams ("lanes"); the resolves stay ordered on the caller's stream
//     k_resolve                   sums the S sample slots of each pixel in the spec's blocked order (deterministic)
//
//   k_finish (once)               mean radiance -> float4 framebuffer tile
//   n_devices > 1: one host thread + context per device, interleaved row shards, peer-copy gather, k_interleave
//
// No OptiX, no CUDA shims, no Triton, no MFMA (divergent scalar fp32). Results do not depend on
// scheduling: every path owns a counter-based RNG stream and every (pixel, block) its own sum.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtw.h"
#include "rtw_bvh.h"
#include "rtw_device.h"
#include "rtw_kernels.h"

using namespace rtwdev;
using namespace rtwk;

#ifdef RTW_SPLIT_BUILD
// __graft_entry__.build() compiles the shading kernels in translation units of their own, in parallel (rtw_inst_path.hip,
// rtw_inst_shade.hip): here they are only declared
namespace rtwk {
#define RTW_EXT(K_) \
    extern template __global__ void K_<RTW_RNG_PHILOX, 0>(const KArgs); extern template __global__ void K_<RTW_RNG_PHILOX, 1>(const KArgs); extern template __global__ void K_<RTW_RNG_PHILOX, 2>(const KArgs); \
    extern template __global__ void K_<RTW_RNG_TEA_LCG, 0>(const KArgs); extern template __global__ void K_<RTW_RNG_TEA_LCG, 1>(const KArgs); extern template __global__ void K_<RTW_RNG_TEA_LCG, 2>(const KArgs);
RTW_EXT(k_path)
extern template __global__ void k_path<RTW_RNG_PHILOX, 1, 1>(const KArgs);
extern template __global__ void k_path<RTW_RNG_TEA_LCG, 1, 1>(const KArgs);
RTW_EXT(k_first)
RTW_EXT(k_shade)
RTW_EXT(k_bounce)
#undef RTW_EXT
extern template __global__ void k_classify<true>(const KArgs, uint32_t*, uint32_t*, uint32_t);
extern template __global__ void k_classify<false>(const KArgs, uint32_t*, uint32_t*, uint32_t);
}  // namespace rtwk
#endif

constexpr int kBruteMaxPrims = 24;  // at or below: scalar-cache brute lists; above: BVH with the LDS stack

// =============================================================================================
// host side of the library
struct rtw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // scene
    bool has_scene = false;
    DScene sc{};
    void* d_scene = nullptr;   // one allocation holding all scene tables
    int stack_depth = 0;
    size_t lds_bytes = 0;
    size_t n_tree_nodes = 0, n_tree_leaves = 0;  // 4-wide nodes and leaf records of the uploaded scene's tree
    // render pool
    // Two independent "lanes" (stream + path pool): consecutive batches alternate between them so that the
    // bandwidth-bound kernels of one batch overlap the compute- and latency-bound kernels of the other.
    struct Lane {
        hipStream_t st = nullptr;
        hipEvent_t ev_done = nullptr, ev_free = nullptr;
        PathBuf buf[2] = {};
        uint2* hit[2] = {nullptr, nullptr};
        void* slab[2] = {nullptr, nullptr};  // one allocation per ping-pong set: six planes + hit records
        float4* lbuf = nullptr;
        uint32_t* cnt = nullptr;
        size_t cnt_words = 0;
        size_t paths = 0;
        bool probes = false;   // the slabs hold the planes of scenes with listed lights (p2, a full p5) as well
    } lane[4];
    float4* accum = nullptr;   // per pixel: sum of the finished sample blocks
    float4* part = nullptr;    // per pixel: running sum of the current block (wavefront pipeline)
    float4* upart = nullptr;   // per pixel: running sum of the current summation unit's finished blocks (wavefront pipeline)
    size_t accum_pix = 0;
    float4* blocksum = nullptr;  // k_path: [block][pixel] unit sums of one pass
    size_t blocksum_elems = 0;
    uint32_t* d_queue = nullptr; // k_path: job counter [0], k_classify's two counters [1], [2]
    uint32_t* d_order = nullptr; // k_path: job order of the pixel groups
    hipStream_t stream2 = nullptr;  // k_path: the stream of the fine-grained end-game launch
    size_t order_groups = 0;
    std::vector<hipEvent_t> ev_pool;  // timing events, reused across launches and calls
    unsigned long long* d_stats = nullptr;
    float4* d_out = nullptr;
    size_t out_pix = 0;
    int n_cu = 256;
    // n_devices > 1: this context is a group; kids[g] renders the g-th interleaved sub-shard on device_ids[g], the
    // shards are gathered on device_ids[0] (this->device) into `stage` and interleaved into the caller's frame
    std::vector<rtw_ctx*> kids;
    float4* stage = nullptr;
    size_t stage_pix = 0;
    size_t pool_cap = ~(size_t)0;  // paths in flight this device has room for (halved when a pool allocation fails: render_single)
    struct Worker* worker = nullptr;  // a kid's host thread: created with the group, lives until rtw_destroy
};

// One persistent host thread per kid context of a group (n_devices > 1). The caller's thread hands it one shard per render
// call and waits; no thread is created or joined per call.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool has_job = false, done = false, quit = false;
    // the job (written by the caller's thread before has_job is set, read by the worker; results the other way)
    rtw_params P{};
    size_t npix = 0;
    float4* gather_dst = nullptr;  // where the shard goes on the group's first device
    int gather_dev = 0;
    bool want_stats = false;
    rtw_stats st{};
    int rc = RTW_OK;
};

namespace {

int fail(rtw_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RTW_ERR_OOM : RTW_ERR_DEVICE,             \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

void free_lane(rtw_ctx::Lane& L) {
    for (int b = 0; b < 2; b++) {
        if (L.slab[b]) (void)hipFree(L.slab[b]);
        L.slab[b] = nullptr;
        L.buf[b] = PathBuf{};
        L.hit[b] = nullptr;
    }
    if (L.lbuf) (void)hipFree(L.lbuf);
    L.lbuf = nullptr;
    L.paths = 0;
    L.probes = false;
}
void free_pool(rtw_ctx* c) {
    for (auto& L : c->lane) {
        free_lane(L);
        if (L.cnt) (void)hipFree(L.cnt);
        L.cnt = nullptr; L.cnt_words = 0;
    }
}

// Streams that must run side by side get DIFFERENT priorities. The runtime maps streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES, default 4) round-robin in creation order, and two streams that land on one queue serialise: whether the
// two lanes overlapped used to depend on how many streams the process (torch, an earlier render) had created before them
// (measured: scene 1 at 3.3 instead of 4.1 Gsamples/s after one small k_path render). Queues are pooled per priority class,
// so a normal- and a high-priority stream never share one.
hipError_t create_stream(hipStream_t* st, int cls /* 0 normal, 1 high, 2 low */) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    const int normal = (least + greatest) / 2;
    const int prio = cls == 1 ? greatest : (cls == 2 ? least : normal);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio);
}

int ensure_lane(rtw_ctx* c, rtw_ctx::Lane& L, size_t paths, size_t cnt_words, bool probes) {
    if (!L.st) {
        const int idx = (int)(&L - c->lane);
        HIP_TRY(c, create_stream(&L.st, idx & 1));
        HIP_TRY(c, hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&L.ev_free, hipEventDisableTiming));
    }
    if (paths > L.paths || (probes && !L.probes)) {
        paths = std::max(paths, L.paths);
        free_lane(L);
        // the state planes and the hit records of one ping-pong set live in one slab, page-aligned: with listed lights six
        // 16-byte planes + the 8-byte hit records (104 B per path); without (no probe is ever queued: rtw_kernels.h store_path)
        // p2 is not there and p5 is a plane of dwords: 76 B per path - a third less of what decides how large a batch can be
        auto al = [](size_t v) { return (v + 4095) & ~(size_t)4095; };
        const size_t full = al(paths * sizeof(float4));
        const size_t sz[7] = {full, full, probes ? full : 0, full, full, probes ? full : al(paths * sizeof(uint32_t)), al(paths * sizeof(uint2))};
        size_t off[8] = {0};
        for (int k = 0; k < 7; k++) off[k + 1] = off[k] + sz[k];
        for (int b = 0; b < 2; b++) {
            HIP_TRY(c, hipMalloc(&L.slab[b], off[7] + 4096));
            char* base = (char*)L.slab[b];
            L.buf[b].p0 = (float4*)(base + off[0]); L.buf[b].p1 = (float4*)(base + off[1]); L.buf[b].p2 = (float4*)(base + off[2]);
            L.buf[b].p3 = (float4*)(base + off[3]); L.buf[b].p4 = (float4*)(base + off[4]); L.buf[b].p5 = (uint4*)(base + off[5]);
            L.hit[b] = (uint2*)(base + off[6]);
        }
        HIP_TRY(c, hipMalloc(&L.lbuf, paths * sizeof(float4)));
        L.paths = paths;
        L.probes = probes;
    }
    if (cnt_words > L.cnt_words) {
        if (L.cnt) (void)hipFree(L.cnt);
        L.cnt = nullptr; L.cnt_words = 0;
        HIP_TRY(c, hipMalloc(&L.cnt, cnt_words * sizeof(uint32_t)));
        L.cnt_words = cnt_words;
    }
    return RTW_OK;
}

int ensure_pool(rtw_ctx* c, int n_lanes, size_t paths, size_t npix, size_t cnt_words, bool probes = true) {
    for (int l = 0; l < n_lanes; l++) {
        int rc = ensure_lane(c, c->lane[l], paths, cnt_words, probes);
        if (rc) return rc;
    }
    if (n_lanes == 0 && !c->d_queue) HIP_TRY(c, hipMalloc(&c->d_queue, 64));
    if (npix > c->accum_pix) {
        if (c->accum) (void)hipFree(c->accum);
        c->accum = nullptr; c->accum_pix = 0;
        HIP_TRY(c, hipMalloc(&c->accum, npix * sizeof(float4)));
        if (c->part) (void)hipFree(c->part);
        c->part = nullptr;
        HIP_TRY(c, hipMalloc(&c->part, npix * sizeof(float4)));
        if (c->upart) (void)hipFree(c->upart);
        c->upart = nullptr;
        HIP_TRY(c, hipMalloc(&c->upart, npix * sizeof(float4)));
        c->accum_pix = npix;
    }
    if (!c->d_stats) HIP_TRY(c, hipMalloc(&c->d_stats, (kStatRows + 1) * 8 * sizeof(unsigned long long)));
    return RTW_OK;
}

// multiply-high constants for exact 32-bit division by an invariant d >= 1 (Granlund & Montgomery 1994)
void magic_div(uint32_t d, uint32_t& m, uint32_t& s1, uint32_t& s2) {
    uint32_t l = 0;
    while (l < 32 && ((uint64_t)1 << l) < d) l++;
    m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    s1 = l < 1 ? l : 1;
    s2 = l > 0 ? l - 1 : 0;
}

// Tuning knobs, read from the environment once per call site (defaults are what profiles/ was measured with):
//   RTW_POOL_PATHS  paths in flight over all lanes (default 2^30: sized for 288 GB of HBM - a full-HD frame at 512+ spp takes 178 GB of
//                   state without listed lights, 238 GB with; an allocation that fails is halved, see render_single. Round 3, BASELINE
//                   config 3: 2^28 (8 batches of 64 spp) 6.7, 2^29 (4 x 128) 7.0, 2^30 (2 x 256) 7.3 Gsamples/s)
//   RTW_LANES       stream lanes that overlap consecutive batches (default 2; 1..4)
//   RTW_GRID_MULT   persistent workgroups per CU (default 8 with one lane, 4 with two)
//   RTW_TAIL_START  first bounce handled by the fused multi-bounce tail launches (default 6; 20 for tree scenes, 40 for tree scenes with media)
//   RTW_FUSED=1     every bounce through the fused k_bounce
//   RTW_SPLIT_MEDIA=0  scenes with media: every bounce through k_bounce (default: split pipeline, volumes tested in the shading kernels)
//   RTW_BRUTE_MAX   largest primitive count walked with the scalar-cache brute lists (default 24; 0 forces the BVH)
//   RTW_LDS_KB      dynamic LDS per workgroup for traversal stacks + staged tree nodes (default 16)
//   RTW_TAIL_GROUP  bounces per launch of the first tail group (default 2; groups grow by half every second launch)
//   RTW_PAIRED=1    tree scenes, two lanes: batches run in pairs whose trace launches alternate, so that a k_trace_bvh always has the
//                   other batch's k_shade beside it (measured 12 % slower than the free-running lanes: off; RTW_PAIRED_TRACE_WAVES
//                   = waves per SIMD of the trace kernel in that mode, default 4)
//   RTW_FIRST_GROUP_LOG2  k_first: 2^n neighbouring threads start samples of one pixel (default 3; 0 = one sample of 64 pixels per wave)
//   RTW_STAGGER     how far the second lane starts behind the first, in percent of a batch (its first batch is cut short by that
//                   much; 0 = no offset). Default: 50 for the candidate-list scenes under RTW_PATH=0, 0 for tree scenes (there the
//                   extra batch costs more than the offset gains: scenes 1, 2, 4 +1-4 % at 512+ spp, +6-13 % at 128-256 spp)
//   RTW_PATH        1 (default): scenes walked with the brute lists render through k_path (paths in registers, in-wave
//                   regeneration); 0: always the wavefront pipeline
//   RTW_PATH_TREE   1: tree scenes render through k_path_tree (k_path's idea with a per-lane walk state machine and a vote on the
//                   kind of step; bit-identical, measured 25-45 % slower than the wavefront kernels on scenes 1, 2, 4); 0 (default): wavefront
//   RTW_PATH_UNIT_BLOCKS 16-sample blocks a lane takes as one unit in the bulk launch (default: 8 = 128 samples when a lane has
//                        600+ blocks to do, else 4)
//   RTW_PATH_FINE_BLOCKS blocks at the end of a pass that a second, concurrent launch hands out one by one (default: 8 behind 8-block
//                        units, 16 behind shorter ones)
//   RTW_PATH_JOB_BLOCKS  units per pixel in one k_path job (default 2: a job is 64 pixels x 2 units)
//   RTW_PATH_GRID_MULT   k_path workgroups per CU (default: what the occupancy query admits)
//   RTW_BLOCKSUM_BYTES   cap of the k_path block-sum buffer (default 16 GiB); larger renders run in passes over the samples
//   RTW_KERNEL_TIMING    0: no per-launch events even when the caller asks for rtw_stats (kernel_seconds stay 0)
struct Tuning {
    size_t pool_paths = (size_t)1 << 30;
    int lanes = 2;
    int grid_mult = 0;   // 0 = automatic
    int tail_start = 0;  // 0 = automatic: 6 for the brute-list scenes, 20 for tree scenes (40 with media)
    bool fused = false;
    bool split_media = true;  // RTW_SPLIT_MEDIA=0: scenes with media keep every bounce in k_bounce
    int brute_max = kBruteMaxPrims;
    size_t lds_kb = 16;
    int trace_block = 256;       // threads per workgroup of k_trace_bvh (256, 512, 1024)
    size_t trace_lds_kb = 16;    // its LDS budget: stacks + tree nodes + leaf records
    bool trace_auto = true;      // neither RTW_TRACE_BLOCK nor RTW_TRACE_LDS_KB given: the render picks the pair (see render_single)
    int trace_waves = 5;         // waves per SIMD it is launched for (the kernel is compiled for 6: 78 VGPRs). 5 leaves a SIMD the 96 VGPRs of one
                                 // wave of the other lane's k_shade; medians of 5 renders, 6 -> 5 -> 4: scene 1 7 242 / 7 285 / 7 072 Msamples/s,
                                 // scene 2 3 593 / 3 664 / 3 646, scene 4 2 123 / 2 128 / 2 059 (profiles/r03_trace_waves_sweep.txt)
    int stagger_pct = -1;  // -1 = automatic
    int tail_group = 2;
    bool paired = false;         // RTW_PAIRED=1: tree scenes, two lanes: the batches of a pair alternate their trace launches
    int paired_trace_waves = 4;  // RTW_PAIRED_TRACE_WAVES: waves per SIMD of k_trace_bvh then (room for the other batch's k_shade)
    int first_group_log2 = 3;    // k_first: up to 2^this neighbouring threads take samples of one pixel (RTW_FIRST_GROUP_LOG2; round 2: 16;
                                 // round 3, with the wave-coherent walk: 8 (scene 1 medians of 5: 7 085 against 6 948-6 998 Msamples/s; 4: 7 091;
                                 // 2: 7 066; 1: 7 011; scenes 2 and 4 do not care). Round 2's note on 16:
                                 // k_first -6 ... -14 %; at 64 the later launches lose more - their finished paths then write
                                 // 16-byte results npix apart - than k_first gains)
    int path = 1;
    int path_tree = 0;
    int path_job_blocks = 2;
    int path_unit_blocks = 0;    // 0 = automatic (8 for large renders, else 4)
    int path_fine_blocks = -1;   // -1 = automatic (8 behind 8-block units, 16 behind shorter ones)
    int path_grid_mult = 0;
    size_t blocksum_bytes = (size_t)16 << 30;
    bool kernel_timing = true;
    bool verbose = false;  // RTW_VERBOSE=1: table sizes at upload (stderr)
};
Tuning read_tuning() {
    Tuning t;
    auto geti = [](const char* name, long long& out) {
        const char* e = getenv(name);
        if (!e || !*e) return false;
        out = atoll(e);
        return true;
    };
    long long v;
    if (geti("RTW_POOL_PATHS", v) && v >= 1024) t.pool_paths = (size_t)v;
    if (geti("RTW_LANES", v)) t.lanes = (int)std::max<long long>(1, std::min<long long>(4, v));
    if (geti("RTW_GRID_MULT", v)) t.grid_mult = (int)std::max<long long>(1, v);
    if (geti("RTW_TAIL_START", v)) t.tail_start = (int)std::max<long long>(1, v);
    if (geti("RTW_FUSED", v)) t.fused = v == 1;
    if (geti("RTW_SPLIT_MEDIA", v)) t.split_media = v != 0;
    if (geti("RTW_BRUTE_MAX", v)) t.brute_max = (int)v;
    if (geti("RTW_LDS_KB", v)) t.lds_kb = (size_t)std::max<long long>(0, v);
    if (geti("RTW_TRACE_BLOCK", v) && (v == 256 || v == 512 || v == 1024)) { t.trace_block = (int)v; t.trace_auto = false; }
    if (geti("RTW_TRACE_LDS_KB", v)) { t.trace_lds_kb = (size_t)std::max<long long>(0, std::min<long long>(150, v)); t.trace_auto = false; }
    if (geti("RTW_TRACE_WAVES", v)) t.trace_waves = (int)std::max<long long>(1, std::min<long long>(8, v));
    if (geti("RTW_TAIL_GROUP", v)) t.tail_group = (int)std::max<long long>(1, std::min<long long>(64, v));
    if (geti("RTW_FIRST_GROUP_LOG2", v)) t.first_group_log2 = (int)std::max<long long>(0, std::min<long long>(8, v));
    if (geti("RTW_PAIRED", v)) t.paired = v != 0;
    if (geti("RTW_PAIRED_TRACE_WAVES", v)) t.paired_trace_waves = (int)std::max<long long>(1, std::min<long long>(8, v));
    if (geti("RTW_STAGGER", v)) t.stagger_pct = (int)std::max<long long>(0, std::min<long long>(99, v));
    if (geti("RTW_PATH", v)) t.path = (int)std::max<long long>(0, std::min<long long>(2, v));
    if (geti("RTW_PATH_TREE", v)) t.path_tree = v != 0;
    if (geti("RTW_PATH_JOB_BLOCKS", v)) t.path_job_blocks = (int)std::max<long long>(1, std::min<long long>(1024, v));
    if (geti("RTW_PATH_UNIT_BLOCKS", v)) t.path_unit_blocks = (int)std::max<long long>(0, std::min<long long>(4096, v));
    if (geti("RTW_PATH_FINE_BLOCKS", v)) t.path_fine_blocks = (int)std::max<long long>(-1, std::min<long long>(4096, v));
    if (geti("RTW_PATH_GRID_MULT", v)) t.path_grid_mult = (int)std::max<long long>(1, std::min<long long>(16, v));
    if (geti("RTW_BLOCKSUM_BYTES", v) && v >= (1 << 16)) t.blocksum_bytes = (size_t)v;
    if (geti("RTW_KERNEL_TIMING", v)) t.kernel_timing = v != 0;
    if (geti("RTW_VERBOSE", v)) t.verbose = v != 0;
    return t;
}

// LDS of a tree-walking workgroup of `block` threads: the traversal stacks (16-bit entries unless a reference needs more),
// then as many leading (breadth-first) tree nodes and, once all nodes are in, leaf records as fit `budget` bytes.
size_t tree_lds_layout(size_t n_nodes_all, size_t n_leaves_all, int stack_depth, bool wide, size_t block, size_t budget, int32_t& n_nodes, int32_t& n_leaves) {
    const size_t stack_words = wide ? (size_t)stack_depth * block : ((size_t)stack_depth * block + 1) / 2;
    const size_t stack_bytes = ((stack_words + 3) & ~size_t(3)) * 4;
    size_t room = budget > stack_bytes ? budget - stack_bytes : 0;
    n_nodes = (int32_t)std::min<size_t>(n_nodes_all, room / sizeof(rtwbvh::Q4Node));
    room -= (size_t)n_nodes * sizeof(rtwbvh::Q4Node);
    n_leaves = (size_t)n_nodes == n_nodes_all ? (int32_t)std::min<size_t>(n_leaves_all, room / sizeof(rtwbvh::LeafRec)) : 0;
    return stack_bytes + (size_t)n_nodes * sizeof(rtwbvh::Q4Node) + (size_t)n_leaves * sizeof(rtwbvh::LeafRec);
}

// k_trace_bvh's instantiation for a workgroup size and a tree (mode: see the kernel)
typedef void (*TraceBvhKernel)(const KArgs);
int trace_bvh_mode(const DScene& sc) { return sc.stack_wide ? 0 : sc.n_lds_nodes >= sc.n_nodes ? 2 : 1; }
TraceBvhKernel trace_bvh_kernel(int block, int mode) {
#define RTW_TB(B_) (mode == 0 ? k_trace_bvh<B_, 0> : mode == 1 ? k_trace_bvh<B_, 1> : k_trace_bvh<B_, 2>)
    return block == 1024 ? RTW_TB(1024) : block == 512 ? RTW_TB(512) : RTW_TB(256);
#undef RTW_TB
}

enum { LK_FIRST = RTW_K_FIRST, LK_SHADE = RTW_K_SHADE, LK_TRACE = RTW_K_TRACE, LK_BOUNCE = RTW_K_BOUNCE, LK_PATH = RTW_K_PATH, LK_PATH_TREE = RTW_K_COUNT };
void launch(int which, int rng_kind, const KArgs& a, int grid, size_t lds, hipStream_t s, int block = kBlock) {
    const bool lcg = rng_kind == RTW_RNG_TEA_LCG;
    // kernels that shade exist in six instantiations: RNG kind x feature level
    const int feat = a.sc.has_tex;  // 0 hot, 1 cold features, 2 cold features + the mixture estimator (rtw_kernels.h shade_a)
#define RTW_LAUNCH_SHADING_R(K_, LDS_, R_)                                                                        \
    do {                                                                                                          \
        if (feat == 2) hipLaunchKernelGGL((K_<R_, 2>), dim3(grid), dim3(kBlock), LDS_, s, a);                     \
        else if (feat == 1) hipLaunchKernelGGL((K_<R_, 1>), dim3(grid), dim3(kBlock), LDS_, s, a);                \
        else hipLaunchKernelGGL((K_<R_, 0>), dim3(grid), dim3(kBlock), LDS_, s, a);                               \
    } while (0)
#define RTW_LAUNCH_SHADING(K_, LDS_)                                                                              \
    do {                                                                                                          \
        if (lcg) RTW_LAUNCH_SHADING_R(K_, LDS_, RTW_RNG_TEA_LCG); else RTW_LAUNCH_SHADING_R(K_, LDS_, RTW_RNG_PHILOX); \
    } while (0)
    switch (which) {
    case LK_FIRST: RTW_LAUNCH_SHADING(k_first, lds); break;
    case LK_SHADE: RTW_LAUNCH_SHADING(k_shade, 0); break;
    case LK_TRACE:
        if (a.sc.use_bvh) hipLaunchKernelGGL(trace_bvh_kernel(block, trace_bvh_mode(a.sc)), dim3(grid), dim3(block), lds, s, a);
        else if (a.sc.n_generic == 0) hipLaunchKernelGGL((k_trace<true>), dim3(grid), dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_trace<false>), dim3(grid), dim3(kBlock), lds, s, a);
        break;
    case LK_PATH:
        if (feat == 1 && a.sc.n_vol > 0) {  // media scenes: the cold instantiation allocated for 5 waves (rtw_kernels.h k_path MEDIA5)
            if (lcg) hipLaunchKernelGGL((k_path<RTW_RNG_TEA_LCG, 1, 1>), dim3(grid), dim3(kBlock), lds, s, a);
            else hipLaunchKernelGGL((k_path<RTW_RNG_PHILOX, 1, 1>), dim3(grid), dim3(kBlock), lds, s, a);
        } else {
            RTW_LAUNCH_SHADING(k_path, lds);
        }
        break;
#ifdef RTW_EXPERIMENTS
    case LK_PATH_TREE: RTW_LAUNCH_SHADING(k_path_tree, lds); break;
#endif
    default: RTW_LAUNCH_SHADING(k_bounce, lds); break;
#undef RTW_LAUNCH_SHADING
#undef RTW_LAUNCH_SHADING_R
    }
}

// ---- the exception barrier of the C ABI (include/rtw.h: "no exceptions cross this ABI") -------------------------------
// The implementations below use std::vector / std::string / std::thread and may throw (bad_alloc, system_error). Every
// extern "C" entry point runs its implementation inside guarded(): an exception becomes an error code and a message, as the
// reference's OPTIX_CHECK / CUDA_CHECK exceptions (Director.cpp:106-122) become the Director's exit path.
int fail_nothrow(rtw_ctx* c, int code, const char* msg) noexcept {
    if (c) {
        try { c->err.assign(msg); } catch (...) { c->err.clear(); }  // (clear() does not allocate)
    }
    return code;
}
// Test hook (tests/test_abi.py, tests/test_gpu_round3.py): RTW_TEST_FAULT="<where>:<kind>" makes the named point of the library
// throw (kind bad_alloc | runtime), so that the barrier and the worker threads' containment can be exercised on purpose.
// where: entry (every entry point, before its arguments are looked at), upload (after the scene tables are staged),
// worker (inside a group's worker thread, before its render).
void test_fault(const char* where) {
    const char* e = getenv("RTW_TEST_FAULT");
    if (!e || !*e) return;
    const size_t n = strlen(where);
    if (strncmp(e, where, n) != 0 || e[n] != ':') return;
    if (strcmp(e + n + 1, "bad_alloc") == 0) throw std::bad_alloc();
    throw std::runtime_error(std::string("injected fault at ") + where);
}
template <class F>
int guarded(rtw_ctx* c, F&& f) noexcept {
    try {
        test_fault("entry");
        return f();
    } catch (const std::bad_alloc&) {
        return fail_nothrow(c, RTW_ERR_OOM, "out of host memory (std::bad_alloc)");
    } catch (const std::exception& e) {
        return fail_nothrow(c, RTW_ERR_DEVICE, e.what());
    } catch (...) {
        return fail_nothrow(c, RTW_ERR_DEVICE, "unknown exception");
    }
}

int impl_destroy(rtw_ctx* c);
int impl_render_device(rtw_ctx* c, const rtw_params* P, void* d_rgba, void* hip_stream, rtw_stats* stats);
void worker_main(rtw_ctx* kc);

int create_single(rtw_ctx** out, int device) {
    rtw_ctx* c = new (std::nothrow) rtw_ctx();
    if (!c) return RTW_ERR_OOM;
    c->device = device;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return RTW_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    *out = c;
    return RTW_OK;
}

int impl_create(rtw_ctx** out, int n_devices, const int* device_ids) {
    if (n_devices < 1 || n_devices > 64) return RTW_ERR_INVALID_ARG;
    rtw_ctx* c = nullptr;
    int rc = create_single(&c, device_ids ? device_ids[0] : 0);
    if (rc) return rc;
    if (n_devices > 1) {
        // a group: one single-device context per entry of device_ids (entries may repeat: two shards on one GPU)
        for (int g = 0; g < n_devices; g++) {
            rtw_ctx* k = nullptr;
            rc = create_single(&k, device_ids ? device_ids[g] : g);
            if (rc) { impl_destroy(c); return rc; }
            try {
                c->kids.push_back(k);
            } catch (...) { impl_destroy(k); impl_destroy(c); throw; }
            // the kid's host thread, for the lifetime of the group (std::thread's constructor may throw std::system_error:
            // the caller's guard turns that into an error code once the half-built group is gone)
            try {
                k->worker = new Worker();
                k->worker->th = std::thread(worker_main, k);
            } catch (...) { impl_destroy(c); throw; }
            if (k->device != c->device) {  // direct peer copies for the gather where the link allows them; not an error if not
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, c->device, k->device) == hipSuccess && can) {
                    (void)hipSetDevice(c->device);
                    (void)hipDeviceEnablePeerAccess(k->device, 0);
                    (void)hipGetLastError();  // hipErrorPeerAccessAlreadyEnabled is fine
                }
            }
        }
        (void)hipSetDevice(c->device);
    }
    *out = c;
    return RTW_OK;
}

int impl_destroy(rtw_ctx* c) {
    if (!c) return RTW_ERR_INVALID_ARG;
    for (rtw_ctx* k : c->kids) (void)impl_destroy(k);
    c->kids.clear();
    if (c->worker) {
        if (c->worker->th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(c->worker->m);
                c->worker->quit = true;
            }
            c->worker->cv.notify_all();
            c->worker->th.join();
        }
        delete c->worker;
        c->worker = nullptr;
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_pool(c);
    for (auto& L : c->lane) {
        if (L.ev_done) (void)hipEventDestroy(L.ev_done);
        if (L.ev_free) (void)hipEventDestroy(L.ev_free);
        if (L.st) (void)hipStreamDestroy(L.st);
    }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->accum) (void)hipFree(c->accum);
    if (c->part) (void)hipFree(c->part);
    if (c->upart) (void)hipFree(c->upart);
    if (c->blocksum) (void)hipFree(c->blocksum);
    if (c->d_queue) (void)hipFree(c->d_queue);
    if (c->d_order) (void)hipFree(c->d_order);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stage) (void)hipFree(c->stage);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_scene) (void)hipFree(c->d_scene);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RTW_OK;
}

int impl_upload_scene(rtw_ctx* c, const void* blob, size_t bytes) {
    if (!c) return RTW_ERR_INVALID_ARG;
    if (!c->kids.empty()) {  // group: every device gets its own copy of the scene tables and the tree
        c->has_scene = false;
        for (rtw_ctx* k : c->kids) {
            const int rc = impl_upload_scene(k, blob, bytes);
            if (rc) return fail(c, rc, k->err);
        }
        c->has_scene = true;
        return RTW_OK;
    }
    if (!blob || bytes < sizeof(rtw_scene_header)) return fail(c, RTW_ERR_BAD_SCENE, "scene blob too small");
    rtw_scene_header h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != RTW_SCENE_MAGIC || h.version != RTW_SCENE_VERSION || h.total_bytes > bytes)
        return fail(c, RTW_ERR_BAD_SCENE, "bad scene header (magic/version/size)");
    auto in_range = [&](uint32_t off, uint32_t n, size_t sz) { return (size_t)off + (size_t)n * sz <= bytes; };
    if (!in_range(h.off_prims, h.n_prims, sizeof(rtw_prim)) || !in_range(h.off_xforms, h.n_xforms, sizeof(rtw_xform)) ||
        !in_range(h.off_materials, h.n_materials, sizeof(rtw_material)) || !in_range(h.off_textures, h.n_textures, sizeof(rtw_texture)) ||
        !in_range(h.off_lights, h.n_lights, sizeof(rtw_light)) || h.n_xforms < 1)
        return fail(c, RTW_ERR_BAD_SCENE, "scene table out of range");
    if (h.camera_type < RTW_CAM_PERSPECTIVE || h.camera_type > RTW_CAM_ORTHOGRAPHIC) return fail(c, RTW_ERR_BAD_SCENE, "unknown camera type");
    if ((h.off_prims | h.off_xforms | h.off_materials | h.off_textures | h.off_lights | h.off_texdata) & 15u)
        return fail(c, RTW_ERR_BAD_SCENE, "scene table not 16-byte aligned");
    const char* b = (const char*)blob;
    std::vector<rtw_prim> prims(h.n_prims);
    std::vector<rtw_xform> xforms(h.n_xforms);
    std::vector<rtw_material> mats(h.n_materials);
    std::vector<rtw_texture> texs(h.n_textures);
    std::vector<rtw_light> lights(h.n_lights);
    if (h.n_prims) memcpy(prims.data(), b + h.off_prims, h.n_prims * sizeof(rtw_prim));
    memcpy(xforms.data(), b + h.off_xforms, h.n_xforms * sizeof(rtw_xform));
    if (h.n_materials) memcpy(mats.data(), b + h.off_materials, h.n_materials * sizeof(rtw_material));
    if (h.n_textures) memcpy(texs.data(), b + h.off_textures, h.n_textures * sizeof(rtw_texture));
    if (h.n_lights) memcpy(lights.data(), b + h.off_lights, h.n_lights * sizeof(rtw_light));

    for (const rtw_xform& x : xforms)
        for (int k = 0; k < 12; k++) if (!std::isfinite(x.m[k]) || !std::isfinite(x.inv[k])) return fail(c, RTW_ERR_BAD_SCENE, "transform not finite");
    for (const rtw_material& m : mats)
        if (m.texture >= (int32_t)h.n_textures) return fail(c, RTW_ERR_BAD_SCENE, "material texture out of range");
    // texture data section and texture records (the test-side checker applies the same rules)
    std::vector<uint32_t> texdata;
    if (h.off_texdata) {
        if ((h.off_texdata & 3u) || (size_t)h.off_texdata + (size_t)h.texdata_bytes > bytes) return fail(c, RTW_ERR_BAD_SCENE, "texture data section out of range");
        texdata.resize(h.texdata_bytes / 4u);
        if (!texdata.empty()) memcpy(texdata.data(), b + h.off_texdata, texdata.size() * 4u);
    }
    for (uint32_t i = 0; i < h.n_textures; i++) {
        const rtw_texture& t = texs[i];
        if (t.type == RTW_TEX_CHECKER) {
            if (t.odd < 0 || t.even < 0 || (uint32_t)t.odd >= h.n_textures || (uint32_t)t.even >= h.n_textures ||
                texs[t.odd].type == RTW_TEX_CHECKER || texs[t.even].type == RTW_TEX_CHECKER)
                return fail(c, RTW_ERR_BAD_SCENE, "checker texture children out of range or nested");
        } else if (t.type == RTW_TEX_NOISE) {
            if ((size_t)t.data + 1536u > texdata.size()) return fail(c, RTW_ERR_BAD_SCENE, "noise texture tables out of range");
        } else if (t.type == RTW_TEX_IMAGE) {
            if ((size_t)t.data + 2u > texdata.size()) return fail(c, RTW_ERR_BAD_SCENE, "image texture out of range");
            const uint32_t iw = texdata[t.data], ih = texdata[t.data + 1];
            if (iw == 0 || ih == 0 || iw > 32768u || ih > 32768u || (size_t)t.data + 2u + (size_t)iw * ih > texdata.size())
                return fail(c, RTW_ERR_BAD_SCENE, "image texture out of range");
        } else if (t.type != RTW_TEX_CONSTANT && t.type != RTW_TEX_NULL) {
            return fail(c, RTW_ERR_BAD_SCENE, "unknown texture type");
        }
    }
    // RTW_EST_CORRECTED: light definitions moved onto the emitting rectangles they describe, and which primitives those
    // are (same matching rule as the CPU checker: same normal axis and in-plane extent, plane within 1 % of the longer edge)
    std::vector<rtw_light> clights(lights);
    std::vector<uint8_t> listed(h.n_prims, 0);
    for (size_t i = 0; i < clights.size(); i++) {
        rtw_light& lt = clights[i];
        for (uint32_t j = 0; j < h.n_prims; j++) {
            const rtw_prim& pr = prims[j];
            if (pr.type < RTW_PRIM_RECT_X || pr.type > RTW_PRIM_RECT_Z || pr.xform != 0) continue;
            if (pr.material < 0 || (uint32_t)pr.material >= h.n_materials || mats[pr.material].type != RTW_MAT_DIFFUSE_LIGHT) continue;
            const int ax = pr.type - RTW_PRIM_RECT_X, aa = ax == 0 ? 1 : 0, ab = ax == 2 ? 1 : 2;
            const float ea = pr.p[1] - pr.p[0], eb = pr.p[3] - pr.p[2];
            float u[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
            u[aa] = ea; v[ab] = eb;
            bool same = lt.position[aa] == pr.p[0] && lt.position[ab] == pr.p[2];
            for (int k = 0; k < 3; k++) if (lt.vec_u[k] != u[k] || lt.vec_v[k] != v[k]) same = false;
            if (!same || !(std::fabs(lt.position[ax] - pr.p[4]) <= 0.01f * std::fmax(ea, eb))) continue;
            lt.position[ax] = pr.p[4];
            listed[j] = 1;
            break;
        }
    }
    std::vector<HitRec> shade(h.n_prims);
    std::vector<int32_t> order;
    int has_motion = 0, has_tex = 0;
    for (uint32_t i = 0; i < h.n_prims; i++) {
        const rtw_prim& p = prims[i];
        if (p.type < RTW_PRIM_SPHERE || p.type > RTW_PRIM_VOLUME_SPHERE) return fail(c, RTW_ERR_BAD_SCENE, "unknown primitive type");
        if (p.xform < 0 || (uint32_t)p.xform >= h.n_xforms) return fail(c, RTW_ERR_BAD_SCENE, "primitive xform out of range");
        if (p.material < 0 || (uint32_t)p.material >= h.n_materials) return fail(c, RTW_ERR_BAD_SCENE, "primitive material out of range");
        for (int k = 0; k < 12; k++) if (!std::isfinite(p.p[k])) return fail(c, RTW_ERR_BAD_SCENE, "primitive parameter not finite");
        if (p.type == RTW_PRIM_MOVING_SPHERE) has_motion = 1;
        const rtw_material& m = mats[p.material];
        HitRec s{};
        s.mat_type = m.type; s.bsdf_eval = m.bsdf_eval; s.param = m.fuzz_or_eta; s.xform = p.xform;
        {
            // shading-normal data, same fp32 operations and order as the device/oracle would use per hit
            const rtw_xform& xf = xforms[p.xform];
            auto xfn = [&](float nx, float ny, float nz, float* o3) {
                float v[3] = {std::fmaf(xf.inv[0], nx, std::fmaf(xf.inv[4], ny, xf.inv[8] * nz)),
                              std::fmaf(xf.inv[1], nx, std::fmaf(xf.inv[5], ny, xf.inv[9] * nz)),
                              std::fmaf(xf.inv[2], nx, std::fmaf(xf.inv[6], ny, xf.inv[10] * nz))};
                float dd = std::fmaf(v[2], v[2], std::fmaf(v[1], v[1], v[0] * v[0]));
                float inv = 1.0f / std::sqrt(dd);
                o3[0] = v[0] * inv; o3[1] = v[1] * inv; o3[2] = v[2] * inv;
            };
            if (p.type == RTW_PRIM_SPHERE || p.type == RTW_PRIM_MOVING_SPHERE) {
                s.kind = p.type == RTW_PRIM_MOVING_SPHERE ? HK_MOVING_SPHERE : (p.xform != 0 ? HK_SPHERE_XFORM : HK_SPHERE);
                s.nx = p.p[0]; s.ny = p.p[1]; s.nz = p.p[2];
                s.inv_r = 1.0f / p.p[3];
            } else {
                s.kind = HK_CONST_NORMAL;
                float n[3] = {0.f, 0.f, 0.f};
                if (p.type == RTW_PRIM_RECT_X) n[0] = 1.f;
                else if (p.type == RTW_PRIM_RECT_Y) n[1] = 1.f;
                else if (p.type == RTW_PRIM_RECT_Z) n[2] = 1.f;
                else n[0] = 1.f;  // volumes report (1,0,0)
                if (p.flip && !rtwbvh::is_volume(p.type)) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
                if (p.xform != 0) { float w[3]; xfn(n[0], n[1], n[2], w); n[0] = w[0]; n[1] = w[1]; n[2] = w[2]; }
                s.nx = n[0]; s.ny = n[1]; s.nz = n[2];
                // onb::buildFromW (lib/onb.cuh:20-32) with the device's fp32 operation order
                auto nrm = [](const float* a3, float* o3) {
                    float dd = std::fmaf(a3[2], a3[2], std::fmaf(a3[1], a3[1], a3[0] * a3[0]));
                    float inv = 1.0f / std::sqrt(dd);
                    o3[0] = a3[0] * inv; o3[1] = a3[1] * inv; o3[2] = a3[2] * inv;
                };
                auto crs = [](const float* a3, const float* b3, float* o3) {
                    o3[0] = std::fmaf(a3[1], b3[2], -(a3[2] * b3[1]));
                    o3[1] = std::fmaf(a3[2], b3[0], -(a3[0] * b3[2]));
                    o3[2] = std::fmaf(a3[0], b3[1], -(a3[1] * b3[0]));
                };
                float w3[3], a3[3], t3[3], v3_[3], u3[3];
                nrm(n, w3);
                if (w3[0] > 0.9f || w3[0] < -0.9f) { a3[0] = 0.f; a3[1] = 1.f; a3[2] = 0.f; } else { a3[0] = 1.f; a3[1] = 0.f; a3[2] = 0.f; }
                crs(w3, a3, t3);
                nrm(t3, v3_);
                crs(w3, v3_, u3);
                s.ux = u3[0]; s.uy = u3[1]; s.uz = u3[2];
                s.vx = v3_[0]; s.vy = v3_[1]; s.vz = v3_[2];
                s.wx = w3[0]; s.wy = w3[1]; s.wz = w3[2];
            }
        }
        if (m.texture >= 0) {
            if ((uint32_t)m.texture >= h.n_textures) return fail(c, RTW_ERR_BAD_SCENE, "material texture out of range");
            const rtw_texture& t = texs[m.texture];
            if (t.type == RTW_TEX_CONSTANT) { s.r = t.color[0]; s.g = t.color[1]; s.b = t.color[2]; }
            else if (t.type != RTW_TEX_NULL) { s.kind |= (m.texture + 1) << 8; has_tex = 1; }  // checker / noise / image: evaluated per hit
        }
        if (listed[i]) s.kind |= 0x80;
        shade[i] = s;
    }
    const Tuning tune = read_tuning();
    const bool use_bvh = (int)h.n_prims > tune.brute_max;
    // order[]: volumes (index order), then -- small scenes only -- the moving spheres, which keep the generic test
    for (uint32_t i = 0; i < h.n_prims; i++) if (rtwbvh::is_volume(prims[i].type)) order.push_back((int32_t)i);
    const int n_vol = (int)order.size();
    if (!use_bvh)
        for (uint32_t i = 0; i < h.n_prims; i++) if (prims[i].type == RTW_PRIM_MOVING_SPHERE) order.push_back((int32_t)i);
    const int n_generic = (int)order.size() - n_vol;

    // small scenes: regroup the remaining primitives by instance transform, rectangles by axis
    std::vector<BruteGroup> groups;
    std::vector<BruteRec> recs;
    if (!use_bvh) {
        std::vector<int> xf_seen;
        for (uint32_t i = 0; i < h.n_prims; i++) {
            const int t = prims[i].type;
            if (rtwbvh::is_volume(t) || t == RTW_PRIM_MOVING_SPHERE) continue;
            if (std::find(xf_seen.begin(), xf_seen.end(), prims[i].xform) == xf_seen.end()) xf_seen.push_back(prims[i].xform);
        }
        for (int xf : xf_seen) {
            BruteGroup g{};
            g.xform = xf;
            g.first = (int32_t)recs.size();
            const int kinds[4] = {RTW_PRIM_RECT_X, RTW_PRIM_RECT_Y, RTW_PRIM_RECT_Z, RTW_PRIM_SPHERE};
            int32_t* counts[4] = {&g.n_rx, &g.n_ry, &g.n_rz, &g.n_sph};
            for (int k = 0; k < 4; k++)
                for (uint32_t i = 0; i < h.n_prims; i++) {
                    const rtw_prim& p = prims[i];
                    if (p.type != kinds[k] || p.xform != xf) continue;
                    BruteRec r{};
                    if (k < 3) { r.a = p.p[0]; r.b = p.p[1]; r.c = p.p[2]; r.d = p.p[3]; r.e = p.p[4]; }
                    else { r.a = p.p[0]; r.b = p.p[1]; r.c = p.p[2]; r.d = p.p[3]; }
                    r.prim = (int32_t)i;
                    recs.push_back(r);
                    (*counts[k])++;
                }
            groups.push_back(g);
        }
    }
    // k_path's LDS image of the same lists (rtw_device.h walk_lds): groups, their world->object matrices, records
    std::vector<uint32_t> walk;
    if (!use_bvh && n_generic == 0 && !groups.empty()) {
        static_assert(sizeof(BruteGroup) == 32 && sizeof(BruteRec) == 32, "walk image layout");
        const size_t ng = groups.size();
        walk.resize((5 * ng + 2 * recs.size() + 4) * 4, 0u);  // + 4 words: the reader fetches up to two records ahead
        memcpy(walk.data(), groups.data(), ng * sizeof(BruteGroup));
        for (size_t g = 0; g < ng; g++) memcpy(walk.data() + (2 * ng + 3 * g) * 4, xforms[groups[g].xform].inv, 12 * sizeof(float));
        if (!recs.empty()) memcpy(walk.data() + 5 * ng * 4, recs.data(), recs.size() * sizeof(BruteRec));
        for (size_t i = 0; i < recs.size(); i++) walk[(5 * ng + 2 * i) * 4 + 5] = (uint32_t)recs[i].prim + 1u;  // the walk's tie key
        // (a rectangle with lo > hi can never be hit under either form of the range test, so none needs removing)
        if (walk.size() / 4 > (size_t)kWalkMaxWords) walk.clear();
    }
    rtwbvh::Bvh bvh;
    if (use_bvh) {
        bvh = rtwbvh::build_bvh(prims.data(), h.n_prims, xforms.data());
        if (bvh.stack_need > 95) return fail(c, RTW_ERR_UNSUPPORTED, "tree deeper than the LDS traversal stack");
        if (bvh.max_exp > 60) return fail(c, RTW_ERR_UNSUPPORTED, "scene extent beyond 1e20");
    }

    // one device allocation, 256-byte aligned sub-tables
    auto al = [](size_t v) { return (v + 255u) & ~size_t(255); };
    size_t o_prims = 0;
    size_t o_xf = al(o_prims + prims.size() * sizeof(rtw_prim));
    size_t o_shade = al(o_xf + xforms.size() * sizeof(rtw_xform));
    size_t o_lights = al(o_shade + shade.size() * sizeof(HitRec));
    size_t o_clights = al(o_lights + std::max<size_t>(1, lights.size()) * sizeof(rtw_light));
    size_t o_nodes = al(o_clights + std::max<size_t>(1, lights.size()) * sizeof(rtw_light));
    // the wave-coherent walk's nodes: fp32 child boxes, padded for its plane arithmetic. It computes a plane's distance as
    // fma(plane, 1/d, -(o * 1/d)): the rounding of o * 1/d displaces a plane by about ulp(|o|) in space, so the boxes grow by
    // 2^-19 of the largest coordinate around (scene bounds, camera origin; the primitives' own bounds carry 1e-4 already)
    if (use_bvh) {
        float big = 1.0f;
        for (uint32_t i = 0; i < h.n_prims; i++) {
            const rtwbvh::Box wb = rtwbvh::world_bounds(prims[i], xforms[prims[i].xform]);
            for (int a = 0; a < 3; a++) big = std::max(big, std::max(std::fabs(wb.mn[a]), std::fabs(wb.mx[a])));
        }
        for (int a = 0; a < 3; a++) big = std::max(big, std::fabs(h.camera.origin[a]) + std::fabs(h.camera.lens_radius));
        const float pad = 1.0e-4f + big * (1.0f / 524288.0f);
        for (rtwbvh::WNode& w : bvh.wq4)
            for (int k = 0; k < 4; k++)
                if (w.ref[k] != rtwbvh::kQ4Empty)
                    for (int a = 0; a < 3; a++) { w.box[k][a] -= pad; w.box[k][3 + a] += pad; }
    }
    size_t o_wnodes = al(o_nodes + std::max<size_t>(1, bvh.q4.size()) * sizeof(rtwbvh::Q4Node));
    size_t o_tree = al(o_wnodes + std::max<size_t>(1, bvh.wq4.size()) * sizeof(rtwbvh::WNode));
    size_t o_order = al(o_tree + std::max<size_t>(1, bvh.leaves.size()) * sizeof(rtwbvh::LeafRec));
    size_t o_groups = al(o_order + std::max<size_t>(1, order.size()) * sizeof(int32_t));
    size_t o_recs = al(o_groups + std::max<size_t>(1, groups.size()) * sizeof(BruteGroup));
    size_t o_texs = al(o_recs + (recs.size() + 1) * sizeof(BruteRec));  // + 1: traverse_brute reads one record ahead
    size_t o_texdata = al(o_texs + std::max<size_t>(1, texs.size()) * sizeof(rtw_texture));
    size_t o_walk = al(o_texdata + std::max<size_t>(1, texdata.size()) * sizeof(uint32_t));
    size_t total = al(o_walk + std::max<size_t>(1, walk.size()) * sizeof(uint32_t));
    std::vector<char> stage(total, 0);
    if (!prims.empty()) memcpy(stage.data() + o_prims, prims.data(), prims.size() * sizeof(rtw_prim));
    memcpy(stage.data() + o_xf, xforms.data(), xforms.size() * sizeof(rtw_xform));
    if (!shade.empty()) memcpy(stage.data() + o_shade, shade.data(), shade.size() * sizeof(HitRec));
    if (!lights.empty()) memcpy(stage.data() + o_lights, lights.data(), lights.size() * sizeof(rtw_light));
    if (!clights.empty()) memcpy(stage.data() + o_clights, clights.data(), clights.size() * sizeof(rtw_light));
    if (!bvh.q4.empty()) memcpy(stage.data() + o_nodes, bvh.q4.data(), bvh.q4.size() * sizeof(rtwbvh::Q4Node));
    if (!bvh.wq4.empty()) memcpy(stage.data() + o_wnodes, bvh.wq4.data(), bvh.wq4.size() * sizeof(rtwbvh::WNode));
    if (!bvh.leaves.empty()) memcpy(stage.data() + o_tree, bvh.leaves.data(), bvh.leaves.size() * sizeof(rtwbvh::LeafRec));
    if (!order.empty()) memcpy(stage.data() + o_order, order.data(), order.size() * sizeof(int32_t));
    if (!groups.empty()) memcpy(stage.data() + o_groups, groups.data(), groups.size() * sizeof(BruteGroup));
    if (!recs.empty()) memcpy(stage.data() + o_recs, recs.data(), recs.size() * sizeof(BruteRec));
    if (!texs.empty()) memcpy(stage.data() + o_texs, texs.data(), texs.size() * sizeof(rtw_texture));
    if (!texdata.empty()) memcpy(stage.data() + o_texdata, texdata.data(), texdata.size() * sizeof(uint32_t));
    if (!walk.empty()) memcpy(stage.data() + o_walk, walk.data(), walk.size() * sizeof(uint32_t));

    test_fault("upload");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_scene) { (void)hipFree(c->d_scene); c->d_scene = nullptr; }
    c->has_scene = false;
    HIP_TRY(c, hipMalloc(&c->d_scene, total));
    HIP_TRY(c, hipMemcpy(c->d_scene, stage.data(), total, hipMemcpyHostToDevice));
    char* d = (char*)c->d_scene;
    DScene sc{};
    sc.prims = (const rtw_prim*)(d + o_prims);
    sc.xforms = (const rtw_xform*)(d + o_xf);
    sc.hitrec = (const HitRec*)(d + o_shade);
    sc.lights = (const rtw_light*)(d + o_lights);
    sc.clights = (const rtw_light*)(d + o_clights);
    sc.estimator = RTW_EST_REFERENCE; sc.ray_tmin = 1e-6f; sc.probe_eps = 500 * 1.0e-7f;  // set per render
    sc.nodes = (const u32x4*)(d + o_nodes);
    sc.leaves = (const u32x4*)(d + o_tree);
    sc.wnodes = (const u32x4*)(d + o_wnodes);
    sc.order = (const int32_t*)(d + o_order);
    sc.groups = (const BruteGroup*)(d + o_groups);
    sc.recs = (const BruteRec*)(d + o_recs);
    sc.texs = (const rtw_texture*)(d + o_texs);
    sc.texdata = (const uint32_t*)(d + o_texdata);
    sc.walk = (const u32x4*)(d + o_walk);
    sc.n_walk_words = (int32_t)(walk.size() / 4);
    sc.noise_lds_data = -1;  // the first noise texture some primitive shows gets its tables staged in LDS
    for (uint32_t i = 0; i < h.n_prims && sc.noise_lds_data < 0; i++) {
        int ti = mats[prims[i].material].texture;
        if (ti < 0) continue;
        if (texs[ti].type == RTW_TEX_CHECKER) ti = texs[texs[ti].odd].type == RTW_TEX_NOISE ? texs[ti].odd : texs[ti].even;
        if (texs[ti].type == RTW_TEX_NOISE) sc.noise_lds_data = (int32_t)texs[ti].data;
    }
    // selects the kernel instantiations that contain the cold features: textures, media, (k_path) moving spheres in the brute lists,
    // camera kinds other than the reference's lens-free perspective camera
    sc.has_tex = (has_tex || n_vol > 0 || n_generic > 0 || h.camera_type != RTW_CAM_PERSPECTIVE || h.camera.lens_radius != 0.0f) ? 1 : 0;
    sc.n_groups = (int)groups.size();
    sc.n_generic = n_generic;
    sc.n_prims = (int)h.n_prims;
    sc.n_vol = n_vol;
    sc.n_tree = (int)bvh.prim_order.size();
    sc.n_lights = (int)h.n_lights;
    {   // generous bounds of the whole scene (volumes and motion sweeps included), padded by 1 % of the diagonal
        rtwbvh::Box all;
        for (uint32_t i = 0; i < h.n_prims; i++) all.add(rtwbvh::world_bounds(prims[i], xforms[prims[i].xform]));
        float diag = 0.f;
        for (int a = 0; a < 3; a++) diag += (all.mx[a] - all.mn[a]) * (all.mx[a] - all.mn[a]);
        const float pad = 0.01f * std::sqrt(diag) + 1.0f;
        for (int a = 0; a < 3; a++) { sc.bmin[a] = all.mn[a] - pad; sc.bmax[a] = all.mx[a] + pad; }
    }
    sc.sky_light = h.sky_light;
    sc.use_bvh = use_bvh ? 1 : 0;
    sc.has_motion = has_motion;
    sc.cam = h.camera;
    sc.pdf = h.pdf;
    sc.cam_type = h.camera_type;
    // LDS per block: the traversal stacks (16-bit entries when every reference fits), then as many leading (breadth-first)
    // tree nodes and, behind them, leaf records as fit the budget
    c->stack_depth = use_bvh ? bvh.stack_need + 2 : 0;  // + the two rows under the stack that end a walk
    sc.stack_depth = c->stack_depth;
    sc.n_lds_nodes = 0; sc.n_lds_leaves = 0; sc.stack_wide = 0;
    c->lds_bytes = 0;
    if (use_bvh) {
        sc.stack_wide = (std::max(bvh.q4.size(), (size_t)bvh.n_slots) << 2) >= 0x7ff0u ? 1 : 0;  // 16-bit entries are read sign-extended
        c->n_tree_nodes = bvh.q4.size(); c->n_tree_leaves = bvh.n_slots;
        sc.n_nodes = (int32_t)bvh.q4.size();
        c->lds_bytes = tree_lds_layout(c->n_tree_nodes, c->n_tree_leaves, c->stack_depth, sc.stack_wide != 0, kBlock, tune.lds_kb * 1024, sc.n_lds_nodes, sc.n_lds_leaves);
        if (tune.verbose) fprintf(stderr, "[rtw] tree (SAH bins %d, collapse %d, sample-walk cost %.3f): %zu nodes, %zu leaf records, stack %d x %d bit; LDS %zu B: %d nodes, %d leaf records\n", bvh.bins, bvh.collapse_kind, bvh.cost,
                                  bvh.q4.size(), (size_t)bvh.n_slots, c->stack_depth, sc.stack_wide ? 32 : 16, c->lds_bytes, sc.n_lds_nodes, sc.n_lds_leaves);
    }
    c->sc = sc;
    c->has_scene = true;
    return RTW_OK;
}

int check_render_args(rtw_ctx* c, const rtw_params* P) {
    if (!c->has_scene) return fail(c, RTW_ERR_NO_SCENE, "rtw_render before rtw_upload_scene");
    if (!P) return fail(c, RTW_ERR_INVALID_ARG, "null params");
    if (P->width <= 0 || P->height <= 0 || P->spp <= 0 || P->max_depth < 0 || P->row0 < 0 || P->row1 > P->height || P->row0 > P->row1)
        return fail(c, RTW_ERR_INVALID_ARG, "bad render params");
    if (P->rng_kind != RTW_RNG_PHILOX && P->rng_kind != RTW_RNG_TEA_LCG) return fail(c, RTW_ERR_INVALID_ARG, "bad rng_kind");
    if (P->sample_offset < 0 || P->samples_per_pass < 0 || P->row_stride < 0) return fail(c, RTW_ERR_INVALID_ARG, "bad sample_offset/samples_per_pass/row_stride");
    if (P->estimator < RTW_EST_REFERENCE || P->estimator > RTW_EST_MIXTURE) return fail(c, RTW_ERR_INVALID_ARG, "bad estimator");
    return RTW_OK;
}

size_t shard_rows(const rtw_params* P) {
    const size_t k = P->row_stride > 1 ? (size_t)P->row_stride : 1;
    return ((size_t)(P->row1 - P->row0) + k - 1) / k;
}

// One device: the whole render of the shard P describes, result in d_rgba (device memory of c->device).
int render_single(rtw_ctx* c, const rtw_params* P, void* d_rgba, hipStream_t s, rtw_stats* stats) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (!s) s = c->stream;

    const uint32_t row_stride = P->row_stride > 1 ? (uint32_t)P->row_stride : 1u;
    const size_t rows = shard_rows(P);
    const size_t npix = rows * (size_t)P->width;
    if (stats) memset(stats, 0, sizeof *stats);
    if (npix == 0) return RTW_OK;
    if (npix > 0xffffffffull / 2) return fail(c, RTW_ERR_UNSUPPORTED, "tile too large");

    const Tuning tune = read_tuning();
    const bool timing = stats != nullptr && tune.kernel_timing;
    // timing events come from a pool kept in the context: (kernel kind, start, stop) triples of this call
    size_t ev_used = 0;
    auto new_event = [&](hipEvent_t& e) -> hipError_t {
        if (ev_used == c->ev_pool.size()) {
            hipEvent_t n = nullptr;
            hipError_t er = hipEventCreate(&n);
            if (er != hipSuccess) return er;
            c->ev_pool.push_back(n);
        }
        e = c->ev_pool[ev_used++];
        return hipSuccess;
    };
    struct Timed { int kind; hipEvent_t a, b; };
    std::vector<Timed> ev_k;
    auto timed_launch = [&](hipStream_t ls, int kind, const KArgs& ka, int grid_, size_t lds_, int block_ = kBlock) -> hipError_t {
        if (!timing) {
            launch(kind, P->rng_kind, ka, grid_, lds_, ls, block_);
            return hipSuccess;
        }
        Timed t{kind == LK_PATH_TREE ? (int)RTW_K_PATH : kind, nullptr, nullptr};
        hipError_t er = new_event(t.a);
        if (er == hipSuccess) er = new_event(t.b);
        if (er == hipSuccess) er = hipEventRecord(t.a, ls);
        if (er != hipSuccess) return er;
        launch(kind, P->rng_kind, ka, grid_, lds_, ls, block_);
        ev_k.push_back(t);
        return hipEventRecord(t.b, ls);
    };
#define HIP_TRY_C(expr) HIP_TRY(c, expr)
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    HIP_TRY_C(new_event(ev_begin));
    HIP_TRY_C(new_event(ev_end));

    // kernel arguments common to both pipelines
    KArgs base{};
    base.sc = c->sc;
    if (P->estimator != RTW_EST_REFERENCE) {  // the corrected estimators live in the cold-feature instantiations
        base.sc.estimator = P->estimator; base.sc.has_tex = P->estimator == RTW_EST_MIXTURE ? 2 : 1;
        base.sc.ray_tmin = 1.0e-3f; base.sc.probe_eps = 1.0e-3f;
    }
    base.npix = (uint32_t)npix;
    base.width = (uint32_t)P->width;
    base.height = (uint32_t)P->height;
    base.row0 = (uint32_t)P->row0;
    base.row_stride = row_stride;
    magic_div((uint32_t)P->width, base.divw_m, base.divw_s1, base.divw_s2);
    magic_div(row_stride, base.divs_m, base.divs_s1, base.divs_s2);
    base.seed = P->seed;
    base.max_depth = (uint32_t)P->max_depth;
    base.stack_stride = kBlock;
    base.spp = (uint32_t)P->spp;
    const size_t lds = c->lds_bytes;
    const unsigned pix_grid = (unsigned)std::min<size_t>((npix + kBlock - 1) / kBlock, (size_t)c->n_cu * 8);
    uint64_t launches = 0;
    const bool path_small = !c->sc.use_bvh && c->sc.n_prims <= kPathMaxPrims && (c->sc.n_walk_words > 0 || c->sc.has_tex);
#ifdef RTW_EXPERIMENTS
    const bool path_tree = c->sc.use_bvh && tune.path_tree != 0;
#else
    const bool path_tree = false;  // (k_path_tree exists in -DRTW_EXPERIMENTS builds only: RTW_PATH_TREE is ignored here)
#endif
    // k_path packs a unit's pixel as x | y << 16: frames wider or taller than 65535 take the wavefront kernels
    const bool fits16 = P->width <= 65535 && P->height <= 65535;
    const bool use_path = P->max_depth > 0 && tune.path != 0 && fits16 && (path_small || path_tree);

    if (use_path) {
        // ---- k_path: paths in registers, lanes regenerate; only the unit sums (16 B per pixel and 64 samples) reach HBM
        int rc = ensure_pool(c, 0, 0, npix, 0);
        if (rc) return rc;
        const size_t n_blocks = ((size_t)P->spp + kSumBlock - 1) / kSumBlock;
        int wg_per_cu = tune.path_grid_mult;
        const size_t path_lds = path_tree ? lds : 0;
        if (wg_per_cu <= 0) {
            const bool lcg = P->rng_kind == RTW_RNG_TEA_LCG;
            const int feat = base.sc.has_tex;
            int nb = 0;
            hipError_t qe;
#define RTW_OCC_R(K_, R_) (feat == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, K_<R_, 2>, kBlock, path_lds)   \
                           : feat == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, K_<R_, 1>, kBlock, path_lds) \
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, K_<R_, 0>, kBlock, path_lds))
#define RTW_OCC(K_) (lcg ? RTW_OCC_R(K_, RTW_RNG_TEA_LCG) : RTW_OCC_R(K_, RTW_RNG_PHILOX))
#ifdef RTW_EXPERIMENTS
            if (path_tree) qe = RTW_OCC(k_path_tree); else
#endif
            if (feat == 1 && base.sc.n_vol > 0)
                qe = lcg ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_path<RTW_RNG_TEA_LCG, 1, 1>, kBlock, path_lds)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_path<RTW_RNG_PHILOX, 1, 1>, kBlock, path_lds);
            else
                qe = RTW_OCC(k_path);
#undef RTW_OCC
#undef RTW_OCC_R
            wg_per_cu = (qe == hipSuccess && nb > 0) ? std::min(nb, 8) : 4;
        }
        const size_t n_groups = (npix + 63) / 64;
        if (n_groups > c->order_groups) {
            if (c->d_order) (void)hipFree(c->d_order);
            c->d_order = nullptr; c->order_groups = 0;
            HIP_TRY(c, hipMalloc(&c->d_order, 3 * n_groups * sizeof(uint32_t)));
            c->order_groups = n_groups;
        }
        if (!c->stream2) HIP_TRY(c, create_stream(&c->stream2, 2));  // low priority: it fills the slots the bulk launch vacates
        HIP_TRY_C(hipEventRecord(ev_begin, s));
        HIP_TRY_C(hipMemsetAsync(c->accum, 0, npix * sizeof(float4), s));
        HIP_TRY_C(hipMemsetAsync(c->d_stats, 0, (kStatRows + 1) * 8 * sizeof(unsigned long long), s));
        {   // job order: longest units first (k_classify)
            HIP_TRY_C(hipMemsetAsync(c->d_queue, 0, 64, s));
            KArgs a = base;
            const dim3 cg((unsigned)std::min<size_t>((n_groups + 3) / 4, (size_t)c->n_cu * 8));
            if (path_small && c->sc.n_walk_words > 0) hipLaunchKernelGGL(k_classify<true>, cg, dim3(kBlock), 0, s, a, c->d_order, c->d_queue + 1, (uint32_t)n_groups);
            else hipLaunchKernelGGL(k_classify<false>, cg, dim3(kBlock), 0, s, a, c->d_order, c->d_queue + 1, (uint32_t)n_groups);
        }
        // A launch ends when its slowest unit ends, and a unit through a glass sphere runs several milliseconds. So the bulk of
        // a pass is handed out in units of `unit_blocks` blocks (one lane keeps a pixel for 64 samples: little bookkeeping), and
        // its last `fine_blocks` blocks in single-block units by a SECOND launch on a second stream: its workgroups move into the
        // slots the first launch's workgroups vacate as they run dry, so the machine stays full until only 16-sample units are
        // left (measured on the 1/8 shard of the metric frame: see DESIGN.md section 6).
        // Unit size: every unit costs a little (queue, camera-ray set-up, a 16-byte store per block either way) and a launch
        // ends with its longest units, so long renders want long units and short ones short units. Measured on the metric
        // frame (1 620 blocks per lane): 8-block units 0.556 s, 4-block 0.562 s, 2-block 0.581 s; on its 1/8 shard (202
        // blocks per lane): 0.0773, 0.0722, 0.0736 s; on the 1/2 shard 8 and 4 are level.
        const size_t blocks_per_lane = npix * n_blocks / ((size_t)c->n_cu * (size_t)wg_per_cu * kBlock);
        const size_t U = tune.path_unit_blocks > 0 ? (size_t)tune.path_unit_blocks : (blocks_per_lane >= 600 ? 8 : 4);
        // the end-game region: the last 8 blocks of a pass behind 8-block units, the last 16 behind 4-block units (a shard-sized
        // render: the bulk launch drains for a unit's length at its end, and the single-block work beside it must last that long;
        // 1/8 shard of the metric frame, medians of 12 runs: F = 8 0.0739 s, 16 0.0725, 24 0.0728, 32 0.0727, 48 0.0729; the full
        // frame does not care: 0.5597 against 0.5593)
        const size_t F = tune.path_fine_blocks >= 0 ? (size_t)tune.path_fine_blocks : (U >= 8 ? 8 : 16);
        // Sums in memory (the arithmetic spec's three levels, rtw.h): a bulk launch whose lane units are whole summation units
        // (U a multiple of 8 blocks) stores ONE float4 per unit and pixel, everything else one per block; k_resolve_blocks adds
        // them up in the spec's order. A pass covers a multiple of 8 blocks, so no summation unit straddles two passes.
        const bool unit_sums = !path_tree && (U % kSumUnitBlocks) == 0;
        auto coarse_of = [&](size_t nb) { return nb > 4 * F ? ((nb - F) / U) * U : (size_t)0; };  // short passes are all fine units
        auto slots_of = [&](size_t nb) { const size_t nc = coarse_of(nb); return unit_sums ? nc / kSumUnitBlocks + (nb - nc) : nb; };
        const size_t cap_slots = std::max<size_t>(1, tune.blocksum_bytes / (npix * sizeof(float4)));
        size_t pass_blocks = n_blocks;
        if (slots_of(n_blocks) > cap_slots) {
            pass_blocks = kSumUnitBlocks;
            while (pass_blocks + kSumUnitBlocks < n_blocks && slots_of(pass_blocks + kSumUnitBlocks) <= cap_slots) pass_blocks += kSumUnitBlocks;
        }
        size_t need_slots = 0;
        for (size_t b0 = 0; b0 < n_blocks; b0 += pass_blocks) need_slots = std::max(need_slots, slots_of(std::min(pass_blocks, n_blocks - b0)));
        if (need_slots * npix > c->blocksum_elems) {
            if (c->blocksum) (void)hipFree(c->blocksum);
            c->blocksum = nullptr; c->blocksum_elems = 0;
            HIP_TRY(c, hipMalloc(&c->blocksum, need_slots * npix * sizeof(float4)));
            c->blocksum_elems = need_slots * npix;
        }
        for (size_t b0 = 0; b0 < n_blocks; b0 += pass_blocks) {
            const size_t nb = std::min(pass_blocks, n_blocks - b0);
            const size_t nb_coarse = coarse_of(nb);
            const size_t slots_coarse = unit_sums ? nb_coarse / kSumUnitBlocks : nb_coarse;
            HIP_TRY_C(hipMemsetAsync(c->d_queue, 0, 4, s));
            HIP_TRY_C(hipMemsetAsync(c->d_queue + 4, 0, 4, s));
            hipEvent_t ev_a = nullptr, ev_b = nullptr;  // (from the call's event pool, like the wavefront lanes' start event)
            HIP_TRY_C(new_event(ev_a));
            HIP_TRY_C(new_event(ev_b));
            HIP_TRY_C(hipEventRecord(ev_a, s));
            // the two launches of a pass overlap, so they are timed as one: from before the first to after both (on s, which waits
            // for the second stream's launch below); rocprofv3 lists them as two dispatches whose durations both span the pass
            Timed tp{(int)RTW_K_PATH, nullptr, nullptr};
            if (timing) {
                HIP_TRY_C(new_event(tp.a));
                HIP_TRY_C(new_event(tp.b));
                HIP_TRY_C(hipEventRecord(tp.a, s));
            }
            for (int part = 0; part < 2; part++) {
                const size_t first = part == 0 ? 0 : nb_coarse, count = part == 0 ? nb_coarse : nb - nb_coarse;
                if (count == 0) continue;
                const size_t ub = part == 0 ? U : 1;
                const size_t n_units = (count + ub - 1) / ub;                     // units per pixel in this launch
                const size_t jb = std::min<size_t>((size_t)tune.path_job_blocks, n_units);  // units per pixel and job
                const size_t n_ranges = (n_units + jb - 1) / jb;
                const size_t n_jobs = n_groups * n_ranges;
                if (n_jobs > 0xfffffff0ull) return fail(c, RTW_ERR_UNSUPPORTED, "too many k_path jobs");
                KArgs a = base;
                a.stats = c->d_stats;
                a.sample0 = (uint32_t)P->sample_offset;
                a.queue = c->d_queue + (part == 0 ? 0 : 4);
                a.order = c->d_order;
                a.order_counts = c->d_queue + 1;
                a.blocksum = c->blocksum + (part == 0 ? 0 : slots_coarse) * npix;
                a.unit_sums = (part == 0 && unit_sums) ? 1u : 0u;
                a.n_jobs = (uint32_t)n_jobs; a.n_ranges = (uint32_t)n_ranges; a.units_per_job = (uint32_t)jb;
                a.block0 = (uint32_t)(b0 + first); a.n_blocks_pass = (uint32_t)count; a.unit_blocks = (uint32_t)ub;
                const int grid = (int)std::min<size_t>((size_t)c->n_cu * (size_t)wg_per_cu, (n_jobs + 3) / 4);
                hipStream_t ls = part == 0 ? s : c->stream2;
                if (part == 1) HIP_TRY_C(hipStreamWaitEvent(ls, ev_a, 0));
                launch(path_tree ? LK_PATH_TREE : LK_PATH, P->rng_kind, a, grid, path_lds, ls);
                launches++;
                if (part == 1) {
                    HIP_TRY_C(hipEventRecord(ev_b, ls));
                    HIP_TRY_C(hipStreamWaitEvent(s, ev_b, 0));
                }
            }
            if (timing) {
                HIP_TRY_C(hipEventRecord(tp.b, s));
                ev_k.push_back(tp);
            }
            // coarse region: whole unit sums (unit_sums) or block sums from block b0 on; fine region: block sums from b0 + nb_coarse on
            if (unit_sums)
                hipLaunchKernelGGL(k_resolve_blocks, dim3(pix_grid), dim3(kBlock), 0, s, (const float4*)c->blocksum, c->accum, (uint32_t)npix, (uint32_t)slots_coarse,
                                   (uint32_t)(nb - nb_coarse), (uint32_t)(b0 + nb_coarse));
            else
                hipLaunchKernelGGL(k_resolve_blocks, dim3(pix_grid), dim3(kBlock), 0, s, (const float4*)c->blocksum, c->accum, (uint32_t)npix, 0u, (uint32_t)nb, (uint32_t)b0);
        }
        hipLaunchKernelGGL(k_finish, dim3(pix_grid), dim3(kBlock), 0, s, (const float4*)c->accum, (const float4*)nullptr, (const float4*)nullptr, (float4*)d_rgba, (uint32_t)npix, (float)P->spp);
    } else {
    // ---- wavefront pipeline (tree scenes; RTW_PATH=0)
    // samples per pass: keep about pool_target paths in flight, split over the lanes
    const int want_lanes = tune.lanes;
    size_t S = P->samples_per_pass > 0 ? (size_t)P->samples_per_pass
                                       : std::max<size_t>(1, std::min(tune.pool_paths, c->pool_cap) / (size_t)want_lanes / npix);
    S = std::min<size_t>(S, (size_t)P->spp);
    while (S > 1 && npix * S > 0xfffffff0ull) S--;
    if (P->samples_per_pass <= 0 && want_lanes > 1 && (size_t)P->spp >= (size_t)want_lanes) {
        // equal batches, as many as a multiple of the lanes: every lane gets the same number of batches of the same size (a
        // render of 128 spp whose pool would take it in one batch would leave the second lane idle; 512 spp in batches of 129
        // would end with a short fourth one). Sizes are kept multiples of 16 where that fits (k_first's sample grouping).
        size_t nb = ((size_t)P->spp + S - 1) / S;
        nb = (nb + (size_t)want_lanes - 1) / (size_t)want_lanes * (size_t)want_lanes;
        size_t s_eq = ((size_t)P->spp + nb - 1) / nb;
        if (((s_eq + 15) & ~(size_t)15) <= S) s_eq = (s_eq + 15) & ~(size_t)15;
        S = std::max<size_t>(1, std::min(S, s_eq));
    }
    const size_t paths_max = npix * S;
    // Persistent compacting grid: G workgroups (8 per CU when a lane has the GPU to itself, 4 when two lanes share it).
    // Output region b belongs to workgroup b, which is handed every G-th 256-path chunk of its input: at most
    // ceil(chunks / G) + 1 chunks (the work list of a later launch has up to one partial chunk per region more than
    // the first), so a region of (ceil(chunks / G) + 2) * 256 slots cannot overflow.
    const size_t n_batches = ((size_t)P->spp + S - 1) / S;
    const int n_lanes = (int)std::min<size_t>((size_t)want_lanes, std::max<size_t>(n_batches, 1));
    const uint32_t grid_mult = tune.grid_mult > 0 ? (uint32_t)tune.grid_mult : (n_lanes > 1 ? 4u : 8u);
    auto grid_for = [&](size_t paths) {
        const size_t chunks = (paths + kBlock - 1) / kBlock;
        return (uint32_t)std::min<size_t>(std::min<size_t>(chunks, (size_t)c->n_cu * grid_mult), (size_t)kMaxRegions);
    };
    auto cap_for = [&](size_t paths) {
        const size_t chunks = (paths + kBlock - 1) / kBlock;
        const size_t g = grid_for(paths);
        return ((chunks + g - 1) / g + 2) * (size_t)kBlock;
    };
    // RTW_PAIRED: batches run two at a time with their trace launches alternating (see the batch loop)
#ifdef RTW_EXPERIMENTS
    const bool paired = tune.paired && n_lanes == 2 && c->sc.use_bvh;
#else
    const bool paired = false;  // (measured 12 % slower: -DRTW_EXPERIMENTS builds only)
#endif
    // k_trace_bvh: large workgroups share one LDS copy of the tree (nodes, then leaf records) between more waves
    int trace_block = tune.trace_block;
    size_t trace_budget = tune.trace_lds_kb * 1024;
    // LDS the trace launch may plan with per CU: all 160 KB when the knobs say so; 142 KB when the render chooses - the other
    // lane's kernels (k_shade: 9 KB per workgroup) must find room beside a resident k_trace_bvh, or the two lanes take turns
    size_t trace_cu_lds = (size_t)160 * 1024;
    if (c->sc.use_bvh && tune.trace_auto) {
        // Every node of the tree in the workgroup's LDS image takes the global loads - and the vmcnt waits behind them - out of the
        // walk loop (k_trace_bvh mode 2), but only pays while the kernel keeps its waves AND leaves the other lane room: a 256-thread
        // workgroup at 6 per CU has 14 KB for stacks + nodes (scene 1's stacks alone: 21 levels = 10.75 KB), a 512-thread workgroup
        // shares one image between twice the waves. Measured (round 3, scene 1: 240 nodes = 15.4 KB; k_trace_bvh per 512-spp render):
        // 256 threads / 16 KB (88 nodes in LDS, the old default) 0.152-0.155 s; 256 / 26 KB (all nodes, 4 waves per SIMD) 0.154;
        // 512 / 37 KB (all nodes, 3 workgroups = 141 KB per CU) 0.144-0.147; 512 / 40.5 KB (152 KB per CU: no room left for the
        // other lane's workgroups) 0.155; 512 / 43 KB (three planned, two fit) 0.166; 512 / 44 KB (two planned) 0.143-0.145.
        // Scene 2: 0.163 -> 0.153. Scene 4 (1 419 nodes = 91 KB) stays at 256 / 16 KB.
        const size_t cu_lds = (size_t)142 * 1024;
        const size_t blocks[2] = {256, 512};
        for (size_t blk : blocks) {
            int32_t nn = 0, nl = 0;
            const size_t want = std::max<size_t>(1, (size_t)(4 * tune.trace_waves) / (blk / 64));  // workgroups per CU at the wanted occupancy
            const size_t fixed = (kMaxRegions + 1 + blk) * 4 + 64;                                // the work list's static LDS
            // stacks + every node, no leaf records (a partial leaf image makes a wave run both of leaf_test's fetch paths)
            const size_t need = tree_lds_layout(c->n_tree_nodes, 0, c->stack_depth, c->sc.stack_wide != 0, blk, (size_t)150 * 1024, nn, nl);
            if ((size_t)nn != c->n_tree_nodes) continue;
            const size_t fit = cu_lds / (need + fixed);
            if (fit >= want || (blk == 512 && fit >= 2)) { trace_block = (int)blk; trace_budget = need; trace_cu_lds = cu_lds; break; }
        }
    }
    int32_t trace_nodes = 0, trace_leaves = 0;
    size_t trace_lds = 0;
    int trace_grid = 0;
    if (c->sc.use_bvh) {
        trace_lds = tree_lds_layout(c->n_tree_nodes, c->n_tree_leaves, c->stack_depth, c->sc.stack_wide != 0, (size_t)trace_block, trace_budget,
                                    trace_nodes, trace_leaves);
        const size_t per_wg = trace_lds + (kMaxRegions + 1 + (size_t)trace_block) * 4 + 64;
        const size_t by_lds = std::max<size_t>(1, trace_cu_lds / per_wg);
        const size_t by_waves = std::max<size_t>(1, (size_t)(4 * (paired ? tune.paired_trace_waves : tune.trace_waves)) / ((size_t)trace_block / 64));
        trace_grid = (int)((size_t)c->n_cu * std::min(by_lds, by_waves));
        if (trace_lds > 48 * 1024) {  // beyond the default dynamic-LDS limit of a launch
            DScene ts = c->sc;
            ts.n_lds_nodes = trace_nodes;
            const void* f = (const void*)trace_bvh_kernel(trace_block, trace_bvh_mode(ts));
            HIP_TRY(c, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)trace_lds));
        }
    }
    const uint32_t regions_max = grid_for(paths_max);
    const size_t region_cap_max = cap_for(paths_max);
    bool split_first = false;
    // Launch schedule of one batch. Wide bounces: one k_shade + one k_trace per bounce (split pipeline; scenes
    // whose intersection programs draw random numbers keep trace and shade fused in k_bounce instead).
    // Thin tail: k_bounce with several bounces in registers, in growing groups.
    struct Step { int kind, depth, n_iter; };
    std::vector<Step> sched;
    {
        // the fused tail kernel walks the tree one lane per path (lane utilisation 0.2): tree scenes stay in the split
        // pipeline longer, and longest where media keep many paths alive deep (scene 4, 3.9 segments per sample: 20 -> 40
        // +5 %; scenes 1 and 2, 2.6 and 3.2: best at 20, -2 % at 30)
        const int tail_start = tune.tail_start > 0 ? tune.tail_start : (c->sc.use_bvh ? (c->sc.n_vol > 0 ? 40 : 20) : 6);
        const bool split = !tune.fused && (c->sc.n_vol == 0 || tune.split_media);
        int d = 0, grp = tune.tail_group, rep = 0;
        while (d < P->max_depth) {
            if (d < tail_start) {
                if (split) {
                    // k_first has already traced and shaded depth 0
                    if (d > 0) { sched.push_back({LK_TRACE, d, 1}); sched.push_back({LK_SHADE, d, 1}); }
                } else {
                    sched.push_back({LK_BOUNCE, d, 1});
                }
                d++;
            } else {
                const int n = std::min(grp, P->max_depth - d);
                if (++rep == 2) { rep = 0; grp += grp / 2; }
                sched.push_back({LK_BOUNCE, d, n});
                d += n;
            }
        }
        // a batch must not end with probes still queued: a zero-bounce k_bounce resolves them and retires the zombies
        if (split && (sched.empty() || sched.back().kind == LK_SHADE)) sched.push_back({LK_BOUNCE, P->max_depth, 0});
        split_first = split;
    }
    const size_t cnt_words = (size_t)regions_max * (sched.size() + 2);
    int rc = ensure_pool(c, n_lanes, (size_t)regions_max * region_cap_max, npix, cnt_words, c->sc.n_lights > 0);
    if (rc == RTW_ERR_OOM && P->samples_per_pass <= 0 && S > 1) {
        // The pool is sized for an MI355X to itself (2^29 paths: up to 120 GiB). A device with less to give - another process on
        // it, a smaller part - gets half as many paths in flight, and half again, until the allocation fits: smaller batches,
        // the same image (a path's draws and a pixel's summation order do not depend on the batch size).
        free_pool(c);
        (void)hipGetLastError();
        c->pool_cap = std::max<size_t>((size_t)want_lanes * npix, npix * S * (size_t)want_lanes / 2);
        return render_single(c, P, d_rgba, s, stats);
    }
    if (rc) return rc;

    HIP_TRY_C(hipEventRecord(ev_begin, s));
    HIP_TRY_C(hipMemsetAsync(c->accum, 0, npix * sizeof(float4), s));
    HIP_TRY_C(hipMemsetAsync(c->part, 0, npix * sizeof(float4), s));
    HIP_TRY_C(hipMemsetAsync(c->upart, 0, npix * sizeof(float4), s));
    HIP_TRY_C(hipMemsetAsync(c->d_stats, 0, (kStatRows + 1) * 8 * sizeof(unsigned long long), s));

    if (P->max_depth > 0) {
        // the lanes start once the accumulators are cleared
        hipEvent_t ev_ready = nullptr;
        HIP_TRY_C(new_event(ev_ready));
        HIP_TRY_C(hipEventRecord(ev_ready, s));
        size_t bi = 0;
        // The second lane's first batch is cut short so that the lanes run half a batch apart: one lane's bandwidth-bound
        // k_shade launches then meet the other's issue-bound k_first / k_trace instead of its own kind (5 runs each on one
        // box: 9.35-9.58 Gsamples/s with the offset, 8.93-9.59 without).
        const int stagger_pct = tune.stagger_pct >= 0 ? tune.stagger_pct : (c->sc.use_bvh ? 0 : 50);
        // One batch in flight on a lane: its arguments and which of the lane's two path buffers is current.
        struct BatchRun { rtw_ctx::Lane* L; KArgs a; int cur; size_t ci; uint32_t regions; size_t Sb, s0; };
        auto batch_size = [&](size_t b, size_t s0_) {
            size_t want = S;
            if (stagger_pct > 0 && b > 0 && b < (size_t)n_lanes && S > 1)  // lane k starts k/n_lanes of a batch late (at 50 %)
                want = std::max<size_t>(1, S - S * b * (size_t)stagger_pct * 2 / (100 * (size_t)n_lanes));
            return std::min(want, (size_t)P->spp - s0_);
        };
        // camera rays (+ the primary segment) of a batch on its lane
        auto begin_batch = [&](size_t b, size_t s0_, size_t Sb_, BatchRun& R) -> hipError_t {
            R.L = &c->lane[b % (size_t)n_lanes]; R.Sb = Sb_; R.s0 = s0_;
            rtw_ctx::Lane& L = *R.L;
            const size_t paths = npix * Sb_;
            R.regions = grid_for(paths);
            // this lane's pool is free again once the resolve of its previous batch has run on the main stream
            hipError_t er = hipStreamWaitEvent(L.st, b < (size_t)n_lanes ? ev_ready : L.ev_free, 0);
            if (er == hipSuccess) er = hipMemsetAsync(L.cnt, 0, (size_t)R.regions * (sched.size() + 2) * sizeof(uint32_t), L.st);
            if (er != hipSuccess) return er;
            KArgs& a = R.a;
            a = base;
            a.lbuf = L.lbuf;
            a.stats = c->d_stats;
            a.n_regions = R.regions;
            a.n_paths = (uint32_t)paths;
            a.sample0 = (uint32_t)(P->sample_offset + (int)s0_);
            a.region_cap = (uint32_t)cap_for(paths);
            a.trace_first = split_first ? 1u : 0u;
            a.first_group_log2 = 0;
            while (a.first_group_log2 < (uint32_t)tune.first_group_log2 && (Sb_ >> (a.first_group_log2 + 1)) << (a.first_group_log2 + 1) == Sb_) a.first_group_log2++;
            // k_first fills buffer 0 (and the hit buffer); every compacting launch then flips the buffers
            R.cur = 0;
            R.ci = 0;  // index of the region-counter row describing buffer `cur`
            a.out = L.buf[0];
            a.hit_out = L.hit[0];
            a.cnt_out = L.cnt;
            a.depth = 0; a.n_iter = 1;
            launches++;
            return timed_launch(L.st, LK_FIRST, a, (int)R.regions, c->sc.use_bvh ? (size_t)(kBlock / 64) * (size_t)c->stack_depth * sizeof(uint32_t) : 0);  // every compacting launch uses exactly this grid: workgroup b owns region b
        };
        // step si of the schedule for a batch; a trace launch may wait for an event of the other lane and record one
        auto run_step = [&](BatchRun& R, size_t si, hipEvent_t wait_for, hipEvent_t record) -> hipError_t {
            rtw_ctx::Lane& L = *R.L;
            KArgs& a = R.a;
            const Step& st = sched[si];
            const int grid = (int)R.regions;
            a.in = L.buf[R.cur];
            a.hit = L.hit[R.cur];
            a.hit_out = L.hit[R.cur];  // k_trace fills the records of the buffer it reads
            a.cnt_in = L.cnt + R.ci * R.regions;
            a.depth = (uint32_t)st.depth;
            a.n_iter = (uint32_t)st.n_iter;
            hipError_t er = hipSuccess;
            if (wait_for) er = hipStreamWaitEvent(L.st, wait_for, 0);
            if (er != hipSuccess) return er;
            if (st.kind == LK_TRACE) {
                if (c->sc.use_bvh) {  // its own workgroup size, LDS image and grid: waves own streams of chunks, not regions
                    KArgs at = a;
                    at.sc.n_lds_nodes = trace_nodes; at.sc.n_lds_leaves = trace_leaves;
                    er = timed_launch(L.st, LK_TRACE, at, trace_grid, trace_lds, trace_block);
                } else {
                    er = timed_launch(L.st, LK_TRACE, a, grid, lds);
                }
            } else {
                a.out = L.buf[R.cur ^ 1];
                a.hit_out = L.hit[R.cur ^ 1];
                a.cnt_out = L.cnt + (R.ci + 1) * R.regions;
                er = timed_launch(L.st, st.kind, a, grid, st.kind == LK_BOUNCE ? lds : 0);
                R.cur ^= 1;
                R.ci++;
            }
            launches++;
            if (er == hipSuccess && record) er = hipEventRecord(record, L.st);
            return er;
        };
        auto end_batch = [&](BatchRun& R) -> hipError_t {
            rtw_ctx::Lane& L = *R.L;
            hipError_t er = hipEventRecord(L.ev_done, L.st);
            // batches are resolved into the accumulators in order, on the main stream
            if (er == hipSuccess) er = hipStreamWaitEvent(s, L.ev_done, 0);
            if (er != hipSuccess) return er;
            hipLaunchKernelGGL(k_resolve, dim3(pix_grid), dim3(kBlock), 0, s, (const float4*)L.lbuf, c->accum, c->upart, c->part, (uint32_t)npix, (uint32_t)R.Sb, (uint32_t)R.s0);
            return hipEventRecord(L.ev_free, s);
        };
        for (size_t s0 = 0; s0 < (size_t)P->spp;) {
            BatchRun A;
            const size_t SbA = batch_size(bi, s0);
            HIP_TRY_C(begin_batch(bi, s0, SbA, A));
            const size_t s1 = s0 + SbA;
            if (paired && s1 < (size_t)P->spp) {
                // Two batches in step: the trace launches of the pair never run at the same time - each starts when the
                // other batch's trace launch of the same depth (or of the depth before) has ended - so an issue-bound
                // k_trace_bvh always has a bandwidth-bound k_shade of the other batch beside it and not its own kind.
                BatchRun B;
                const size_t SbB = batch_size(bi + 1, s1);
                HIP_TRY_C(begin_batch(bi + 1, s1, SbB, B));
                hipEvent_t ev_prev = nullptr;
                for (size_t si = 0; si < sched.size(); si++) {
                    if (sched[si].kind == LK_TRACE) {
                        hipEvent_t ea = nullptr, eb = nullptr;
                        HIP_TRY_C(new_event(ea));
                        HIP_TRY_C(new_event(eb));
                        HIP_TRY_C(run_step(A, si, ev_prev, ea));
                        HIP_TRY_C(run_step(B, si, ea, eb));
                        ev_prev = eb;
                    } else {
                        HIP_TRY_C(run_step(A, si, nullptr, nullptr));
                        HIP_TRY_C(run_step(B, si, nullptr, nullptr));
                    }
                }
                HIP_TRY_C(end_batch(A));
                HIP_TRY_C(end_batch(B));
                s0 = s1 + SbB;
                bi += 2;
            } else {
                for (size_t si = 0; si < sched.size(); si++) HIP_TRY_C(run_step(A, si, nullptr, nullptr));
                HIP_TRY_C(end_batch(A));
                s0 = s1;
                bi++;
            }
        }
    }
    hipLaunchKernelGGL(k_finish, dim3(pix_grid), dim3(kBlock), 0, s, (const float4*)c->accum, (const float4*)c->upart, (const float4*)c->part, (float4*)d_rgba, (uint32_t)npix, (float)P->spp);
    }
    HIP_TRY_C(hipGetLastError());
    HIP_TRY_C(hipEventRecord(ev_end, s));
    HIP_TRY_C(hipEventSynchronize(ev_end));

    unsigned long long hs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        unsigned long long rows_[kStatRows * 8];
        HIP_TRY_C(hipMemcpy(rows_, c->d_stats, sizeof rows_, hipMemcpyDeviceToHost));
        for (uint32_t r = 0; r < kStatRows; r++)
            for (int k = 0; k < 8; k++) hs[k] += rows_[r * 8 + k];
    }
#ifdef RTW_PHASE_TIMERS
    if (use_path) {
        unsigned long long ph[8];
        HIP_TRY_C(hipMemcpy(ph, c->d_stats + kStatRows * 8, sizeof ph, hipMemcpyDeviceToHost));
        double tot = 0;
        for (int q = 0; q < 6; q++) tot += (double)ph[q];
        const char* nm[6] = {"refill", "regen", "walk_r", "shade_a", "walk_s", "shade_b"};
        fprintf(stderr, "[rtw] k_path wave-cycles by phase:");
        for (int q = 0; q < 6; q++) fprintf(stderr, " %s %.1f%%", nm[q], 100.0 * (double)ph[q] / tot);
        fprintf(stderr, " (total %.3g wave-cycles, %.0f per 64 segments)\n", tot, tot / ((double)hs[0] / 64.0));
        if (path_tree) fprintf(stderr, "[rtw] k_path_tree wave steps per 64 segments: inner %.1f  leaf %.1f  shade %.2f\n", (double)ph[6] / ((double)hs[0] / 64.0),
                               (double)(ph[7] % 1000000ull) / ((double)hs[0] / 64.0), (double)(ph[7] / 1000000ull) / ((double)hs[0] / 64.0));
    }
#endif
#ifdef RTW_SUBPHASE_TIMERS
    if (use_path) {
        static std::vector<unsigned long long> tab((size_t)kSubWaves * kSubRows);
        HIP_TRY_C(hipMemcpyFromSymbol(tab.data(), HIP_SYMBOL(g_sub_cyc), tab.size() * sizeof(unsigned long long)));
        double sum[kSubRows] = {0};
        for (size_t w = 0; w < (size_t)kSubWaves; w++) for (int q = 0; q < 14; q++) sum[q] += (double)tab[w * kSubRows + q];
        double tot = 0;
        for (int q = 0; q < 14; q++) tot += sum[q];
        const char* nm[9] = {"outside", "hitrec+philox", "lambert", "light", "metal", "diel", "iso", "nee", "entry"};
        fprintf(stderr, "[rtw] k_path wave-cycles by sub-phase of the closest-hit program:");
        for (int q = 0; q < 9; q++) fprintf(stderr, " %s %.1f%%", nm[q], 100.0 * sum[q] / tot);
        fprintf(stderr, " (total %.3g)\n", tot);
        std::fill(tab.begin(), tab.end(), 0ull);
        HIP_TRY_C(hipMemcpyToSymbol(HIP_SYMBOL(g_sub_cyc), tab.data(), tab.size() * sizeof(unsigned long long)));
    }
#endif
#ifdef RTW_TRACE_COUNT
    {
        unsigned long long w[7];
        HIP_TRY_C(hipMemcpy(w, c->d_stats + kStatRows * 8, sizeof w, hipMemcpyDeviceToHost));
        const double rays = (double)w[6];
        fprintf(stderr, "[rtw] k_trace_bvh: rays %.4g; per ray: node visits %.2f, primitive tests %.2f; wave steps per 64 rays: inner %.2f (lanes busy %.2f), leaf %.2f (lanes busy %.2f), outer %.2f\n",
                rays, (double)hs[6] / rays, (double)hs[7] / rays, (double)w[0] * 64.0 / rays, (double)hs[6] / ((double)w[0] * 64.0), (double)w[1] * 64.0 / rays,
                (double)hs[7] / ((double)w[1] * 64.0), (double)hs[2 + RTW_K_BOUNCE] * 64.0 / rays);
        const double tt = (double)(w[2] + w[3] + w[4] + w[5]);
        fprintf(stderr, "[rtw] k_trace_bvh wave-cycles: refill %.1f%% inner %.1f%% leaf %.1f%% finish %.1f%% (%.0f per 64 rays; per inner wave step %.0f, per leaf wave step %.0f)\n",
                100.0 * (double)w[2] / tt, 100.0 * (double)w[3] / tt, 100.0 * (double)w[4] / tt, 100.0 * (double)w[5] / tt, tt * 64.0 / rays,
                (double)w[3] / (double)w[0], (double)w[4] / (double)w[1]);
    }
#endif
    if (stats) {
        float ms = 0.f;
        HIP_TRY_C(hipEventElapsedTime(&ms, ev_begin, ev_end));
        stats->seconds = (double)ms * 1e-3;
        stats->bounce_seconds = stats->seconds;  // the lanes overlap: the loop time is the elapsed time of the call
        for (const Timed& t : ev_k) {
            float m = 0.f;
            HIP_TRY_C(hipEventElapsedTime(&m, t.a, t.b));
            stats->kernel_seconds[t.kind] += (double)m * 1e-3;
            stats->kernel_launches[t.kind]++;
        }
        for (int k = 0; k < RTW_K_COUNT; k++) stats->kernel_segments[k] = hs[2 + k];
        stats->bounce_launches = launches;
        stats->samples = (uint64_t)npix * (uint64_t)P->spp;
        stats->segments = hs[0];
        stats->shadow_rays = hs[1];
        stats->algorithmic_bytes = 128ull * stats->segments + 32ull * stats->samples;
    }
#undef HIP_TRY_C
    return RTW_OK;
}

// rows of the gathered shards -> rows of the frame: frame row r is row r / n of shard r % n
__global__ void __launch_bounds__(256) k_interleave(const float4* __restrict__ stage, float4* __restrict__ out, uint32_t width, uint32_t rows, uint32_t n,
                                                    const uint32_t* __restrict__ shard_off) {
    const size_t total = (size_t)rows * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)r * width);
        const uint32_t g = r % n, l = r / n;
        out[i] = stage[((size_t)shard_off[g] + l) * width + x];
    }
}

// n_devices > 1: kid g renders rows row0 + g*k, row0 + (g + n)*k ... of the shard (k = the caller's row stride) on its own
// device from its own (persistent) host thread and then PUSHES its float4 shard to device_ids[0] with one
// hipMemcpyPeerAsync on its own stream - n concurrent transfers, one per inbound xGMI link of the gathering device (single
// process: a peer copy is the xGMI transfer an RCCL send/recv pair would issue) - and the rows are interleaved there.
int run_shard(rtw_ctx* kc, Worker& w) {
    test_fault("worker");
    if (hipSetDevice(kc->device) != hipSuccess) return fail(kc, RTW_ERR_DEVICE, "hipSetDevice failed");
    if (w.npix == 0) return RTW_OK;
    if (w.npix > kc->out_pix) {
        if (kc->d_out) (void)hipFree(kc->d_out);
        kc->d_out = nullptr; kc->out_pix = 0;
        if (hipMalloc(&kc->d_out, w.npix * sizeof(float4)) != hipSuccess) return fail(kc, RTW_ERR_OOM, "shard buffer");
        kc->out_pix = w.npix;
    }
    const int rc = render_single(kc, &w.P, kc->d_out, nullptr, w.want_stats ? &w.st : nullptr);
    if (rc) return rc;
    HIP_TRY(kc, hipMemcpyPeerAsync(w.gather_dst, w.gather_dev, kc->d_out, kc->device, w.npix * sizeof(float4), kc->stream));
    HIP_TRY(kc, hipStreamSynchronize(kc->stream));
    return RTW_OK;
}
void worker_main(rtw_ctx* kc) {
    Worker& w = *kc->worker;
    for (;;) {
        try {
            std::unique_lock<std::mutex> lk(w.m);
            w.cv.wait(lk, [&] { return w.has_job || w.quit; });
            if (w.quit) return;
            lk.unlock();
            int rc;
            try {
                rc = run_shard(kc, w);
            } catch (const std::bad_alloc&) {
                rc = fail_nothrow(kc, RTW_ERR_OOM, "out of host memory in a worker thread (std::bad_alloc)");
            } catch (const std::exception& e) {
                rc = fail_nothrow(kc, RTW_ERR_DEVICE, e.what());
            } catch (...) {
                rc = fail_nothrow(kc, RTW_ERR_DEVICE, "unknown exception in a worker thread");
            }
            lk.lock();
            w.rc = rc; w.has_job = false; w.done = true;
            lk.unlock();
            w.cv.notify_all();
        } catch (...) {
            // a failing lock: nothing sane is left to do with this thread's queue; report once and leave
            w.rc = RTW_ERR_DEVICE; w.has_job = false; w.done = true;
            w.cv.notify_all();
            return;
        }
    }
}
int render_group(rtw_ctx* c, const rtw_params* P, void* d_rgba, hipStream_t s, rtw_stats* stats) {
    const size_t n = c->kids.size();
    const size_t k = P->row_stride > 1 ? (size_t)P->row_stride : 1;
    const size_t rows = shard_rows(P);
    if (stats) memset(stats, 0, sizeof *stats);
    if (rows == 0) return RTW_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!s) s = c->stream;
    const size_t npix = rows * (size_t)P->width;
    if (npix > c->stage_pix) {
        if (c->stage) (void)hipFree(c->stage);
        c->stage = nullptr; c->stage_pix = 0;
        HIP_TRY(c, hipMalloc(&c->stage, npix * sizeof(float4) + (n + 1) * sizeof(uint32_t) + 256));
        c->stage_pix = npix;
    }
    std::vector<uint32_t> off(n + 1, 0);
    std::vector<size_t> krows(n, 0);
    for (size_t g = 0; g < n; g++) {
        krows[g] = rows > g ? (rows - g + n - 1) / n : 0;
        off[g + 1] = off[g] + (uint32_t)krows[g];
    }
    // hand every kid its shard, then wait for all of them (a failing or throwing kid reports through its rc)
    for (size_t g = 0; g < n; g++) {
        Worker& w = *c->kids[g]->worker;
        std::lock_guard<std::mutex> lk(w.m);
        w.P = *P;
        w.P.row0 = P->row0 + (int32_t)(g * k);
        w.P.row_stride = (int32_t)(k * n);
        if (w.P.row0 > w.P.row1) w.P.row0 = w.P.row1;
        w.npix = krows[g] * (size_t)P->width;
        w.gather_dst = c->stage + (size_t)off[g] * P->width;
        w.gather_dev = c->device;
        w.want_stats = stats != nullptr;
        memset(&w.st, 0, sizeof w.st);
        w.rc = RTW_OK; w.done = false; w.has_job = true;
    }
    for (size_t g = 0; g < n; g++) c->kids[g]->worker->cv.notify_all();
    int first_rc = RTW_OK;
    size_t first_bad = 0;
    for (size_t g = 0; g < n; g++) {
        Worker& w = *c->kids[g]->worker;
        std::unique_lock<std::mutex> lk(w.m);
        w.cv.wait(lk, [&] { return w.done; });
        if (w.rc != RTW_OK && first_rc == RTW_OK) { first_rc = w.rc; first_bad = g; }
    }
    if (first_rc != RTW_OK)
        return fail(c, first_rc, std::string("device ") + std::to_string(c->kids[first_bad]->device) + ": " + c->kids[first_bad]->err);
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t* d_off = (uint32_t*)((char*)c->stage + ((npix * sizeof(float4) + 255) & ~(size_t)255));
    HIP_TRY(c, hipMemcpyAsync(d_off, off.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_interleave, dim3((unsigned)std::min<size_t>((npix + 255) / 256, (size_t)c->n_cu * 8)), dim3(256), 0, s, (const float4*)c->stage,
                       (float4*)d_rgba, (uint32_t)P->width, (uint32_t)rows, (uint32_t)n, (const uint32_t*)d_off);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(s));
    if (stats) {
        for (size_t g = 0; g < n; g++) {
            const rtw_stats& ks = c->kids[g]->worker->st;
            stats->samples += ks.samples; stats->segments += ks.segments; stats->shadow_rays += ks.shadow_rays;
            stats->algorithmic_bytes += ks.algorithmic_bytes; stats->bounce_launches += ks.bounce_launches;
            stats->seconds = std::max(stats->seconds, ks.seconds);
            stats->bounce_seconds = std::max(stats->bounce_seconds, ks.bounce_seconds);
            for (int q = 0; q < RTW_K_COUNT; q++) {
                stats->kernel_seconds[q] += ks.kernel_seconds[q]; stats->kernel_launches[q] += ks.kernel_launches[q];
                stats->kernel_segments[q] += ks.kernel_segments[q];
            }
        }
    }
    return RTW_OK;
}

int impl_render_device(rtw_ctx* c, const rtw_params* P, void* d_rgba, void* hip_stream, rtw_stats* stats) {
    if (!c) return RTW_ERR_INVALID_ARG;
    int rc = check_render_args(c, P);
    if (rc) return rc;
    if (!d_rgba) return fail(c, RTW_ERR_INVALID_ARG, "null params or output");
    if (!c->kids.empty()) return render_group(c, P, d_rgba, (hipStream_t)hip_stream, stats);
    return render_single(c, P, d_rgba, (hipStream_t)hip_stream, stats);
}

int impl_render(rtw_ctx* c, const rtw_params* P, float* rgba_out, rtw_stats* stats) {
    if (!c) return RTW_ERR_INVALID_ARG;
    if (!rgba_out) return fail(c, RTW_ERR_INVALID_ARG, "null output");
    if (!P) return fail(c, RTW_ERR_INVALID_ARG, "null params");
    if (P->width <= 0 || P->row0 < 0 || P->row1 < P->row0) return fail(c, RTW_ERR_INVALID_ARG, "bad render params");
    const size_t npix = shard_rows(P) * (size_t)P->width;
    HIP_TRY(c, hipSetDevice(c->device));
    if (npix > c->out_pix) {
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_out = nullptr; c->out_pix = 0;
        HIP_TRY(c, hipMalloc(&c->d_out, std::max<size_t>(npix, 1) * sizeof(float4)));
        c->out_pix = npix;
    }
    int rc = impl_render_device(c, P, c->d_out, nullptr, stats);
    if (rc) return rc;
    if (npix) HIP_TRY(c, hipMemcpy(rgba_out, c->d_out, npix * sizeof(float4), hipMemcpyDeviceToHost));  // Director.cpp:999-1000
    return RTW_OK;
}

int impl_denoise(rtw_ctx* c, const float* rgba_in, float* rgba_out, int32_t width, int32_t height, int32_t iterations, float sigma) {
    if (!c) return RTW_ERR_INVALID_ARG;
    if (!rgba_in || !rgba_out || rgba_in == rgba_out || width <= 0 || height <= 0 || iterations < 1 || iterations > 8 || !(sigma > 0.f) ||
        (int64_t)width * height > (1 << 28))
        return fail(c, RTW_ERR_INVALID_ARG, "rtw_denoise: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)width * height * sizeof(float4);
    float4* d[2] = {nullptr, nullptr};
    auto cleanup = [&]() { for (float4* p : d) if (p) (void)hipFree(p); };
    for (int k = 0; k < 2; k++)
        if (hipMalloc(&d[k], bytes) != hipSuccess) { cleanup(); return fail(c, RTW_ERR_OOM, "rtw_denoise: device allocation failed"); }
    hipError_t e = hipMemcpyAsync(d[0], rgba_in, bytes, hipMemcpyHostToDevice, c->stream);
    const int n = width * height;
    const unsigned grid = (unsigned)std::min<int64_t>(((int64_t)n + kBlock - 1) / kBlock, (int64_t)c->n_cu * 8);
    int cur = 0;
    float s_i = sigma;
    for (int it = 0; it < iterations && e == hipSuccess; it++) {
        hipLaunchKernelGGL(k_atrous, dim3(grid), dim3(kBlock), 0, c->stream, (const float4*)d[cur], d[cur ^ 1], (int)width, (int)height, 1 << it,
                           1.0f / (s_i * s_i));
        e = hipGetLastError();
        cur ^= 1;
        s_i = s_i * 0.5f;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rgba_out, d[cur], bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, RTW_ERR_DEVICE, std::string("rtw_denoise: ") + hipGetErrorString(e));
    return RTW_OK;
}

int impl_debug_intersect(rtw_ctx* c, const float* rays, const float* ray_time, const float* gather_time, int n, float* out_t, int32_t* out_prim) {
    if (!c) return RTW_ERR_INVALID_ARG;
    if (!c->has_scene) return fail(c, RTW_ERR_NO_SCENE, "no scene");
    if (n < 0 || (n > 0 && (!rays || !out_t || !out_prim))) return fail(c, RTW_ERR_INVALID_ARG, "bad arguments");
    if (n == 0) return RTW_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float *d_rays = nullptr, *d_rt = nullptr, *d_gt = nullptr, *d_t = nullptr;
    int32_t* d_p = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_rays); (void)hipFree(d_rt); (void)hipFree(d_gt); (void)hipFree(d_t); (void)hipFree(d_p); };
#define HIP_TRY_D(expr)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) { cleanup(); return fail(c, RTW_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    HIP_TRY_D(hipMalloc(&d_rays, (size_t)n * 8 * sizeof(float)));
    HIP_TRY_D(hipMalloc(&d_t, (size_t)n * sizeof(float)));
    HIP_TRY_D(hipMalloc(&d_p, (size_t)n * sizeof(int32_t)));
    HIP_TRY_D(hipMemcpy(d_rays, rays, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice));
    if (ray_time) { HIP_TRY_D(hipMalloc(&d_rt, (size_t)n * sizeof(float))); HIP_TRY_D(hipMemcpy(d_rt, ray_time, (size_t)n * sizeof(float), hipMemcpyHostToDevice)); }
    if (gather_time) { HIP_TRY_D(hipMalloc(&d_gt, (size_t)n * sizeof(float))); HIP_TRY_D(hipMemcpy(d_gt, gather_time, (size_t)n * sizeof(float), hipMemcpyHostToDevice)); }
    const size_t lds = c->lds_bytes;
    hipLaunchKernelGGL(k_debug_intersect, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), lds, c->stream, c->sc, (const float*)d_rays,
                       (const float*)d_rt, (const float*)d_gt, n, d_t, d_p, (uint32_t)kBlock);
    HIP_TRY_D(hipGetLastError());
    HIP_TRY_D(hipStreamSynchronize(c->stream));
    HIP_TRY_D(hipMemcpy(out_t, d_t, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY_D(hipMemcpy(out_prim, d_p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    cleanup();
#undef HIP_TRY_D
    return RTW_OK;
}

}  // namespace

extern "C" {

int rtw_abi_version(void) { return RTW_ABI_VERSION; }

const char* rtw_last_error(rtw_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rtw_create(rtw_ctx** out, int n_devices, const int* device_ids) {
    if (!out) return RTW_ERR_INVALID_ARG;
    *out = nullptr;
    return guarded(nullptr, [&] { return impl_create(out, n_devices, device_ids); });
}
int rtw_destroy(rtw_ctx* c) { return guarded(nullptr, [&] { return impl_destroy(c); }); }
int rtw_upload_scene(rtw_ctx* c, const void* blob, size_t bytes) { return guarded(c, [&] { return impl_upload_scene(c, blob, bytes); }); }
int rtw_render_device(rtw_ctx* c, const rtw_params* P, void* d_rgba, void* hip_stream, rtw_stats* stats) {
    return guarded(c, [&] { return impl_render_device(c, P, d_rgba, hip_stream, stats); });
}
int rtw_render(rtw_ctx* c, const rtw_params* P, float* rgba_out, rtw_stats* stats) { return guarded(c, [&] { return impl_render(c, P, rgba_out, stats); }); }
int rtw_denoise(rtw_ctx* c, const float* rgba_in, float* rgba_out, int32_t width, int32_t height, int32_t iterations, float sigma) {
    return guarded(c, [&] { return impl_denoise(c, rgba_in, rgba_out, width, height, iterations, sigma); });
}
int rtw_debug_intersect(rtw_ctx* c, const float* rays, const float* ray_time, const float* gather_time, int n, float* out_t, int32_t* out_prim) {
    return guarded(c, [&] { return impl_debug_intersect(c, rays, ray_time, gather_time, n, out_t, out_prim); });
}

}  // extern "C"
