// rtw_kernels.h — the kernels (gfx950). See rtw_hip.hip for the two pipelines and their launch schedules.
//
//   k_path       small scenes: whole paths in registers, lanes own (pixel, sample block) units and regenerate; LDS candidate
//                lists and hit records; only 16-byte unit sums reach HBM
//   k_path_tree  the same for tree scenes, lanes vote on the kind of step (slower than the wavefront kernels: -DRTW_EXPERIMENTS builds only)
//   k_first      generate primary rays (raygen.cu:123-147, camera.cu:11-19), trace and shade the primary segment
//   k_shade      closest-hit / miss programs for one bounce of every live path: (media: the volume pass), material
//                scatter, texture, light sampling (the shadow ray is QUEUED in the path state, not traced), Russian
//                roulette, wave64 ballot/popcount compaction of the survivors into the workgroup's own output region
//   k_trace<>    surfaces only, small scenes under RTW_PATH=0: the radiance ray's closest hit and the queued shadow ray's
//                any-hit in one shared walk over the scalar-cache candidate lists (or one traverse<> each with moving spheres)
//   k_trace_bvh  surfaces only, tree scenes: waves own streams of chunks, idle lanes refill, majority-vote stepping
//   k_bounce     fused trace+shade(+inline shadow probe) for several bounces in registers: the thin tail of a batch
//   k_resolve / k_resolve_blocks / k_finish   deterministic per-pixel sums in the spec's blocked order, mean radiance
//
// Path state: six 16-byte SoA planes per path (96 B), read and written with dwordx4 accesses that
// are contiguous across a wave:
//   p0 = origin.xyz, dir.x      p1 = dir.yz, ray_time, gk (24-bit gather-time fraction | zombie bit 31)
//   p2 = shadow dir.xyz, shadow tmax (<0: none queued)
//   p3 = T.xyz, L.x             p4 = L.yz, c.xy          p5 = c.z, w0, rng a, rng b
// (scenes without listed lights queue no probes: p2 is not used, p4 = L.yz, w0, rng a, and rng b alone takes p5's place as a
// plane of dwords: 68 bytes per path)
// c is the queued light-sample contribution (already multiplied by the throughput of its segment); it
// is added to L by the next k_shade if the shadow probe found no occluder, before anything else
// touches L, so the floating-point order of the reference's `sampleRadiance += radiance*throughput`
// (raygen.cu:60) is preserved bit for bit. A "zombie" is a path that ended (roulette / depth) with a
// shadow probe still queued: it survives one more trace pass, collects the contribution and retires.
#pragma once
#include "rtw_device.h"

namespace rtwk {
using namespace rtwdev;

#ifndef RTW_MIN_WAVES
#define RTW_MIN_WAVES 1
#endif
// Phase fences. An `asm volatile` statement (here: nothing but a comment in the ISA) is a point the instruction scheduler does
// not move work across. k_path's loop body is one long straight-line region to it, and without fences it hoists loads and
// address arithmetic of later phases over earlier ones until the 96-register budget of 5 waves per SIMD overflows (88 bytes
// of scratch per lane in the hot loop); with a fence at every phase boundary the same code allocates without a spill and
// the metric workload runs 3.5 % faster. scripts/isa_phases.sh counts instructions between the comments.
#ifdef RTW_SUBPHASE_TIMERS
// Diagnostic build only (scripts/build_variant.sh sub -DRTW_SUBPHASE_TIMERS): wave-cycles between the sub-phase fences of the
// closest-hit program, accumulated per wave in a private row of a device-global table (row 14: id of the open interval, row
// 15: its start); the bookkeeping itself lies outside the measured intervals. Printed by the host after a k_path render.
constexpr int kSubRows = 16, kSubWaves = 8192;
__device__ unsigned long long g_sub_cyc[kSubWaves][kSubRows];
RTW_DEV constexpr int rtw_sub_id(const char* n) {  // "sa_hitrec" 1, "sa_lambert" 2, "sa_light" 3, "sa_metal" 4, "sa_diel" 5, "sa_iso" 6, "sa_nee" 7, other 0
    return n[0] != 's' || n[1] != 'a' ? 0 : n[3] == 'h' ? 1 : n[3] == 'l' && n[4] == 'a' ? 2 : n[3] == 'l' ? 3 : n[3] == 'm' ? 4 : n[3] == 'd' ? 5 : n[3] == 'i' ? 6 : n[3] == 'n' ? 7 : 0;
}
RTW_DEV void rtw_sub_stamp(int id) {
    const unsigned long long t_in = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63u) == 0u) {
        unsigned long long* row = g_sub_cyc[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kSubWaves];
        const unsigned long long open_id = row[14], t0 = row[15];
        if (t0 != 0ull) row[open_id] += t_in - t0;
        row[14] = (unsigned long long)id;
        __builtin_amdgcn_s_waitcnt(0);
        row[15] = __builtin_amdgcn_s_memtime();
    }
}
#define RTW_MARK2(name) asm volatile("; MARK " name)
#define RTW_SUB(id) rtw_sub_stamp(id)
#else
#define RTW_MARK2(name) asm volatile("; MARK " name)
#define RTW_SUB(id)
#endif
constexpr int kBlock = 256;                       // 4 wave64 per workgroup
#ifndef RTW_MAX_REGIONS
#define RTW_MAX_REGIONS 1024
#endif
// 1 024 regions = 4 compacting workgroups per CU (what two lanes use; a single lane used to take 8). Round 3 halved the table: it is
// static LDS of every wavefront kernel (4 KB instead of 8), and LDS is what decides whether the other lane's workgroups find room
// on a CU beside a resident k_trace_bvh (scene 4 +5 %, scene 1 +2 %).
constexpr uint32_t kMaxRegions = RTW_MAX_REGIONS;            // region counters scanned in LDS by every workgroup (>= the compacting grid)
constexpr uint32_t kZombie = 0x80000000u;
constexpr uint32_t kNeePrev = 0x40000000u;       // gk bit 30 (corrected estimator): a light sample was taken at the previous vertex

struct PathBuf {
    float4* p0; float4* p1; float4* p2; float4* p3; float4* p4; uint4* p5;
};

struct KArgs {
    DScene sc;
    PathBuf in, out;
    const uint2* hit;           // per input slot: t, (prim+1) | probe queued<<30 | occluded<<31
    uint2* hit_out;             // hit records of the OUTPUT slots (k_trace: of the slots it reads)
    float4* lbuf;               // per path id: final radiance of the sample
    const uint32_t* cnt_in;     // live paths per region (input)
    uint32_t* cnt_out;          // live paths per region (output): region b is written by workgroup b alone
    unsigned long long* stats;  // kStatRows rows of 8: [0] segments, [1] shadow probes, [2 + kind] units per kernel kind
    uint32_t n_regions, n_paths, npix, width, height, row0, sample0, seed, depth, max_depth, stack_stride, region_cap, n_iter, trace_first;
    uint32_t row_stride;        // >= 1: local row l of the shard is image row row0 + l*row_stride
    uint32_t first_group_log2;  // k_first: 2^this consecutive threads take samples of ONE pixel (a power of two dividing the batch)
    uint32_t divw_m, divw_s1, divw_s2;  // exact division by width (multiply-high + shifts)
    uint32_t divs_m, divs_s1, divs_s2;  // exact division by row_stride
    // k_path (register-resident paths with in-wave regeneration)
    uint32_t* queue;            // [0]: next job of the launch (one returning atomic per job and wave)
    float4* blocksum;           // [block - block0][shard-local pixel]: sum of the block's samples in sample order
    uint32_t n_jobs, n_ranges, units_per_job, block0, n_blocks_pass, spp;
    uint32_t unit_blocks;       // consecutive sample blocks a lane takes as one unit
    uint32_t unit_sums;         // 1: the lane units are whole aligned summation units (rtw.h RTW_SUM_UNIT_BLOCKS): one 16-byte sum is
                                // stored per summation unit, slot [block / 8][pixel]; 0: one per block, slot [block][pixel]
    const uint32_t* order;      // job order of the 64-pixel groups: three lists, longest units first (k_classify)
    const uint32_t* order_counts;  // lengths of the lists of classes 2, 1, 0
};

struct Path {
    v3 o, d;
    float ray_time;
    uint32_t gk;
    v3 ldir;
    float ltmax;
    v3 T, L, c;
    uint32_t w0, a, b;
};

// k_path's per-workgroup LDS copy of the constants its loop reads every iteration (camera frame of the perspective camera,
// the pdf rectangle, light 0): as kernel arguments they would sit in ~35 SGPRs for the whole launch, which the register
// allocator pays for with spills into VGPR lanes (v_writelane / v_readlane in the loop) and a scalar load of the light record
// per vertex; from LDS they are a few broadcast reads where they are used
struct PathConsts {
    float cam_o[4], cam_ll[4], cam_h[4], cam_v[4];
    float rect[5]; int32_t gen; float pad[2];
    float lnrm[3], larea, lemi[3], pad2;
    // what the end of a sample block needs (k_path: a branch that one or two lanes of a wave take in two iterations out of three):
    // read from here, these launch constants do not sit in SGPRs - which the loop has none to spare of - between those visits
    uint32_t npix, width, row0, row_stride, divs_m, divs_s1, divs_s2, spp, bs_lo, bs_hi, unit_shift, pad3;
    // what handing out units needs (the refill at the top of the loop: every few iterations of a wave)
    uint32_t n_jobs, n_ranges, units_per_job, unit_blocks, n_blocks_pass, block0, divw_m, divw_s1, divw_s2, q_lo, q_hi, pad4;
};

// probes = the scene lists lights: only then can a vertex queue a shadow probe (shade_a), so plane p2 - the probe's
// direction and length - is neither written nor read in scenes without (16 of a path's 96 bytes, each way)
RTW_DEV void load_trace_part(const PathBuf& B, size_t s, Path& p, bool probes) {
    const float4 a = B.p0[s], b = B.p1[s];
    p.o = V(a.x, a.y, a.z); p.d = V(a.w, b.x, b.y); p.ray_time = b.z; p.gk = __float_as_uint(b.w);
    p.ldir = V(0.f, 0.f, 0.f); p.ltmax = -1.0f;
    if (probes) {
        const float4 c = B.p2[s];
        p.ldir = V(c.x, c.y, c.z); p.ltmax = c.w;
    }
}
// Planes p3 ... p5 of a path. With probes: p3 = T.xyz, L.x; p4 = L.yz, c.xy; p5 = c.z, w0, a, b. Without (no listed light: c is
// never set): p3 the same; p4 = L.yz, w0, a; and b alone in a plane of dwords that takes p5's place (the same allocation read
// as uint32[]) - 36 bytes instead of 48, each way, in the kernels that run at the HBM copy rate.
RTW_DEV void load_state_part(const PathBuf& B, size_t s, Path& p, bool probes) {
    const float4 d = B.p3[s], e = B.p4[s];
    p.T = V(d.x, d.y, d.z);
    if (probes) {
        const uint4 f = B.p5[s];
        p.L = V(d.w, e.x, e.y); p.c = V(e.z, e.w, __uint_as_float(f.x));
        p.w0 = f.y; p.a = f.z; p.b = f.w;
    } else {
        p.L = V(d.w, e.x, e.y); p.c = V(0.f, 0.f, 0.f);
        p.w0 = __float_as_uint(e.z); p.a = __float_as_uint(e.w); p.b = ((const uint32_t*)B.p5)[s];
    }
}
RTW_DEV void load_path(const PathBuf& B, size_t s, Path& p, bool probes) {
    load_trace_part(B, s, p, probes);
    load_state_part(B, s, p, probes);
}
// k_shade's view: everything but the queued probe's direction (plane p2 belongs to k_trace; whether a probe was
// queued and what it found comes back in the hit record)
RTW_DEV void load_shade_part(const PathBuf& B, size_t s, Path& p, bool probes) {
    const float4 a = B.p0[s], b = B.p1[s];
    p.o = V(a.x, a.y, a.z); p.d = V(a.w, b.x, b.y); p.ray_time = b.z; p.gk = __float_as_uint(b.w);
    p.ldir = V(0.f, 0.f, 0.f); p.ltmax = -1.0f;
    load_state_part(B, s, p, probes);
}
RTW_DEV void store_path(const PathBuf& B, size_t s, const Path& p, bool probes) {
    B.p0[s] = make_float4(p.o.x, p.o.y, p.o.z, p.d.x);
    B.p1[s] = make_float4(p.d.y, p.d.z, p.ray_time, __uint_as_float(p.gk));
    B.p3[s] = make_float4(p.T.x, p.T.y, p.T.z, p.L.x);
    if (probes) {
        B.p2[s] = make_float4(p.ldir.x, p.ldir.y, p.ldir.z, p.ltmax);
        B.p4[s] = make_float4(p.L.y, p.L.z, p.c.x, p.c.y);
        B.p5[s] = make_uint4(__float_as_uint(p.c.z), p.w0, p.a, p.b);
    } else {
        B.p4[s] = make_float4(p.L.y, p.L.z, __uint_as_float(p.w0), __uint_as_float(p.a));
        ((uint32_t*)B.p5)[s] = p.b;
    }
}

// ------------------------------------------------------------------ work list
// Region r holds cnt_in[r] live paths = ceil(cnt/256) chunks. Every workgroup scans the (<= kMaxRegions)
// counters into an LDS prefix array once, then strides over the virtual chunk ids: no empty
// iterations, perfect balance, and a launch with nothing alive costs one scan.
struct WorkList {
    uint32_t* pref;  // [kMaxRegions+1]: (exclusive prefix of chunk counts) << 8 | (live paths of the region) & 255
    uint32_t total_chunks;
};
#define RTW_WORKLIST_SHARED_T(BLOCK_)               \
    __shared__ uint32_t s_pref[kMaxRegions + 1];   \
    __shared__ uint32_t s_part[BLOCK_];
#define RTW_WORKLIST_SHARED RTW_WORKLIST_SHARED_T(kBlock)

// BLOCK = threads of the calling workgroup (the chunks of the list are kBlock slots whatever the workgroup's size)
template <uint32_t BLOCK = kBlock>
RTW_DEV WorkList worklist_init(const uint32_t* cnt_in, uint32_t n_regions, uint32_t* s_pref, uint32_t* s_part) {
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kPer = kMaxRegions / BLOCK;
    uint32_t loc[kPer], low[kPer];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
        const uint32_t r = tid * kPer + j;
        const uint32_t raw = r < n_regions ? cnt_in[r] : 0u;
        low[j] = raw & 255u;
        loc[j] = sum;
        sum += (raw + kBlock - 1) / kBlock;
    }
    s_part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < BLOCK; off <<= 1) {
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    const uint32_t excl = s_part[tid] - sum;
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) s_pref[tid * kPer + j] = ((excl + loc[j]) << 8) | low[j];
    if (tid == BLOCK - 1) s_pref[kMaxRegions] = s_part[tid] << 8;
    __syncthreads();
    WorkList w;
    w.pref = s_pref; w.total_chunks = s_pref[kMaxRegions] >> 8;
    return w;
}
// virtual chunk id -> (region, chunk within the region, number of live paths in that chunk: 256 but for a region's last)
RTW_DEV void worklist_lookup(const WorkList& w, uint32_t n_regions, uint32_t vc, uint32_t& region, uint32_t& chunk, uint32_t& n_valid) {
    uint32_t lo = 0, hi = n_regions;  // largest r with prefix[r] <= vc
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((w.pref[mid] >> 8) <= vc) lo = mid; else hi = mid;
    }
    const uint32_t e = w.pref[lo];
    const uint32_t n_chunks = (w.pref[lo + 1] >> 8) - (e >> 8);
    const uint32_t rem = e & 255u;
    region = lo;
    chunk = vc - (e >> 8);
    n_valid = (chunk + 1u < n_chunks || rem == 0u) ? kBlock : rem;
}

struct NoRng {
    uint32_t a, b;
    RTW_DEV float next1() { return 0.5f; }
    RTW_DEV float randf1() { return 0.5f; }
};

// ------------------------------------------------------------------ shading
struct Nee {
    bool has;
    v3 dir, rad;  // rad = f * Le * (w * (wL.n) / pdfL), not yet multiplied by the throughput
    float tmin, tmax;
};

// RTW_EST_MIXTURE: the solid-angle density, seen from so, of "one of the nl listed lights uniformly, then a point on its
// parallelogram uniformly" in the unit direction w (what the reference's rect_*_value stubs, pdf/rectPdf.cu:75-122, were
// meant to return); the oracle's light_list_pdf, operation for operation.
RTW_DEV float light_list_pdf(const rtw_light* __restrict__ lights, const int nl, const v3 so, const v3 w) {
    float sum = 0.0f;
    for (int i = 0; i < nl; i++) {
        const rtw_light lt = lights[i];
        const v3 n = ld3(lt.normal), pos = ld3(lt.position), eu = ld3(lt.vec_u), evv = ld3(lt.vec_v);
        const float denom = dot3(w, n);
        const float dn = dot3(vsub(pos, so), n);
        if (denom == 0.0f) continue;
        const float t = dn / denom;
        if (!(t > 1.0e-6f)) continue;
        const v3 rel = vsub(vfma(w, t, so), pos);
        const float a = dot3(rel, eu) / dot3(eu, eu);
        const float b = dot3(rel, evv) / dot3(evv, evv);
        if (!(a >= -1.0e-4f && a <= 1.0001f && b >= -1.0e-4f && b <= 1.0001f)) continue;
        sum += (t * t) / (lt.area * __builtin_fabsf(denom));
    }
    return sum / (float)nl;
}

// Closest-hit / miss program up to and including the light sample (shaders/closehit.cu:45-94,
// miss/miss.cu:8-30, material/*.cu, pdf/mixturePdf.cu:25-38, pdf/rectPdf.cu:124-193).
// TEX: 0 = the hot instantiation; 1 = the one for scenes with non-constant textures or media and for the corrected
// estimators (the cold features); 2 = 1 + the mixture estimator (its light-list pdf loop costs the others registers:
// inlined into instantiation 1 it put 20 bytes of scratch into k_shade and k_path and 4-7 % onto scenes 2, 3 and 4)
template <int KIND, int TEX>
RTW_DEV int shade_a(const DScene& sc, Rng<KIND>& g, const v3 origin, const v3 dir, const float gather_time, const float t, const int prim,
                    v3& so, v3& sd, v3& att, v3& radiance, Nee& nee, const uint32_t* noise_lds, uint32_t& nee_prev, const u32x4* hr_lds = nullptr,
                    const PathConsts* pc = nullptr) {
    // nee_prev (corrected estimator only): in - a light sample was taken at the previous vertex; out - one was taken here
    const int est = TEX ? sc.estimator : 0;
    const uint32_t had_nee = nee_prev;
    nee_prev = 0u;
    radiance = V(0.f, 0.f, 0.f);
    att = V(0.f, 0.f, 0.f);
    so = origin; sd = dir;
    nee.has = false;
    nee.dir = V(0.f, 0.f, 0.f); nee.rad = V(0.f, 0.f, 0.f); nee.tmin = 0.f; nee.tmax = -1.f;
    if (prim < 0) {
        if (sc.sky_light) {  // miss.cu:8-21
            v3 u = normalize3(dir);
            float tt = 0.5f * (u.y + 1.0f);
            float w = 1.0f - tt;
            radiance = V(fma_(tt, 0.5f, w), fma_(tt, 0.7f, w), fma_(tt, 1.0f, w));
        }
        return EV_MISS;
    }
    RTW_MARK2("sa_hitrec");
    RTW_SUB(1);
    g.align_block();
    const HitRec hr = load_hitrec(sc, prim, hr_lds);
    // the segment's first Philox block, generated while the whole wave is on one code path and - not depending on the hit
    // record - while its LDS reads are in flight (a light or normal-material vertex draws nothing: its block is simply unused)
    g.warm();
    v3 hp, hn;
    hit_attributes(sc, hr, prim, origin, dir, t, gather_time, hp, hn);
    const int mtype = hr.mat_type;
    const float mparam = hr.param;
    v3 tex = V(hr.r, hr.g, hr.b);
    if (TEX && hr.tex_dyn >= 0) tex = texture_eval(sc, hr, prim, origin, dir, t, 0.0f, hp, hn, noise_lds);  // checker / noise / image
    int ev;
    bool specular = false;
    RTW_MARK2("sa_lambert");
    if (mtype == RTW_MAT_LAMBERTIAN) {
        RTW_SUB(2);
        // lambertianMaterial.cu:41-71, onb.cuh:20-32, sampling.cuh:49-60 (Q1)
        v3 u, v, w;
        if (hr.kind == HK_CONST_NORMAL) {  // basis baked per primitive at upload, same operations
            u = V(hr.ux, hr.uy, hr.uz); v = V(hr.vx, hr.vy, hr.vz); w = V(hr.wx, hr.wy, hr.wz);
        } else {
            w = normalize3(hn);
            v3 a = (w.x > 0.9f || w.x < -0.9f) ? V(0.f, 1.f, 0.f) : V(1.f, 0.f, 0.f);
            v = normalize3(cross3(w, a));
            u = cross3(w, v);
        }
        float r1, r2;
        if (KIND == RTW_RNG_PHILOX) {
            // the segment's first block is in registers (align_block + warm above) and these are its draws 0 and 1: no
            // block check, no lane select (every next1() call site otherwise carries its own copy of the refill)
            r1 = g.block_draw(0); r2 = g.block_draw(1);
            g.a += 2u;
        } else {
            r1 = g.next1();
            r2 = g.next1();
        }
        float sn, cs;
        sincos2pi(r1, sn, cs);
        float sq = __builtin_sqrtf(r2);
        float lx = est ? cs * sq : (cs * 2.0f) * sq;  // corrected: cosine-weighted, without the stray 2 (Q1)
        float ly = est ? sn * sq : (sn * 2.0f) * sq;
        float lz = __builtin_sqrtf(1.0f - r2);
        float pdf = lz * RTW_1_PI_F;
        v3 sdir = V(fma_(lz, w.x, fma_(ly, v.x, lx * u.x)),
                    fma_(lz, w.y, fma_(ly, v.y, lx * u.y)),
                    fma_(lz, w.z, fma_(ly, v.z, lx * u.z)));
        sdir = normalize3(sdir);
        so = hp; sd = sdir;
        float cosine = dot3(hn, sdir);
        ev = EV_HIT;
        if (cosine <= 0.0f || pdf <= 0.0f) ev = EV_CANCEL;
        else att = tex;
    RTW_MARK2("sa_light");
    } else if (mtype == RTW_MAT_DIFFUSE_LIGHT) {
        // diffuseLight.cu:48-69
        RTW_SUB(3);
        if (dot3(hn, dir) < 0.0f) radiance = tex;
        // corrected: the light sample of the previous vertex already accounted for this emitter
        if (est == RTW_EST_CORRECTED && had_nee != 0u && hr.listed != 0) radiance = V(0.f, 0.f, 0.f);
        ev = EV_CANCEL;
    RTW_MARK2("sa_metal");
    } else if (mtype == RTW_MAT_METAL) {
        // metalMaterial.cu:32-64 (Q5)
        RTW_SUB(4);
        specular = true;
        v3 refl = reflect3(est ? normalize3(dir) : dir, hn);  // corrected: unit incoming direction (Q5)
        v3 ball = random_in_unit_sphere(g);
        v3 sdir = normalize3(vfma(ball, mparam, refl));
        so = hp; sd = sdir;
        att = tex;
        ev = (dot3(sdir, hn) <= 0.0f) ? EV_CANCEL : EV_HIT;
    RTW_MARK2("sa_diel");
    } else if (mtype == RTW_MAT_DIELECTRIC) {
        // dielectricMaterial.cu:37-114
        RTW_SUB(5);
        specular = true;
        v3 unit = normalize3(dir);
        v3 ln;
        float eta_i, eta_t;
        if (dot3(dir, hn) < 0.0f) { ln = hn; eta_i = 1.0f; eta_t = mparam; }
        else { ln = vneg(hn); eta_i = mparam; eta_t = 1.0f; }
        float cos_i = __builtin_fminf(dot3(vneg(unit), ln), 1.0f);
        float sin_i = __builtin_sqrtf(fma_(-cos_i, cos_i, 1.0f));
        float ratio = eta_i / eta_t;
        v3 sdir;
        if (ratio * sin_i > 1.0f) {
            sdir = reflect3(unit, ln);
        } else {
            float r0 = (eta_i - eta_t) / (eta_i + eta_t);
            r0 = r0 * r0;
            float om = 1.0f - cos_i;
            float om2 = om * om;
            float p5 = (om2 * om2) * om;
            float refl_prob = fma_(1.0f - r0, p5, r0);
            if (g.next1() < refl_prob) {
                sdir = reflect3(unit, ln);
            } else {
                float sin_t = __builtin_fminf(ratio * sin_i, 1.0f);
                float cos_t = __builtin_sqrtf(fma_(-sin_t, sin_t, 1.0f));
                v3 a = vscale(vfma(ln, cos_i, unit), ratio);
                sdir = vfma(ln, -cos_t, a);
            }
        }
        so = hp; sd = sdir;
        att = V(1.f, 1.f, 1.f);
        ev = EV_HIT;
    RTW_MARK2("sa_iso");
    } else if (mtype == RTW_MAT_ISOTROPIC) {
        // isotropicMaterial.cu:30-51 (Q14)
        RTW_SUB(6);
        specular = true;
        sd = random_in_unit_sphere(g);
        so = hp;
        att = tex;
        ev = EV_HIT;
    } else {
        // normalMaterial.cu:21-31
        specular = true;
        att = vfma(hn, 0.5f, V(0.5f, 0.5f, 0.5f));
        ev = EV_FINISH;
    }

    RTW_MARK2("sa_nee");
    RTW_SUB(7);
    // next-event estimation, closehit.cu:70-94: sample the light; the visibility probe comes later
    const int nl = sc.n_lights;
    if (est == RTW_EST_CORRECTED && ev == EV_HIT && !specular && nl > 0) {
        // each listed light over its own parallelogram (moved onto the emitting rectangle at upload), area-measure
        // estimator, no heuristic weight
        nee_prev = 1u;
        int il = 0;
        if (nl > 1) {
            il = (int)__builtin_floorf(g.next1() * (float)nl);
            il = il < 0 ? 0 : (il > nl - 1 ? nl - 1 : il);
        }
        const rtw_light lt = sc.clights[il];
        const float ra = g.next1();
        const float rb = g.next1();
        const v3 rp = vfma(ld3(lt.vec_v), rb, vfma(ld3(lt.vec_u), ra, ld3(lt.position)));
        v3 ldir = vsub(rp, so);
        const float ldist = length3(ldir);
        if (ldist > 1.0e-6f && hr.bsdf_eval == 0) {
            ldir = vscale(ldir, 1.0f / ldist);
            const float costa = dot3(vneg(ldir), ld3(lt.normal));
            const float ndl = dot3(ldir, hn);
            const v3 f = vscale(att, RTW_1_PI_F);
            if (costa > 1.0e-6f && ndl > 0.0f && (f.x != 0.0f || f.y != 0.0f || f.z != 0.0f)) {
                const float eps = sc.probe_eps;
                const float lpdf = (ldist * ldist) / (lt.area * costa);
                const float k = ndl / lpdf;
                const v3 lem = vscale(ld3(lt.emission), (float)nl);
                nee.has = true;
                nee.dir = ldir;
                nee.tmin = eps;
                nee.tmax = ldist - eps;
                nee.rad = vscale(vmul(f, lem), k);
            }
        }
    } else if (TEX == 2 && est == RTW_EST_MIXTURE && ev == EV_HIT && !specular && nl > 0 && hr.bsdf_eval == 0) {
        // the book's estimator (rtw.h): direction from the light list or the cosine lobe, probability 1/2 each; the
        // throughput carries albedo * p_cos / (p_cos / 2 + p_light / 2); no probe, every emitter hit counts
        int il = 0;
        if (nl > 1) {
            il = (int)__builtin_floorf(g.next1() * (float)nl);
            il = il < 0 ? 0 : (il > nl - 1 ? nl - 1 : il);
        }
        const float u0 = g.next1();
        const float ra = g.next1();
        const float rb = g.next1();
        if (u0 < 0.5f) {
            const rtw_light lt = sc.clights[il];
            const v3 rp = vfma(ld3(lt.vec_v), rb, vfma(ld3(lt.vec_u), ra, ld3(lt.position)));
            const v3 ldir = vsub(rp, so);
            const float ldist = length3(ldir);
            if (ldist > 1.0e-6f) sd = vscale(ldir, 1.0f / ldist);
        }
        const float ndl = dot3(sd, hn);
        const float pb = __builtin_fmaxf(0.0f, ndl) * RTW_1_PI_F;
        const float pl = light_list_pdf(sc.clights, nl, so, sd);
        const float pm = 0.5f * (pb + pl);
        if (pb > 0.0f && pm > 0.0f) att = vscale(att, pb / pm);
        else ev = EV_CANCEL;
    } else if (est == RTW_EST_REFERENCE && ev == EV_HIT && !specular && nl > 0) {
        int il = 0;
        if (nl > 1) {
            il = (int)__builtin_floorf(g.next1() * (float)nl);
            il = il < 0 ? 0 : (il > nl - 1 ? nl - 1 : il);
        }
        // one light (the usual case): the record index is wave-uniform, so it is read through the scalar cache
        v3 lnrm, lemi;
        float larea;
        if (nl > 1) {
            const RTW_CONST rtw_light* lt = as_const(sc.lights + il);
            lnrm = V(lt->normal[0], lt->normal[1], lt->normal[2]);
            lemi = V(lt->emission[0], lt->emission[1], lt->emission[2]);
            larea = lt->area;
        } else if (pc != nullptr) {
            lnrm = ld3(pc->lnrm); lemi = ld3(pc->lemi); larea = pc->larea;
        } else {
            const RTW_CONST rtw_light* lt = as_const(sc.lights);
            lnrm = V(lt->normal[0], lt->normal[1], lt->normal[2]);
            lemi = V(lt->emission[0], lt->emission[1], lt->emission[2]);
            larea = lt->area;
        }
        int gen;  // mixturePdf.cu:25-38: always child p1 (Q4)
        float rc0, rc1, rc2, rc3, rc4;
        if (pc != nullptr) {
            gen = __builtin_amdgcn_readfirstlane(pc->gen);
            rc0 = pc->rect[0]; rc1 = pc->rect[1]; rc2 = pc->rect[2]; rc3 = pc->rect[3]; rc4 = pc->rect[4];
        } else {
            gen = sc.pdf.gen;
            if (gen == RTW_PDF_MIXTURE || gen == RTW_PDF_MIXTURE_BIAS) gen = sc.pdf.p1_gen;
            rc0 = sc.pdf.rect[0]; rc1 = sc.pdf.rect[1]; rc2 = sc.pdf.rect[2]; rc3 = sc.pdf.rect[3]; rc4 = sc.pdf.rect[4];
        }
        float lpdf = 0.0f, ldist = 0.0f;
        v3 ldir = V(0.f, 0.f, 0.f), lem = V(0.f, 0.f, 0.f);
        if (gen == RTW_PDF_RECT_X || gen == RTW_PDF_RECT_Y || gen == RTW_PDF_RECT_Z) {
            // rectPdf.cu:124-193
            float ra, rb;
            if (KIND == RTW_RNG_PHILOX && nl == 1) {
                // only a Lambertian vertex gets here, two draws into its block: these are draws 2 and 3 of the same block
                ra = g.block_draw(2); rb = g.block_draw(3);
                g.a += 2u;
            } else {
                ra = g.next1();
                rb = g.next1();
            }
            float pa = fma_(ra, rc1 - rc0, rc0);
            float pb = fma_(rb, rc3 - rc2, rc2);
            float k = rc4;
            v3 rp = (gen == RTW_PDF_RECT_X) ? V(k, pa, pb) : (gen == RTW_PDF_RECT_Y) ? V(pa, k, pb) : V(pa, pb, k);
            ldir = vsub(rp, so);
            ldist = length3(ldir);
            if (ldist > 1.0e-6f) {
                ldir = vscale(ldir, 1.0f / ldist);
                float costa = dot3(vneg(ldir), lnrm);
                if (costa > 1.0e-6f) {
                    lem = vscale(lemi, (float)nl);
                    lpdf = (ldist * ldist) / (larea * costa);
                }
            }
        }
        if (lpdf > 0.0f && hr.bsdf_eval == 0) {
            // lambertianMaterial.cu:74-81
            v3 f = vscale(att, RTW_1_PI_F);
            float ndl = dot3(ldir, hn);
            float bpdf = __builtin_fmaxf(0.0f, ndl * RTW_1_PI_F);
            if (0.0f < bpdf && (f.x != 0.0f || f.y != 0.0f || f.z != 0.0f)) {
                const float eps = sc.probe_eps;
                float a2 = lpdf * lpdf;
                float weight = a2 / fma_(bpdf, bpdf, a2);  // raydata.cuh:167-171
                float k = (weight * ndl) / lpdf;
                nee.has = true;
                nee.dir = ldir;
                nee.tmin = eps;
                nee.tmax = ldist - eps;  // closehit.cu:95-101
                nee.rad = vscale(vmul(f, lem), k);  // closehit.cu:111-113
            }
        }
    }
    return ev;
}

// Tail of rayColor's loop body (raygen.cu:60-84): accumulate, continue or stop, Russian roulette.
// cap_rr (RTW_EST_MIXTURE): that estimator's weights reach 2, so its throughput can exceed 1 and the survival probability
// is capped at 1 (the reference's throughput never exceeds 1: its roulette needs no cap and gets none)
template <int KIND>
RTW_DEV bool shade_b(const uint32_t depth, const uint32_t max_depth, Rng<KIND>& g, const int ev, const v3 so, const v3 sd, const v3 att,
                     const v3 radiance, v3& origin, v3& dir, v3& T, v3& L, const bool cap_rr = false) {
    L = vadd(L, vmul(radiance, T));  // raygen.cu:60
    if (ev != EV_HIT) return false;
    origin = so; dir = sd;
    T = vmul(T, att);
    if (2u <= depth) {  // raygen.cu:74-82
        float p = __builtin_fmaxf(__builtin_fmaxf(T.x, T.y), T.z);
        if (cap_rr) p = __builtin_fminf(p, 1.0f);
        if (p < g.rr_draw()) return false;
        T = vscale(T, 1.0f / p);
    }
    return depth + 1u < max_depth;
}

template <int KIND>
RTW_DEV void rng_from_path(Rng<KIND>& g, uint32_t seed, const Path& p) {
    if (KIND == RTW_RNG_TEA_LCG) g.init(seed, 0, 0, p.a, p.b);
    else g.init(seed, p.w0, p.b, p.a, p.b);  // Philox: w0 = global pixel, b = sample index
}
// n / d for any 32-bit n, with (m, s1, s2) precomputed on the host for the invariant divisor d (Granlund-Montgomery)
RTW_DEV uint32_t fastdiv(uint32_t n, uint32_t m, uint32_t s1, uint32_t s2) {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> s1)) >> s2;
}
template <int KIND>
RTW_DEV uint32_t path_id_of(const KArgs& A, const Path& p) {
    if (KIND == RTW_RNG_TEA_LCG) return p.w0;
    // Philox paths carry the global pixel (their stream key); the radiance slot is indexed by the shard-local pixel
    uint32_t local = p.w0 - A.row0 * A.width;
    if (A.row_stride > 1) {
        const uint32_t y = fastdiv(p.w0, A.divw_m, A.divw_s1, A.divw_s2);
        const uint32_t x = p.w0 - y * A.width;
        local = fastdiv(y - A.row0, A.divs_m, A.divs_s1, A.divs_s2) * A.width + x;
    }
    return (p.b - A.sample0) * A.npix + local;
}
RTW_DEV float gather_time_of(const KArgs& A, uint32_t gk) {
    return fma_((float)(gk & 0x00ffffffu) * (1.0f / 16777216.0f), A.sc.cam.time1 - A.sc.cam.time0, A.sc.cam.time0);
}
RTW_DEV void finish_path(const KArgs& A, uint32_t path_id, v3 L) {
    // removeNaNs, raygen.cu:17-24
    float lx = (L.x == L.x) ? L.x : 0.f, ly = (L.y == L.y) ? L.y : 0.f, lz = (L.z == L.z) ? L.z : 0.f;
    A.lbuf[path_id] = make_float4(lx, ly, lz, 0.f);
}
// Stream compaction without global atomics: output region b belongs to workgroup b, whose four waves share a
// cursor in LDS (agent-scope atomics are resolved beyond the XCD's L2 on this part, ~0.4 us each and serialised per
// address). The host sizes a region for the most paths one workgroup of the persistent grid can be handed.
// s_cursor[0]: the output cursor; [1], [2]: this workgroup's segment / shadow-ray counts. The render statistics live in
// kStatRows rows of 8 counters (row = workgroup & 63, summed by the host) so that the one global atomic per workgroup
// and counter does not queue behind every other workgroup's on a single address.
constexpr uint32_t kStatRows = 64;
#define RTW_CURSOR_SHARED __shared__ uint32_t s_cursor[3];
RTW_DEV void cursor_init(uint32_t* s_cursor) {
    if (threadIdx.x < 3) s_cursor[threadIdx.x] = 0u;
    __syncthreads();
}
RTW_DEV unsigned long long* stat_row(const KArgs& A) { return A.stats + (size_t)(blockIdx.x & (kStatRows - 1u)) * 8u; }
RTW_DEV void cursor_publish(const KArgs& A, uint32_t* s_cursor, uint32_t n_seg, uint32_t n_shadow, int kind) {
    for (int off = 32; off > 0; off >>= 1) {
        n_seg += __shfl_down(n_seg, off);
        n_shadow += __shfl_down(n_shadow, off);
    }
    if ((threadIdx.x & 63u) == 0 && (n_seg | n_shadow)) {
        atomicAdd(&s_cursor[1], n_seg);
        atomicAdd(&s_cursor[2], n_shadow);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        A.cnt_out[blockIdx.x] = s_cursor[0];
        const uint32_t a = s_cursor[1], b = s_cursor[2];
        unsigned long long* row = stat_row(A);
        if (a) { atomicAdd(&row[0], (unsigned long long)a); atomicAdd(&row[2 + kind], (unsigned long long)a); }
        if (b) atomicAdd(&row[1], (unsigned long long)b);
    }
}
RTW_DEV void compact_store(const KArgs& A, uint32_t* s_cursor, bool keep, const Path& p) {
    const unsigned long long ballot = __ballot(keep);
    if (!ballot) return;
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0) base = atomicAdd(&s_cursor[0], (uint32_t)__popcll(ballot));
    base = __builtin_amdgcn_readfirstlane(base);
    if (keep) store_path(A.out, (size_t)blockIdx.x * A.region_cap + base + before, p, A.sc.n_lights > 0);
}
// The trace pass of one path: radiance ray closest hit + queued shadow probe any-hit, result into the hit buffer.
template <bool DUAL>
RTW_DEV void trace_path(const KArgs& A, const Path& p, size_t slot, const TravMem& tm, uint32_t& n_rays) {
    NoRng ng;
    const float gt = gather_time_of(A, p.gk);
    const bool do_r = !(p.gk & kZombie);
    const bool do_s = p.ltmax >= 0.0f;
    float th = 0.f;
    int prim = -1;
    uint32_t occl = 0;
    if (DUAL) {
        // small scenes: one shared walk for both rays
        bool oc;
        traverse_dual_brute(A.sc, p.o, p.d, p.ldir, do_r, do_s, A.sc.probe_eps, p.ltmax, th, prim, oc);
        if (!do_r) { th = 0.f; prim = -1; }
        if (do_s) occl = (oc ? 0x80000000u : 0u) | 0x40000000u;  // bit 30: a probe was queued
    } else {
        if (do_r) traverse<NoRng, false, true>(A.sc, p.o, p.d, A.sc.ray_tmin, 1.e27f, p.ray_time, gt, ng, tm, th, prim);
        if (do_s) {  // traceOcclusion, closehit.cu:16-42 (ray time 0)
            float st;
            int sprim;
            traverse<NoRng, true, true>(A.sc, p.o, p.ldir, A.sc.probe_eps, p.ltmax, 0.0f, gt, ng, tm, st, sprim);
            occl = (sprim >= 0 ? 0x80000000u : 0u) | 0x40000000u;
        }
    }
    n_rays += (do_r || do_s) ? 1u : 0u;  // unit of the trace kernels: one path slot (its radiance ray and its queued probe)
    A.hit_out[slot] = make_uint2(__float_as_uint(th), (uint32_t)(prim + 1) | occl);
}

// LDS copy of one noise texture's tables (6 KB) for the kernels instantiated with TEX
#define RTW_NOISE_SHARED __shared__ uint32_t s_noise[TEX ? 1536 : 1];
template <bool TEX>
RTW_DEV const uint32_t* stage_noise(const DScene& sc, uint32_t* s_noise) {
    if (!TEX || sc.noise_lds_data < 0) return nullptr;
    for (uint32_t i = threadIdx.x; i < 1536u; i += kBlock) s_noise[i] = sc.texdata[sc.noise_lds_data + i];
    __syncthreads();
    return s_noise;
}

// __raygen__Program (raygen.cu:123-147) + perspectiveCamera (camera.cu:11-19) + color() (raygen.cu:89-95): the camera
// path of sample `sample` of image pixel (x, y); shared by k_first and k_path so that both draw the same numbers.
template <int KIND, bool LDS_CAM = false>
RTW_DEV void raygen(const KArgs& A, const uint32_t x, const uint32_t y, const uint32_t sample, const uint32_t path_id, Path& p, Rng<KIND>& g,
                    const PathConsts* pc = nullptr) {
    const uint32_t pixel = A.width * y + x;
    float r0, r1, r2, r3, r4;
    if (KIND == RTW_RNG_TEA_LCG) {
        uint32_t s = tea<64>(pixel, sample);  // raygen.cu:129
        r0 = lcg_rnd(s); r1 = lcg_rnd(s);
        r2 = 0.0f; r3 = 0.0f;
        // camera.cu:11-19: the perspective camera draws its lens sample (used or not); scene/camera.cuh:35-56: the other two take no seed
        if (LDS_CAM || A.sc.cam_type == RTW_CAM_PERSPECTIVE) { r2 = lcg_rnd(s); r3 = lcg_rnd(s); }
        g.init(A.seed, pixel, sample, s, s);  // prd.seed = seed; rayColor's local copy (Q7)
        r4 = lcg_rnd(s);
        p.w0 = path_id;
    } else {
        uint32_t o[4];
        philox4x32_10(pixel, sample, 0u, 0u, A.seed, 0u, o);
        r0 = u24(o[0]); r1 = u24(o[1]); r2 = u24(o[2]); r3 = u24(o[3]);
        // fifth raygen draw (gather time) from the block's spare low bytes: one raygen block per camera path
        r4 = (float)(((o[0] & 0xffu) << 16) | ((o[1] & 0xffu) << 8) | (o[2] & 0xffu)) * (1.0f / 16777216.0f);
        g.init(A.seed, pixel, sample, 0u, sample);
        p.w0 = pixel;
    }
    const rtw_camera& cam = A.sc.cam;
    const float s = ((float)x + r0) / (float)A.width;
    const float t = ((float)y + r1) / (float)A.height;
    if (LDS_CAM) {  // k_path, perspective camera without a lens: the frame comes from LDS (same values, same operations)
        p.o = ld3(pc->cam_o);
        p.d = vfma(ld3(pc->cam_h), s, ld3(pc->cam_ll));
        p.d = vfma(ld3(pc->cam_v), t, p.d);
        p.d = vsub(p.d, p.o);
        p.gk = (uint32_t)(r4 * 16777216.0f);
        p.ray_time = (KIND == RTW_RNG_TEA_LCG || A.sc.has_motion) ? g.ray_time(0u) : 0.0f;
        p.ldir = V(0.f, 0.f, 0.f); p.ltmax = -1.0f;
        p.T = V(1.f, 1.f, 1.f); p.L = V(0.f, 0.f, 0.f); p.c = V(0.f, 0.f, 0.f);
        p.a = g.a; p.b = g.b;
        return;
    }
    p.o = ld3(cam.origin);
    if (cam.lens_radius != 0.0f) {  // sampling.cuh:15-22; the two draws are consumed either way
        float sn, cs;
        sincos2pi(r2, sn, cs);
        float sq = __builtin_sqrtf(r3);
        float rx = cam.lens_radius * (sn * sq);
        float ry = cam.lens_radius * (cs * sq);
        p.o = vadd(p.o, vfma(ld3(cam.v), ry, vscale(ld3(cam.u), rx)));
    }
    p.d = vfma(ld3(cam.horizontal), s, ld3(cam.lower_left));
    p.d = vfma(ld3(cam.vertical), t, p.d);
    if (A.sc.cam_type == RTW_CAM_ENVIRONMENT) {  // scene/camera.cuh:35-47 (pi t = 2 pi (t / 2))
        float sx, cx, sy, cy;
        sincos2pi(s, sx, cx);
        sincos2pi(t * 0.5f, sy, cy);
        const v3 a = V(cx * sy, -cy, sx * sy);
        p.o = ld3(cam.origin);
        p.d = normalize3(vfma(ld3(cam.w), a.z, vfma(ld3(cam.v), a.y, vscale(ld3(cam.u), a.x))));
    } else if (A.sc.cam_type == RTW_CAM_ORTHOGRAPHIC) {  // scene/camera.cuh:49-54
        p.o = vadd(p.d, ld3(cam.origin));
        p.d = vneg(normalize3(ld3(cam.w)));
    } else {
        p.d = vsub(p.d, p.o);
    }
    p.gk = (uint32_t)(r4 * 16777216.0f);
    p.ray_time = (KIND == RTW_RNG_TEA_LCG || A.sc.has_motion) ? g.ray_time(0u) : 0.0f;  // raygen.cu:48
    p.ldir = V(0.f, 0.f, 0.f); p.ltmax = -1.0f;
    p.T = V(1.f, 1.f, 1.f); p.L = V(0.f, 0.f, 0.f); p.c = V(0.f, 0.f, 0.f);
    p.a = g.a; p.b = g.b;
}

// ------------------------------------------------------------------ k_first
#ifndef RTW_FIRST_WAVES
#define RTW_FIRST_WAVES RTW_MIN_WAVES
#endif
#ifndef RTW_FIRST_COLD_WAVES
#define RTW_FIRST_COLD_WAVES RTW_MIN_WAVES
#endif
template <int KIND, int TEX>
__global__ void __launch_bounds__(kBlock, TEX ? RTW_FIRST_COLD_WAVES : RTW_FIRST_WAVES) k_first(const KArgs A) {
    extern __shared__ uint32_t s_stack[];
    RTW_CURSOR_SHARED
    RTW_NOISE_SHARED
    const uint32_t tid = threadIdx.x;
    cursor_init(s_cursor);
    const uint32_t* noise_lds = stage_noise<(TEX != 0)>(A.sc, s_noise);
    // (no per-lane stacks and no LDS image of the tree here: the camera rays of a wave walk the tree as one, with one stack per
    // wave - stack_depth dwords of dynamic LDS each - and the node records as scalar operands; scenes without a tree walk the
    // candidate lists. What this kernel does not take of a CU's LDS, the other lane's k_trace_bvh can.)
    uint32_t n_seg = 0, n_shadow = 0;
    const uint32_t total_chunks = (A.n_paths + kBlock - 1) / kBlock;
    for (uint32_t vc = blockIdx.x; vc < total_chunks; vc += gridDim.x) {
        // Which camera path a thread starts. Path ids are sample-major (id = sample slot * npix + pixel: what k_resolve
        // reads), but neighbouring THREADS take 2^first_group_log2 samples of one pixel before the next pixel: a wave's
        // camera rays then leave through one pixel (or a few) and differ only in their lens sample, so they walk the
        // tree together and mostly meet the same material. The image cannot depend on this: a path's draws are keyed
        // by (pixel, sample).
        const uint32_t t_lin = vc * kBlock + tid;
        const uint32_t s_lo = t_lin & ((1u << A.first_group_log2) - 1u), q = t_lin >> A.first_group_log2;
        const uint32_t s_hi = q / A.npix;
        const uint32_t pl = q - s_hi * A.npix;
        const uint32_t slot = (s_hi << A.first_group_log2) | s_lo;
        const uint32_t path_id = slot * A.npix + pl;
        bool keep = false;
        Path p;
        p.gk = 0; p.ltmax = -1.f;
        if (t_lin < A.n_paths) {
        const uint32_t yl = pl / A.width;
        const uint32_t x = pl - yl * A.width;
        const uint32_t y = A.row0 + yl * A.row_stride;
        Rng<KIND> g;
        raygen<KIND>(A, x, y, A.sample0 + slot, path_id, p, g);
        keep = true;
        if (A.trace_first) {
            // split pipeline: trace and shade the primary segment here (primary rays are coherent, so the fused
            // form costs no divergence and saves writing and re-reading 104 B per camera path, half of which
            // leave the scene at once in a 16:9 Cornell frame)
            float th;
            int prim;
            NoRng ng;
            const float gt = gather_time_of(A, p.gk);
            // camera rays of a chunk are neighbours: in a 16:9 Cornell frame 4 waves in 10 look past the box entirely,
            // and a wave-uniform test against the scene bounds spares them the walk over the candidate lists
            th = 1.e27f; prim = -1;
            if (__ballot(may_hit_scene(A.sc, p.o, p.d)) != 0ull) {
                if (A.sc.use_bvh) {
                    // rtw_device.h traverse_wave; volumes first, per lane, as in traverse<>
                    bool vol = false;
                    if (TEX && A.sc.n_vol > 0) vol = volume_pass<Rng<KIND>, false>(A.sc, p.o, p.d, A.sc.ray_tmin, p.ray_time, gt, g, th, prim);
                    traverse_wave(A.sc, s_stack + (tid >> 6) * (uint32_t)A.sc.stack_depth, 1u, true, p.o, p.d, A.sc.ray_tmin, p.ray_time, gt, th, prim, vol);
                } else if (TEX && A.sc.n_vol > 0) {
                    traverse_brute<Rng<KIND>, false, false>(A.sc, p.o, p.d, A.sc.ray_tmin, 1.e27f, p.ray_time, gt, g, th, prim);
                } else {
                    traverse_brute<NoRng, false, true>(A.sc, p.o, p.d, A.sc.ray_tmin, 1.e27f, p.ray_time, gt, ng, th, prim);
                }
            }
            v3 so, sd, att, radiance;
            Nee nee;
            uint32_t nee_prev = 0u;
            const int ev = shade_a<KIND, TEX>(A.sc, g, p.o, p.d, gt, th, prim, so, sd, att, radiance, nee, noise_lds, nee_prev);
            if (nee_prev) p.gk |= kNeePrev;
            n_seg++;
            if (nee.has) {
                n_shadow++;
                bool fogged = false;
                if (TEX && A.sc.n_vol > 0) {  // the probe's volume share, in draw order (see k_shade)
                    float tv = nee.tmax;
                    int pv = -1;
                    fogged = volume_pass<Rng<KIND>, true>(A.sc, so, nee.dir, nee.tmin, 0.0f, gt, g, tv, pv);
                }
                if (fogged) nee.has = false;
                else { p.ldir = nee.dir; p.ltmax = nee.tmax; p.c = vmul(nee.rad, p.T); }
            }
            const bool alive = shade_b<KIND>(0u, A.max_depth, g, ev, so, sd, att, radiance, p.o, p.d, p.T, p.L, TEX == 2 && A.sc.estimator == RTW_EST_MIXTURE);
            p.a = g.a; p.b = g.b;
            if (alive) {
                p.ray_time = (KIND == RTW_RNG_TEA_LCG || A.sc.has_motion) ? g.ray_time(1u) : 0.0f;
                p.b = g.b;
            } else if (nee.has) {
                p.gk |= kZombie;
            } else {
                finish_path(A, path_id, p.L);
                keep = false;
            }
        }
        }
        compact_store(A, s_cursor, keep, p);
    }
    cursor_publish(A, s_cursor, n_seg, n_shadow, RTW_K_FIRST);
}

// ------------------------------------------------------------------ k_trace
// Surfaces only: no intersection program that draws random numbers runs here. In scenes with media the volume
// primitives are tested by the shading kernels, which hold the generator (volume_pass in rtw_device.h).
// DUAL = 1: small static scenes, both rays share one walk over the scalar-cache candidate lists (64 VGPRs, 8 waves);
// DUAL = 0: BVH / moving-sphere scenes, one traversal per ray (the 64-byte node records want the larger budget).
#ifndef RTW_TRACE_BVH_WAVES
#define RTW_TRACE_BVH_WAVES 6
#endif
template <bool DUAL>
__global__ void __launch_bounds__(kBlock, DUAL ? 8 : RTW_TRACE_BVH_WAVES) k_trace(const KArgs A) {
    extern __shared__ uint32_t s_stack[];
    RTW_WORKLIST_SHARED
    const uint32_t tid = threadIdx.x;
    const TravMem tm = trav_mem(A.sc, s_stack, kBlock, tid);
    uint32_t n_rays = 0;
    const WorkList wl = worklist_init(A.cnt_in, A.n_regions, s_pref, s_part);
    // the work-list lookup of the next chunk (a dependent chain of LDS reads) is issued behind this chunk's loads
    uint32_t vc = blockIdx.x, region = 0, chunk = 0, n_valid = 0;
    if (vc < wl.total_chunks) worklist_lookup(wl, A.n_regions, vc, region, chunk, n_valid);
    while (vc < wl.total_chunks) {
        const bool valid = tid < n_valid;
        const size_t slot = (size_t)region * A.region_cap + chunk * kBlock + tid;
        Path p;
        p.gk = 0; p.ltmax = -1.f;
        if (valid) load_trace_part(A.in, slot, p, A.sc.n_lights > 0);
        vc += gridDim.x;
        if (vc < wl.total_chunks) worklist_lookup(wl, A.n_regions, vc, region, chunk, n_valid);
        if (valid) trace_path<DUAL>(A, p, slot, tm, n_rays);
    }
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if ((tid & 63u) == 0 && n_rays) atomicAdd(&stat_row(A)[2 + RTW_K_TRACE], (unsigned long long)n_rays);
}

// ------------------------------------------------------------------ k_trace_bvh
// BVH scenes: the number of traversal steps differs widely between the rays of a wave, so a wave that takes 64
// rays and waits for the longest one idles most of its lanes. Here every WAVE owns a stream of 256-slot chunks
// of the work-list and a lane that has finished its path (radiance ray, then the queued shadow probe) takes the
// next slot of the stream as soon as kRefillIdle lanes are waiting (ballot + mbcnt ranks, no atomics).
// Results are per-slot and independent of the order in which slots are taken.
#ifndef RTW_REFILL_IDLE
#define RTW_REFILL_IDLE 16
#endif
#ifndef RTW_LEAF_BIAS
#define RTW_LEAF_BIAS 1
#endif
#ifndef RTW_LEAF_BIAS_DEN
#define RTW_LEAF_BIAS_DEN 1
#endif
// MODE 0: 32-bit stack entries (trees beyond 8 K references; the step with branches); 1: 16-bit entries, nodes beyond the
// LDS image come from global memory; 2: 16-bit entries and every node in LDS (the walk loop then holds no global load,
// so the stores of finished rays and the loads of a refill are never waited for inside it)
template <int BLOCK, int MODE>
__global__ void __launch_bounds__(BLOCK, RTW_TRACE_BVH_WAVES) k_trace_bvh(const KArgs A) {
    extern __shared__ uint32_t s_stack[];
    RTW_WORKLIST_SHARED_T(BLOCK)
    const uint32_t tid = threadIdx.x;
    const TravMem tm = trav_mem(A.sc, s_stack, BLOCK, tid);
    const WorkList wl = worklist_init<BLOCK>(A.cnt_in, A.n_regions, s_pref, s_part);
    constexpr uint32_t kWaves = BLOCK / 64;
    const uint32_t stride = gridDim.x * kWaves;
    uint32_t vc = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + (tid >> 6));  // this wave's next chunk
    uint32_t chunk_n = 0, next = 0;   // wave-uniform: size of the current chunk, slots of it already handed out
    size_t chunk_base = 0;
    bool exhausted = false;
    uint32_t n_rays = 0;
    // per-lane path and walk state
    bool active = false;
    size_t slot = 0;
    v3 o = V(0.f, 0.f, 0.f), d = o, inv = o, ldir = o;
    float ltmax = -1.f, tmin = 0.f, ray_time = 0.f, best_t = 0.f, th = 0.f, gt = 0.f;
    int best_prim = -1, prim = -1, sp = 0;
    uint32_t cur = kBvhDone, pend = 0, occl = 0;
    bool shadow_phase = false;
    const uint32_t root = A.sc.n_tree > 0 ? 0u : kBvhDone;
#ifdef RTW_TRACE_COUNT
    uint32_t c_inner = 0, c_prim = 0, c_outer = 0, c_winner = 0, c_wleaf = 0, c_rays = 0;
    // wave cycles by part of the loop: 0 refill, 1 inner steps, 2 leaf step, 3 finished rays
    unsigned long long tc[4] = {0, 0, 0, 0}, tc_t0 = __builtin_amdgcn_s_memtime();
#define RTW_TC(I_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tc[I_] += now_ - tc_t0; tc_t0 = now_; }
#else
#define RTW_TC(I_)
#endif
    for (;;) {
#ifdef RTW_TRACE_COUNT
        c_outer++;
#endif
        RTW_TC(3)
        const unsigned long long idle = __ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u || (n_idle >= (uint32_t)RTW_REFILL_IDLE && !exhausted)) {
            unsigned long long need = idle;
            while (need != 0ull && !exhausted) {
                if (next >= chunk_n) {
                    if (vc >= wl.total_chunks) { exhausted = true; break; }
                    uint32_t region, chunk, n_valid;
                    worklist_lookup(wl, A.n_regions, vc, region, chunk, n_valid);
                    vc += stride;
                    chunk_base = (size_t)region * A.region_cap + (size_t)chunk * kBlock;
                    chunk_n = n_valid;
                    next = 0;
                    continue;
                }
                const uint32_t avail = chunk_n - next;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                const bool wants = ((need >> (tid & 63u)) & 1ull) != 0ull;
                if (wants && rank < avail) {
                    slot = chunk_base + next + rank;
                    Path p;
                    load_trace_part(A.in, slot, p, A.sc.n_lights > 0);
                    o = p.o; ldir = p.ldir; ltmax = p.ltmax; gt = gather_time_of(A, p.gk);
                    th = 0.f; prim = -1; occl = 0;
                    const bool do_r = !(p.gk & kZombie);
                    if (do_r) {
                        d = p.d; tmin = A.sc.ray_tmin; best_t = 1.e27f; ray_time = p.ray_time; shadow_phase = false;
                    } else {
                        d = ldir; tmin = A.sc.probe_eps; best_t = ltmax; ray_time = 0.0f; shadow_phase = true;
                    }
                    inv = recip3(d);
                    best_prim = -1; sp = 0; cur = root; pend = 0u;
                    if (do_r || ltmax >= 0.0f) {
                        n_rays++;  // the kernel's unit: one path slot (radiance ray + queued probe)
#ifdef RTW_TRACE_COUNT
                        c_rays++;
#endif
                        active = true;
                    } else {
                        A.hit_out[slot] = make_uint2(0u, 0u);  // a zombie without a probe (does not occur; kept total)
                    }
                }
                next += min((uint32_t)__popcll(need), avail);
                need = __ballot(!active);
            }
            if (__ballot(active) == 0ull) break;
        }
        // One kind of step at a time, chosen by majority: while the lanes standing at inner nodes are not
        // outnumbered they keep stepping down (a tight loop); then the lanes holding a leaf test its primitives.
        // Executing only the majority's branch keeps most lanes busy whatever the mix; the minority waits and grows
        // until it is the majority. A lane that comes to a leaf puts it aside (pend) and walks on with the next
        // node of its stack: it only has to wait at its second leaf. The nodes it visits meanwhile are culled
        // against a best_t that has not seen the leaf yet - a few more visits, never a different result.
        RTW_TC(0)
        const uint32_t n_act = (uint32_t)__popcll(__ballot(active));
#define RTW_SET_ASIDE if (active && pend == 0u && ((cur & 3u) - 1u) < 2u) { pend = cur; cur = bvh_pop(tm, sp); }
        RTW_SET_ASIDE
        bool at_inner = active && (cur & 3u) == 0u;
        for (;;) {
            const uint32_t n_in = (uint32_t)__popcll(__ballot(at_inner));
            if (n_in == 0u || n_in * (uint32_t)RTW_LEAF_BIAS < (n_act - n_in) * (uint32_t)RTW_LEAF_BIAS_DEN) break;
            if (at_inner) {
                if (MODE == 0) {
                    cur = bvh_inner_step(A.sc, tm, o, inv, tmin, best_t, cur, sp);
                    RTW_SET_ASIDE
                } else {
                    bvh_step16<MODE == 2>(A.sc, tm, o, inv, tmin, best_t, cur, pend, sp);
                }
                at_inner = (cur & 3u) == 0u;
#ifdef RTW_TRACE_COUNT
                c_inner++;
#endif
            }
#ifdef RTW_TRACE_COUNT
            c_winner++;
#endif
        }
        RTW_TC(1)
        const bool at_leaf = active && pend != 0u;
#ifdef RTW_TRACE_COUNT
        if (__ballot(at_leaf) != 0ull) c_wleaf++;
#endif
        if (at_leaf) {
            uint32_t lslot = pend >> 2;
            const uint32_t cnt = pend & 3u;
            pend = 0u;
            bool stop = false;
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)RTW_LEAF_MAX; k++) {
                if (k < cnt && !stop) {
#ifdef RTW_TRACE_COUNT
                    c_prim++;
#endif
                    int pi;
                    float t;
                    if (leaf_test(A.sc, tm, lslot, o, d, inv, tmin, ray_time, gt, t, pi, lslot)) {
                        // same acceptance rule as traverse<>: closest, ties to the lowest primitive index; any hit ends a probe
                        if (t < best_t || (!shadow_phase && t == best_t && best_prim >= 0 && pi < best_prim)) {
                            best_t = t; best_prim = pi;
                            stop = shadow_phase;
                        }
                    }
                }
            }
            if (stop) cur = kBvhDone;
            RTW_SET_ASIDE
        }
#undef RTW_SET_ASIDE
        RTW_TC(2)
        if (active) {
            if (cur == kBvhDone && pend == 0u) {
                // this ray is done
                if (!shadow_phase) {
                    th = best_t; prim = best_prim;
                    if (ltmax >= 0.0f) {  // traceOcclusion, closehit.cu:16-42 (ray time 0)
                        d = ldir; inv = recip3(d); tmin = A.sc.probe_eps; best_t = ltmax; ray_time = 0.0f;
                        best_prim = -1; sp = 0; cur = root; shadow_phase = true;
#ifdef RTW_TRACE_COUNT
                        c_rays++;
#endif
                    } else {
                        A.hit_out[slot] = make_uint2(__float_as_uint(th), (uint32_t)(prim + 1));
                        active = false;
                    }
                } else {
                    occl = (best_prim >= 0 ? 0x80000000u : 0u) | 0x40000000u;  // bit 30: a probe was queued
                    A.hit_out[slot] = make_uint2(__float_as_uint(th), (uint32_t)(prim + 1) | occl);
                    active = false;
                }
            }
        }
    }
#ifdef RTW_TRACE_COUNT
    for (int off = 32; off > 0; off >>= 1) { c_inner += __shfl_down(c_inner, off); c_prim += __shfl_down(c_prim, off); c_rays += __shfl_down(c_rays, off); }
    RTW_TC(3)
    if ((tid & 63u) == 0) for (int q = 0; q < 4; q++) atomicAdd(&A.stats[kStatRows * 8 + 2 + q], tc[q]);
    if ((tid & 63u) == 0) { atomicAdd(&A.stats[kStatRows * 8 + 0], (unsigned long long)c_winner); atomicAdd(&A.stats[kStatRows * 8 + 1], (unsigned long long)c_wleaf); atomicAdd(&A.stats[6], (unsigned long long)c_inner); atomicAdd(&A.stats[7], (unsigned long long)c_prim); atomicAdd(&A.stats[2 + RTW_K_BOUNCE], (unsigned long long)c_outer); atomicAdd(&A.stats[kStatRows * 8 + 6], (unsigned long long)c_rays); }
#endif
#undef RTW_TC
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off);
    if ((tid & 63u) == 0 && n_rays) atomicAdd(&stat_row(A)[2 + RTW_K_TRACE], (unsigned long long)n_rays);
}

// ------------------------------------------------------------------ k_shade
#ifndef RTW_SHADE_WAVES
#define RTW_SHADE_WAVES RTW_MIN_WAVES
#endif
// the instantiations with the cold features (textures, media, the corrected estimators) are allocated for 5 waves per SIMD (96 VGPRs
// instead of 111): k_shade waits on its loads, and the smaller allocation fits beside more of the other lane's k_trace_bvh waves.
// Round 3, same box, bit-exact: scene 4 2 072 -> 2 127 Msamples/s (k_shade 0.094 -> 0.085 s), scene 2 3 554 -> 3 629; at 6 waves
// the spills cost 11-15 %. k_first's cold instantiations stay at 4: at 5 scene 2 gains another 2.5 %, scene 4 loses 2 %.
#ifndef RTW_SHADE_COLD_WAVES
#define RTW_SHADE_COLD_WAVES 5
#endif
// RTW_SHADE_SORT=1 (experiment, off): deal a chunk's paths to the threads by hit material. Measured: k_shade 4-7 % SLOWER on
// scenes 1, 2, 4 with one lane or two. Round 3 repeated it with the material class carried in the hit record (no lookup at all) and
// the next chunk's hit records fetched a chunk ahead: still 4 % slower on scene 1, 1-2 % on scenes 2 and 4 (bit-exact). The kernel
// waits on its path loads (61 % of its wave-cycles, rocprofv3 round 3) and issues VALU for 22 % of them: the serialised material
// branches hide behind the loads, and the sort's two barriers and permuted loads do not.
#if !defined(RTW_SHADE_SORT) || !defined(RTW_EXPERIMENTS)
#undef RTW_SHADE_SORT
#define RTW_SHADE_SORT 0
#endif
template <int KIND, int TEX>
__global__ void __launch_bounds__(kBlock, TEX ? RTW_SHADE_COLD_WAVES : RTW_SHADE_WAVES) k_shade(const KArgs A) {
    RTW_WORKLIST_SHARED
    RTW_CURSOR_SHARED
    RTW_NOISE_SHARED
#if RTW_SHADE_SORT
    __shared__ uint32_t s_sort_cnt[8 * (kBlock / 64)];
    __shared__ uint16_t s_sort_perm[kBlock];
#endif
    const uint32_t tid = threadIdx.x;
    cursor_init(s_cursor);
    const uint32_t* noise_lds = stage_noise<(TEX != 0)>(A.sc, s_noise);
    uint32_t n_seg = 0, n_shadow = 0;
    const WorkList wl = worklist_init(A.cnt_in, A.n_regions, s_pref, s_part);
    // the work-list lookup of the next chunk (a dependent chain of LDS reads) is issued behind this chunk's loads
    uint32_t vc = blockIdx.x, region = 0, chunk = 0, n_valid = 0;
    if (vc < wl.total_chunks) worklist_lookup(wl, A.n_regions, vc, region, chunk, n_valid);
    while (vc < wl.total_chunks) {
        const bool valid = tid < n_valid;
        bool keep = false;
        Path p;
        p.gk = 0; p.ltmax = -1.f;
        uint2 h = make_uint2(0u, 0u);
#if RTW_SHADE_SORT
        // The chunk's paths are dealt to the threads by the material they hit (a counting sort over 8 keys through LDS:
        // hit record -> material type -> rank by ballot, two barriers), so that a wave shades one or two kinds of vertex
        // instead of all of them one after the other: unsorted, k_shade ran at a lane utilisation of 0.29 on scene 1.
        // Which thread shades which path cannot show in the image (a path's draws are its own).
        {
            const size_t base_slot = (size_t)region * A.region_cap + chunk * kBlock;
            uint32_t key = 7u;  // threads beyond the chunk's paths go last
            if (valid) {
                const uint2 h0 = A.hit[base_slot + tid];
                const int prim0 = (int)(h0.y & 0x3fffffffu) - 1;
                key = prim0 < 0 ? 6u : min((uint32_t)*as_const(&A.sc.hitrec[prim0].mat_type), 5u);
            }
            uint32_t rank = 0u, mine = 0u;
            for (uint32_t k = 0; k < 8u; k++) {
                const unsigned long long m = __ballot(key == k);
                if ((tid & 63u) == 0u) s_sort_cnt[k * (kBlock / 64) + (tid >> 6)] = (uint32_t)__popcll(m);
                if (key == k) { rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); mine = k * (kBlock / 64) + (tid >> 6); }
            }
            __syncthreads();
            uint32_t before = 0u;
            for (uint32_t j = 0; j < mine; j++) before += s_sort_cnt[j];  // key-major, wave-minor: <= 31 broadcast reads
            s_sort_perm[before + rank] = (uint16_t)tid;
            __syncthreads();
            if (valid) {
                const size_t slot = base_slot + s_sort_perm[tid];
                load_shade_part(A.in, slot, p, A.sc.n_lights > 0);
                h = A.hit[slot];
            }
            // (no barrier is needed before the next chunk's sort: its counts are written while at most s_sort_perm is still being
            // read, and its first barrier stands before s_sort_perm is written again)
        }
#else
        if (valid) {
            const size_t slot = (size_t)region * A.region_cap + chunk * kBlock + tid;
            load_shade_part(A.in, slot, p, A.sc.n_lights > 0);
            h = A.hit[slot];
        }
#endif
        vc += gridDim.x;
        if (vc < wl.total_chunks) worklist_lookup(wl, A.n_regions, vc, region, chunk, n_valid);
        if (valid) {
            // the light sample queued by the previous bounce (closehit.cu:103-113), now that its probe is back
            if ((h.y & 0xc0000000u) == 0x40000000u) p.L = vadd(p.L, p.c);
            if (p.gk & kZombie) {
                finish_path(A, path_id_of<KIND>(A, p), p.L);
            } else {
                Rng<KIND> g;
                rng_from_path<KIND>(g, A.seed, p);
                const float gt = gather_time_of(A, p.gk);
                int prim = (int)(h.y & 0x3fffffffu) - 1;
                float th = __uint_as_float(h.x);
                if (TEX && A.sc.n_vol > 0) {
                    // scenes with media: k_trace found the closest surface; the volumes are tested here, where the
                    // generator lives, before this segment's closest-hit draws - the reference's order
                    float tv = 1.e27f;
                    int pv = -1;
                    if (volume_pass<Rng<KIND>, false>(A.sc, p.o, p.d, A.sc.ray_tmin, p.ray_time, gt, g, tv, pv) && !(prim >= 0 && th < tv)) { th = tv; prim = pv; }
                }
                v3 so, sd, att, radiance;
                Nee nee;
                uint32_t nee_prev = (p.gk & kNeePrev) ? 1u : 0u;
                const int ev = shade_a<KIND, TEX>(A.sc, g, p.o, p.d, gt, th, prim, so, sd, att, radiance, nee, noise_lds, nee_prev);
                p.gk = (p.gk & ~kNeePrev) | (nee_prev ? kNeePrev : 0u);
                n_seg++;
                p.ltmax = -1.0f;
                if (nee.has) {
                    n_shadow++;
                    bool fogged = false;
                    if (TEX && A.sc.n_vol > 0) {  // the probe's volume share (any hit), right after the light-sample draws
                        float tv = nee.tmax;
                        int pv = -1;
                        fogged = volume_pass<Rng<KIND>, true>(A.sc, so, nee.dir, nee.tmin, 0.0f, gt, g, tv, pv);
                    }
                    if (fogged) nee.has = false;
                    else { p.ldir = nee.dir; p.ltmax = nee.tmax; p.c = vmul(nee.rad, p.T); }
                }
                const bool alive = shade_b<KIND>(A.depth, A.max_depth, g, ev, so, sd, att, radiance, p.o, p.d, p.T, p.L, TEX == 2 && A.sc.estimator == RTW_EST_MIXTURE);
                if (ev == EV_HIT) p.o = so;  // a queued probe starts at the hit point even when the path stops here
                p.a = g.a; p.b = g.b;
                if (alive) {
                    p.ray_time = (KIND == RTW_RNG_TEA_LCG || A.sc.has_motion) ? g.ray_time(A.depth + 1u) : 0.0f;
                    p.b = g.b;
                    keep = true;
                } else if (nee.has) {
                    p.gk |= kZombie;
                    keep = true;
                } else {
                    finish_path(A, path_id_of<KIND>(A, p), p.L);
                }
            }
        }
        compact_store(A, s_cursor, keep, p);
    }
    cursor_publish(A, s_cursor, n_seg, n_shadow, RTW_K_SHADE);
}

// ------------------------------------------------------------------ k_bounce (fused)
template <int KIND, int TEX>
__global__ void __launch_bounds__(kBlock, RTW_MIN_WAVES) k_bounce(const KArgs A) {
    extern __shared__ uint32_t s_stack[];
    RTW_WORKLIST_SHARED
    RTW_CURSOR_SHARED
    RTW_NOISE_SHARED
    const uint32_t tid = threadIdx.x;
    cursor_init(s_cursor);
    const uint32_t* noise_lds = stage_noise<(TEX != 0)>(A.sc, s_noise);
    const TravMem tm = trav_mem(A.sc, s_stack, kBlock, tid);
    uint32_t n_seg = 0, n_shadow = 0;
    const WorkList wl = worklist_init(A.cnt_in, A.n_regions, s_pref, s_part);
    for (uint32_t vc = blockIdx.x; vc < wl.total_chunks; vc += gridDim.x) {
        uint32_t region, chunk, n_valid;
        worklist_lookup(wl, A.n_regions, vc, region, chunk, n_valid);
        const uint32_t idx = chunk * kBlock + tid;
        const bool valid = tid < n_valid;
        bool keep = false;
        Path p;
        p.gk = 0; p.ltmax = -1.f;
        if (valid) {
            load_path(A.in, (size_t)region * A.region_cap + idx, p, A.sc.n_lights > 0);
            Rng<KIND> g;
            rng_from_path<KIND>(g, A.seed, p);
            const float gt = gather_time_of(A, p.gk);
            if (p.ltmax >= 0.0f) {  // a probe queued by k_first / k_shade: its volume share was decided there, with the
                                    // generator in the reference's draw order; what is left is the surfaces
                float st;
                int sprim;
                traverse<Rng<KIND>, true, true>(A.sc, p.o, p.ldir, A.sc.probe_eps, p.ltmax, 0.0f, gt, g, tm, st, sprim);
                if (sprim < 0) p.L = vadd(p.L, p.c);
                p.ltmax = -1.0f;
            }
            if (p.gk & kZombie) {
                finish_path(A, path_id_of<KIND>(A, p), p.L);
            } else {
                // A.n_iter consecutive bounces in registers: the thin tail of a batch is latency-bound, so fewer,
                // longer launches win; scenes with volumes run every bounce here (n_iter = 1 while wide).
                uint32_t depth = A.depth;
                bool alive = false;
                uint32_t nee_prev = (p.gk & kNeePrev) ? 1u : 0u;
                for (uint32_t it = 0; it < A.n_iter; it++) {
                    float t;
                    int prim;
                    traverse<Rng<KIND>, false, false>(A.sc, p.o, p.d, A.sc.ray_tmin, 1.e27f, p.ray_time, gt, g, tm, t, prim);
                    v3 so, sd, att, radiance;
                    Nee nee;
                    const int ev = shade_a<KIND, TEX>(A.sc, g, p.o, p.d, gt, t, prim, so, sd, att, radiance, nee, noise_lds, nee_prev);
                    n_seg++;
                    if (nee.has) {
                        float st;
                        int sprim;
                        traverse<Rng<KIND>, true, false>(A.sc, so, nee.dir, nee.tmin, nee.tmax, 0.0f, gt, g, tm, st, sprim);
                        n_shadow++;
                        if (sprim < 0) radiance = vadd(radiance, nee.rad);
                    }
                    alive = shade_b<KIND>(depth, A.max_depth, g, ev, so, sd, att, radiance, p.o, p.d, p.T, p.L, TEX == 2 && A.sc.estimator == RTW_EST_MIXTURE);
                    depth++;
                    if (!alive) break;
                    p.ray_time = (KIND == RTW_RNG_TEA_LCG || A.sc.has_motion) ? g.ray_time(depth) : 0.0f;
                }
                p.a = g.a; p.b = g.b;
                p.gk = (p.gk & ~kNeePrev) | (nee_prev ? kNeePrev : 0u);
                if (alive) keep = true;
                else finish_path(A, path_id_of<KIND>(A, p), p.L);
            }
        }
        compact_store(A, s_cursor, keep, p);
    }
    cursor_publish(A, s_cursor, n_seg, n_shadow, RTW_K_BOUNCE);
}

// position gi of the job order -> pixel group: the three lists of k_classify, longest units first
RTW_DEV uint32_t order_lookup(const KArgs& A, uint32_t gi) {
    const uint32_t n_groups = (A.npix + 63u) / 64u;
    const uint32_t c2 = A.order_counts[0], c1 = A.order_counts[1];
    const uint32_t idx = gi < c2 ? gi : (gi < c2 + c1 ? n_groups + (gi - c2) : 2u * n_groups + (gi - c2 - c1));
    return __builtin_amdgcn_readfirstlane(A.order[idx]);
}

// ------------------------------------------------------------------ k_classify
// Scheduling hint for k_path's job queue, nothing more (no result depends on the order). A launch ends when its slowest
// unit ends, and a unit's length follows its pixel: 64 samples through a glass sphere take ~3x the segments of 64 samples
// on a wall, and a pixel that looks past the scene one segment per sample. Longest-first order keeps the tail short:
//   class 2  a quarter of the group's pixel-centre camera rays meet a specular surface first (metal, dielectric: long chains)
//   class 1  some ray can reach the scene
//   class 0  every ray stays outside the scene bounds (one-segment paths)
// order[] holds three lists (class 2 at 0, class 1 at n_groups, class 0 at 2 n_groups) and counters[] their lengths; the
// queue serves them in that order. Measured on the metric workload's 1/8 shard: the launch's fixed cost (tail) fell from
// 5.6 ms to [see DESIGN.md].
template <bool WALK>
__global__ void __launch_bounds__(kBlock) k_classify(const KArgs A, uint32_t* __restrict__ order, uint32_t* __restrict__ counters, uint32_t n_groups) {
    __shared__ u32x4 s_walk[WALK ? kWalkMaxWords : 1];
    if (WALK) {
        for (uint32_t i = threadIdx.x; i < (uint32_t)A.sc.n_walk_words; i += kBlock) s_walk[i] = A.sc.walk[i];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (kBlock / 64u);
    for (uint32_t g = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6); g < n_groups; g += waves) {
        const uint32_t p_local = g * 64u + lane;
        bool may = false, specular = false;
        if (p_local < A.npix) {
            const uint32_t yl = fastdiv(p_local, A.divw_m, A.divw_s1, A.divw_s2);
            const uint32_t x = p_local - yl * A.width, y = A.row0 + yl * A.row_stride;
            const rtw_camera& cam = A.sc.cam;
            if (A.sc.cam_type != RTW_CAM_PERSPECTIVE || cam.lens_radius != 0.0f) {
                may = true;  // (no cheap classification: everything counts as ordinary)
            } else {
                const float s = ((float)x + 0.5f) / (float)A.width, t = ((float)y + 0.5f) / (float)A.height;
                const v3 o = ld3(cam.origin);
                const v3 d = vsub(vfma(ld3(cam.vertical), t, vfma(ld3(cam.horizontal), s, ld3(cam.lower_left))), o);
                may = may_hit_scene(A.sc, o, d);
                if (WALK && may) {
                    float th;
                    int prim;
                    walk_lds<false>(s_walk, A.sc.n_groups, o, d, A.sc.ray_tmin, 1.e27f, th, prim);
                    if (prim >= 0) {
                        const int mt = A.sc.hitrec[prim].mat_type;
                        specular = mt == RTW_MAT_METAL || mt == RTW_MAT_DIELECTRIC;
                    }
                }
            }
        }
        const bool any = __ballot(may) != 0ull;
        const uint32_t n_spec = (uint32_t)__popcll(__ballot(specular));
        if (lane == 0) {
            const uint32_t cls = !any ? 0u : (n_spec >= 16u ? 2u : 1u);
            order[(2u - cls) * n_groups + atomicAdd(&counters[2u - cls], 1u)] = g;
        }
    }
}

// ------------------------------------------------------------------ k_path
// Small scenes (the candidate lists): the whole path lives in registers. A lane owns one UNIT at a time - one pixel,
// unit_blocks consecutive blocks of kSumBlock samples - and walks the unit's paths one segment per loop iteration: camera
// ray when it has none (regeneration), closest hit, closest-hit / miss program, the light sample's shadow probe, roulette.
// A finished sample is added to the current block's sum in sample order; a finished block stores its 16-byte sum; after the
// unit's last block the lane takes the next unit of its wave's stream (ballot + mbcnt ranks, no atomics). A wave's stream
// is a queue of jobs (64 neighbouring pixels x units_per_job units), one returning atomic per job, so lanes never wait for
// each other: every iteration runs with all 64 lanes on live paths until the launch runs dry. Nothing but the block sums
// reaches HBM. The image does not depend on which lane ran which unit: a block's sum is a function of (pixel, block) alone,
// and the per-pixel sums are taken block by block in order (k_resolve_blocks) - the summation order of the arithmetic spec.
// The host runs a pass as two overlapping launches (rtw_hip.hip): the bulk in 4-block units, the last blocks one by one.
constexpr uint32_t kSumBlock = RTW_SUM_BLOCK;
constexpr uint32_t kSumUnitBlocks = RTW_SUM_UNIT_BLOCKS;
constexpr int kPathMaxPrims = 64;  // k_path walks the brute lists only: scenes of at most this many primitives
// RTW_MARK: a phase fence (see RTW_MARK2 above); the diagnostic build -DRTW_PHASE_TIMERS (never shipped) also accumulates
// s_memtime deltas per phase and wave there (printed by the host)
#if defined(RTW_PHASE_TIMERS)
#define RTW_MARK(name) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph_cyc[ph_cur] += now_ - ph_t0; ph_t0 = now_; ph_cur = rtw_phase_id(name); }
RTW_DEV constexpr int rtw_phase_id(const char* n) { return n[0] == 'r' && n[2] == 'f' ? 0 : n[0] == 'r' ? 1 : n[0] == 'w' && n[5] == 'r' ? 2 : n[0] == 's' && n[6] == 'a' ? 3 : n[0] == 'w' ? 4 : 5; }
#else
#define RTW_MARK(name) asm volatile("; MARK " name)
#endif
// waves per SIMD the register allocation aims at, for the instantiation without the cold features. Under the default scheduler: 5
// (<= 96 VGPRs; +9 % over 4 on the metric workload, 6 added nothing). Round 3: the product build compiles this kernel's translation
// unit with LLVM's register-pressure-first scheduler (-mllvm -amdgpu-sched-strategy=iterative-minreg, __graft_entry__.py UNIT_FLAGS)
// and -DRTW_PATH_WAVES=6: 80 VGPRs + 48 B of scratch, 0.5292 s per render of the metric frame against 0.5565 (the phase fences
// still pay under it: without them 0.549 at 5 waves). The cold instantiation needs the registers more than the occupancy
// (the environment-camera scenes lose 27 % at 5 waves: left at RTW_MIN_WAVES; scenes with media take MEDIA5 below)
#ifndef RTW_PATH_WAVES
#define RTW_PATH_WAVES 5
#endif
#ifndef RTW_PATH_COLD_WAVES
#define RTW_PATH_COLD_WAVES RTW_MIN_WAVES
#endif
// MEDIA5: the cold instantiation once more, allocated for 5 waves per SIMD: scenes with media wait on the volume records' loads and
// gain from the occupancy (BASELINE config 4: +4.5 %), the other users of the cold features (other cameras, corrected
// estimators, thin lenses) are VALU-bound like the hot instantiation and lose up to 27 % to the spills
template <int KIND, int TEX, int MEDIA5 = 0>
__global__ void __launch_bounds__(kBlock, TEX ? (MEDIA5 ? 5 : RTW_PATH_COLD_WAVES) : RTW_PATH_WAVES) k_path(const KArgs A) {
    RTW_NOISE_SHARED
    __shared__ u32x4 s_hitrec[kPathMaxPrims * 6];
    const uint32_t tid = threadIdx.x;
    const uint32_t* noise_lds = stage_noise<(TEX != 0)>(A.sc, s_noise);
    {   // the hit records of a small scene live in LDS for the whole launch
        const u32x4* src = (const u32x4*)A.sc.hitrec;
        for (uint32_t i = tid; i < (uint32_t)A.sc.n_prims * 6u; i += kBlock) s_hitrec[i] = src[i];
    }
    // running sum of the lane's current summation unit (the sums of its finished blocks, in order): touched once per 16 samples,
    // so it lives in LDS and costs the loop no register
    __shared__ float s_usum[3][kBlock];
    __shared__ PathConsts s_pc;
    if (tid == 0) {
        for (int q = 0; q < 3; q++) {
            s_pc.cam_o[q] = A.sc.cam.origin[q]; s_pc.cam_ll[q] = A.sc.cam.lower_left[q];
            s_pc.cam_h[q] = A.sc.cam.horizontal[q]; s_pc.cam_v[q] = A.sc.cam.vertical[q];
        }
        for (int q = 0; q < 5; q++) s_pc.rect[q] = A.sc.pdf.rect[q];
        int gen = A.sc.pdf.gen;
        if (gen == RTW_PDF_MIXTURE || gen == RTW_PDF_MIXTURE_BIAS) gen = A.sc.pdf.p1_gen;
        s_pc.gen = gen;
        if (A.sc.n_lights > 0) {
            for (int q = 0; q < 3; q++) { s_pc.lnrm[q] = A.sc.lights[0].normal[q]; s_pc.lemi[q] = A.sc.lights[0].emission[q]; }
            s_pc.larea = A.sc.lights[0].area;
        }
        s_pc.npix = A.npix; s_pc.width = A.width; s_pc.row0 = A.row0; s_pc.row_stride = A.row_stride;
        s_pc.divs_m = A.divs_m; s_pc.divs_s1 = A.divs_s1; s_pc.divs_s2 = A.divs_s2; s_pc.spp = A.spp;
        s_pc.bs_lo = (uint32_t)(uint64_t)A.blocksum; s_pc.bs_hi = (uint32_t)((uint64_t)A.blocksum >> 32);
        s_pc.unit_shift = A.unit_sums ? 3u : 0u; s_pc.pad3 = 0u;
        s_pc.n_jobs = A.n_jobs; s_pc.n_ranges = A.n_ranges; s_pc.units_per_job = A.units_per_job; s_pc.unit_blocks = A.unit_blocks;
        s_pc.n_blocks_pass = A.n_blocks_pass; s_pc.block0 = A.block0; s_pc.divw_m = A.divw_m; s_pc.divw_s1 = A.divw_s1; s_pc.divw_s2 = A.divw_s2;
        s_pc.q_lo = (uint32_t)(uint64_t)A.queue; s_pc.q_hi = (uint32_t)((uint64_t)A.queue >> 32); s_pc.pad4 = 0u;
    }
    // the LDS camera serves the perspective camera without a lens (the reference's only camera); other kinds read the arguments
    // (the host sends other camera kinds and thin lenses through the cold instantiation, which reads the arguments)
    const PathConsts* pc_cam = TEX ? nullptr : &s_pc;
    __shared__ u32x4 s_walk[TEX ? 1 : kWalkMaxWords];
    if (!TEX) for (uint32_t i = tid; i < (uint32_t)A.sc.n_walk_words; i += kBlock) s_walk[i] = A.sc.walk[i];
    __syncthreads();
    const uint32_t lane = tid & 63u;
    // wave-uniform: the job stream
    uint32_t u_next = 0, u_end = 0, job_g = 0, job_b = 0;
    bool exhausted = false;
    // per lane, the unit: pixel (x | y << 16 of the full image), block of the pass, current sample (relative to sample_offset)
    bool need = true;
    uint32_t pxy = 0, blk = 0, blk_end = 0, s_cur = 0;
    v3 usum = V(0.f, 0.f, 0.f);
    // per lane, the path. The instantiation without the cold features (TEX == false) has no motion, no media and the
    // reference estimator: ray time, gather time and the ray-time stream are dead there and take no registers.
    bool alive = false;
    uint32_t depth = 0, rng_a = 0;
    uint32_t gk = 0, rng_b = 0, nee_prev = 0;
    float ray_time = 0.f;
    v3 o = V(0.f, 0.f, 0.f), d = o, T = o, L = o;
    uint32_t n_seg = 0, n_shadow = 0;
#ifdef RTW_PHASE_TIMERS
    unsigned long long ph_cyc[6] = {0, 0, 0, 0, 0, 0}, ph_t0 = __builtin_amdgcn_s_memtime();
    int ph_cur = 0;
#endif
    for (;;) {
        RTW_MARK("refill");
        unsigned long long need_mask = __ballot(need);
        while (need_mask != 0ull && !exhausted) {
            const PathConsts* kc = &s_pc;  // (launch constants from the LDS page: see PathConsts)
            if (u_next >= u_end) {
                uint32_t q = 0;
                if (lane == 0) q = atomicAdd((uint32_t*)(((uint64_t)kc->q_hi << 32) | kc->q_lo), 1u);
                q = __builtin_amdgcn_readfirstlane(q);
                if (q >= (uint32_t)__builtin_amdgcn_readfirstlane((int)kc->n_jobs)) { exhausted = true; break; }
                const uint32_t n_ranges = (uint32_t)__builtin_amdgcn_readfirstlane((int)kc->n_ranges), upj = (uint32_t)__builtin_amdgcn_readfirstlane((int)kc->units_per_job);
                const uint32_t gi = q / n_ranges;
                job_g = order_lookup(A, gi);
                job_b = (q - gi * n_ranges) * upj;
                u_next = 0; u_end = 64u * upj;
                continue;
            }
            const uint32_t avail = u_end - u_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
            if (need && rank < avail) {
                const uint32_t ub = kc->unit_blocks, nbp = kc->n_blocks_pass;
                const uint32_t u = u_next + rank;
                const uint32_t b = (job_b + (u >> 6)) * ub;  // first block of the unit (blocks of this pass)
                const uint32_t p_local = job_g * 64u + (u & 63u);
                if (p_local < kc->npix && b < nbp) {
                    need = false;
                    blk = b;
                    blk_end = min(b + ub, nbp);
                    const uint32_t yl = fastdiv(p_local, kc->divw_m, kc->divw_s1, kc->divw_s2);
                    pxy = (p_local - yl * kc->width) | ((kc->row0 + yl * kc->row_stride) << 16);
                    s_cur = (kc->block0 + b) * kSumBlock;
                    usum = V(0.f, 0.f, 0.f);
                    alive = false;
                }
            }
            u_next += min((uint32_t)__popcll(need_mask), avail);
            need_mask = __ballot(need);
        }
        if (__ballot(!need) == 0ull) break;
        RTW_MARK("regen");
        const bool busy = !need;
        const uint32_t px = pxy & 0xffffu, py = pxy >> 16;
        Rng<KIND> g;
        if (busy && !alive) {  // regeneration: the next camera path of this lane's unit
            Path p;
            raygen<KIND, !TEX>(A, px, py, A.sample0 + s_cur, 0u, p, g, pc_cam);
            o = p.o; d = p.d; T = p.T; L = p.L; rng_a = p.a;
            if (TEX) { ray_time = p.ray_time; gk = p.gk; rng_b = p.b; nee_prev = 0; }
            depth = 0; alive = true;
        }
        if (busy) {
            const uint32_t pixel = A.width * py + px;
            if (KIND == RTW_RNG_TEA_LCG) g.init(A.seed, 0, 0, rng_a, rng_b);
            else g.init(A.seed, pixel, A.sample0 + s_cur, rng_a, A.sample0 + s_cur);
        }
        RTW_MARK("walk_r");
        const float gt = TEX ? gather_time_of(A, gk) : 0.0f;
        // closest hit (raygen.cu:41-54). Camera rays of neighbouring pixels that all look past the scene skip the walk.
        float th = 1.e27f;
        int prim = -1;
        const bool walk = busy && (depth > 0u || may_hit_scene(A.sc, o, d));
        if (__ballot(walk) != 0ull) {
            if (busy) {
                if (TEX) traverse_brute<Rng<KIND>, false, false>(A.sc, o, d, A.sc.ray_tmin, 1.e27f, ray_time, gt, g, th, prim);
                else walk_lds<false>(s_walk, A.sc.n_groups, o, d, A.sc.ray_tmin, 1.e27f, th, prim);
            }
        }
        RTW_MARK("shade_a");
#ifdef RTW_SUBPHASE_TIMERS
        rtw_sub_stamp(8);   // 8: shade_a entry (miss test, branch to the hit code)
#endif
        if (busy) {
            v3 so, sd, att, radiance;
            Nee nee;
            const int ev = shade_a<KIND, TEX>(A.sc, g, o, d, gt, th, prim, so, sd, att, radiance, nee, noise_lds, nee_prev, s_hitrec, &s_pc);
            n_seg++;
#ifdef RTW_SUBPHASE_TIMERS
            rtw_sub_stamp(0);  // 0: everything outside the closest-hit program
#endif
            RTW_MARK("walk_s");
            if (nee.has) {  // traceOcclusion, closehit.cu:16-42
                float st;
                int sprim;
                if (TEX) traverse_brute<Rng<KIND>, true, false>(A.sc, so, nee.dir, nee.tmin, nee.tmax, 0.0f, gt, g, st, sprim);
                else walk_lds<true>(s_walk, A.sc.n_groups, so, nee.dir, nee.tmin, nee.tmax, st, sprim);
                n_shadow++;
                if (sprim < 0) radiance = vadd(radiance, nee.rad);
            }
            RTW_MARK("shade_b");
            alive = shade_b<KIND>(depth, A.max_depth, g, ev, so, sd, att, radiance, o, d, T, L, TEX == 2 && A.sc.estimator == RTW_EST_MIXTURE);
            depth++;
            rng_a = g.a;
            if (alive) {
                if (TEX) {
                    ray_time = (KIND == RTW_RNG_TEA_LCG || A.sc.has_motion) ? g.ray_time(depth) : 0.0f;
                    rng_b = g.b;
                }
            } else {
                // removeNaNs (raygen.cu:17-24), then the unit's running sum, in sample order
                usum = vadd(usum, V((L.x == L.x) ? L.x : 0.f, (L.y == L.y) ? L.y : 0.f, (L.z == L.z) ? L.z : 0.f));
                s_cur++;
                if ((s_cur % kSumBlock) == 0u || s_cur >= A.spp) {  // a block is complete: its sum goes out, the unit moves on
                    const PathConsts* kc = &s_pc;  // (launch constants from the LDS page: see PathConsts)
                    const uint32_t c_row0 = kc->row0, c_stride = kc->row_stride, c_width = kc->width, c_npix = kc->npix;
                    const uint32_t yl_ = c_stride > 1 ? fastdiv(py - c_row0, kc->divs_m, kc->divs_s1, kc->divs_s2) : py - c_row0;
                    const uint32_t pix = yl_ * c_width + px;
                    const uint32_t b_done = blk;
                    blk++;
                    need = blk >= blk_end || s_cur >= A.spp;
                    float4* const bsum_base = (float4*)(((uint64_t)kc->bs_hi << 32) | kc->bs_lo);
                    // the spec's second level: block sums add up in order inside aligned units of kSumUnitBlocks blocks (the launch's
                    // first block is unit-aligned). A launch whose lane units are whole summation units (A.unit_sums: shift 3) keeps
                    // the unit's running sum in LDS and stores the unit's sum, slot [block / 8][pixel]; any other stores every
                    // block's sum, slot [block][pixel] (0 + x = x: the same bits as storing x).
                    const uint32_t ush = kc->unit_shift;
                    v3 prev = V(0.f, 0.f, 0.f);
                    if (ush != 0u && (b_done % kSumUnitBlocks) != 0u) prev = V(s_usum[0][tid], s_usum[1][tid], s_usum[2][tid]);
                    const v3 u = vadd(prev, usum);
                    if (ush == 0u || need || (blk % kSumUnitBlocks) == 0u) bsum_base[(size_t)(b_done >> ush) * c_npix + pix] = make_float4(u.x, u.y, u.z, 0.f);
                    else { s_usum[0][tid] = u.x; s_usum[1][tid] = u.y; s_usum[2][tid] = u.z; }
                    usum = V(0.f, 0.f, 0.f);
                }
            }
        }
    }
#ifdef RTW_PHASE_TIMERS
    RTW_MARK("refill");
    if (lane == 0) for (int q = 0; q < 6; q++) atomicAdd(&A.stats[kStatRows * 8 + q], ph_cyc[q]);
#endif
    for (int off = 32; off > 0; off >>= 1) {
        n_seg += __shfl_down(n_seg, off);
        n_shadow += __shfl_down(n_shadow, off);
    }
    if (lane == 0) {
        unsigned long long* row = stat_row(A);
        if (n_seg) { atomicAdd(&row[0], (unsigned long long)n_seg); atomicAdd(&row[2 + RTW_K_PATH], (unsigned long long)n_seg); }
        if (n_shadow) atomicAdd(&row[1], (unsigned long long)n_shadow);
    }
}

#ifdef RTW_EXPERIMENTS  // measured slower than the default paths: built by scripts/build_variant.sh only, never by build()
// ------------------------------------------------------------------ k_path_tree
// Tree scenes, same idea as k_path (paths in registers, lanes own units of one pixel x one sample block and regenerate,
// only the unit sums reach HBM), but a ray's walk through the BVH takes a different number of steps in every lane, so a
// lane is a small state machine and the wave executes ONE kind of step at a time, chosen by vote:
//   inner step   lanes standing at an inner node test its two children (bvh_inner_step, LDS stack column per lane)
//   leaf step    lanes standing at a leaf test one of its primitives
//   shade step   lanes whose radiance ray has come back run the closest-hit / miss program, roulette, and - when the
//                sample ends - add it to the unit and start the next camera path; it is the expensive step, so it waits
//                until RTW_TREE_SHADE_AT lanes want it (or nobody can do anything else)
// A path alternates: [shadow probe of the previous vertex, if a light sample is pending] -> radiance ray -> shade.
// The probe's contribution c = f * Le * w * T is held in registers and added to L when the probe comes back free: the
// same floating-point order as the wavefront kernels (and the oracle). Media are tested in the shade step, where the
// generator is, exactly as k_shade does.
#ifndef RTW_TREE_SHADE_AT
#define RTW_TREE_SHADE_AT 40
#endif
#ifndef RTW_TREE_WAVES
#define RTW_TREE_WAVES RTW_MIN_WAVES
#endif
template <int KIND, int TEX>
__global__ void __launch_bounds__(kBlock, RTW_TREE_WAVES) k_path_tree(const KArgs A) {
    extern __shared__ uint32_t s_stack[];
    RTW_NOISE_SHARED
    const uint32_t tid = threadIdx.x;
    const uint32_t* noise_lds = stage_noise<(TEX != 0)>(A.sc, s_noise);
    const TravMem tm = trav_mem(A.sc, s_stack, kBlock, tid);
    const uint32_t lane = tid & 63u;
    const uint32_t root = A.sc.n_tree > 0 ? 0u : kBvhDone;
    // wave-uniform: the job stream
    uint32_t u_next = 0, u_end = 0, job_g = 0, job_b = 0;
    bool exhausted = false;
    // per lane: the unit
    bool need = true;
    uint32_t pxy = 0, blk = 0, blk_end = 0, s_cur = 0;
    v3 usum = V(0.f, 0.f, 0.f);
    // per lane: the path
    enum { PH_SHADE = 0, PH_PROBE = 1, PH_RAY = 2 };  // what the lane waits for: the shade step, or its walk (probe / radiance ray)
    uint32_t phase = PH_SHADE;
    bool alive = false;      // false in PH_SHADE: the sample is over (or none was started): finish it and regenerate
    bool fresh = true;       // no sample to finish (a new unit)
    uint32_t depth = 0, rng_a = 0, rng_b = 0, gk = 0, nee_prev = 0;
    float ray_time = 0.f;
    v3 o = V(0.f, 0.f, 0.f), d = o, T = o, L = o;
    v3 ldir = o, c = o;      // pending light sample: probe direction, contribution
    float ltmax = -1.f;
    // per lane: the walk
    uint32_t cur = kBvhDone, pend = 0;
    int sp = 0, best_prim = -1;
    float best_t = 0.f, tmin = 0.f, wtime = 0.f;
    v3 wd = o, inv = o;
    uint32_t n_seg = 0, n_shadow = 0;
#ifdef RTW_PHASE_TIMERS
    unsigned long long ph_cyc[6] = {0, 0, 0, 0, 0, 0}, ph_t0 = __builtin_amdgcn_s_memtime();
    int ph_cur = 0;
    unsigned long long st_cnt[3] = {0, 0, 0};
#endif
    for (;;) {
        RTW_MARK("refill");
        // ---- units
        unsigned long long need_mask = __ballot(need);
        while (need_mask != 0ull && !exhausted) {
            if (u_next >= u_end) {
                uint32_t q = 0;
                if (lane == 0) q = atomicAdd(A.queue, 1u);
                q = __builtin_amdgcn_readfirstlane(q);
                if (q >= A.n_jobs) { exhausted = true; break; }
                const uint32_t gi = q / A.n_ranges;
                job_g = order_lookup(A, gi);
                job_b = (q - gi * A.n_ranges) * A.units_per_job;
                u_next = 0; u_end = 64u * A.units_per_job;
                continue;
            }
            const uint32_t avail = u_end - u_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
            if (need && rank < avail) {
                const uint32_t u = u_next + rank;
                const uint32_t b = (job_b + (u >> 6)) * A.unit_blocks;
                const uint32_t p_local = job_g * 64u + (u & 63u);
                if (p_local < A.npix && b < A.n_blocks_pass) {
                    need = false;
                    blk = b;
                    blk_end = min(b + A.unit_blocks, A.n_blocks_pass);
                    const uint32_t yl = fastdiv(p_local, A.divw_m, A.divw_s1, A.divw_s2);
                    pxy = (p_local - yl * A.width) | ((A.row0 + yl * A.row_stride) << 16);
                    s_cur = (A.block0 + b) * kSumBlock;
                    usum = V(0.f, 0.f, 0.f);
                    phase = PH_SHADE; alive = false; fresh = true;
                }
            }
            u_next += min((uint32_t)__popcll(need_mask), avail);
            need_mask = __ballot(need);
        }
        const unsigned long long busy_mask = __ballot(!need);
        if (busy_mask == 0ull) break;
        const bool busy = !need;
        // ---- vote
        const bool want_shade = busy && phase == PH_SHADE;
        const uint32_t n_busy = (uint32_t)__popcll(busy_mask);
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(want_shade));
        if (n_shade >= (uint32_t)RTW_TREE_SHADE_AT || n_shade == n_busy) {
            // ---- shade step
            RTW_MARK("shade_a");
#ifdef RTW_PHASE_TIMERS
            st_cnt[0]++;
#endif
            if (want_shade) {
                const uint32_t px = pxy & 0xffffu, py = pxy >> 16;
                const uint32_t pixel = A.width * py + px;
                Rng<KIND> g;
                if (alive) {
                    if (KIND == RTW_RNG_TEA_LCG) g.init(A.seed, 0, 0, rng_a, rng_b);
                    else g.init(A.seed, pixel, A.sample0 + s_cur, rng_a, A.sample0 + s_cur);
                    const float gt = gather_time_of(A, gk);
                    int prim = best_prim;
                    float th = best_t;
                    if (TEX && A.sc.n_vol > 0) {  // media: tested here, before this segment's closest-hit draws (see k_shade)
                        float tv = 1.e27f;
                        int pv = -1;
                        if (volume_pass<Rng<KIND>, false>(A.sc, o, d, A.sc.ray_tmin, ray_time, gt, g, tv, pv) && !(prim >= 0 && th < tv)) { th = tv; prim = pv; }
                    }
                    v3 so, sd, att, radiance;
                    Nee nee;
                    const int ev = shade_a<KIND, TEX>(A.sc, g, o, d, gt, th, prim, so, sd, att, radiance, nee, noise_lds, nee_prev);
                    n_seg++;
                    ltmax = -1.0f;
                    if (nee.has) {
                        n_shadow++;
                        bool fogged = false;
                        if (TEX && A.sc.n_vol > 0) {  // the probe's volume share (any hit), right after the light-sample draws
                            float tv = nee.tmax;
                            int pv = -1;
                            fogged = volume_pass<Rng<KIND>, true>(A.sc, so, nee.dir, nee.tmin, 0.0f, gt, g, tv, pv);
                        }
                        if (!fogged) { ldir = nee.dir; ltmax = nee.tmax; c = vmul(nee.rad, T); }
                    }
                    alive = shade_b<KIND>(depth, A.max_depth, g, ev, so, sd, att, radiance, o, d, T, L, TEX == 2 && A.sc.estimator == RTW_EST_MIXTURE);
                    if (ev == EV_HIT) o = so;  // a pending probe starts at the hit point even when the path stops here
                    depth++;
                    rng_a = g.a; rng_b = g.b;
                    if (alive) {
                        ray_time = (KIND == RTW_RNG_TEA_LCG || A.sc.has_motion) ? g.ray_time(depth) : 0.0f;
                        rng_b = g.b;
                    }
                    fresh = false;
                    if (ltmax >= 0.0f) phase = PH_PROBE;        // (a dead path with a probe pending comes back here once it is traced)
                    else if (alive) phase = PH_RAY;
                }
                if (phase == PH_SHADE) {
                    // the sample is over (or the unit is new): removeNaNs (raygen.cu:17-24), the unit's running sum, the next camera path
                    if (!fresh) {
                        usum = vadd(usum, V((L.x == L.x) ? L.x : 0.f, (L.y == L.y) ? L.y : 0.f, (L.z == L.z) ? L.z : 0.f));
                        s_cur++;
                        if ((s_cur % kSumBlock) == 0u || s_cur >= A.spp) {
                            const uint32_t yl_ = A.row_stride > 1 ? fastdiv(py - A.row0, A.divs_m, A.divs_s1, A.divs_s2) : py - A.row0;
                            A.blocksum[(size_t)blk * A.npix + (yl_ * A.width + px)] = make_float4(usum.x, usum.y, usum.z, 0.f);
                            usum = V(0.f, 0.f, 0.f);
                            blk++;
                            need = blk >= blk_end || s_cur >= A.spp;
                        }
                    }
                    if (!need) {
                        Path p;
                        raygen<KIND>(A, px, py, A.sample0 + s_cur, 0u, p, g);
                        o = p.o; d = p.d; T = p.T; L = p.L; rng_a = p.a; rng_b = p.b; ray_time = p.ray_time; gk = p.gk;
                        depth = 0; nee_prev = 0; alive = true; fresh = false; ltmax = -1.0f;
                        phase = PH_RAY;
                    }
                }
                // start the walk the lane now waits for
                if (!need) {
                    if (phase == PH_PROBE) { wd = ldir; tmin = A.sc.probe_eps; best_t = ltmax; wtime = 0.0f; }
                    else { wd = d; tmin = A.sc.ray_tmin; best_t = 1.e27f; wtime = ray_time; }
                    inv = recip3(wd);
                    best_prim = -1; sp = 0; cur = root;
                }
            }
            continue;
        }
        // ---- walk step: inner nodes while the lanes standing at one are not outnumbered, then the primitives of the leaves
        // put aside (a lane that comes to a leaf walks on with the next node of its stack: see k_trace_bvh)
        RTW_MARK("walk_r");
        const bool walking = busy && phase != PH_SHADE;
        const uint32_t n_walk = n_busy - n_shade;
#define RTW_SET_ASIDE if (walking && pend == 0u && ((cur & 3u) - 1u) < 2u) { pend = cur; cur = bvh_pop(tm, sp); }
        RTW_SET_ASIDE
        bool at_inner = walking && (cur & 3u) == 0u;
        for (;;) {
            const uint32_t n_in = (uint32_t)__popcll(__ballot(at_inner));
            if (n_in == 0u || n_in * (uint32_t)RTW_LEAF_BIAS < (n_walk - n_in) * (uint32_t)RTW_LEAF_BIAS_DEN) break;
            if (at_inner) {
                if (tm.wide) {
                    cur = bvh_inner_step(A.sc, tm, o, inv, tmin, best_t, cur, sp);
                    RTW_SET_ASIDE
                } else {
                    bvh_step16<false>(A.sc, tm, o, inv, tmin, best_t, cur, pend, sp);
                }
                at_inner = (cur & 3u) == 0u;
            }
#ifdef RTW_PHASE_TIMERS
            st_cnt[1]++;
#endif
        }
#ifdef RTW_PHASE_TIMERS
        st_cnt[2]++;
#endif
        RTW_MARK("walk_s");
        if (walking && pend != 0u) {
            uint32_t lslot = pend >> 2;
            const uint32_t cnt = pend & 3u;
            pend = 0u;
            bool stop = false;
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)RTW_LEAF_MAX; k++) {
                if (k < cnt && !stop) {
                    int pi;
                    float t;
                    if (leaf_test(A.sc, tm, lslot, o, wd, inv, tmin, wtime, gather_time_of(A, gk), t, pi, lslot)) {
                        // same acceptance rule as traverse<>: closest, ties to the lowest primitive index; any hit ends a probe
                        if (t < best_t || (phase == PH_RAY && t == best_t && best_prim >= 0 && pi < best_prim)) {
                            best_t = t; best_prim = pi;
                            stop = phase == PH_PROBE;
                        }
                    }
                }
            }
            if (stop) cur = kBvhDone;
            RTW_SET_ASIDE
        }
#undef RTW_SET_ASIDE
        RTW_MARK("shade_b");
        if (walking && cur == kBvhDone && pend == 0u) {
            if (phase == PH_PROBE) {
                // traceOcclusion (closehit.cu:16-42) is back: a free path to the light adds the held contribution
                if (best_prim < 0) L = vadd(L, c);
                ltmax = -1.0f;
                if (alive) {
                    phase = PH_RAY;
                    wd = d; tmin = A.sc.ray_tmin; best_t = 1.e27f; wtime = ray_time;
                    inv = recip3(wd);
                    best_prim = -1; sp = 0; cur = root;
                } else {
                    phase = PH_SHADE;  // the path ended at that vertex: finish the sample
                }
            } else {
                phase = PH_SHADE;      // best_t / best_prim hold the closest hit
            }
        }
    }
#ifdef RTW_PHASE_TIMERS
    RTW_MARK("refill");
    if (lane == 0) { for (int q = 0; q < 6; q++) atomicAdd(&A.stats[kStatRows * 8 + q], ph_cyc[q]); atomicAdd(&A.stats[kStatRows * 8 + 6], st_cnt[1]); atomicAdd(&A.stats[kStatRows * 8 + 7], st_cnt[0] * 1000000ull + st_cnt[2]); }
#endif
    for (int off = 32; off > 0; off >>= 1) {
        n_seg += __shfl_down(n_seg, off);
        n_shadow += __shfl_down(n_shadow, off);
    }
    if (lane == 0) {
        unsigned long long* row = stat_row(A);
        if (n_seg) { atomicAdd(&row[0], (unsigned long long)n_seg); atomicAdd(&row[2 + RTW_K_PATH], (unsigned long long)n_seg); }
        if (n_shadow) atomicAdd(&row[1], (unsigned long long)n_shadow);
    }
}

#endif  // RTW_EXPERIMENTS

#ifndef RTW_TEMPLATES_ONLY  // (the translation units that only instantiate the shading kernels leave the plain kernels to rtw_hip.hip)
// per-pixel sums of one k_path pass in the arithmetic spec's order (rtw.h: samples in order inside blocks, block sums in order
// inside aligned units of kSumUnitBlocks blocks, unit sums in order). slots: n_unit_slots whole unit sums [unit][pixel], then
// n_block_slots block sums [block][pixel] whose first block (index first_block of the render call) is unit-aligned.
__global__ void __launch_bounds__(kBlock) k_resolve_blocks(const float4* __restrict__ slots, float4* __restrict__ accum, uint32_t npix, uint32_t n_unit_slots,
                                                           uint32_t n_block_slots, uint32_t first_block) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        float4 a = accum[i];
        for (uint32_t u = 0; u < n_unit_slots; u++) {
            const float4 l = slots[(size_t)u * npix + i];
            a.x += l.x; a.y += l.y; a.z += l.z;
        }
        float ux = 0.f, uy = 0.f, uz = 0.f;
        for (uint32_t b = 0; b < n_block_slots; b++) {
            const float4 l = slots[(size_t)(n_unit_slots + b) * npix + i];
            ux += l.x; uy += l.y; uz += l.z;
            if (((first_block + b + 1u) % kSumUnitBlocks) == 0u || b + 1u == n_block_slots) {
                a.x += ux; a.y += uy; a.z += uz;
                ux = 0.f; uy = 0.f; uz = 0.f;
            }
        }
        accum[i] = a;
    }
}

// One a-trous pass (rtw.h rtw_denoise): 25 taps in row-major order, clamped at the borders, plain fp32 in a fixed order
__global__ void __launch_bounds__(kBlock) k_atrous(const float4* __restrict__ in, float4* __restrict__ out, int width, int height, int step, float inv_sigma2) {
    const int n = width * height;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / width, x = i - y * width;
        const float4 c = in[i];
        const float kern[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
        float sr = 0.f, sg = 0.f, sb = 0.f, sw = 0.f;
        for (int dy = -2; dy <= 2; dy++) {
            int yy = y + dy * step;
            yy = yy < 0 ? 0 : (yy > height - 1 ? height - 1 : yy);
            for (int dx = -2; dx <= 2; dx++) {
                int xx = x + dx * step;
                xx = xx < 0 ? 0 : (xx > width - 1 ? width - 1 : xx);
                const float4 q = in[yy * width + xx];
                const float dr = c.x - q.x, dg = c.y - q.y, db = c.z - q.z;
                const float d2 = (dr * dr + dg * dg) + db * db;
                const float w = (kern[dy + 2] * kern[dx + 2]) / (1.0f + d2 * inv_sigma2);
                sr = sr + w * q.x; sg = sg + w * q.y; sb = sb + w * q.z; sw = sw + w;
            }
        }
        out[i] = make_float4(sr / sw, sg / sw, sb / sw, c.w);
    }
}

// Sums the S sample slots of every pixel in the arithmetic spec's order (rtw.h RTW_SUM_BLOCK, RTW_SUM_UNIT_BLOCKS): slot s is
// sample first_sample + s of the render call; samples add up in order inside aligned blocks of kSumBlock (running block sum in
// `part`), a finished block's sum is added to the running unit sum (`upart`), a finished unit's (kSumUnitBlocks blocks) to
// `accum`. Fixed order => reproducible bits, whatever the batch size.
__global__ void __launch_bounds__(kBlock) k_resolve(const float4* __restrict__ lbuf, float4* __restrict__ accum, float4* __restrict__ upart, float4* __restrict__ part,
                                                    uint32_t npix, uint32_t nslots, uint32_t first_sample) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        float4 a = accum[i], u = upart[i], b = part[i];
        for (uint32_t s = 0; s < nslots; s++) {
            const uint32_t rel = first_sample + s;
            if (rel != 0u && (rel % kSumBlock) == 0u) {
                u.x += b.x; u.y += b.y; u.z += b.z;
                b = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((rel % (kSumBlock * kSumUnitBlocks)) == 0u) {
                    a.x += u.x; a.y += u.y; a.z += u.z;
                    u = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            float4 l = lbuf[(size_t)s * npix + i];
            b.x += l.x; b.y += l.y; b.z += l.z;
        }
        accum[i] = a;
        upart[i] = u;
        part[i] = b;
    }
}

// mean radiance: the last (possibly partial) block's sum joins its unit's, the last unit's sum the total, then the division by spp
__global__ void __launch_bounds__(kBlock) k_finish(const float4* __restrict__ accum, const float4* __restrict__ upart, const float4* __restrict__ part,
                                                   float4* __restrict__ out, uint32_t npix, float spp) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        float4 a = accum[i];
        if (part != nullptr) {
            float4 u = upart[i];
            const float4 b = part[i];
            u.x += b.x; u.y += b.y; u.z += b.z;
            a.x += u.x; a.y += u.y; a.z += u.z;
        }
        out[i] = make_float4(a.x / spp, a.y / spp, a.z / spp, 1.0f);
    }
}

__global__ void __launch_bounds__(kBlock) k_debug_intersect(const DScene sc, const float* __restrict__ rays, const float* __restrict__ ray_time,
                                                           const float* __restrict__ gather_time, int n, float* __restrict__ out_t,
                                                           int32_t* __restrict__ out_prim, uint32_t stack_stride) {
    extern __shared__ uint32_t s_stack[];
    const TravMem tm = trav_mem(sc, s_stack, stack_stride, threadIdx.x);  // before the early exit: it holds a barrier
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 8 * (size_t)i;
    NoRng g;
    float t;
    int prim;
    traverse<NoRng, false, true>(sc, V(r[0], r[1], r[2]), V(r[3], r[4], r[5]), r[6], r[7], ray_time ? ray_time[i] : 0.f,
                                 gather_time ? gather_time[i] : 0.f, g, tm, t, prim);
    out_t[i] = t;
    out_prim[i] = prim;
}

#endif  // RTW_TEMPLATES_ONLY

}  // namespace rtwk
