// rtw_hip.hip — wavefront Monte-Carlo path tracer for MI355X (gfx950) behind the C ABI of include/rtw.h.
//
// What the reference does in ONE OptiX megakernel launch (Director.cpp:982-984: raygen -> traverse ->
// closest-hit/miss -> callables, one thread per pixel, one sample) is done here as a wavefront loop:
//
//   per batch of S samples per pixel (P = pixels*S paths in flight, sized for HBM not for cache):
//     k_bounce<FIRST>   generate primary rays in registers, trace+shade segment 0
//     k_bounce  x (max_depth-1)   one launch per bounce: load 64 B path state (4 coalesced 16 B planes),
//                        closest hit, material scatter, light sample + shadow probe, Russian roulette,
//                        then wave64 ballot/popcount compaction of the survivors into the other
//                        ping-pong buffer; finished paths drop their radiance into lbuf[path]
//     k_resolve         sums the S sample slots of each pixel in sample order (deterministic)
//   k_finish            mean radiance -> float4 framebuffer tile
//
// No OptiX, no CUDA shims, no Triton, no MFMA (divergent scalar fp32). Results do not depend on
// scheduling: every path owns a counter-based RNG stream and its own lbuf slot.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtw.h"
#include "rtw_bvh.h"
#include "rtw_device.h"

using namespace rtwdev;

namespace {

#ifndef RTW_MIN_WAVES
#define RTW_MIN_WAVES 1  // __launch_bounds__ second argument: minimum waves per SIMD the register allocator must allow
#endif
constexpr int kBlock = 256;                 // 4 wave64 per workgroup
constexpr uint32_t kMinRegionCap = 64 * kBlock;  // a region holds at least 64 chunks of 256 paths
constexpr uint32_t kMaxRegions = 2048;           // region counters scanned in LDS by every workgroup
constexpr int kBruteMaxPrims = 24;          // at or below: scalar-cache brute force; above: BVH

struct BounceArgs {
    DScene sc;
    const float4* in0; const float4* in1; const float4* in2; const uint4* in3;
    float4* out0; float4* out1; float4* out2; uint4* out3;
    float4* lbuf;
    const uint32_t* cnt_in;
    uint32_t* cnt_out;
    unsigned long long* stats;
    uint32_t n_regions, n_paths, npix, width, height, row0, sample0, seed, depth, max_depth, stack_stride, region_cap, n_iter, pad;
};

// ---------------------------------------------------------------------------------------------
// One path segment: optixTraverse + closest-hit / miss + the tail of rayColor's loop body
// (raygen/raygen.cu:36-84, shaders/closehit.cu:45-121, miss/miss.cu:8-30).
// Returns true when the path continues into the next bounce.
// Stage 1 of a segment: the radiance ray's closest hit (optixTraverse, raygen.cu:41-54).
template <int KIND>
RTW_DEV void trace_stage(const BounceArgs& A, const uint32_t depth, Rng<KIND>& g, const v3 origin, const v3 dir, float gather_time,
                         uint32_t* stack, float& t, int& prim, float& ray_time) {
    const DScene& sc = A.sc;
    ray_time = 0.0f;
    if (KIND == RTW_RNG_TEA_LCG || sc.has_motion) ray_time = g.ray_time(depth);  // raygen.cu:48
    traverse<Rng<KIND>, false, false>(sc, origin, dir, 1e-6f, 1.e27f, ray_time, gather_time, g, stack, A.stack_stride, t, prim);
}

// Stage 2: closest-hit / miss programs + the tail of rayColor's loop body.
template <int KIND>
RTW_DEV bool shade_stage(const BounceArgs& A, const uint32_t depth, Rng<KIND>& g, v3& origin, v3& dir, v3& T, v3& L, float gather_time,
                         uint32_t* stack, uint32_t& n_shadow, const float t, const int prim, const float ray_time) {
    const DScene& sc = A.sc;
    v3 radiance = V(0.f, 0.f, 0.f);
    int ev;
    v3 att = V(0.f, 0.f, 0.f), so = origin, sd = dir;
    if (prim < 0) {
        if (sc.sky_light) {  // miss.cu:8-21
            v3 u = normalize3(dir);
            float tt = 0.5f * (u.y + 1.0f);
            float w = 1.0f - tt;
            radiance = V(fma_(tt, 0.5f, w), fma_(tt, 0.7f, w), fma_(tt, 1.0f, w));
        }
        ev = EV_MISS;
    } else {
        const HitRec hr = load_hitrec(sc, prim);
        v3 hp, hn;
        hit_attributes(sc, hr, prim, origin, dir, t, gather_time, hp, hn);
        const int mtype = hr.mat_type;
        const float mparam = hr.param;
        const v3 tex = V(hr.r, hr.g, hr.b);
        const int bsdf_eval = hr.bsdf_eval;
        bool specular = false;
        if (mtype == RTW_MAT_LAMBERTIAN) {
            // lambertianMaterial.cu:41-71, onb.cuh:20-32, sampling.cuh:49-60 (Q1)
            v3 w = normalize3(hn);
            v3 a = (w.x > 0.9f || w.x < -0.9f) ? V(0.f, 1.f, 0.f) : V(1.f, 0.f, 0.f);
            v3 v = normalize3(cross3(w, a));
            v3 u = cross3(w, v);
            float r1 = g.next1();
            float r2 = g.next1();
            float sn, cs;
            sincos2pi(r1, sn, cs);
            float sq = __builtin_sqrtf(r2);
            float lx = (cs * 2.0f) * sq;
            float ly = (sn * 2.0f) * sq;
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
        } else if (mtype == RTW_MAT_DIFFUSE_LIGHT) {
            // diffuseLight.cu:48-69
            if (dot3(hn, dir) < 0.0f) radiance = tex;
            ev = EV_CANCEL;
        } else if (mtype == RTW_MAT_METAL) {
            // metalMaterial.cu:32-64 (Q5)
            specular = true;
            v3 refl = reflect3(dir, hn);
            v3 ball = random_in_unit_sphere(g);
            v3 sdir = normalize3(vfma(ball, mparam, refl));
            so = hp; sd = sdir;
            att = tex;
            ev = (dot3(sdir, hn) <= 0.0f) ? EV_CANCEL : EV_HIT;
        } else if (mtype == RTW_MAT_DIELECTRIC) {
            // dielectricMaterial.cu:37-114
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
        } else if (mtype == RTW_MAT_ISOTROPIC) {
            // isotropicMaterial.cu:30-51 (Q14)
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

        // next-event estimation, closehit.cu:70-118
        const int nl = sc.n_lights;
        if (ev == EV_HIT && !specular && nl > 0) {
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
            } else {
                const RTW_CONST rtw_light* lt = as_const(sc.lights);
                lnrm = V(lt->normal[0], lt->normal[1], lt->normal[2]);
                lemi = V(lt->emission[0], lt->emission[1], lt->emission[2]);
                larea = lt->area;
            }
            int gen = sc.pdf.gen;  // mixturePdf.cu:25-38: always child p1 (Q4)
            if (gen == RTW_PDF_MIXTURE || gen == RTW_PDF_MIXTURE_BIAS) gen = sc.pdf.p1_gen;
            float lpdf = 0.0f, ldist = 0.0f;
            v3 ldir = V(0.f, 0.f, 0.f), lem = V(0.f, 0.f, 0.f);
            if (gen == RTW_PDF_RECT_X || gen == RTW_PDF_RECT_Y || gen == RTW_PDF_RECT_Z) {
                // rectPdf.cu:124-193
                float ra = g.next1();
                float rb = g.next1();
                float pa = fma_(ra, sc.pdf.rect[1] - sc.pdf.rect[0], sc.pdf.rect[0]);
                float pb = fma_(rb, sc.pdf.rect[3] - sc.pdf.rect[2], sc.pdf.rect[2]);
                float k = sc.pdf.rect[4];
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
            if (lpdf > 0.0f && bsdf_eval == 0) {
                // lambertianMaterial.cu:74-81
                v3 f = vscale(att, RTW_1_PI_F);
                float ndl = dot3(ldir, hn);
                float bpdf = __builtin_fmaxf(0.0f, ndl * RTW_1_PI_F);
                if (0.0f < bpdf && (f.x != 0.0f || f.y != 0.0f || f.z != 0.0f)) {
                    const float eps = 500 * 1.0e-7f;
                    float st;
                    int sprim;
                    traverse<Rng<KIND>, true, false>(sc, so, ldir, eps, ldist - eps, 0.0f, gather_time, g, stack, A.stack_stride, st, sprim);
                    n_shadow++;
                    if (sprim < 0) {
                        float a2 = lpdf * lpdf;
                        float weight = a2 / fma_(bpdf, bpdf, a2);  // raydata.cuh:167-171
                        float k = (weight * ndl) / lpdf;
                        radiance = vadd(radiance, vscale(vmul(f, lem), k));
                    }
                }
            }
        }
    }
    L = V(fma_(radiance.x, T.x, L.x), fma_(radiance.y, T.y, L.y), fma_(radiance.z, T.z, L.z));  // raygen.cu:60
    if (ev != EV_HIT) return false;
    origin = so; dir = sd;
    T = vmul(T, att);
    if (2u <= depth) {  // raygen.cu:74-82
        float p = __builtin_fmaxf(__builtin_fmaxf(T.x, T.y), T.z);
        if (p < g.next1()) return false;
        T = vscale(T, 1.0f / p);
    }
    return depth + 1u < A.max_depth;
}

// ---------------------------------------------------------------------------------------------
template <int KIND, bool FIRST>
__global__ void __launch_bounds__(kBlock, RTW_MIN_WAVES) k_bounce(const BounceArgs A) {
    extern __shared__ uint32_t s_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t* my_stack = s_stack + tid;
    uint32_t n_seg = 0, n_shadow = 0;

    // Work list. Region r holds cnt_in[r] live paths = ceil(cnt/256) chunks. Every workgroup scans the
    // (<= kMaxRegions) counters into an LDS prefix array once, then strides over the virtual chunk
    // ids: no empty iterations, perfect balance, and a launch with nothing alive costs one scan.
    __shared__ uint32_t s_pref[kMaxRegions + 1];
    __shared__ uint32_t s_raw[kMaxRegions];
    __shared__ uint32_t s_part[kBlock];
    uint32_t total_chunks;
    const uint32_t chunks_per_region = A.region_cap / kBlock;
    if (FIRST) {
        total_chunks = (A.n_paths + kBlock - 1) / kBlock;
    } else {
        constexpr uint32_t kPer = kMaxRegions / kBlock;
        uint32_t loc[kPer];
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t r = tid * kPer + j;
            const uint32_t raw = r < A.n_regions ? A.cnt_in[r] : 0u;
            s_raw[r] = raw;
            const uint32_t c = (raw + kBlock - 1) / kBlock;
            loc[j] = sum;
            sum += c;
        }
        s_part[tid] = sum;
        __syncthreads();
        for (uint32_t off = 1; off < kBlock; off <<= 1) {
            uint32_t v = tid >= off ? s_part[tid - off] : 0u;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        const uint32_t excl = s_part[tid] - sum;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) s_pref[tid * kPer + j] = excl + loc[j];
        if (tid == kBlock - 1) s_pref[kMaxRegions] = s_part[tid];
        __syncthreads();
        total_chunks = s_pref[kMaxRegions];
    }
    for (uint32_t vc = blockIdx.x; vc < total_chunks; vc += gridDim.x) {
        uint32_t region, chunk, n_in;
        if (FIRST) {
            region = vc / chunks_per_region;
            chunk = vc - region * chunks_per_region;
            const uint32_t lo = region * A.region_cap;
            n_in = min(A.region_cap, A.n_paths - lo);
        } else {
            // largest r with s_pref[r] <= vc
            uint32_t lo = 0, hi = A.n_regions;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_pref[mid] <= vc) lo = mid; else hi = mid;
            }
            region = lo;
            chunk = vc - s_pref[lo];
            n_in = s_raw[region];
        }
        const uint32_t idx = chunk * kBlock + tid;
        const bool valid = idx < n_in;
        const size_t slot_in = (size_t)region * A.region_cap + idx;

        bool alive = false;
        v3 origin = V(0, 0, 0), dir = V(0, 0, 0), T = V(1, 1, 1), L = V(0, 0, 0);
        uint32_t w0 = 0, gk = 0;
        float valid_gather = 0.f;
        Rng<KIND> g;
        g.init(A.seed, 0, 0, 0, 0);
        if (valid) {
            float gather_time;
            if (FIRST) {
                // __raygen__Program (raygen.cu:123-147) + perspectiveCamera (camera.cu:11-19) + color() (raygen.cu:89-95)
                const uint32_t path_id = (uint32_t)slot_in;
                const uint32_t slot = path_id / A.npix;
                const uint32_t pl = path_id - slot * A.npix;
                const uint32_t yl = pl / A.width;
                const uint32_t x = pl - yl * A.width;
                const uint32_t y = A.row0 + yl;
                const uint32_t pixel = A.width * y + x;
                const uint32_t sample = A.sample0 + slot;
                float r0, r1, r2, r3, r4;
                if (KIND == RTW_RNG_TEA_LCG) {
                    uint32_t s = tea<64>(pixel, sample);  // raygen.cu:129
                    r0 = lcg_rnd(s); r1 = lcg_rnd(s); r2 = lcg_rnd(s); r3 = lcg_rnd(s);
                    g.init(A.seed, pixel, sample, s, s);  // prd.seed = seed; rayColor's local copy (Q7)
                    r4 = lcg_rnd(s);
                    w0 = path_id;
                } else {
                    uint32_t o[4];
                    philox4x32_10(pixel, sample, 0u, 0u, A.seed, 0u, o);
                    r0 = u24(o[0]); r1 = u24(o[1]); r2 = u24(o[2]); r3 = u24(o[3]);
                    philox4x32_10(pixel, sample, 1u, 0u, A.seed, 0u, o);
                    r4 = u24(o[0]);
                    g.init(A.seed, pixel, sample, 0u, sample);
                    w0 = pixel;
                }
                const rtw_camera& cam = A.sc.cam;
                float s = ((float)x + r0) / (float)A.width;
                float t = ((float)y + r1) / (float)A.height;
                origin = ld3(cam.origin);
                if (cam.lens_radius != 0.0f) {  // sampling.cuh:15-22; the two draws are consumed either way
                    float sn, cs;
                    sincos2pi(r2, sn, cs);
                    float sq = __builtin_sqrtf(r3);
                    float rx = cam.lens_radius * (sn * sq);
                    float ry = cam.lens_radius * (cs * sq);
                    origin = vadd(origin, vfma(ld3(cam.v), ry, vscale(ld3(cam.u), rx)));
                }
                dir = vfma(ld3(cam.horizontal), s, ld3(cam.lower_left));
                dir = vfma(ld3(cam.vertical), t, dir);
                dir = vsub(dir, origin);
                gk = (uint32_t)(r4 * 16777216.0f);
                gather_time = fma_(r4, cam.time1 - cam.time0, cam.time0);
            } else {
                float4 p0 = A.in0[slot_in], p1 = A.in1[slot_in], p2 = A.in2[slot_in];
                uint4 p3 = A.in3[slot_in];
                origin = V(p0.x, p0.y, p0.z);
                dir = V(p0.w, p1.x, p1.y);
                T = V(p1.z, p1.w, p2.x);
                L = V(p2.y, p2.z, p2.w);
                w0 = p3.x; gk = p3.w;
                if (KIND == RTW_RNG_TEA_LCG) g.init(A.seed, 0, 0, p3.y, p3.z);
                else g.init(A.seed, w0, p3.z, p3.y, p3.z);  // word 2 carries the sample index
                gather_time = fma_((float)gk * (1.0f / 16777216.0f), A.sc.cam.time1 - A.sc.cam.time0, A.sc.cam.time0);
            }
            valid_gather = gather_time;
        }
        float gather_time = valid_gather;
        bool live = valid;
        if (valid) {
            // A.n_iter consecutive bounces in registers: 1 for the wide early bounces (compaction after every
            // segment keeps the lanes full), several for the thin tail (a launch with few paths is
            // latency-bound, so fewer, longer launches win).
            uint32_t depth = A.depth;
            for (uint32_t it = 0; it < A.n_iter; it++) {
                float t, ray_time;
                int prim;
                trace_stage<KIND>(A, depth, g, origin, dir, gather_time, my_stack, t, prim, ray_time);
                alive = shade_stage<KIND>(A, depth, g, origin, dir, T, L, gather_time, my_stack, n_shadow, t, prim, ray_time);
                n_seg++;
                depth++;
                if (!alive) break;
            }
        }
        if (live && !alive) {
            uint32_t path_id = w0;
            if (KIND == RTW_RNG_PHILOX) path_id = (g.b - A.sample0) * A.npix + (w0 - A.row0 * A.width);
            // removeNaNs, raygen.cu:17-24
            float lx = (L.x == L.x) ? L.x : 0.f, ly = (L.y == L.y) ? L.y : 0.f, lz = (L.z == L.z) ? L.z : 0.f;
            A.lbuf[path_id] = make_float4(lx, ly, lz, 0.f);
        }
        // wave64 ballot + popcount prefix; one atomic per wave reserves its slice of the region's output
        const unsigned long long ballot = __ballot(alive);
        if (ballot) {
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.cnt_out[region], (uint32_t)__popcll(ballot));
            base = __builtin_amdgcn_readfirstlane(base);
            if (alive) {
                const size_t so = (size_t)region * A.region_cap + base + before;
                A.out0[so] = make_float4(origin.x, origin.y, origin.z, dir.x);
                A.out1[so] = make_float4(dir.y, dir.z, T.x, T.y);
                A.out2[so] = make_float4(T.z, L.x, L.y, L.z);
                A.out3[so] = make_uint4(w0, g.a, g.b, gk);
            }
        }
    }
    // statistics: wave reduction, one atomic pair per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_seg += __shfl_down(n_seg, off);
        n_shadow += __shfl_down(n_shadow, off);
    }
    if (lane == 0 && (n_seg | n_shadow)) {
        atomicAdd(&A.stats[0], (unsigned long long)n_seg);
        atomicAdd(&A.stats[1], (unsigned long long)n_shadow);
    }
}

// sums the S sample slots of every pixel in ascending sample order (fixed order => reproducible bits)
__global__ void __launch_bounds__(kBlock) k_resolve(const float4* __restrict__ lbuf, float4* __restrict__ accum, uint32_t npix, uint32_t nslots) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        float4 a = accum[i];
        for (uint32_t s = 0; s < nslots; s++) {
            float4 l = lbuf[(size_t)s * npix + i];
            a.x += l.x; a.y += l.y; a.z += l.z;
        }
        accum[i] = a;
    }
}

__global__ void __launch_bounds__(kBlock) k_finish(const float4* __restrict__ accum, float4* __restrict__ out, uint32_t npix, float spp) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        float4 a = accum[i];
        out[i] = make_float4(a.x / spp, a.y / spp, a.z / spp, 1.0f);
    }
}

struct NoRng {
    uint32_t a, b;
    RTW_DEV float next1() { return 0.5f; }
    RTW_DEV float randf1() { return 0.5f; }
};

__global__ void __launch_bounds__(kBlock) k_debug_intersect(const DScene sc, const float* __restrict__ rays, const float* __restrict__ ray_time,
                                                           const float* __restrict__ gather_time, int n, float* __restrict__ out_t,
                                                           int32_t* __restrict__ out_prim, uint32_t stack_stride) {
    extern __shared__ uint32_t s_stack[];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 8 * (size_t)i;
    NoRng g;
    float t;
    int prim;
    traverse<NoRng, false, true>(sc, V(r[0], r[1], r[2]), V(r[3], r[4], r[5]), r[6], r[7], ray_time ? ray_time[i] : 0.f,
                                 gather_time ? gather_time[i] : 0.f, g, s_stack + threadIdx.x, stack_stride, t, prim);
    out_t[i] = t;
    out_prim[i] = prim;
}

}  // namespace

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
    // render pool
    size_t pool_paths = 0;
    float4* planes[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    uint4* plane3[2] = {nullptr, nullptr};
    float4* lbuf = nullptr;
    float4* accum = nullptr;
    size_t accum_pix = 0;
    uint32_t* cnt = nullptr;
    size_t cnt_words = 0;
    unsigned long long* d_stats = nullptr;
    float4* d_out = nullptr;
    size_t out_pix = 0;
    int n_cu = 256;
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

void free_pool(rtw_ctx* c) {
    for (int b = 0; b < 2; b++) {
        for (int k = 0; k < 3; k++) { if (c->planes[b][k]) (void)hipFree(c->planes[b][k]); c->planes[b][k] = nullptr; }
        if (c->plane3[b]) (void)hipFree(c->plane3[b]);
        c->plane3[b] = nullptr;
    }
    if (c->lbuf) (void)hipFree(c->lbuf);
    c->lbuf = nullptr;
    c->pool_paths = 0;
}

int ensure_pool(rtw_ctx* c, size_t paths, size_t npix, size_t cnt_words) {
    if (paths > c->pool_paths) {
        free_pool(c);
        for (int b = 0; b < 2; b++) {
            for (int k = 0; k < 3; k++) HIP_TRY(c, hipMalloc(&c->planes[b][k], paths * sizeof(float4)));
            HIP_TRY(c, hipMalloc(&c->plane3[b], paths * sizeof(uint4)));
        }
        HIP_TRY(c, hipMalloc(&c->lbuf, paths * sizeof(float4)));
        c->pool_paths = paths;
    }
    if (npix > c->accum_pix) {
        if (c->accum) (void)hipFree(c->accum);
        c->accum = nullptr; c->accum_pix = 0;
        HIP_TRY(c, hipMalloc(&c->accum, npix * sizeof(float4)));
        c->accum_pix = npix;
    }
    if (cnt_words > c->cnt_words) {
        if (c->cnt) (void)hipFree(c->cnt);
        c->cnt = nullptr; c->cnt_words = 0;
        HIP_TRY(c, hipMalloc(&c->cnt, cnt_words * sizeof(uint32_t)));
        c->cnt_words = cnt_words;
    }
    if (!c->d_stats) HIP_TRY(c, hipMalloc(&c->d_stats, 4 * sizeof(unsigned long long)));
    return RTW_OK;
}

size_t pool_target_paths() {
    const char* e = getenv("RTW_POOL_PATHS");
    if (e && *e) {
        long long v = atoll(e);
        if (v >= 1024) return (size_t)v;
    }
    return (size_t)1 << 26;  // 67 M paths in flight: 9 GiB of ping-pong state + radiance slots (HBM-sized on purpose: long batches amortise the thin tail launches)
}

template <int KIND, bool FIRST>
void launch_bounce(const BounceArgs& a, int grid, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((k_bounce<KIND, FIRST>), dim3(grid), dim3(kBlock), lds, s, a);
}

}  // namespace

extern "C" {

int rtw_abi_version(void) { return RTW_ABI_VERSION; }

const char* rtw_last_error(rtw_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rtw_create(rtw_ctx** out, int n_devices, const int* device_ids) {
    if (!out) return RTW_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_devices != 1) return RTW_ERR_UNSUPPORTED;  // one context per GPU; ranks are separate processes
    rtw_ctx* c = new (std::nothrow) rtw_ctx();
    if (!c) return RTW_ERR_OOM;
    c->device = device_ids ? device_ids[0] : 0;
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

int rtw_destroy(rtw_ctx* c) {
    if (!c) return RTW_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_pool(c);
    if (c->accum) (void)hipFree(c->accum);
    if (c->cnt) (void)hipFree(c->cnt);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_scene) (void)hipFree(c->d_scene);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RTW_OK;
}

int rtw_upload_scene(rtw_ctx* c, const void* blob, size_t bytes) {
    if (!c) return RTW_ERR_INVALID_ARG;
    if (!blob || bytes < sizeof(rtw_scene_header)) return fail(c, RTW_ERR_BAD_SCENE, "scene blob too small");
    rtw_scene_header h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != RTW_SCENE_MAGIC || h.version != RTW_ABI_VERSION || h.total_bytes > bytes)
        return fail(c, RTW_ERR_BAD_SCENE, "bad scene header (magic/version/size)");
    auto in_range = [&](uint32_t off, uint32_t n, size_t sz) { return (size_t)off + (size_t)n * sz <= bytes; };
    if (!in_range(h.off_prims, h.n_prims, sizeof(rtw_prim)) || !in_range(h.off_xforms, h.n_xforms, sizeof(rtw_xform)) ||
        !in_range(h.off_materials, h.n_materials, sizeof(rtw_material)) || !in_range(h.off_textures, h.n_textures, sizeof(rtw_texture)) ||
        !in_range(h.off_lights, h.n_lights, sizeof(rtw_light)) || h.n_xforms < 1)
        return fail(c, RTW_ERR_BAD_SCENE, "scene table out of range");
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

    std::vector<HitRec> shade(h.n_prims);
    std::vector<int32_t> order;
    int has_motion = 0;
    for (uint32_t i = 0; i < h.n_prims; i++) {
        const rtw_prim& p = prims[i];
        if (p.type < RTW_PRIM_SPHERE || p.type > RTW_PRIM_VOLUME_SPHERE) return fail(c, RTW_ERR_BAD_SCENE, "unknown primitive type");
        if (p.xform < 0 || (uint32_t)p.xform >= h.n_xforms) return fail(c, RTW_ERR_BAD_SCENE, "primitive xform out of range");
        if (p.material < 0 || (uint32_t)p.material >= h.n_materials) return fail(c, RTW_ERR_BAD_SCENE, "primitive material out of range");
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
            }
        }
        if (m.texture >= 0) {
            if ((uint32_t)m.texture >= h.n_textures) return fail(c, RTW_ERR_BAD_SCENE, "material texture out of range");
            const rtw_texture& t = texs[m.texture];
            if (t.type == RTW_TEX_CONSTANT) { s.r = t.color[0]; s.g = t.color[1]; s.b = t.color[2]; }
            else if (t.type != RTW_TEX_NULL) return fail(c, RTW_ERR_UNSUPPORTED, "only constant and null textures are supported");
        }
        shade[i] = s;
    }
    int brute_max = kBruteMaxPrims;
    if (const char* e = getenv("RTW_BRUTE_MAX")) brute_max = atoi(e);  // experiments: 0 forces the BVH path
    const bool use_bvh = (int)h.n_prims > brute_max;
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
    rtwbvh::Bvh bvh;
    if (use_bvh) {
        bvh = rtwbvh::build_bvh(prims.data(), h.n_prims, xforms.data());
        if (bvh.max_depth > 60) return fail(c, RTW_ERR_UNSUPPORTED, "BVH deeper than the LDS traversal stack");
    }

    // one device allocation, 256-byte aligned sub-tables
    auto al = [](size_t v) { return (v + 255u) & ~size_t(255); };
    size_t o_prims = 0;
    size_t o_xf = al(o_prims + prims.size() * sizeof(rtw_prim));
    size_t o_shade = al(o_xf + xforms.size() * sizeof(rtw_xform));
    size_t o_lights = al(o_shade + shade.size() * sizeof(HitRec));
    size_t o_nodes = al(o_lights + std::max<size_t>(1, lights.size()) * sizeof(rtw_light));
    size_t o_tree = al(o_nodes + std::max<size_t>(1, bvh.nodes.size()) * sizeof(BvhNode));
    size_t o_order = al(o_tree + std::max<size_t>(1, bvh.prim_order.size()) * sizeof(int32_t));
    size_t o_groups = al(o_order + std::max<size_t>(1, order.size()) * sizeof(int32_t));
    size_t o_recs = al(o_groups + std::max<size_t>(1, groups.size()) * sizeof(BruteGroup));
    size_t total = al(o_recs + std::max<size_t>(1, recs.size()) * sizeof(BruteRec));
    std::vector<char> stage(total, 0);
    if (!prims.empty()) memcpy(stage.data() + o_prims, prims.data(), prims.size() * sizeof(rtw_prim));
    memcpy(stage.data() + o_xf, xforms.data(), xforms.size() * sizeof(rtw_xform));
    if (!shade.empty()) memcpy(stage.data() + o_shade, shade.data(), shade.size() * sizeof(HitRec));
    if (!lights.empty()) memcpy(stage.data() + o_lights, lights.data(), lights.size() * sizeof(rtw_light));
    static_assert(sizeof(BvhNode) == sizeof(rtwbvh::Node), "node layout");
    if (!bvh.nodes.empty()) memcpy(stage.data() + o_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode));
    if (!bvh.prim_order.empty()) memcpy(stage.data() + o_tree, bvh.prim_order.data(), bvh.prim_order.size() * sizeof(int32_t));
    if (!order.empty()) memcpy(stage.data() + o_order, order.data(), order.size() * sizeof(int32_t));
    if (!groups.empty()) memcpy(stage.data() + o_groups, groups.data(), groups.size() * sizeof(BruteGroup));
    if (!recs.empty()) memcpy(stage.data() + o_recs, recs.data(), recs.size() * sizeof(BruteRec));

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
    sc.nodes = (const BvhNode*)(d + o_nodes);
    sc.tree_prims = (const int32_t*)(d + o_tree);
    sc.order = (const int32_t*)(d + o_order);
    sc.groups = (const BruteGroup*)(d + o_groups);
    sc.recs = (const BruteRec*)(d + o_recs);
    sc.n_groups = (int)groups.size();
    sc.n_generic = n_generic;
    sc.n_prims = (int)h.n_prims;
    sc.n_vol = n_vol;
    sc.n_tree = (int)bvh.prim_order.size();
    sc.n_lights = (int)h.n_lights;
    sc.sky_light = h.sky_light;
    sc.use_bvh = use_bvh ? 1 : 0;
    sc.has_motion = has_motion;
    sc.cam = h.camera;
    sc.pdf = h.pdf;
    c->sc = sc;
    c->stack_depth = use_bvh ? bvh.max_depth + 2 : 0;
    c->has_scene = true;
    return RTW_OK;
}

int rtw_render_device(rtw_ctx* c, const rtw_params* P, void* d_rgba, void* hip_stream, rtw_stats* stats) {
    if (!c) return RTW_ERR_INVALID_ARG;
    if (!c->has_scene) return fail(c, RTW_ERR_NO_SCENE, "rtw_render before rtw_upload_scene");
    if (!P || !d_rgba) return fail(c, RTW_ERR_INVALID_ARG, "null params or output");
    if (P->width <= 0 || P->height <= 0 || P->spp <= 0 || P->max_depth < 0 || P->row0 < 0 || P->row1 > P->height || P->row0 > P->row1)
        return fail(c, RTW_ERR_INVALID_ARG, "bad render params");
    if (P->rng_kind != RTW_RNG_PHILOX && P->rng_kind != RTW_RNG_TEA_LCG) return fail(c, RTW_ERR_INVALID_ARG, "bad rng_kind");
    if (P->sample_offset < 0 || P->samples_per_pass < 0) return fail(c, RTW_ERR_INVALID_ARG, "bad sample_offset/samples_per_pass");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;

    const size_t rows = (size_t)(P->row1 - P->row0);
    const size_t npix = rows * (size_t)P->width;
    if (stats) memset(stats, 0, sizeof *stats);
    if (npix == 0) return RTW_OK;
    if (npix > 0xffffffffull / 2) return fail(c, RTW_ERR_UNSUPPORTED, "tile too large");

    // samples per pass: keep about pool_target paths in flight
    size_t S = P->samples_per_pass > 0 ? (size_t)P->samples_per_pass : std::max<size_t>(1, pool_target_paths() / npix);
    S = std::min<size_t>(S, (size_t)P->spp);
    while (S > 1 && npix * S > 0xfffffff0ull) S--;
    const size_t paths_max = npix * S;
    // region capacity: a multiple of 256 paths, at least 16384, large enough that <= kMaxRegions regions cover the pool
    size_t region_cap = std::max<size_t>(kMinRegionCap, (((paths_max + kMaxRegions - 1) / kMaxRegions) + kBlock - 1) / kBlock * kBlock);
    const uint32_t regions_max = (uint32_t)((paths_max + region_cap - 1) / region_cap);
    // launch schedule: (first depth, bounces in registers). Early bounces one per launch; the tail in growing groups.
    std::vector<std::pair<int, int>> sched;
    {
        const char* e = getenv("RTW_TAIL_START");
        const int tail_start = (e && *e) ? std::max(1, atoi(e)) : 4;
        int d = 0, grp = 2, rep = 0;
        while (d < P->max_depth) {
            int n = 1;
            if (d >= tail_start) {
                n = std::min(grp, P->max_depth - d);
                if (++rep == 2) { rep = 0; grp += grp / 2; }
            }
            sched.push_back({d, n});
            d += n;
        }
    }
    const size_t cnt_words = (size_t)regions_max * (sched.size() + 2);
    int rc = ensure_pool(c, (size_t)regions_max * region_cap, npix, cnt_words);
    if (rc) return rc;

    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    HIP_TRY(c, hipEventCreate(&ev_begin));
    HIP_TRY(c, hipEventCreate(&ev_end));
    std::vector<hipEvent_t> ev_loop;
    auto cleanup = [&]() {
        (void)hipEventDestroy(ev_begin);
        (void)hipEventDestroy(ev_end);
        for (hipEvent_t e : ev_loop) (void)hipEventDestroy(e);
    };
#define HIP_TRY_C(expr)                                                               \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            cleanup();                                                                \
            return fail(c, RTW_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                             \
    } while (0)

    HIP_TRY_C(hipEventRecord(ev_begin, s));
    HIP_TRY_C(hipMemsetAsync(c->accum, 0, npix * sizeof(float4), s));
    HIP_TRY_C(hipMemsetAsync(c->d_stats, 0, 4 * sizeof(unsigned long long), s));

    const size_t lds = (size_t)c->stack_depth * kBlock * sizeof(uint32_t);
    uint64_t launches = 0;
    if (P->max_depth > 0) {
        for (size_t s0 = 0; s0 < (size_t)P->spp; s0 += S) {
            const size_t Sb = std::min(S, (size_t)P->spp - s0);
            const size_t paths = npix * Sb;
            const uint32_t regions = (uint32_t)((paths + region_cap - 1) / region_cap);
            HIP_TRY_C(hipMemsetAsync(c->cnt, 0, (size_t)regions * (sched.size() + 2) * sizeof(uint32_t), s));
            BounceArgs a{};
            a.sc = c->sc;
            a.lbuf = c->lbuf;
            a.stats = c->d_stats;
            a.n_regions = regions;
            a.n_paths = (uint32_t)paths;
            a.npix = (uint32_t)npix;
            a.width = (uint32_t)P->width;
            a.height = (uint32_t)P->height;
            a.row0 = (uint32_t)P->row0;
            a.sample0 = (uint32_t)(P->sample_offset + (int)s0);
            a.seed = P->seed;
            a.max_depth = (uint32_t)P->max_depth;
            a.stack_stride = kBlock;
            a.region_cap = (uint32_t)region_cap;
            const uint32_t n_chunks = (uint32_t)((paths + kBlock - 1) / kBlock);
            const int grid = (int)std::min<uint32_t>(n_chunks, (uint32_t)c->n_cu * 8u);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            HIP_TRY_C(hipEventCreate(&e0));
            ev_loop.push_back(e0);
            HIP_TRY_C(hipEventCreate(&e1));
            ev_loop.push_back(e1);
            HIP_TRY_C(hipEventRecord(e0, s));
            for (size_t li = 0; li < sched.size(); li++) {
                const int d = sched[li].first;
                const int ib = (int)(li & 1), ob = ib ^ 1;  // launch li reads buffer li&1 (launch 0 reads nothing), writes the other
                a.in0 = c->planes[ib][0]; a.in1 = c->planes[ib][1]; a.in2 = c->planes[ib][2]; a.in3 = c->plane3[ib];
                a.out0 = c->planes[ob][0]; a.out1 = c->planes[ob][1]; a.out2 = c->planes[ob][2]; a.out3 = c->plane3[ob];
                a.cnt_in = c->cnt + li * regions;
                a.cnt_out = c->cnt + (li + 1) * regions;
                a.depth = (uint32_t)d;
                a.n_iter = (uint32_t)sched[li].second;
                if (P->rng_kind == RTW_RNG_PHILOX) {
                    if (d == 0) launch_bounce<RTW_RNG_PHILOX, true>(a, grid, lds, s);
                    else launch_bounce<RTW_RNG_PHILOX, false>(a, grid, lds, s);
                } else {
                    if (d == 0) launch_bounce<RTW_RNG_TEA_LCG, true>(a, grid, lds, s);
                    else launch_bounce<RTW_RNG_TEA_LCG, false>(a, grid, lds, s);
                }
                launches++;
            }
            HIP_TRY_C(hipEventRecord(e1, s));
            hipLaunchKernelGGL(k_resolve, dim3((unsigned)std::min<size_t>((npix + kBlock - 1) / kBlock, (size_t)c->n_cu * 8)), dim3(kBlock), 0, s,
                               (const float4*)c->lbuf, c->accum, (uint32_t)npix, (uint32_t)Sb);
        }
    }
    hipLaunchKernelGGL(k_finish, dim3((unsigned)std::min<size_t>((npix + kBlock - 1) / kBlock, (size_t)c->n_cu * 8)), dim3(kBlock), 0, s,
                       (const float4*)c->accum, (float4*)d_rgba, (uint32_t)npix, (float)P->spp);
    HIP_TRY_C(hipGetLastError());
    HIP_TRY_C(hipEventRecord(ev_end, s));
    HIP_TRY_C(hipEventSynchronize(ev_end));

    unsigned long long hs[4] = {0, 0, 0, 0};
    HIP_TRY_C(hipMemcpy(hs, c->d_stats, sizeof hs, hipMemcpyDeviceToHost));
    if (stats) {
        float ms = 0.f;
        HIP_TRY_C(hipEventElapsedTime(&ms, ev_begin, ev_end));
        stats->seconds = (double)ms * 1e-3;
        double loop_ms = 0.0;
        for (size_t i = 0; i + 1 < ev_loop.size(); i += 2) {
            float m = 0.f;
            HIP_TRY_C(hipEventElapsedTime(&m, ev_loop[i], ev_loop[i + 1]));
            loop_ms += m;
        }
        stats->bounce_seconds = loop_ms * 1e-3;
        stats->bounce_launches = launches;
        stats->samples = (uint64_t)npix * (uint64_t)P->spp;
        stats->segments = hs[0];
        stats->shadow_rays = hs[1];
        stats->algorithmic_bytes = 128ull * stats->segments + 32ull * stats->samples;
    }
    cleanup();
#undef HIP_TRY_C
    return RTW_OK;
}

int rtw_render(rtw_ctx* c, const rtw_params* P, float* rgba_out, rtw_stats* stats) {
    if (!c) return RTW_ERR_INVALID_ARG;
    if (!rgba_out) return fail(c, RTW_ERR_INVALID_ARG, "null output");
    if (!P) return fail(c, RTW_ERR_INVALID_ARG, "null params");
    if (P->width <= 0 || P->row0 < 0 || P->row1 < P->row0) return fail(c, RTW_ERR_INVALID_ARG, "bad render params");
    const size_t npix = (size_t)(P->row1 - P->row0) * (size_t)P->width;
    HIP_TRY(c, hipSetDevice(c->device));
    if (npix > c->out_pix) {
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_out = nullptr; c->out_pix = 0;
        HIP_TRY(c, hipMalloc(&c->d_out, std::max<size_t>(npix, 1) * sizeof(float4)));
        c->out_pix = npix;
    }
    int rc = rtw_render_device(c, P, c->d_out, nullptr, stats);
    if (rc) return rc;
    if (npix) HIP_TRY(c, hipMemcpy(rgba_out, c->d_out, npix * sizeof(float4), hipMemcpyDeviceToHost));  // Director.cpp:999-1000
    return RTW_OK;
}

int rtw_debug_intersect(rtw_ctx* c, const float* rays, const float* ray_time, const float* gather_time, int n, float* out_t, int32_t* out_prim) {
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
    const size_t lds = (size_t)c->stack_depth * kBlock * sizeof(uint32_t);
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

}  // extern "C"
