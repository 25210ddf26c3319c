// rtw_device.h — device-side math, RNG, intersection and shading for the gfx950 wavefront kernels.
//
// Arithmetic contract (DESIGN.md "arithmetic spec"): fp32, compiled -ffp-contract=off, fused
// multiply-add only where __builtin_fmaf is written, IEEE-correct division and sqrt (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), own polynomial sincos / log. The same operations in
// the same order are stated independently in oracle/rtw_oracle.c, which is what the parity tests
// check this file against.
//
// Each function cites the reference device code it replaces (paths under RestOfLife/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtw.h"

#define RTW_DEV __device__ __forceinline__

namespace rtwdev {

struct v3 { float x, y, z; };

RTW_DEV v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
RTW_DEV v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
RTW_DEV v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
RTW_DEV v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
RTW_DEV v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
RTW_DEV v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
RTW_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RTW_DEV v3 vfma(v3 a, float s, v3 b) { return V(fma_(a.x, s, b.x), fma_(a.y, s, b.y), fma_(a.z, s, b.z)); }
RTW_DEV float dot3(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
RTW_DEV v3 cross3(v3 a, v3 b) {
    return V(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
RTW_DEV v3 normalize3(v3 a) { float inv = 1.0f / __builtin_sqrtf(dot3(a, a)); return vscale(a, inv); }
RTW_DEV float length3(v3 a) { return __builtin_sqrtf(dot3(a, a)); }
RTW_DEV v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }

RTW_DEV v3 xf_point(const float* m, v3 p) {
    return V(fma_(m[0], p.x, fma_(m[1], p.y, fma_(m[2], p.z, m[3]))),
             fma_(m[4], p.x, fma_(m[5], p.y, fma_(m[6], p.z, m[7]))),
             fma_(m[8], p.x, fma_(m[9], p.y, fma_(m[10], p.z, m[11]))));
}
RTW_DEV v3 xf_vector(const float* m, v3 d) {
    return V(fma_(m[0], d.x, fma_(m[1], d.y, m[2] * d.z)),
             fma_(m[4], d.x, fma_(m[5], d.y, m[6] * d.z)),
             fma_(m[8], d.x, fma_(m[9], d.y, m[10] * d.z)));
}
// optixTransformNormalFromObjectToWorldSpace: transpose of the world->object linear part
RTW_DEV v3 xf_normal(const float* inv, v3 n) {
    return V(fma_(inv[0], n.x, fma_(inv[4], n.y, inv[8] * n.z)),
             fma_(inv[1], n.x, fma_(inv[5], n.y, inv[9] * n.z)),
             fma_(inv[2], n.x, fma_(inv[6], n.y, inv[10] * n.z)));
}

#define RTW_1_PI_F 0.318309886183790671538f
#define RTW_PIO2_F 1.57079632679489661923f
#define RTW_FLT_MAX 3.402823466e+38f

// sin/cos(2*pi*r), r in [0,1) — stands in for sinf/cosf of lib/sampling.cuh:15-22,49-60
RTW_DEV void sincos2pi(float r, float& s_out, float& c_out) {
    float t = r * 4.0f;
    float q = __builtin_floorf(t + 0.5f);
    float f = t - q;
    float x = f * RTW_PIO2_F;
    float x2 = x * x;
    float sp = fma_(x2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fma_(x2, sp, -1.6666654611e-1f);
    float s = fma_(x * x2, sp, x);
    float cp = fma_(x2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fma_(x2, cp, 4.166664568298827e-2f);
    float c = fma_(x2 * x2, cp, fma_(x2, -0.5f, 1.0f));
    int qi = ((int)q) & 3;
    float so = (qi == 0) ? s : (qi == 1) ? c : (qi == 2) ? -s : -c;
    float co = (qi == 0) ? c : (qi == 1) ? -s : (qi == 2) ? -c : s;
    s_out = so;
    c_out = co;
}

// sin / atan2 / asin of the texture callables (texture/noiseTexture.cu:77, checkeredTexture.cu:9, geometry/sphere.cu:24-30):
// Cephes single-precision algorithms (sinf.c, atanf.c, asinf.c) with every fused step written out; the CPU checker in
// the test tree repeats them operation for operation.
RTW_DEV float sin_spec(float xx) {
    float x = __builtin_fabsf(xx);
    bool neg = xx < 0.0f;
    uint32_t j = (uint32_t)(x * 1.27323954473516f);
    float y = (float)j;
    if (j & 1u) { j += 1u; y += 1.0f; }
    j &= 7u;
    if (j > 3u) { neg = !neg; j -= 4u; }
    x = fma_(-y, 0.78515625f, x);
    x = fma_(-y, 2.4187564849853515625e-4f, x);
    x = fma_(-y, 3.77489497744594108e-8f, x);
    const float z = x * x;
    float r;
    if (j == 1u || j == 2u) {
        float p = fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f);
        p = fma_(p, z, 4.166664568298827e-2f);
        r = fma_(p * z, z, fma_(-0.5f, z, 1.0f));
    } else {
        float p = fma_(-1.9515295891e-4f, z, 8.3321608736e-3f);
        p = fma_(p, z, -1.6666654611e-1f);
        r = fma_(p * z, x, x);
    }
    return neg ? -r : r;
}
RTW_DEV float atan_pos(float x) {
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    const float z = x * x;
    float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fma_(p, z, 1.99777106478e-1f);
    p = fma_(p, z, -3.33329491539e-1f);
    return y + fma_(p * z, x, x);
}
RTW_DEV float atan2_spec(float y, float x) {
    const float pi = 3.14159265358979323846f, pio2 = 1.5707963267948966f;
    if (x == 0.0f) return y > 0.0f ? pio2 : (y < 0.0f ? -pio2 : 0.0f);
    if (y == 0.0f) return x < 0.0f ? pi : 0.0f;
    const float q = y / x;
    float a = atan_pos(__builtin_fabsf(q));
    if (q < 0.0f) a = -a;
    const float w = x < 0.0f ? (y < 0.0f ? -pi : pi) : 0.0f;
    return w + a;
}
RTW_DEV float asin_spec(float xx) {
    float a = __builtin_fabsf(xx);
    if (a > 1.0f) a = 1.0f;
    float r;
    if (a < 1.0e-4f) {
        r = a;
    } else {
        float z, x;
        const bool flag = a > 0.5f;
        if (flag) { z = 0.5f * (1.0f - a); x = __builtin_sqrtf(z); }
        else { x = a; z = x * x; }
        float p = fma_(4.2163199048e-2f, z, 2.4181311049e-2f);
        p = fma_(p, z, 4.5470025998e-2f);
        p = fma_(p, z, 7.4953002686e-2f);
        p = fma_(p, z, 1.6666752422e-1f);
        r = fma_(p * z, x, x);
        if (flag) r = 1.5707963267948966f - (r + r);
    }
    return xx < 0.0f ? -r : r;
}

// natural log (Cephes logf) — stands in for logf of geometry/volumeBox.cu:79, volumeSphere.cu:93
RTW_DEV float log_spec(float x) {
    if (x == 0.0f) return -__builtin_inff();
    uint32_t ix = __float_as_uint(x);
    int e = (int)((ix >> 23) & 255u) - 126;
    ix = (ix & 0x007fffffu) | 0x3f000000u;
    float m = __uint_as_float(ix);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = fma_(7.0376836292e-2f, m, -1.1514610310e-1f);
    y = fma_(y, m, 1.1676998740e-1f);
    y = fma_(y, m, -1.2420140846e-1f);
    y = fma_(y, m, 1.4249322787e-1f);
    y = fma_(y, m, -1.6668057665e-1f);
    y = fma_(y, m, 2.0000714765e-1f);
    y = fma_(y, m, -2.4999993993e-1f);
    y = fma_(y, m, 3.3333331174e-1f);
    y = (y * m) * z;
    float fe = (float)e;
    y = fma_(-2.12194440e-4f, fe, y);
    y = fma_(-0.5f, z, y);
    float r = m + y;
    r = fma_(0.693359375f, fe, r);
    return r;
}

// ------------------------------------------------------------------ RNG
// lib/random.cuh:7-19 (tea<N>)
template <unsigned N>
RTW_DEV uint32_t tea(uint32_t s0, uint32_t s1) {
    uint32_t t = 0;
#pragma unroll 4
    for (unsigned n = 0; n < N; n++) {
        t += 0x9E3779B9u;
        s0 += ((s1 << 4) + 0xa341316cu) ^ (s1 + t) ^ ((s1 >> 5) + 0xc8013ea4u);
        s1 += ((s0 << 4) + 0xad90777du) ^ (s0 + t) ^ ((s0 >> 5) + 0x7e95761eu);
    }
    return s0;
}
// OptiX SDK cuda/random.h lcg/rnd
RTW_DEV float lcg_rnd(uint32_t& s) {
    s = 1664525u * s + 1013904223u;
    return (float)(s & 0x00FFFFFFu) / (float)0x01000000;
}
// lib/random.cuh:22-38 (xorshift32 / randf incl. quirk Q10)
RTW_DEV float xorshift_randf(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    float r = ((float)s) / 4294967296.0f;
    return (r != 1.0f) ? r : (float)0x3F7FFFFF;
}
// 32 x 32 -> 64 multiply in one v_mad_u64_u32 (the compiler emits v_mul_lo_u32 + v_mul_hi_u32; measured 1.45x
// more Philox blocks per second with the single instruction, scripts/ubench/mulwide.hip)
RTW_DEV void mul_wide(uint32_t m, uint32_t x, uint32_t& hi, uint32_t& lo) {
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "s"(m), "v"(x) : "vcc");
    lo = (uint32_t)r;
    hi = (uint32_t)(r >> 32);
}
RTW_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        uint32_t hi0, lo0, hi1, lo1;
        mul_wide(0xD2511F53u, c0, hi0, lo0);
        mul_wide(0xCD9E8D57u, c2, hi1, lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RTW_DEV float u24(uint32_t v) { return (float)(v >> 8) * (1.0f / 16777216.0f); }

// Per-path generator. Streams: 0 raygen seed, 1 prd.seed, 2 rayColor's local seed (ray time).
// KIND == RTW_RNG_TEA_LCG: a = prd.seed word, b = ray-time word.
// KIND == RTW_RNG_PHILOX : a = stream-1 draw counter, b unused; ray time is draw `depth` of stream 2.
template <int KIND>
struct Rng;

template <>
struct Rng<RTW_RNG_TEA_LCG> {
    uint32_t a, b;
    RTW_DEV void init(uint32_t, uint32_t, uint32_t, uint32_t wa, uint32_t wb) { a = wa; b = wb; }
    RTW_DEV float next1() { return lcg_rnd(a); }
    RTW_DEV float randf1() { return xorshift_randf(a); }
    RTW_DEV void align_block() {}
    RTW_DEV void warm() {}
    RTW_DEV float block_draw(int) const { return 0.f; }  // (Philox only)
    RTW_DEV float rr_draw() { return lcg_rnd(a); }  // raygen.cu:77
    RTW_DEV float ray_time(uint32_t) { return lcg_rnd(b); }
};

template <>
struct Rng<RTW_RNG_PHILOX> {
    uint32_t a, b;
    uint32_t key, pixel, sample, cb, seg_base;
    uint32_t c[4];
    RTW_DEV void init(uint32_t k, uint32_t px, uint32_t smp, uint32_t wa, uint32_t wb) {
        key = k; pixel = px; sample = smp; a = wa; b = wb; cb = 0xffffffffu; seg_base = wa;
        c[0] = c[1] = c[2] = c[3] = 0;
    }
    RTW_DEV float next1() {
        uint32_t blk = a >> 2;
        if (blk != cb) { philox4x32_10(pixel, sample, blk, 1u, key, 0u, c); cb = blk; }
        uint32_t l = a & 3u;
        uint32_t v = (l == 0) ? c[0] : (l == 1) ? c[1] : (l == 2) ? c[2] : c[3];
        a++;
        return u24(v);
    }
    RTW_DEV float randf1() { return next1(); }
    // draw i (a constant) of the block in registers; the caller knows which block that is and advances `a` itself
    RTW_DEV float block_draw(int i) const { return u24(c[i]); }
    // the closest-hit program of every segment starts at a fresh block (see the oracle): wave-coherent refills
    RTW_DEV void align_block() { a = (a + 3u) & ~3u; seg_base = a; }
    // generate the block the next draw comes from now, while the whole wave is still on one code path
    // (the material branches then find it cached instead of each running its own copy of the ten rounds)
    RTW_DEV void warm() {
        const uint32_t blk = a >> 2;
        if (blk != cb) { philox4x32_10(pixel, sample, blk, 1u, key, 0u, c); cb = blk; }
    }
    // Russian-roulette draw: a fifth uniform from the low bytes of words 0..2 of the block this segment used last
    // (see the oracle's rng_rr): a Lambertian segment then needs one Philox block, not two.
    RTW_DEV float rr_draw() {
        const uint32_t blk = (a == seg_base) ? (a >> 2) : ((a - 1u) >> 2);
        if (blk != cb) { philox4x32_10(pixel, sample, blk, 1u, key, 0u, c); cb = blk; }
        const uint32_t v = ((c[0] & 0xffu) << 16) | ((c[1] & 0xffu) << 8) | (c[2] & 0xffu);
        return (float)v * (1.0f / 16777216.0f);
    }
    RTW_DEV float ray_time(uint32_t depth) {
        uint32_t o[4];
        philox4x32_10(pixel, sample, depth >> 2, 2u, key, 0u, o);
        uint32_t l = depth & 3u;
        return u24((l == 0) ? o[0] : (l == 1) ? o[1] : (l == 2) ? o[2] : o[3]);
    }
};

// ------------------------------------------------------------------ device scene
// The tree: 64-byte nodes of the 4-wide tree with quantised child boxes, 32-byte leaf records in tree order (rtw_bvh.h
// Q4Node, LeafRec). A reference is idx << 2 | count (count 0 = inner node, 1-2 = leaf of that many records).
// Per-primitive hit record baked at upload (96 B, fetched with a burst of 16-byte loads once
// the closest hit is known): material + its constant texture colour (texture/constantTexture.cu:5-10,
// nullTexture.cu:7-12) and everything the shading normal needs, so that no primitive / transform
// record has to be re-read per lane after traversal.
enum { HK_CONST_NORMAL = 0, HK_SPHERE = 1, HK_MOVING_SPHERE = 2, HK_SPHERE_XFORM = 3 };
struct HitRec {
    int32_t mat_type;   // rtw_material_type
    int32_t bsdf_eval;
    float param;        // fuzz or eta
    int32_t kind;       // HK_* | listed light << 7 | (index of a non-constant texture + 1) << 8
    float r, g, b;      // texture colour
    float inv_r;        // spheres: 1/radius (IEEE division, done once on the host)
    float nx, ny, nz;   // HK_CONST_NORMAL: world shading normal (rectangles, volumes); spheres: centre
    int32_t xform;
    // HK_CONST_NORMAL: the orthonormal basis onb::buildFromW(normal) of lib/onb.cuh:20-32, computed once on the
    // host with the same fp32 operations (u = cross(w,v), v = normalize(cross(w,a)), w = normalize(n))
    float ux, uy, uz; int32_t tex_dyn;  // (filled by load_hitrec from kind's high bits; -1 = constant colour in r, g, b)
    float vx, vy, vz; int32_t listed;    // (from kind bit 7) an emitting rectangle a light definition describes
    float wx, wy, wz, pad2;
};

// Small-scene candidate lists, built at upload (rtw_upload_scene). 32-byte records so that one
// s_load_dwordx8 brings a whole candidate into SGPRs; rectangles are sorted by axis inside a group so
// the inner loops contain no per-candidate kind dispatch at all.
struct BruteGroup {
    int32_t xform;
    int32_t first;                 // first record of the group in recs[]
    int32_t n_rx, n_ry, n_rz, n_sph;
    int32_t pad0, pad1;
};
struct BruteRec {
    float a, b, c, d, e;           // rect: a0,a1,b0,b1,k   sphere: cx,cy,cz,r,-
    int32_t prim;
    int32_t pad0, pad1;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DScene {
    const rtw_prim* __restrict__ prims;
    const rtw_xform* __restrict__ xforms;
    const HitRec* __restrict__ hitrec;     // indexed by primitive
    const rtw_light* __restrict__ lights;
    const u32x4* __restrict__ nodes;         // 4 vectors per node
    const u32x4* __restrict__ leaves;        // 2 vectors per leaf record
    const u32x4* __restrict__ wnodes;        // the same tree with fp32 child boxes, 8 vectors per node (rtw_bvh.h WNode: traverse_wave)
    const int32_t* __restrict__ order;       // candidate order: volumes (index order) then the rest (index order)
    const BruteGroup* __restrict__ groups;   // small scenes: primitives regrouped by instance transform and kind
    const BruteRec* __restrict__ recs;
    const rtw_texture* __restrict__ texs;    // only read for hit records with a non-constant texture
    const uint32_t* __restrict__ texdata;    // noise tables, image texels (rtw.h rtw_texture)
    const u32x4* __restrict__ walk;          // small scenes: the candidate lists as a stream of items (walk_items), n_walk_words 16-byte words; 0 = none
    int32_t n_prims, n_vol, n_tree, n_lights, sky_light, use_bvh, has_motion, n_groups;
    int32_t n_generic;                       // order[n_vol .. n_vol+n_generic): moving spheres, tested through the generic path
    int32_t n_walk_words;
    // BVH: leading (breadth-first) nodes staged in LDS behind the traversal stacks; noise_lds_data: word offset of the
    // noise tables staged in LDS, or -1
    int32_t n_lds_nodes, stack_depth, has_tex, noise_lds_data;
    int32_t n_lds_leaves, stack_wide;        // leaf records staged in LDS; 1 = 32-bit stack entries (references beyond 16 bits)
    int32_t n_nodes;                         // nodes of the 4-wide tree (n_lds_nodes == n_nodes: the LDS image holds them all)
    int32_t estimator;                       // rtw_estimator of the current render (set per call, not at upload)
    float ray_tmin, probe_eps;               // 1e-6 / 5e-5 as the reference; 1e-3 for the corrected estimators
    const rtw_light* __restrict__ clights;   // RTW_EST_CORRECTED: the light list moved onto the emitting rectangles
    float bmin[3], bmax[3];                  // generous world bounds of everything (k_first's wave-uniform miss test)
    rtw_camera cam;
    rtw_pdf pdf;
    int32_t cam_type;                        // rtw_camera_type
};

RTW_DEV bool is_volume(int type) { return type == RTW_PRIM_VOLUME_BOX || type == RTW_PRIM_VOLUME_SPHERE; }

// Scene tables are immutable for the lifetime of a launch, so they are read through the constant
// address space: a wave-uniform index then becomes an s_load into SGPRs (scalar cache, no VGPRs,
// no vector-memory latency), a divergent index an ordinary global_load.
#define RTW_CONST __attribute__((address_space(4)))
template <class T>
RTW_DEV const RTW_CONST T* as_const(const T* p) { return (const RTW_CONST T*)(uint64_t)p; }

RTW_DEV rtw_prim load_prim(const DScene& sc, int i) {
    const RTW_CONST u32x4* q = (const RTW_CONST u32x4*)(uint64_t)(sc.prims + i);
    u32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    rtw_prim r;
    r.type = (int)a.x; r.material = (int)a.y; r.xform = (int)a.z; r.flip = (int)a.w;
    r.p[0] = __uint_as_float(b.x); r.p[1] = __uint_as_float(b.y); r.p[2] = __uint_as_float(b.z); r.p[3] = __uint_as_float(b.w);
    r.p[4] = __uint_as_float(c.x); r.p[5] = __uint_as_float(c.y); r.p[6] = __uint_as_float(c.z); r.p[7] = __uint_as_float(c.w);
    r.p[8] = __uint_as_float(d.x); r.p[9] = __uint_as_float(d.y); r.p[10] = __uint_as_float(d.z); r.p[11] = __uint_as_float(d.w);
    return r;
}
struct M34 { float m[12]; };
RTW_DEV M34 load_m34(const float* p) {
    const RTW_CONST f32x4* q = (const RTW_CONST f32x4*)(uint64_t)p;
    f32x4 a = q[0], b = q[1], c = q[2];
    M34 r;
    r.m[0] = a.x; r.m[1] = a.y; r.m[2] = a.z; r.m[3] = a.w;
    r.m[4] = b.x; r.m[5] = b.y; r.m[6] = b.z; r.m[7] = b.w;
    r.m[8] = c.x; r.m[9] = c.y; r.m[10] = c.z; r.m[11] = c.w;
    return r;
}
RTW_DEV M34 load_xf_m(const DScene& sc, int i) { return load_m34(sc.xforms[i].m); }
RTW_DEV M34 load_xf_inv(const DScene& sc, int i) { return load_m34(sc.xforms[i].inv); }
RTW_DEV int load_i32(const int32_t* p) { return *as_const(p); }
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
RTW_DEV BruteGroup load_group(const DScene& sc, int i) {
    const u32x8 q = *(const RTW_CONST u32x8*)(uint64_t)(sc.groups + i);
    BruteGroup g;
    g.xform = (int)q[0]; g.first = (int)q[1]; g.n_rx = (int)q[2]; g.n_ry = (int)q[3]; g.n_rz = (int)q[4]; g.n_sph = (int)q[5];
    g.pad0 = 0; g.pad1 = 0;
    return g;
}
RTW_DEV BruteRec load_rec(const DScene& sc, int i) {
    const u32x8 q = *(const RTW_CONST u32x8*)(uint64_t)(sc.recs + i);
    BruteRec r;
    r.a = __uint_as_float(q[0]); r.b = __uint_as_float(q[1]); r.c = __uint_as_float(q[2]); r.d = __uint_as_float(q[3]);
    r.e = __uint_as_float(q[4]); r.prim = (int)q[5]; r.pad0 = 0; r.pad1 = 0;
    return r;
}
#ifndef RTW_LEAF_MAX
#define RTW_LEAF_MAX 1  // primitives per leaf of the tree (rtw_bvh.h builds to the same constant)
#endif
static constexpr uint32_t kBvhDone = 0xffffffffu;  // a reference with count bits = 3: neither an inner node nor a leaf that exists
// Per-thread traversal memory: this thread's column of the LDS stack, the block's LDS copy of the top of the tree and
// of the first leaf records. Stack entries are 16 bits wide whenever every reference of the tree fits (the LDS the
// stacks take decides how many waves a CU holds), else 32.
struct TravMem {
    uint16_t* stack16;
    uint32_t* stack32;
    uint32_t stride;
    const u32x4* nodes;   // LDS, n_nodes * 4 vectors
    uint32_t n_nodes;
    const u32x4* leaves;  // LDS, n_leaves * 2 vectors
    uint32_t n_leaves;
    bool wide;
};

// Every thread of the block calls this once before its first traverse<> (it holds a barrier).
RTW_DEV TravMem trav_mem(const DScene& sc, uint32_t* lds, uint32_t block, uint32_t tid) {
    TravMem tm;
    tm.wide = sc.stack_wide != 0;
    // rows 0 and 1 of a column hold "nothing left" for good (a pop of the empty stack reads row 1; bvh_step16 looks two
    // entries down), the stack proper starts at row 2
    if (tm.wide) { lds[tid] = kBvhDone; lds[block + tid] = kBvhDone; }
    else { ((uint16_t*)lds)[tid] = 0xffffu; ((uint16_t*)lds)[block + tid] = 0xffffu; }
    // A lane's 16-bit column inside a row of its wave: lane l sits in dword l % 32, half l / 32. LDS stores and dword reads are
    // served in two groups of 32 lanes (banks of 4 bytes): with lanes 2k and 2k + 1 sharing a dword, two lanes of a group met on
    // one bank whenever their stack heights differed (measured: SQ_LDS_BANK_CONFLICT 1.4x the LDS-active cycles of k_trace_bvh);
    // this way every lane of a group has a bank of its own whatever the heights.
#ifndef RTW_STACK_LINEAR
    tm.stack16 = (uint16_t*)lds + 2u * block + ((tid & ~63u) | ((tid & 31u) << 1) | ((tid >> 5) & 1u));
#else
    tm.stack16 = (uint16_t*)lds + 2u * block + tid;
#endif
    tm.stack32 = lds + 2u * block + tid;
    tm.stride = block;
    const uint32_t stack_words = tm.wide ? (uint32_t)sc.stack_depth * block : ((uint32_t)sc.stack_depth * block + 1u) / 2u;
    u32x4* cache = (u32x4*)(lds + ((stack_words + 3u) & ~3u));
    tm.nodes = cache;
    tm.n_nodes = (uint32_t)sc.n_lds_nodes;
    tm.leaves = cache + tm.n_nodes * 4u;
    tm.n_leaves = (uint32_t)sc.n_lds_leaves;
    if (tm.n_nodes) {
        const uint32_t nv = tm.n_nodes * 4u, nl = tm.n_leaves * 2u;
#ifdef RTW_NODE_SWIZZLE
        // quad k of node i sits at quad k ^ ((i >> 1) & 3) of the node's 64 bytes (see bvh_node_quads)
        for (uint32_t i = tid; i < nv; i += block) cache[(i & ~3u) | ((i ^ (i >> 3)) & 3u)] = sc.nodes[i];
#else
        for (uint32_t i = tid; i < nv; i += block) cache[i] = sc.nodes[i];
#endif
        for (uint32_t i = tid; i < nl; i += block) cache[nv + i] = sc.leaves[i];
        __syncthreads();
    }
    return tm;
}

// geometry/movingSphere.cu:33-39
RTW_DEV v3 moving_center(const rtw_prim& pr, float time) {
    v3 c0 = ld3(&pr.p[0]);
    float t0 = pr.p[7], t1 = pr.p[8];
    if (t0 == t1) return c0;
    v3 c1 = ld3(&pr.p[4]);
    float u = (time - t0) / (t1 - t0);
    return vfma(vsub(c1, c0), u, c0);
}

// geometry/sphere.cu:52-60,93-95
RTW_DEV bool sphere_roots(v3 o, v3 d, v3 c, float r, float tmin, float tmax, float& t_out) {
    v3 oc = vsub(o, c);
    float a = dot3(d, d);
    float b = dot3(oc, d);
    float cc = fma_(-r, r, dot3(oc, oc));
    float disc = fma_(b, b, -(a * cc));
    if (disc < 0.0f) return false;
    float sq = __builtin_sqrtf(disc);
    float t = (-b - sq) / a;
    if (t < tmax && t > tmin) { t_out = t; return true; }
    t = (-b + sq) / a;
    if (t < tmax && t > tmin) { t_out = t; return true; }
    return false;
}

// geometry/volumeBox.cu:29-52
RTW_DEV bool box_boundary(float temp1, float temp2, float tMin, float tMax, float& rec) {
    if (temp1 > temp2) return false;
    if (temp1 < tMax && temp1 > tMin) { rec = temp1; return true; }
    if (temp2 < tMax && temp2 > tMin) { rec = temp2; return true; }
    return false;
}

// Object-space ray of primitive pr for world ray (o,d).
RTW_DEV void object_ray(const DScene& sc, const rtw_prim& pr, v3 o, v3 d, float ray_time, v3& oo, v3& dd, v3& motion) {
    oo = o; dd = d;
    if (pr.xform != 0) {
        M34 inv = load_xf_inv(sc, pr.xform);
        oo = xf_point(inv.m, o);
        dd = xf_vector(inv.m, d);
    }
    motion = V(0.f, 0.f, 0.f);
    if (pr.type == RTW_PRIM_MOVING_SPHERE) {
        // matrix-motion transform translate(lerp(C0,C1,rayTime)), geometry/ioMovingSphere.h:161-203
        v3 c0 = ld3(&pr.p[0]), c1 = ld3(&pr.p[4]);
        motion = vfma(vsub(c1, c0), ray_time, c0);
        oo = vsub(oo, motion);
    }
}

// One primitive's intersection program against an object-space ray; true when it reports a hit in (tmin,tmax_cur).
// inv = (1/dd.x, 1/dd.y, 1/dd.z) by IEEE division: the caller computes it once per object-space ray and
// reuses it for every rectangle / box under the same transform (same bits as dividing per primitive).
template <class RNG>
// bounded (corrected estimators): a medium only scatters inside its extent; the reference does not test that (Q9)
RTW_DEV bool prim_test(const rtw_prim& pr, v3 oo, v3 dd, v3 inv, float tmin, float tmax_cur, float gather_time, RNG& g, float& t_out,
                       bool bounded = false) {
    switch (pr.type) {
    case RTW_PRIM_SPHERE:
        return sphere_roots(oo, dd, ld3(&pr.p[0]), pr.p[3], tmin, tmax_cur, t_out);
    case RTW_PRIM_MOVING_SPHERE:
        return sphere_roots(oo, dd, moving_center(pr, gather_time), pr.p[3], tmin, tmax_cur, t_out);
    case RTW_PRIM_RECT_X:
    case RTW_PRIM_RECT_Y:
    case RTW_PRIM_RECT_Z: {
        // shaders/aarectx.cu:8-22, aarecty.cu:8-22, aarectz.cu:9-23
        float ok, ik, oa, da, ob, db;
        if (pr.type == RTW_PRIM_RECT_X) { ok = oo.x; ik = inv.x; oa = oo.y; da = dd.y; ob = oo.z; db = dd.z; }
        else if (pr.type == RTW_PRIM_RECT_Y) { ok = oo.y; ik = inv.y; oa = oo.x; da = dd.x; ob = oo.z; db = dd.z; }
        else { ok = oo.z; ik = inv.z; oa = oo.x; da = dd.x; ob = oo.y; db = dd.y; }
        float t = (pr.p[4] - ok) * ik;
        if (!(t >= tmin && t < tmax_cur)) return false;
        float a = fma_(t, da, oa);
        float b = fma_(t, db, ob);
        if (!(a >= pr.p[0] && a <= pr.p[1] && b >= pr.p[2] && b <= pr.p[3])) return false;
        t_out = t;
        return true;
    }
    case RTW_PRIM_VOLUME_BOX: {
        // geometry/volumeBox.cu:55-113 (Q8: EPSILON is the integer 0; Q9: extent is not tested)
        v3 t0 = vmul(vsub(ld3(&pr.p[0]), oo), inv);
        v3 t1 = vmul(vsub(ld3(&pr.p[3]), oo), inv);
        float temp1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0.x, t1.x), __builtin_fminf(t0.y, t1.y)), __builtin_fminf(t0.z, t1.z));
        float temp2 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0.x, t1.x), __builtin_fmaxf(t0.y, t1.y)), __builtin_fmaxf(t0.z, t1.z));
        float h1, h2;
        if (!box_boundary(temp1, temp2, -RTW_FLT_MAX, RTW_FLT_MAX, h1)) return false;
        if (!box_boundary(temp1, temp2, h1, RTW_FLT_MAX, h2)) return false;
        if (h1 < tmin) h1 = tmin;
        if (h2 > tmax_cur) h2 = tmax_cur;
        if (h1 >= h2) return false;
        if (h1 < 0.f) h1 = 0.f;
        float len = length3(dd);
        float hit_distance = -(1.0f / pr.p[6]) * log_spec(g.randf1());
        float t = h1 + hit_distance / len;
        if (!(t >= tmin && t < tmax_cur)) return false;
        if (bounded && !(t < h2)) return false;
        t_out = t;
        return true;
    }
    case RTW_PRIM_VOLUME_SPHERE: {
        // geometry/volumeSphere.cu:67-127
        v3 c = ld3(&pr.p[0]);
        float h1, h2;
        if (!sphere_roots(oo, dd, c, pr.p[3], -RTW_FLT_MAX, RTW_FLT_MAX, h1)) return false;
        if (!sphere_roots(oo, dd, c, pr.p[3], h1, RTW_FLT_MAX, h2)) return false;
        if (h1 < tmin) h1 = tmin;
        if (h2 > tmax_cur) h2 = tmax_cur;
        if (h1 >= h2) return false;
        if (h1 < 0.f) h1 = 0.f;
        float len = length3(dd);
        float hit_distance = -(1.0f / pr.p[4]) * log_spec(g.next1());
        float t = h1 + hit_distance / len;
        if (!(t >= tmin && t < tmax_cur)) return false;
        if (bounded && !(t < h2)) return false;
        t_out = t;
        return true;
    }
    default:
        return false;
    }
}

// ---- tree walk, shared by traverse<> and the refilling trace kernels (rtw_kernels.h k_trace_bvh, k_path_tree) ----

// sp is the byte offset of the top of this thread's stack column (0 = empty). A pop never branches: under the stack
// lies a row that says "nothing left" (16-bit entries are sign-extended, so 0xffff reads as kBvhDone; references stay
// below 0x8000 in that mode), and a walk that has popped it is over.
// The four 16-byte quads of node i from the workgroup's LDS image. Lanes read quad k of DIFFERENT nodes in one ds_read_b128, and
// nodes are 64 bytes apart: unswizzled, every lane's quad k lies in the same 4 of a 128-byte bank row's 8 quad positions (node
// parity picks the half), so the instruction is served from 8 of the 32 banks. RTW_NODE_SWIZZLE stores quad k at position
// k ^ ((i >> 1) & 3): one instruction's reads then spread over all positions. Measured (round 3, bit-exact, off by default): scene 1
// 7 334 -> 7 207 Msamples/s, scene 2 3 693 -> 3 615 (medians of 3; k_trace_bvh +2.5 %): the six extra address instructions per
// node cost more than the conflicts - same-node lanes near the root are broadcast reads either way.
RTW_DEV void bvh_node_quads(const u32x4* nodes, uint32_t i, u32x4& q0, u32x4& q1, u32x4& q2, u32x4& q3) {
    const u32x4* q = nodes + 4u * i;
#ifdef RTW_NODE_SWIZZLE
    const uint32_t f = (i >> 1) & 3u;
    q0 = q[f]; q1 = q[f ^ 1u]; q2 = q[f ^ 2u]; q3 = q[f ^ 3u];
#else
    q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
#endif
}
RTW_DEV uint32_t bvh_top(const TravMem& tm, int sp) {
    if (tm.wide) return *(const uint32_t*)((const char*)tm.stack32 + (sp - (int)(tm.stride * 4u)));
    return (uint32_t)(int32_t)*(const int16_t*)((const char*)tm.stack16 + (sp - (int)(tm.stride * 2u)));
}
RTW_DEV uint32_t bvh_pop(const TravMem& tm, int& sp) {
    const uint32_t r = bvh_top(tm, sp);
    sp -= (int)(tm.stride * (tm.wide ? 4u : 2u));
    return r;
}

// One step through inner node `cur`: the four children's boxes against [tmin, best_t]; the nearest child that is hit is
// entered, the others are pushed. Returns the next reference (inner or leaf), or kBvhDone when nothing is left.
// The boxes only cull (exact hits are decided by the primitive tests), so everything here may be approximate as long as
// it errs towards "hit": planes are p + q * step on the node's grid (rounded outwards by the builder, and the primitive
// bounds under them are padded by 1e-4), plane distances q * (step / d) + (p - o) / d with the reciprocal clamped to
// +-1e18 (a zero direction component then gives huge finite distances of the right sign instead of inf - inf).
RTW_DEV uint32_t bvh_inner_step(const DScene& sc, const TravMem& tm, const v3 o, const v3 inv, float tmin, float best_t, uint32_t cur, int& sp) {
    const uint32_t i = cur >> 2;
    u32x4 q0, q1, q2, q3;
    if (i < tm.n_nodes) {
        bvh_node_quads(tm.nodes, i, q0, q1, q2, q3);
    } else {
        const RTW_CONST u32x4* q = (const RTW_CONST u32x4*)(uint64_t)(sc.nodes + 4u * i);
        q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
    }
    // q0 = p.x p.y p.z step.x, q1 = lo.x lo.y lo.z hi.x, q2 = hi.y hi.z step.y step.z, q3 = the references
    const float ix = __builtin_amdgcn_fmed3f(inv.x, -1.0e18f, 1.0e18f), iy = __builtin_amdgcn_fmed3f(inv.y, -1.0e18f, 1.0e18f),
                iz = __builtin_amdgcn_fmed3f(inv.z, -1.0e18f, 1.0e18f);
    const float ax = (__uint_as_float(q0.x) - o.x) * ix, ay = (__uint_as_float(q0.y) - o.y) * iy, az = (__uint_as_float(q0.z) - o.z) * iz;
    const float bx = __uint_as_float(q0.w) * ix, by = __uint_as_float(q2.z) * iy, bz = __uint_as_float(q2.w) * iz;
    // the plane a ray meets first on an axis is lo when it travels upwards
    const uint32_t nx = ix < 0.0f ? q1.w : q1.x, fx = ix < 0.0f ? q1.x : q1.w;
    const uint32_t ny = iy < 0.0f ? q2.x : q1.y, fy = iy < 0.0f ? q1.y : q2.x;
    const uint32_t nz = iz < 0.0f ? q2.y : q1.z, fz = iz < 0.0f ? q1.z : q2.y;
    uint32_t key[4];
#define RTW_Q4_CHILD(C_, REF_)                                                                                                   \
    {                                                                                                                            \
        const float tnx = fma_((float)((nx >> (8 * C_)) & 0xffu), bx, ax), tfx = fma_((float)((fx >> (8 * C_)) & 0xffu), bx, ax); \
        const float tny = fma_((float)((ny >> (8 * C_)) & 0xffu), by, ay), tfy = fma_((float)((fy >> (8 * C_)) & 0xffu), by, ay); \
        const float tnz = fma_((float)((nz >> (8 * C_)) & 0xffu), bz, az), tfz = fma_((float)((fz >> (8 * C_)) & 0xffu), bz, az); \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));                                \
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, best_t));                              \
        const bool hit = tn <= tf && (REF_) != kBvhDone;                                                                        \
        key[C_] = hit ? ((__float_as_uint(tn) & ~3u) | (uint32_t)C_) : 0xffffffffu;                                             \
    }
    RTW_Q4_CHILD(0, q3.x) RTW_Q4_CHILD(1, q3.y) RTW_Q4_CHILD(2, q3.z) RTW_Q4_CHILD(3, q3.w)
#undef RTW_Q4_CHILD
    // tn >= tmin > 0, so the distances order as unsigned integers; the slot in the low bits makes the keys distinct
    const uint32_t kmin = min(min(key[0], key[1]), min(key[2], key[3]));
    // every slot is stored at the top of the stack and the top moves past the ones that stay (hit, not the nearest): no
    // divergent branch per slot; the column has one entry of slack for the last store
    const bool p0 = key[0] != 0xffffffffu && key[0] != kmin, p1 = key[1] != 0xffffffffu && key[1] != kmin,
               p2 = key[2] != 0xffffffffu && key[2] != kmin, p3 = key[3] != 0xffffffffu && key[3] != kmin;
    if (tm.wide) {
        char* b = (char*)tm.stack32;
        const int st = (int)(tm.stride * 4u);
        *(uint32_t*)(b + sp) = q3.x; sp += p0 ? st : 0;
        *(uint32_t*)(b + sp) = q3.y; sp += p1 ? st : 0;
        *(uint32_t*)(b + sp) = q3.z; sp += p2 ? st : 0;
        *(uint32_t*)(b + sp) = q3.w; sp += p3 ? st : 0;
    } else {
        char* b = (char*)tm.stack16;
        const int st = (int)(tm.stride * 2u);
        *(uint16_t*)(b + sp) = (uint16_t)q3.x; sp += p0 ? st : 0;
        *(uint16_t*)(b + sp) = (uint16_t)q3.y; sp += p1 ? st : 0;
        *(uint16_t*)(b + sp) = (uint16_t)q3.z; sp += p2 ? st : 0;
        *(uint16_t*)(b + sp) = (uint16_t)q3.w; sp += p3 ? st : 0;
    }
    const uint32_t top = bvh_top(tm, sp);
    const bool none = kmin == 0xffffffffu;
    sp -= none ? (int)(tm.stride * (tm.wide ? 4u : 2u)) : 0;
    const uint32_t slot = kmin & 3u;
    return none ? top : slot == 0u ? q3.x : slot == 1u ? q3.y : slot == 2u ? q3.z : q3.w;
}

// The same step for trees whose references fit 16 bits (every scene of a few thousand primitives), written without a
// divergent branch: the slot of the nearest child is picked with selects, the stack top and the entry under it are read
// together, and a leaf that comes up while the lane has none put aside goes to `pend` at once (the lane walks on with the
// next entry: see k_trace_bvh). cur must be an inner reference; on return cur is the next reference of any kind, or
// kBvhDone. ALL_LDS: the workgroup's LDS image holds every node (no global fetch, so no vmcnt wait in the loop).
template <bool ALL_LDS, bool ASIDE = true>
RTW_DEV void bvh_step16(const DScene& sc, const TravMem& tm, const v3 o, const v3 inv, float tmin, float best_t, uint32_t& cur, uint32_t& pend, int& sp) {
    const uint32_t i = cur >> 2;
    u32x4 q0, q1, q2, q3;
    if (ALL_LDS || i < tm.n_nodes) {
        bvh_node_quads(tm.nodes, i, q0, q1, q2, q3);
    } else {
        const RTW_CONST u32x4* q = (const RTW_CONST u32x4*)(uint64_t)(sc.nodes + 4u * i);
        q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
    }
    const float ix = __builtin_amdgcn_fmed3f(inv.x, -1.0e18f, 1.0e18f), iy = __builtin_amdgcn_fmed3f(inv.y, -1.0e18f, 1.0e18f),
                iz = __builtin_amdgcn_fmed3f(inv.z, -1.0e18f, 1.0e18f);
    const float ax = (__uint_as_float(q0.x) - o.x) * ix, ay = (__uint_as_float(q0.y) - o.y) * iy, az = (__uint_as_float(q0.z) - o.z) * iz;
    const float bx = __uint_as_float(q0.w) * ix, by = __uint_as_float(q2.z) * iy, bz = __uint_as_float(q2.w) * iz;
    const uint32_t nx = ix < 0.0f ? q1.w : q1.x, fx = ix < 0.0f ? q1.x : q1.w;
    const uint32_t ny = iy < 0.0f ? q2.x : q1.y, fy = iy < 0.0f ? q1.y : q2.x;
    const uint32_t nz = iz < 0.0f ? q2.y : q1.z, fz = iz < 0.0f ? q1.z : q2.y;
    uint32_t key[4];
    // an unused slot holds an inverted box (lo = 255, hi = 0 on every axis): its near plane lies 255 grid steps beyond its
    // far plane, so it never passes tn <= tf and needs no test of its own
#define RTW_Q4_CHILD(C_)                                                                                                         \
    {                                                                                                                            \
        const float tnx = fma_((float)((nx >> (8 * C_)) & 0xffu), bx, ax), tfx = fma_((float)((fx >> (8 * C_)) & 0xffu), bx, ax); \
        const float tny = fma_((float)((ny >> (8 * C_)) & 0xffu), by, ay), tfy = fma_((float)((fy >> (8 * C_)) & 0xffu), by, ay); \
        const float tnz = fma_((float)((nz >> (8 * C_)) & 0xffu), bz, az), tfz = fma_((float)((fz >> (8 * C_)) & 0xffu), bz, az); \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));                                \
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, best_t));                              \
        key[C_] = tn <= tf ? ((__float_as_uint(tn) & ~3u) | (uint32_t)C_) : 0xffffffffu;                                        \
    }
    RTW_Q4_CHILD(0) RTW_Q4_CHILD(1) RTW_Q4_CHILD(2) RTW_Q4_CHILD(3)
#undef RTW_Q4_CHILD
    const uint32_t kmin = min(min(key[0], key[1]), min(key[2], key[3]));
    // (a slot that is not hit has the key ~0, which is kmin only when nothing is hit: `!= kmin` alone would push it then)
    const bool p0 = key[0] != 0xffffffffu && key[0] != kmin, p1 = key[1] != 0xffffffffu && key[1] != kmin,
               p2 = key[2] != 0xffffffffu && key[2] != kmin, p3 = key[3] != 0xffffffffu && key[3] != kmin;
    char* b = (char*)tm.stack16;
    const int st = (int)(tm.stride * 2u);
    *(uint16_t*)(b + sp) = (uint16_t)q3.x; sp += p0 ? st : 0;
    *(uint16_t*)(b + sp) = (uint16_t)q3.y; sp += p1 ? st : 0;
    *(uint16_t*)(b + sp) = (uint16_t)q3.z; sp += p2 ? st : 0;
    *(uint16_t*)(b + sp) = (uint16_t)q3.w; sp += p3 ? st : 0;
    const uint32_t e1 = (uint32_t)(int32_t)*(const int16_t*)(b + (sp - st));       // the top of the stack
    const bool none = kmin == 0xffffffffu;
    const uint32_t r01 = (kmin & 1u) ? q3.y : q3.x, r23 = (kmin & 1u) ? q3.w : q3.z;
    const uint32_t sel = (kmin & 2u) ? r23 : r01;
    const uint32_t c1 = none ? e1 : sel;     // the next reference
    if (ASIDE) {
        const uint32_t e2 = (uint32_t)(int32_t)*(const int16_t*)(b + (sp - 2 * st));   // what lies under the top
        const uint32_t c2 = none ? e2 : e1;  // the reference after c1, should c1 be a leaf that is put aside
        sp -= none ? st : 0;
        const bool aside = pend == 0u && ((c1 & 3u) - 1u) < 2u;
        pend = aside ? c1 : pend;
        cur = aside ? c2 : c1;
        sp -= aside ? st : 0;
    } else {  // (walks that test a leaf as soon as it comes up: traverse<>)
        sp -= none ? st : 0;
        cur = c1;
    }
}

RTW_DEV bool uses_inv(int type) { return type >= RTW_PRIM_RECT_X && type <= RTW_PRIM_VOLUME_BOX; }
RTW_DEV v3 recip3(v3 d) { return V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
// The walks that test surfaces only hand this to prim_test (volume kinds, which draw, never reach a tree)
struct NoDraw {
    RTW_DEV float randf1() { return 0.0f; }
    RTW_DEV float next1() { return 0.0f; }
};

// One leaf record (slot k of the leaf table) against a ray, for every walk of the tree (surfaces only: no intersection
// program that reaches a tree draws random numbers). Untransformed spheres, moving spheres and rectangles - most of any
// scene - are tested straight from their slots, the rectangle's axes picked with selects; primitives under an instance
// transform take the general route through object_ray / prim_test. The arithmetic per kind is prim_test's, so the hit
// distances are the same bits. inv = recip3(d), computed once per ray. next = the slot after this record.
RTW_DEV bool leaf_test(const DScene& sc, const TravMem& tm, uint32_t k, const v3 o, const v3 d, const v3 inv, float tmin, float ray_time,
                       float gather_time, float& t, int& prim, uint32_t& next) {
    u32x4 a, b, c;
    c = u32x4{0u, 0u, 0u, 0u};
    // (the slot after the record's first is only a moving sphere's second centre; scenes without one never fetch it)
    if (k + 1u < tm.n_leaves) {
        a = tm.leaves[2u * k]; b = tm.leaves[2u * k + 1u];
        if (sc.has_motion) c = tm.leaves[2u * k + 2u];
    } else {
        const RTW_CONST u32x4* q = (const RTW_CONST u32x4*)(uint64_t)(sc.leaves + 2u * k);
        a = q[0]; b = q[1];
        if (sc.has_motion) c = q[2];
    }
    prim = (int)b.y;
    const uint32_t tx = b.z;  // type | xform << 8
    next = k + ((tx & 0xffu) == (uint32_t)RTW_PRIM_MOVING_SPHERE ? 2u : 1u);
    const v3 c0 = V(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
    if (tx <= (uint32_t)RTW_PRIM_MOVING_SPHERE) {  // RTW_PRIM_SPHERE (0) or RTW_PRIM_MOVING_SPHERE (1), untransformed
        // One call of the intersection program (geometry/sphere.cu:52-60,93-95) for both kinds - a wave that holds both
        // would otherwise run its square root and divisions twice - on per-lane (origin, centre): a static sphere's are
        // the ray's and its own; a moving sphere's are the ray under the matrix-motion transform
        // translate(lerp(C0, C1, rayTime)) (geometry/ioMovingSphere.h:161-203) and the program's own centre at the gather
        // time (geometry/movingSphere.cu:33-39): what object_ray + prim_test compute.
        v3 oo = o, ctr = c0;
        if (tx == (uint32_t)RTW_PRIM_MOVING_SPHERE) {
            const v3 dc = vsub(V(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)), c0);
            const float t0 = __uint_as_float(b.x), t1 = __uint_as_float(b.w);
            oo = vsub(o, vfma(dc, ray_time, c0));
            if (t0 != t1) ctr = vfma(dc, (gather_time - t0) / (t1 - t0), c0);
        }
        return sphere_roots(oo, d, ctr, __uint_as_float(a.w), tmin, RTW_FLT_MAX, t);
    }
    if (tx - (uint32_t)RTW_PRIM_RECT_X <= (uint32_t)(RTW_PRIM_RECT_Z - RTW_PRIM_RECT_X)) {
        // shaders/aarectx.cu:8-22, aarecty.cu:8-22, aarectz.cu:9-23
        const bool isx = tx == (uint32_t)RTW_PRIM_RECT_X, isz = tx == (uint32_t)RTW_PRIM_RECT_Z;
        const float ok = isx ? o.x : isz ? o.z : o.y, ik = isx ? inv.x : isz ? inv.z : inv.y;
        const float oa = isx ? o.y : o.x, da = isx ? d.y : d.x;
        const float ob = isz ? o.y : o.z, db = isz ? d.y : d.z;
        const float tt = (__uint_as_float(b.x) - ok) * ik;
        const float aa = fma_(tt, da, oa);
        const float bb = fma_(tt, db, ob);
        t = tt;
        return tt >= tmin && tt < RTW_FLT_MAX && aa >= __uint_as_float(a.x) && aa <= __uint_as_float(a.y) && bb >= __uint_as_float(a.z) &&
               bb <= __uint_as_float(a.w);
    }
    const int type = (int)(tx & 0xffu);
    rtw_prim pr;
    if (type != RTW_PRIM_SPHERE && !(type >= RTW_PRIM_RECT_X && type <= RTW_PRIM_RECT_Z)) pr = load_prim(sc, prim);
    else {
        pr.type = type; pr.material = 0; pr.xform = (int)(tx >> 8); pr.flip = 0;
        pr.p[0] = __uint_as_float(a.x); pr.p[1] = __uint_as_float(a.y); pr.p[2] = __uint_as_float(a.z); pr.p[3] = __uint_as_float(a.w);
        pr.p[4] = __uint_as_float(b.x);
        for (int j = 5; j < 12; j++) pr.p[j] = 0.0f;
    }
    v3 po, pd, mt;
    object_ray(sc, pr, o, d, ray_time, po, pd, mt);
    v3 pinv = inv;
    if (pr.xform != 0 && uses_inv(pr.type)) pinv = recip3(pd);
    NoDraw ng;
    return prim_test(pr, po, pd, pinv, tmin, RTW_FLT_MAX, gather_time, ng, t);
}

// Conservative: false only when the ray certainly stays outside the scene bounds (NaNs from 0 * inf answer "may hit").
RTW_DEV bool may_hit_scene(const DScene& sc, const v3 o, const v3 d) {
    // hardware reciprocals (1 ulp): the bounds are padded by 1 % of the scene diagonal, and the answer only decides
    // whether a wave walks its candidate lists, never what a walk returns
    const v3 inv = V(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const float ax = (sc.bmin[0] - o.x) * inv.x, bx = (sc.bmax[0] - o.x) * inv.x;
    const float ay = (sc.bmin[1] - o.y) * inv.y, by = (sc.bmax[1] - o.y) * inv.y;
    const float az = (sc.bmin[2] - o.z) * inv.z, bz = (sc.bmax[2] - o.z) * inv.z;
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fmaxf(__builtin_fminf(az, bz), 0.0f));
    const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    const bool any_nan = !(ax == ax && bx == bx && ay == ay && by == by && az == az && bz == bz);
    return any_nan || !(tnear > tfar * 1.001f);
}


// The volume primitives' share of a traversal: tested first, in index order, each against the interval the earlier
// ones left (their intersection programs draw random numbers: geometry/volumeBox.cu:79, volumeSphere.cu:93).
// Surfaces then only win with a strictly smaller t, so a kernel that holds the generator can run this pass on its
// own and combine it with a surface-only closest hit found elsewhere: min over both, the volume keeping ties.
template <class RNG, bool ANY_HIT>
RTW_DEV bool volume_pass(const DScene& sc, v3 o, v3 d, float tmin, float ray_time, float gather_time, RNG& g, float& best_t, int& best_prim) {
    bool hit = false;
    for (int k = 0; k < sc.n_vol; k++) {
        int pi = load_i32(sc.order + k);
        const rtw_prim pr = load_prim(sc, pi);
        v3 po, pd, mt;
        object_ray(sc, pr, o, d, ray_time, po, pd, mt);
        float t;
        if (prim_test(pr, po, pd, recip3(pd), tmin, best_t, gather_time, g, t, sc.estimator != 0)) {
            best_t = t;
            best_prim = pi;
            hit = true;
            if (ANY_HIT) return true;
        }
    }
    return hit;
}

// Small scenes (the scalar-cache candidate lists): every lane walks the same records, so they arrive as SGPR operands.
// Per instance transform: one object-space ray, one reciprocal direction, then straight-line tests (rect x / y / z
// lists, spheres). The record after the one under test is already on its way (the scalar loads return out of order, so a
// wave cannot keep more than "everything issued so far" apart: one s_waitcnt per record, hidden behind the test).
// GENERIC = false: the caller knows the scene has no moving spheres (sc.n_generic == 0)
template <class RNG, bool ANY_HIT, bool SKIP_VOLUMES, bool GENERIC = true>
RTW_DEV void traverse_brute(const DScene& sc, v3 o, v3 d, float tmin, float tmax, float ray_time, float gather_time, RNG& g,
                            float& best_t, int& best_prim) {
    best_t = tmax;
    best_prim = -1;
    bool best_is_vol = false;
    if (!SKIP_VOLUMES) {
        best_is_vol = volume_pass<RNG, ANY_HIT>(sc, o, d, tmin, ray_time, gather_time, g, best_t, best_prim);
        if (ANY_HIT && best_is_vol) return;
    }
// straight-line candidate test: every comparison is evaluated, the update is one predicated select
#define RTW_TAKE(HIT_, T_, PI_)                                                                      \
    {                                                                                                \
        const bool tie_ = !ANY_HIT && ((T_) == best_t) & ((PI_) < best_prim) & !best_is_vol;         \
        const bool take_ = (HIT_) & (((T_) < best_t) | tie_);                                        \
        best_t = take_ ? (T_) : best_t;                                                              \
        best_prim = take_ ? (PI_) : best_prim;                                                       \
        best_is_vol = take_ ? false : best_is_vol;                                                   \
    }
    int ri = 0;
    BruteRec R = load_rec(sc, 0);  // (the table always holds one record more than it lists)
    for (int gi = 0; gi < sc.n_groups; gi++) {
        const BruteGroup G = load_group(sc, gi);
        v3 oo = o, dd = d;
        if (G.xform != 0) { M34 im = load_xf_inv(sc, G.xform); oo = xf_point(im.m, o); dd = xf_vector(im.m, d); }
        if (G.n_rx + G.n_ry + G.n_rz > 0) {
            const v3 inv = recip3(dd);
#define RTW_RECT_LOOP(N_, OK_, IK_, OA_, DA_, OB_, DB_)                                              \
            for (int i = 0; i < (N_); i++) {                                                         \
                const BruteRec Rn = load_rec(sc, ++ri);                                              \
                const float t = (R.e - (OK_)) * (IK_);                                               \
                const float a = fma_(t, (DA_), (OA_));                                               \
                const float b = fma_(t, (DB_), (OB_));                                               \
                const bool hit = (t >= tmin) & (a >= R.a) & (a <= R.b) & (b >= R.c) & (b <= R.d);    \
                RTW_TAKE(hit, t, R.prim)                                                             \
                R = Rn;                                                                              \
            }
            RTW_RECT_LOOP(G.n_rx, oo.x, inv.x, oo.y, dd.y, oo.z, dd.z)   // shaders/aarectx.cu:8-22
            RTW_RECT_LOOP(G.n_ry, oo.y, inv.y, oo.x, dd.x, oo.z, dd.z)   // shaders/aarecty.cu:8-22
            RTW_RECT_LOOP(G.n_rz, oo.z, inv.z, oo.x, dd.x, oo.y, dd.y)   // shaders/aarectz.cu:9-23
#undef RTW_RECT_LOOP
        }
        for (int i = 0; i < G.n_sph; i++) {
            const BruteRec Rn = load_rec(sc, ++ri);
            float t = 0.f;
            const bool hit = sphere_roots(oo, dd, V(R.a, R.b, R.c), R.d, tmin, RTW_FLT_MAX, t);
            RTW_TAKE(hit, t, R.prim)
            R = Rn;
        }
    }
#undef RTW_TAKE
    if (!GENERIC || (ANY_HIT && best_prim >= 0)) return;
    // moving spheres (own motion transform per candidate): generic path
    for (int k = 0; k < sc.n_generic; k++) {
        const int pi = load_i32(sc.order + sc.n_vol + k);
        const rtw_prim pr = load_prim(sc, pi);
        v3 po, pd, mt;
        object_ray(sc, pr, o, d, ray_time, po, pd, mt);
        float t;
        if (prim_test(pr, po, pd, V(0.f, 0.f, 0.f), tmin, RTW_FLT_MAX, gather_time, g, t)) {
            if (t < best_t || (!ANY_HIT && t == best_t && best_prim >= 0 && !best_is_vol && pi < best_prim)) {
                best_t = t; best_prim = pi; best_is_vol = false;
                if (ANY_HIT) return;
            }
        }
    }
}

// ---- k_path's walk: the candidate lists in LDS ----------------------------------------------------------------------
// The same groups, records and order as traverse_brute, copied by the workgroup into LDS at kernel start and read by
// every lane of a wave at the same address (broadcast reads, returned in order): while one record is under test the
// next one is already on its way (two register sets, A and B, take turns), so a walk waits for LDS once, not once per
// candidate. Image, in 16-byte words: groups (2 words each: BruteGroup), then per group the 3 rows of its world->object
// matrix, then the records (2 words each: BruteRec with prim + 1 in place of prim), then two words of padding.
// The arithmetic per ray is traverse_brute's; the tests are stated so that they need fewer instructions:
//   lo <= a <= hi        as  med3(a, lo, hi) == a   (finite lo <= hi; a NaN fails both forms)
//   closest hit, ties to the lowest primitive index: (t, prim + 1) < (best_t, best_prim + 1) as ONE unsigned 64-bit
//                        compare of (bits(t) << 32 | prim + 1): t > 0 where it matters, so bits(t) orders like t
//   any hit              no running minimum at all: occluded |= hit & (t < tmax)
constexpr int kWalkMaxWords = 400;  // 6.4 KB of LDS; scenes whose lists are larger use the wavefront kernels
struct WalkRec { u32x4 q0, q1; };
RTW_DEV bool in_range(float a, uint32_t lo, uint32_t hi) { return __builtin_amdgcn_fmed3f(a, __uint_as_float(lo), __uint_as_float(hi)) == a; }

template <bool ANY_HIT>
RTW_DEV void walk_lds(const u32x4* __restrict__ w, const int n_groups, const v3 o, const v3 d, const float tmin, const float tmax, float& best_t, int& best_prim) {
    // closest hit: key = bits(best_t) << 32 | best_prim + 1; any hit: key_lo = 1 once occluded
    uint32_t key_hi = __float_as_uint(tmax), key_lo = 0u;
    const u32x4* xf = w + 2 * n_groups;
    const u32x4* r = w + 5 * n_groups;  // the record stream: groups follow each other without gaps
    WalkRec A, B;
    A.q0 = r[0]; A.q1 = r[1];
#define RTW_RECT_TEST(R_, OK_, IK_, OA_, DA_, OB_, DB_)                                              \
    {                                                                                                \
        const float t = (__uint_as_float(R_.q1.x) - (OK_)) * (IK_);                                  \
        const float a = fma_(t, (DA_), (OA_));                                                       \
        const float b = fma_(t, (DB_), (OB_));                                                       \
        const bool hit = (t >= tmin) & in_range(a, R_.q0.x, R_.q0.y) & in_range(b, R_.q0.z, R_.q0.w); \
        if (ANY_HIT) {                                                                               \
            key_lo = (hit & (t < tmax)) ? 1u : key_lo;                                               \
        } else {                                                                                     \
            const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | R_.q1.y; \
            const bool take = hit & (key < (((unsigned long long)key_hi << 32) | key_lo));           \
            key_hi = take ? __float_as_uint(t) : key_hi;                                             \
            key_lo = take ? R_.q1.y : key_lo;                                                        \
        }                                                                                            \
    }
// n records of one kind: A holds the first on entry and the first record after the list on exit
#define RTW_REC_LIST(N_, TEST_)                                                                      \
    {                                                                                                \
        int n_ = (N_);                                                                               \
        for (; n_ >= 2; n_ -= 2) {                                                                   \
            B.q0 = r[2]; B.q1 = r[3];                                                                \
            TEST_(A)                                                                                 \
            A.q0 = r[4]; A.q1 = r[5];                                                                \
            TEST_(B)                                                                                 \
            r += 4;                                                                                  \
        }                                                                                            \
        if (n_ > 0) {                                                                                \
            B.q0 = r[2]; B.q1 = r[3];                                                                \
            TEST_(A)                                                                                 \
            A = B;                                                                                   \
            r += 2;                                                                                  \
        }                                                                                            \
    }
    for (int gi = 0; gi < n_groups; gi++) {
        const u32x4 g0 = w[2 * gi], g1 = w[2 * gi + 1];
        const int g_xform = (int)__builtin_amdgcn_readfirstlane(g0.x);
        const int n_rx = (int)__builtin_amdgcn_readfirstlane(g0.z), n_ry = (int)__builtin_amdgcn_readfirstlane(g0.w);
        const int n_rz = (int)__builtin_amdgcn_readfirstlane(g1.x), n_sph = (int)__builtin_amdgcn_readfirstlane(g1.y);
        v3 oo = o, dd = d;
        if (g_xform != 0) {
            const u32x4 r0 = xf[3 * gi], r1 = xf[3 * gi + 1], r2 = xf[3 * gi + 2];
            float m[12];
            m[0] = __uint_as_float(r0.x); m[1] = __uint_as_float(r0.y); m[2] = __uint_as_float(r0.z); m[3] = __uint_as_float(r0.w);
            m[4] = __uint_as_float(r1.x); m[5] = __uint_as_float(r1.y); m[6] = __uint_as_float(r1.z); m[7] = __uint_as_float(r1.w);
            m[8] = __uint_as_float(r2.x); m[9] = __uint_as_float(r2.y); m[10] = __uint_as_float(r2.z); m[11] = __uint_as_float(r2.w);
            oo = xf_point(m, o);
            dd = xf_vector(m, d);
        }
        if (n_rx + n_ry + n_rz > 0) {
            const v3 inv = recip3(dd);
#define RTW_TEST_X(R_) RTW_RECT_TEST(R_, oo.x, inv.x, oo.y, dd.y, oo.z, dd.z)   /* shaders/aarectx.cu:8-22 */
#define RTW_TEST_Y(R_) RTW_RECT_TEST(R_, oo.y, inv.y, oo.x, dd.x, oo.z, dd.z)   /* shaders/aarecty.cu:8-22 */
#define RTW_TEST_Z(R_) RTW_RECT_TEST(R_, oo.z, inv.z, oo.x, dd.x, oo.y, dd.y)   /* shaders/aarectz.cu:9-23 */
            RTW_REC_LIST(n_rx, RTW_TEST_X)
            RTW_REC_LIST(n_ry, RTW_TEST_Y)
            RTW_REC_LIST(n_rz, RTW_TEST_Z)
#undef RTW_TEST_X
#undef RTW_TEST_Y
#undef RTW_TEST_Z
        }
#define RTW_TEST_SPH(R_)                                                                             \
        {                                                                                            \
            float t = 0.f;                                                                           \
            const bool hit = sphere_roots(oo, dd, V(__uint_as_float(R_.q0.x), __uint_as_float(R_.q0.y), __uint_as_float(R_.q0.z)), \
                                          __uint_as_float(R_.q0.w), tmin, RTW_FLT_MAX, t);           \
            if (ANY_HIT) {                                                                           \
                key_lo = (hit & (t < tmax)) ? 1u : key_lo;                                           \
            } else {                                                                                 \
                const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | R_.q1.y; \
                const bool take = hit & (key < (((unsigned long long)key_hi << 32) | key_lo));       \
                key_hi = take ? __float_as_uint(t) : key_hi;                                         \
                key_lo = take ? R_.q1.y : key_lo;                                                    \
            }                                                                                        \
        }
        RTW_REC_LIST(n_sph, RTW_TEST_SPH)
#undef RTW_TEST_SPH
    }
#undef RTW_REC_LIST
#undef RTW_RECT_TEST
    best_t = ANY_HIT ? tmax : __uint_as_float(key_hi);
    best_prim = (int)key_lo - 1;
}

// Closest / any hit (optixTraverse at raygen.cu:41-54 and closehit.cu:27-40).
// Candidate order (observable only through volume RNG draws and exact ties in t): volume
// primitives in index order, then everything else with ties resolved to the lowest index.
// tm: this thread's column of the LDS traversal stack and the block's LDS node cache (trav_mem).
template <class RNG, bool ANY_HIT, bool SKIP_VOLUMES>
RTW_DEV void traverse(const DScene& sc, v3 o, v3 d, float tmin, float tmax, float ray_time, float gather_time, RNG& g,
                      const TravMem& tm, float& best_t, int& best_prim) {
    if (!sc.use_bvh) {
        traverse_brute<RNG, ANY_HIT, SKIP_VOLUMES>(sc, o, d, tmin, tmax, ray_time, gather_time, g, best_t, best_prim);
        return;
    }
    best_t = tmax;
    best_prim = -1;
    // volumes first, in index order
    bool best_is_vol = false;
    if (!SKIP_VOLUMES) {
        best_is_vol = volume_pass<RNG, ANY_HIT>(sc, o, d, tmin, ray_time, gather_time, g, best_t, best_prim);
        if (ANY_HIT && best_is_vol) return;
    }
    // The closest hit is the minimum over (t, primitive index) of the non-volume candidates, which is
    // what the oracle's index-order scan with a strict '<' yields; written this way the candidates
    // may be visited in any order, so they are culled by the tree freely.
#define RTW_ACCEPT(T_, PI_)                                                                          \
    if ((T_) < best_t || (!ANY_HIT && (T_) == best_t && best_prim >= 0 && !best_is_vol && (PI_) < best_prim)) { \
        best_t = (T_); best_prim = (PI_); best_is_vol = false;                                       \
        if (ANY_HIT) return;                                                                         \
    }
    if (sc.n_tree <= 0) return;
    // "while-while" walk of the 4-wide tree: a tight loop descends through inner nodes (one 64-byte record per step
    // carries four children's boxes), leaves wait on the per-lane LDS stack column like inner nodes and are tested when
    // they come up. The boxes only cull, so the visiting order does not change the result.
    const v3 inv = recip3(d);
    int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        if (tm.wide) { while ((cur & 3u) == 0u) cur = bvh_inner_step(sc, tm, o, inv, tmin, best_t, cur, sp); }
        else { uint32_t none_aside = 0u; while ((cur & 3u) == 0u) bvh_step16<false, false>(sc, tm, o, inv, tmin, best_t, cur, none_aside, sp); }
        if (cur == kBvhDone) break;
        uint32_t slot = cur >> 2;
        const uint32_t cnt = cur & 3u;
        for (uint32_t k = 0; k < cnt; k++) {
            int pi;
            float t;
            if (leaf_test(sc, tm, slot, o, d, inv, tmin, ray_time, gather_time, t, pi, slot)) { RTW_ACCEPT(t, pi) }
        }
        cur = bvh_pop(tm, sp);
        if (cur == kBvhDone) break;
    }
#undef RTW_ACCEPT
}

// ---- camera rays: the wave walks the tree as one ---------------------------------------------------------------------
// k_first's rays leave through a few neighbouring pixels, so they visit nearly the same nodes. Here the WAVE stands at one
// node at a time: the node record (fp32 child boxes: rtw_bvh.h WNode) arrives through the scalar cache as instruction
// operands, every lane tests its own ray against the four boxes (6 fused multiply-adds, min / max and one compare per child
// instead of 12 conversions + multiply-adds and the per-lane stack traffic of bvh_step16), ballots say which children anyone
// needs, the nearest (by the entry distance of the first lane that wants it) is entered and the others go onto the wave's own
// stack in LDS - one 32-bit entry per level, in the space of the wave's per-lane stack columns, which this walk does not use.
// Leaves are tested by every lane with the record as scalar operands. A lane that missed a box runs through its subtree with
// the others and finds nothing there (boxes contain their primitives), and the closest hit is the minimum over (t, primitive
// index) of the candidates a lane hits whatever the order: per lane the result is traverse<>'s, bit for bit.
// wave_stack: this wave's level 0; level_stride: dwords between levels. live = false: the lane takes part in nothing.
RTW_DEV void traverse_wave(const DScene& sc, uint32_t* wave_stack, const uint32_t level_stride, const bool live, const v3 o, const v3 d, const float tmin,
                           const float ray_time, const float gather_time, float& best_t, int& best_prim, bool best_is_vol) {
    if (sc.n_tree <= 0) return;
    const v3 inv = recip3(d);
    const float ix = __builtin_amdgcn_fmed3f(inv.x, -1.0e18f, 1.0e18f), iy = __builtin_amdgcn_fmed3f(inv.y, -1.0e18f, 1.0e18f),
                iz = __builtin_amdgcn_fmed3f(inv.z, -1.0e18f, 1.0e18f);
    const float nx = -(o.x * ix), ny = -(o.y * iy), nz = -(o.z * iz);
    TravMem tm0;  // leaf records come through the scalar cache here, not from the workgroup's LDS image
    tm0.stack16 = nullptr; tm0.stack32 = nullptr; tm0.stride = 0; tm0.nodes = nullptr; tm0.n_nodes = 0; tm0.leaves = nullptr; tm0.n_leaves = 0; tm0.wide = false;
    uint32_t cur = 0u;  // wave-uniform from here on: the reference the wave stands at
    uint32_t sp = 0u;
    for (;;) {
        while ((cur & 3u) == 0u) {
            const RTW_CONST u32x4* q = (const RTW_CONST u32x4*)(uint64_t)(sc.wnodes + 8u * (cur >> 2));
            const u32x4 b0 = q[0], b1 = q[1], b2 = q[2], b3 = q[3], b4 = q[4], b5 = q[5], rf = q[6];
            uint32_t key[4];
#define RTW_W_CHILD(C_, LX_, LY_, LZ_, HX_, HY_, HZ_, REF_)                                                                                        \
            {                                                                                                                                    \
                key[C_] = 0xffffffffu;                                                                                                           \
                if ((REF_) != kBvhDone) {                                                                                                        \
                    const float ax = fma_(__uint_as_float(LX_), ix, nx), bx = fma_(__uint_as_float(HX_), ix, nx);                                 \
                    const float ay = fma_(__uint_as_float(LY_), iy, ny), by = fma_(__uint_as_float(HY_), iy, ny);                                 \
                    const float az = fma_(__uint_as_float(LZ_), iz, nz), bz = fma_(__uint_as_float(HZ_), iz, nz);                                 \
                    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fmaxf(__builtin_fminf(az, bz), tmin)); \
                    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fminf(__builtin_fmaxf(az, bz), best_t)); \
                    const unsigned long long any = __ballot(live && tn <= tf);                                                                   \
                    if (any != 0ull)                                                                                                             \
                        key[C_] = ((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tn), (int)__builtin_ctzll(any)) & ~3u) | (uint32_t)C_; \
                }                                                                                                                                \
            }
            RTW_W_CHILD(0, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, rf.x)
            RTW_W_CHILD(1, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, rf.y)
            RTW_W_CHILD(2, b3.x, b3.y, b3.z, b3.w, b4.x, b4.y, rf.z)
            RTW_W_CHILD(3, b4.z, b4.w, b5.x, b5.y, b5.z, b5.w, rf.w)
#undef RTW_W_CHILD
            // tn >= tmin > 0: the entry distances order as unsigned integers; children nobody wants sort last
            uint32_t k0 = key[0], k1 = key[1], k2 = key[2], k3 = key[3], t_;
#define RTW_CSWAP(A_, B_) { t_ = min(A_, B_); B_ = max(A_, B_); A_ = t_; }
            RTW_CSWAP(k0, k1) RTW_CSWAP(k2, k3) RTW_CSWAP(k0, k2) RTW_CSWAP(k1, k3) RTW_CSWAP(k1, k2)
#undef RTW_CSWAP
#define RTW_W_REF(K_) (((K_) & 3u) == 0u ? rf.x : ((K_) & 3u) == 1u ? rf.y : ((K_) & 3u) == 2u ? rf.z : rf.w)
            // the farthest wanted child goes down first, the nearest is entered
            if (k3 != 0xffffffffu) { wave_stack[sp * level_stride] = RTW_W_REF(k3); sp++; }
            if (k2 != 0xffffffffu) { wave_stack[sp * level_stride] = RTW_W_REF(k2); sp++; }
            if (k1 != 0xffffffffu) { wave_stack[sp * level_stride] = RTW_W_REF(k1); sp++; }
            if (k0 != 0xffffffffu) {
                cur = RTW_W_REF(k0);
            } else if (sp > 0u) {
                sp--;
                cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_stack[sp * level_stride]);
            } else {
                cur = kBvhDone;
            }
#undef RTW_W_REF
        }
        if (cur == kBvhDone) break;
        uint32_t slot = cur >> 2;
        const uint32_t cnt = cur & 3u;
        for (uint32_t k = 0; k < cnt; k++) {
            int pi;
            float t;
            uint32_t next;
            const bool hit = leaf_test(sc, tm0, slot, o, d, inv, tmin, ray_time, gather_time, t, pi, next);
            if (live && hit && (t < best_t || (t == best_t && best_prim >= 0 && !best_is_vol && pi < best_prim))) {
                best_t = t; best_prim = pi; best_is_vol = false;
            }
            slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
        }
        if (sp == 0u) break;
        sp--;
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_stack[sp * level_stride]);
    }
}

// Radiance ray (closest hit) and queued shadow probe (any hit) of one path in ONE walk over the small-scene
// candidate lists: the two rays share their origin, so each group's object-space origin and each record's
// scalar load and plane offset (k - o) are computed once. Per ray the arithmetic is exactly that of
// traverse<>: same operations, same order, same tie rule (the images stay bit-identical).
// Surfaces only (volumes are never in these lists); not for scenes with moving spheres (n_generic == 0).
RTW_DEV void traverse_dual_brute(const DScene& sc, const v3 o, const v3 dr, const v3 ds, const bool do_r, const bool do_s,
                                 const float smin, const float smax, float& best_t, int& best_prim, bool& occluded) {
    const float rmin = sc.ray_tmin;
    best_t = 1.e27f;
    best_prim = -1;
    occluded = false;
    for (int gi = 0; gi < sc.n_groups; gi++) {
        const BruteGroup G = load_group(sc, gi);
        v3 oo = o, ddr = dr, dds = ds;
        if (G.xform != 0) {
            M34 im = load_xf_inv(sc, G.xform);
            oo = xf_point(im.m, o);
            ddr = xf_vector(im.m, dr);
            dds = xf_vector(im.m, ds);
        }
        int ri = G.first;
        if (G.n_rx + G.n_ry + G.n_rz > 0) {
            const v3 invr = recip3(ddr);
            const v3 invs = recip3(dds);
#define RTW_DUAL_RECT_LOOP(N_, OK_, IKR_, IKS_, OA_, DAR_, DAS_, OB_, DBR_, DBS_)                      \
            for (int i = 0; i < (N_); i++, ri++) {                                                   \
                const BruteRec R = load_rec(sc, ri);                                                 \
                const float kmo = R.e - (OK_);                                                       \
                {                                                                                    \
                    const float t = kmo * (IKR_);                                                    \
                    const float a = fma_(t, (DAR_), (OA_));                                          \
                    const float b = fma_(t, (DBR_), (OB_));                                          \
                    const bool hit = do_r & (t >= rmin) & (a >= R.a) & (a <= R.b) & (b >= R.c) & (b <= R.d); \
                    const bool tie = (t == best_t) & (R.prim < best_prim);                           \
                    const bool take = hit & ((t < best_t) | tie);                                    \
                    best_t = take ? t : best_t;                                                      \
                    best_prim = take ? R.prim : best_prim;                                           \
                }                                                                                    \
                {                                                                                    \
                    const float t = kmo * (IKS_);                                                    \
                    const float a = fma_(t, (DAS_), (OA_));                                          \
                    const float b = fma_(t, (DBS_), (OB_));                                          \
                    occluded |= do_s & (t >= smin) & (t < smax) & (a >= R.a) & (a <= R.b) & (b >= R.c) & (b <= R.d); \
                }                                                                                    \
            }
            RTW_DUAL_RECT_LOOP(G.n_rx, oo.x, invr.x, invs.x, oo.y, ddr.y, dds.y, oo.z, ddr.z, dds.z)
            RTW_DUAL_RECT_LOOP(G.n_ry, oo.y, invr.y, invs.y, oo.x, ddr.x, dds.x, oo.z, ddr.z, dds.z)
            RTW_DUAL_RECT_LOOP(G.n_rz, oo.z, invr.z, invs.z, oo.x, ddr.x, dds.x, oo.y, ddr.y, dds.y)
#undef RTW_DUAL_RECT_LOOP
        }
        for (int i = 0; i < G.n_sph; i++, ri++) {
            const BruteRec R = load_rec(sc, ri);
            const v3 c = V(R.a, R.b, R.c);
            float t = 0.f;
            const bool hit = do_r & sphere_roots(oo, ddr, c, R.d, rmin, RTW_FLT_MAX, t);
            const bool tie = (t == best_t) & (R.prim < best_prim);
            const bool take = hit & ((t < best_t) | tie);
            best_t = take ? t : best_t;
            best_prim = take ? R.prim : best_prim;
            float ts = 0.f;
            occluded |= do_s & sphere_roots(oo, dds, c, R.d, smin, smax, ts);
        }
    }
}

// lds: the workgroup's LDS copy of the hit records (k_path, small scenes), or nullptr: a divergent index then costs an
// LDS read instead of a trip through the vector memory pipeline
RTW_DEV HitRec load_hitrec(const DScene& sc, int prim, const u32x4* lds = nullptr) {
    u32x4 a, b, c, d, e, f;
    d = e = f = u32x4{0u, 0u, 0u, 0u};
    bool basis;
    if (lds != nullptr) {
        const u32x4* q = lds + 6 * prim;
        a = q[0]; b = q[1]; c = q[2];
        basis = (int)(a.w & 127u) == HK_CONST_NORMAL && (int)a.x == RTW_MAT_LAMBERTIAN;
        if (basis) { d = q[3]; e = q[4]; f = q[5]; }
    } else {
        const RTW_CONST u32x4* q = (const RTW_CONST u32x4*)(uint64_t)(sc.hitrec + prim);
        a = q[0]; b = q[1]; c = q[2];
        basis = (int)(a.w & 127u) == HK_CONST_NORMAL && (int)a.x == RTW_MAT_LAMBERTIAN;
        if (basis) { d = q[3]; e = q[4]; f = q[5]; }
    }
    HitRec h;
    h.mat_type = (int)a.x; h.bsdf_eval = (int)a.y; h.param = __uint_as_float(a.z); h.kind = (int)(a.w & 127u); h.listed = (int)((a.w >> 7) & 1u); h.tex_dyn = (int)(a.w >> 8) - 1;
    h.r = __uint_as_float(b.x); h.g = __uint_as_float(b.y); h.b = __uint_as_float(b.z); h.inv_r = __uint_as_float(b.w);
    h.nx = __uint_as_float(c.x); h.ny = __uint_as_float(c.y); h.nz = __uint_as_float(c.z); h.xform = (int)c.w;
    h.ux = __uint_as_float(d.x); h.uy = __uint_as_float(d.y); h.uz = __uint_as_float(d.z);
    h.vx = __uint_as_float(e.x); h.vy = __uint_as_float(e.y); h.vz = __uint_as_float(e.z);
    h.wx = __uint_as_float(f.x); h.wy = __uint_as_float(f.y); h.wz = __uint_as_float(f.z);
    h.pad2 = 0.f;
    return h;
}

// Attributes of the committed hit (registers 0..7 of optixReportIntersection): world point, shading normal.
// The world point is the world ray evaluated at t (same real point as the reference's
// optixTransformPointFromObjectToWorldSpace(o_obj + t*d_obj)); normals follow sphere.cu:63-67 (Q13),
// movingSphere.cu:83-85, aarect{x,y,z}.cu:25-30, volumeBox.cu:86-93, volumeSphere.cu:97-105.
RTW_DEV void hit_attributes(const DScene& sc, const HitRec& h, int prim, v3 o, v3 d, float t, float gather_time, v3& point, v3& normal) {
    const v3 pw = vfma(d, t, o);
    point = pw;
    if (h.kind == HK_CONST_NORMAL) {
        normal = V(h.nx, h.ny, h.nz);
    } else if (h.kind == HK_SPHERE) {
        normal = vscale(vsub(pw, V(h.nx, h.ny, h.nz)), h.inv_r);
    } else {
        const rtw_prim pr = load_prim(sc, prim);
        v3 c = (h.kind == HK_MOVING_SPHERE) ? moving_center(pr, gather_time) : ld3(&pr.p[0]);
        v3 n = vscale(vsub(pw, c), h.inv_r);
        if (pr.xform != 0) { M34 xi = load_xf_inv(sc, pr.xform); n = xf_normal(xi.m, n); }
        normal = n;
    }
}

// ---- non-constant textures (cold path: only kernels instantiated with TEX = true contain it) ----
// geometry/sphere.cu:24-30 on the (unnormalised) shading normal
RTW_DEV void sphere_uv(v3 n, float& u, float& v) {
    const float phi = atan2_spec(n.z, n.x);
    const float theta = asin_spec(n.y);
    u = 1.0f - (phi + 3.14159265358979323846f) / 6.28318530717958647692f;
    v = (theta + 1.57079632679489661923f) / 3.14159265358979323846f;
}
// u, v of the committed hit: spheres from the shading normal, rectangles from the object-space hit point
// (aarectx.cu:33-34 and siblings), volumes 0
RTW_DEV void hit_uv(const DScene& sc, const HitRec& h, int prim, v3 o, v3 d, float t, float ray_time, v3 normal, float& u, float& v) {
    u = 0.0f; v = 0.0f;
    if (h.kind != HK_CONST_NORMAL) { sphere_uv(normal, u, v); return; }
    const rtw_prim pr = load_prim(sc, prim);
    if (pr.type < RTW_PRIM_RECT_X || pr.type > RTW_PRIM_RECT_Z) return;
    v3 po, pd, mt;
    object_ray(sc, pr, o, d, ray_time, po, pd, mt);
    float oa, da, ob, db;
    if (pr.type == RTW_PRIM_RECT_X) { oa = po.y; da = pd.y; ob = po.z; db = pd.z; }
    else if (pr.type == RTW_PRIM_RECT_Y) { oa = po.x; da = pd.x; ob = po.z; db = pd.z; }
    else { oa = po.x; da = pd.x; ob = po.y; db = pd.y; }
    const float a = fma_(t, da, oa), b = fma_(t, db, ob);
    u = (a - pr.p[0]) / (pr.p[1] - pr.p[0]);
    v = (b - pr.p[2]) / (pr.p[3] - pr.p[2]);
}
RTW_DEV float perlin_noise(const uint32_t* __restrict__ tab, v3 p) {  // noiseTexture.cu:20-52
    const float* ranvec = (const float*)tab;
    const int32_t* px = (const int32_t*)(tab + 768);
    const int32_t* py = px + 256;
    const int32_t* pz = py + 256;
    const float fx = __builtin_floorf(p.x), fy = __builtin_floorf(p.y), fz = __builtin_floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const float uu = (u * u) * (3.0f - 2.0f * u);
    const float vv = (v * v) * (3.0f - 2.0f * v);
    const float ww = (w * w) * (3.0f - 2.0f * w);
    // the six permutation entries the eight corners share (noiseTexture.cu:48-50 reads each of them four times)
    const int x0 = px[i & 255], x1 = px[(i + 1) & 255];
    const int y0 = py[j & 255], y1 = py[(j + 1) & 255];
    const int z0 = pz[k & 255], z1 = pz[(k + 1) & 255];
    float accum = 0.0f;
#pragma unroll
    for (int c8 = 0; c8 < 8; c8++) {  // di, dj, dk nested in this order, as the reference accumulates
        const int di = c8 >> 2, dj = (c8 >> 1) & 1, dk = c8 & 1;
        const int idx = ((di ? x1 : x0) ^ (dj ? y1 : y0) ^ (dk ? z1 : z0)) & 255;
        const v3 c = V(ranvec[3 * idx], ranvec[3 * idx + 1], ranvec[3 * idx + 2]);
        const v3 wv = V(u - (float)di, v - (float)dj, w - (float)dk);
        const float wi = di ? uu : 1.0f - uu, wj = dj ? vv : 1.0f - vv, wk = dk ? ww : 1.0f - ww;
        accum = accum + ((wi * wj) * wk) * dot3(c, wv);
    }
    return accum;
}
RTW_DEV float perlin_turb(const uint32_t* __restrict__ tab, v3 p) {  // noiseTexture.cu:54-67
    float accum = 0.0f, weight = 1.0f;
    v3 tp = p;
#pragma unroll 1
    for (int i = 0; i < 7; i++) {
        accum = accum + weight * perlin_noise(tab, tp);
        weight = weight * 0.5f;
        tp = vscale(tp, 2.0f);
    }
    return __builtin_fabsf(accum);
}
RTW_DEV v3 image_fetch(const uint32_t* __restrict__ img, float u, float v) {  // imageTexture.cu:11-17: clamp, bilinear, fp32 weights
    const uint32_t W = img[0], H = img[1];
    const uint32_t* tex = img + 2;
    if (!(u == u)) u = 0.0f;
    if (!(v == v)) v = 0.0f;
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    x = __builtin_fminf(__builtin_fmaxf(x, -1.0f), (float)W);
    y = __builtin_fminf(__builtin_fmaxf(y, -1.0f), (float)H);
    const float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
    const float a = x - x0, b = y - y0;
    const int ix = (int)x0, iy = (int)y0, w1 = (int)W - 1, h1 = (int)H - 1;
    const int ix0 = ix < 0 ? 0 : (ix > w1 ? w1 : ix), ix1 = ix + 1 < 0 ? 0 : (ix + 1 > w1 ? w1 : ix + 1);
    const int iy0 = iy < 0 ? 0 : (iy > h1 ? h1 : iy), iy1 = iy + 1 < 0 ? 0 : (iy + 1 > h1 ? h1 : iy + 1);
    const uint32_t t00 = tex[(size_t)iy0 * W + ix0], t10 = tex[(size_t)iy0 * W + ix1], t01 = tex[(size_t)iy1 * W + ix0], t11 = tex[(size_t)iy1 * W + ix1];
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float rgb[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float c00 = (float)((t00 >> (8 * ch)) & 255u) / 255.0f, c10 = (float)((t10 >> (8 * ch)) & 255u) / 255.0f;
        const float c01 = (float)((t01 >> (8 * ch)) & 255u) / 255.0f, c11 = (float)((t11 >> (8 * ch)) & 255u) / 255.0f;
        rgb[ch] = ((w00 * c00 + w10 * c10) + w01 * c01) + w11 * c11;
    }
    return V(rgb[0], rgb[1], rgb[2]);
}
// texture/checkeredTexture.cu:8-19, noiseTexture.cu:69-78, imageTexture.cu, constantTexture.cu, nullTexture.cu
// noise_lds: the workgroup's LDS copy of the tables of the noise texture at sc.noise_lds_data (or nullptr): a Perlin
// turbulence value is 7 x (6 + 24) table lookups at data-dependent addresses
RTW_DEV v3 texture_eval(const DScene& sc, const HitRec& h, int prim, v3 o, v3 d, float t, float ray_time, v3 p, v3 normal, const uint32_t* noise_lds) {
    rtw_texture tx = sc.texs[h.tex_dyn];
    if (tx.type == RTW_TEX_CHECKER) {
        const float sines = (sin_spec(10.0f * p.x) * sin_spec(10.0f - p.y)) * sin_spec(10.0f * p.z);
        tx = sc.texs[sines < 0.0f ? tx.odd : tx.even];
    }
    if (tx.type == RTW_TEX_CONSTANT) return V(tx.color[0], tx.color[1], tx.color[2]);
    if (tx.type == RTW_TEX_NOISE) {
        float tb;
        if (noise_lds != nullptr && (int32_t)tx.data == sc.noise_lds_data) tb = perlin_turb(noise_lds, vscale(p, tx.scale));
        else tb = perlin_turb(sc.texdata + tx.data, vscale(p, tx.scale));
        const float sn = sin_spec(tx.scale * p.z + 5.0f * tb);
        const float g = 0.5f * (1.0f + sn);
        return V(g, g, g);
    }
    if (tx.type == RTW_TEX_IMAGE) {
        float u, v;
        hit_uv(sc, h, prim, o, d, t, ray_time, normal, u, v);
        return image_fetch(sc.texdata + tx.data, u, v);
    }
    return V(0.f, 0.f, 0.f);
}

// lib/sampling.cuh:25-34
template <class RNG>
RTW_DEV v3 random_in_unit_sphere(RNG& g) {
    v3 p;
    do {
        float a = g.next1();
        float b = g.next1();
        float c = g.next1();
        p = V(fma_(2.0f, a, -1.0f), fma_(2.0f, b, -1.0f), fma_(2.0f, c, -1.0f));
    } while (dot3(p, p) >= 1.0f);
    return p;
}

// sutil reflect
RTW_DEV v3 reflect3(v3 i, v3 n) {
    float k = -2.0f * dot3(n, i);
    return vfma(n, k, i);
}

enum { EV_MISS = 0, EV_HIT = 1, EV_FINISH = 2, EV_CANCEL = 3 };

}  // namespace rtwdev
