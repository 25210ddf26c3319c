/* rtw_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, one-path-at-a-time restatement in plain C of the reference's device hot path
 * (safes/RayTracing-Weekend, RestOfLife/...).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's library; the product (librtw_hip.so, Director) never does.
 *
 * PINNING STATUS.  The reference holds no tests, golden images or known-answer vectors for this path
 *  and its device code needs OptiX 8 + nvcc, so it cannot be built here:
 *   - RNG (tea / xorshift32 / randf) IS pinned: oracle/Makefile compiles the reference's own
 *     lib/random.cuh where it lies into oracle/_ref/ and tests/golden/rng_kat.json holds its outputs;
 *     the OptiX-SDK lcg/rnd (not in tree) is restated from its published definition;
 *     Philox4x32-10 is pinned by the Random123 known-answer vectors.
 *   - Everything else (pixels) is PARITY UNPINNED against OptiX: the checker is this restatement,
 *     function by function, each citing the reference file:line it follows.
 *   - rtw_params.estimator != 0 (corrected estimators) and rtwo_denoise are this build's own definitions, not the
 *     reference's: there the file is the CPU twin of the HIP code, nothing more.
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md "arithmetic spec"): fp32 only,
 * no compiler contraction (-ffp-contract=off), fused multiply-adds only where fmaf() is written,
 * IEEE-correct / and sqrtf, own polynomial sincos/log, so CPU and GPU results can be bit-identical.
 * Where the reference's expression order is unspecified or fast-math dependent the order chosen
 * here is the definition (arguments are drawn left to right).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rtw.h"

/* Test hook (tests/test_oracle_physics.py): rtwo_set_debug(1) switches Russian roulette off, to check that the roulette of
 * raygen.cu:74-82 leaves the expectation alone. Never set by the parity tests. */
static int g_debug_no_roulette = 0;
void rtwo_set_debug(int no_roulette) { g_debug_no_roulette = no_roulette; }

/* ------------------------------------------------------------------ vec3 */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* a*s + b */
static inline v3 vfma(v3 a, float s, v3 b) { return V(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
/* sutil normalize: v * (1/sqrt(dot)) */
static inline v3 normalize3(v3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return vscale(a, inv); }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }

/* 3x4 row-major transforms */
static inline v3 xf_point(const float* m, v3 p) {
    return V(fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[2], p.z, m[3]))),
             fmaf(m[4], p.x, fmaf(m[5], p.y, fmaf(m[6], p.z, m[7]))),
             fmaf(m[8], p.x, fmaf(m[9], p.y, fmaf(m[10], p.z, m[11]))));
}
static inline v3 xf_vector(const float* m, v3 d) {
    return V(fmaf(m[0], d.x, fmaf(m[1], d.y, m[2] * d.z)),
             fmaf(m[4], d.x, fmaf(m[5], d.y, m[6] * d.z)),
             fmaf(m[8], d.x, fmaf(m[9], d.y, m[10] * d.z)));
}
/* normal object->world = (inverse linear part)^T * n   (optixTransformNormalFromObjectToWorldSpace) */
static inline v3 xf_normal(const float* inv, v3 n) {
    return V(fmaf(inv[0], n.x, fmaf(inv[4], n.y, inv[8] * n.z)),
             fmaf(inv[1], n.x, fmaf(inv[5], n.y, inv[9] * n.z)),
             fmaf(inv[2], n.x, fmaf(inv[6], n.y, inv[10] * n.z)));
}

/* ------------------------------------------------------------------ transcendental spec */
#define RTW_PI_F 3.14159265358979323846f
#define RTW_1_PI_F 0.318309886183790671538f
#define RTW_PIO2_F 1.57079632679489661923f

/* sin(2*pi*r), cos(2*pi*r) for r in [0,1): quadrant reduction is exact, Cephes sinf/cosf minimax
 * polynomials on [-pi/4, pi/4].  Stands in for sinf/cosf(2*pi*r) of lib/sampling.cuh:15-22,49-60. */
void rtwo_sincos2pi(float r, float* s_out, float* c_out) {
    float t = r * 4.0f;
    float q = floorf(t + 0.5f);
    float f = t - q;
    float x = f * RTW_PIO2_F;
    float x2 = x * x;
    float sp = fmaf(x2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(x2, sp, -1.6666654611e-1f);
    float s = fmaf(x * x2, sp, x);
    float cp = fmaf(x2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(x2, cp, 4.166664568298827e-2f);
    float c = fmaf(x2 * x2, cp, fmaf(x2, -0.5f, 1.0f));
    int qi = ((int)q) & 3;
    float so, co;
    if (qi == 0) { so = s; co = c; }
    else if (qi == 1) { so = c; co = -s; }
    else if (qi == 2) { so = -s; co = -c; }
    else { so = -c; co = s; }
    *s_out = so;
    *c_out = co;
}

/* natural log, Cephes logf. x == 0 -> -inf; stands in for logf of geometry/volumeBox.cu:79. */
float rtwo_logf(float x) {
    if (x == 0.0f) return -INFINITY;
    uint32_t ix;
    memcpy(&ix, &x, 4);
    int e = (int)((ix >> 23) & 255u) - 126;
    ix = (ix & 0x007fffffu) | 0x3f000000u; /* mantissa in [0.5,1) */
    float m;
    memcpy(&m, &ix, 4);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = (y * m) * z;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* ------------------------------------------------------------------ RNG */
/* lib/random.cuh:7-19 (== OptiX SDK cuda/random.h tea<N>) */
uint32_t rtwo_tea(uint32_t N, uint32_t s0, uint32_t s1) {
    uint32_t t = 0;
    for (uint32_t n = 0; n < N; n++) {
        t += 0x9E3779B9u;
        s0 += ((s1 << 4) + 0xa341316cu) ^ (s1 + t) ^ ((s1 >> 5) + 0xc8013ea4u);
        s1 += ((s0 << 4) + 0xad90777du) ^ (s0 + t) ^ ((s0 >> 5) + 0x7e95761eu);
    }
    return s0;
}
/* lib/random.cuh:22-28 */
uint32_t rtwo_xorshift32(uint32_t* s) {
    uint32_t v = *s;
    v ^= v << 13;
    v ^= v >> 17;
    v ^= v << 5;
    *s = v;
    return v;
}
/* lib/random.cuh:31-38, including quirk Q10 (integer 0x3F7FFFFF converted to float) */
float rtwo_randf(uint32_t* s) {
    float r = ((float)rtwo_xorshift32(s)) / 4294967296.0f;
    if (r != 1.0f) return r;
    return (float)0x3F7FFFFF;
}
/* OptiX SDK 8.0.0 SDK/cuda/random.h lcg()/rnd() (third-party, not in tree; published definition):
 * prev = 1664525*prev + 1013904223; return (prev & 0x00FFFFFF) / 0x01000000 */
float rtwo_lcg_rnd(uint32_t* s) {
    *s = 1664525u * (*s) + 1013904223u;
    return (float)((*s) & 0x00FFFFFFu) / (float)0x01000000;
}
/* Random123 philox4x32-10 */
void rtwo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Per-path generator state.  Three streams mirror the reference's three seed copies:
 * 0 = raygen's `seed` (raygen.cu:129-135,95), 1 = prd.seed (materials, NEE, RR, volumes),
 * 2 = rayColor's local `seed` (ray time only, raygen.cu:30,48). */
typedef struct {
    int kind;
    uint32_t lcg[3];
    uint32_t key0, pixel, sample;
    uint32_t draw[3];
    uint32_t seg_base; /* Philox stream 1: draw counter at the start of the current closest-hit program */
} rng_t;

static float rng_next(rng_t* g, int stream) {
    if (g->kind == RTW_RNG_TEA_LCG) return rtwo_lcg_rnd(&g->lcg[stream]);
    uint32_t d = g->draw[stream]++;
    uint32_t ctr[4] = {g->pixel, g->sample, d >> 2, (uint32_t)stream};
    uint32_t key[2] = {g->key0, 0u};
    uint32_t out[4];
    rtwo_philox4x32_10(ctr, key, out);
    return (float)(out[d & 3u] >> 8) * (1.0f / 16777216.0f);
}
/* The Russian-roulette draw (raygen.cu:77). TEA+LCG: the next value of prd.seed, as in the reference.
 * Philox: a fifth uniform cut from the low bytes of words 0..2 of the block the segment used last (its four
 * 24-bit uniforms use the high bytes), so that a Lambertian segment (2 scatter + 2 light + 1 roulette draws)
 * costs one Philox block instead of two. The draw counter does not move. */
static float rng_rr(rng_t* g) {
    if (g->kind == RTW_RNG_TEA_LCG) return rtwo_lcg_rnd(&g->lcg[1]);
    uint32_t a = g->draw[1];
    uint32_t blk = (a == g->seg_base) ? (a >> 2) : ((a - 1u) >> 2);
    uint32_t ctr[4] = {g->pixel, g->sample, blk, 1u};
    uint32_t key[2] = {g->key0, 0u};
    uint32_t out[4];
    rtwo_philox4x32_10(ctr, key, out);
    uint32_t v = ((out[0] & 0xffu) << 16) | ((out[1] & 0xffu) << 8) | (out[2] & 0xffu);
    return (float)v * (1.0f / 16777216.0f);
}
/* randf(thePrd->seed) of geometry/volumeBox.cu:79: xorshift on the SAME prd.seed word */
static float rng_randf(rng_t* g) {
    if (g->kind == RTW_RNG_TEA_LCG) return rtwo_randf(&g->lcg[1]);
    return rng_next(g, 1);
}

/* ------------------------------------------------------------------ scene view */
typedef struct {
    const rtw_scene_header* h;
    const rtw_prim* prims;
    const rtw_xform* xforms;
    const rtw_material* mats;
    const rtw_texture* texs;
    const rtw_light* lights;
    const uint32_t* texdata; /* texture data section, texdata_words 4-byte words */
    uint32_t texdata_words;
    /* RTW_EST_CORRECTED only (corrected_lights_build): the light list moved onto the emitting rectangles it describes,
     * and which primitives those are */
    rtw_light* clights;
    uint8_t* listed;
    int bounded_media; /* corrected estimators: a medium scatters only inside its extent (the reference does not test it, Q9) */
} scene_t;

static int scene_open(scene_t* s, const void* blob, size_t bytes) {
    if (!blob || bytes < sizeof(rtw_scene_header)) return RTW_ERR_BAD_SCENE;
    const rtw_scene_header* h = (const rtw_scene_header*)blob;
    if (h->magic != RTW_SCENE_MAGIC || h->version != RTW_SCENE_VERSION || h->total_bytes > bytes)
        return RTW_ERR_BAD_SCENE;
    const char* b = (const char*)blob;
    if ((size_t)h->off_prims + (size_t)h->n_prims * sizeof(rtw_prim) > bytes) return RTW_ERR_BAD_SCENE;
    if ((size_t)h->off_xforms + (size_t)h->n_xforms * sizeof(rtw_xform) > bytes) return RTW_ERR_BAD_SCENE;
    if ((size_t)h->off_materials + (size_t)h->n_materials * sizeof(rtw_material) > bytes) return RTW_ERR_BAD_SCENE;
    if ((size_t)h->off_textures + (size_t)h->n_textures * sizeof(rtw_texture) > bytes) return RTW_ERR_BAD_SCENE;
    if ((size_t)h->off_lights + (size_t)h->n_lights * sizeof(rtw_light) > bytes) return RTW_ERR_BAD_SCENE;
    if (h->n_xforms < 1) return RTW_ERR_BAD_SCENE;
    if (h->camera_type < RTW_CAM_PERSPECTIVE || h->camera_type > RTW_CAM_ORTHOGRAPHIC) return RTW_ERR_BAD_SCENE;
    /* include/rtw.h: tables sit at 16-byte aligned offsets (the tables are read in place here) */
    if ((h->off_prims | h->off_xforms | h->off_materials | h->off_textures | h->off_lights | h->off_texdata) & 15u) return RTW_ERR_BAD_SCENE;
    if (((uintptr_t)blob & 3u) != 0) return RTW_ERR_BAD_SCENE;
    s->h = h;
    s->clights = NULL;
    s->listed = NULL;
    s->bounded_media = 0;
    s->prims = (const rtw_prim*)(b + h->off_prims);
    s->xforms = (const rtw_xform*)(b + h->off_xforms);
    s->mats = (const rtw_material*)(b + h->off_materials);
    s->texs = (const rtw_texture*)(b + h->off_textures);
    s->lights = (const rtw_light*)(b + h->off_lights);
    for (uint32_t i = 0; i < h->n_prims; i++) {
        if (s->prims[i].type < RTW_PRIM_SPHERE || s->prims[i].type > RTW_PRIM_VOLUME_SPHERE) return RTW_ERR_BAD_SCENE;
        if (s->prims[i].xform < 0 || (uint32_t)s->prims[i].xform >= h->n_xforms) return RTW_ERR_BAD_SCENE;
        if (s->prims[i].material < 0 || (uint32_t)s->prims[i].material >= h->n_materials) return RTW_ERR_BAD_SCENE;
        for (int k = 0; k < 12; k++) if (!isfinite(s->prims[i].p[k])) return RTW_ERR_BAD_SCENE;
    }
    for (uint32_t i = 0; i < h->n_xforms; i++)
        for (int k = 0; k < 12; k++) if (!isfinite(s->xforms[i].m[k]) || !isfinite(s->xforms[i].inv[k])) return RTW_ERR_BAD_SCENE;
    s->texdata = NULL;
    s->texdata_words = 0;
    if (h->off_texdata) {
        if ((h->off_texdata & 3u) || (size_t)h->off_texdata + (size_t)h->texdata_bytes > bytes) return RTW_ERR_BAD_SCENE;
        s->texdata = (const uint32_t*)(b + h->off_texdata);
        s->texdata_words = h->texdata_bytes / 4u;
    }
    for (uint32_t i = 0; i < h->n_materials; i++)
        if (s->mats[i].texture >= (int32_t)h->n_textures) return RTW_ERR_BAD_SCENE;
    for (uint32_t i = 0; i < h->n_textures; i++) {
        const rtw_texture* t = &s->texs[i];
        if (t->type == RTW_TEX_CHECKER) {
            /* children: any texture but another checker (the reference's callables cannot nest either) */
            if (t->odd < 0 || t->even < 0 || (uint32_t)t->odd >= h->n_textures || (uint32_t)t->even >= h->n_textures) return RTW_ERR_BAD_SCENE;
            if (s->texs[t->odd].type == RTW_TEX_CHECKER || s->texs[t->even].type == RTW_TEX_CHECKER) return RTW_ERR_BAD_SCENE;
        } else if (t->type == RTW_TEX_NOISE) {
            if ((size_t)t->data + 1536u > s->texdata_words) return RTW_ERR_BAD_SCENE;
        } else if (t->type == RTW_TEX_IMAGE) {
            if ((size_t)t->data + 2u > s->texdata_words) return RTW_ERR_BAD_SCENE;
            uint32_t iw = s->texdata[t->data], ih = s->texdata[t->data + 1];
            if (iw == 0 || ih == 0 || iw > 32768u || ih > 32768u || (size_t)t->data + 2u + (size_t)iw * ih > s->texdata_words) return RTW_ERR_BAD_SCENE;
        } else if (t->type != RTW_TEX_CONSTANT && t->type != RTW_TEX_NULL) {
            return RTW_ERR_BAD_SCENE;
        }
    }
    return RTW_OK;
}

/* ------------------------------------------------------------------ sin / atan2 / asin (texture callables)
 * The reference calls sinf (texture/noiseTexture.cu:77, checkeredTexture.cu:9), atan2f and asinf
 * (geometry/sphere.cu:24-30) from libdevice under fast-math. Restated with the published Cephes single-precision
 * algorithms (sinf.c, atanf.c, asinf.c), every fused step written as fmaf so that the HIP build repeats it bit for
 * bit. The numbers differ from libdevice's in the last ulps: texture parity against OptiX is unpinned. */
float rtwo_sinf(float xx) {
    float x = fabsf(xx);
    int sign = xx < 0.0f ? -1 : 1;
    /* octant: j = (int)(x * 4/pi), made even */
    uint32_t j = (uint32_t)(x * 1.27323954473516f);
    float y = (float)j;
    if (j & 1u) { j += 1u; y += 1.0f; }
    j &= 7u;
    if (j > 3u) { sign = -sign; j -= 4u; }
    /* extended-precision modular arithmetic: x - y * pi/4 in three steps */
    x = fmaf(-y, 0.78515625f, x);
    x = fmaf(-y, 2.4187564849853515625e-4f, x);
    x = fmaf(-y, 3.77489497744594108e-8f, x);
    float z = x * x, r;
    if (j == 1u || j == 2u) {
        float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
        p = fmaf(p, z, 4.166664568298827e-2f);
        r = fmaf(p * z, z, fmaf(-0.5f, z, 1.0f));
    } else {
        float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
        p = fmaf(p, z, -1.6666654611e-1f);
        r = fmaf(p * z, x, x);
    }
    return sign < 0 ? -r : r;
}

static float atan_pos(float x) { /* x >= 0 */
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return y + fmaf(p * z, x, x);
}

float rtwo_atan2f(float y, float x) {
    const float pi = 3.14159265358979323846f, pio2 = 1.5707963267948966f;
    if (x == 0.0f) return y > 0.0f ? pio2 : (y < 0.0f ? -pio2 : 0.0f);
    if (y == 0.0f) return x < 0.0f ? pi : 0.0f;
    float q = y / x;
    float a = atan_pos(fabsf(q));
    if (q < 0.0f) a = -a;
    float w = x < 0.0f ? (y < 0.0f ? -pi : pi) : 0.0f;
    return w + a;
}

float rtwo_asinf(float xx) { /* arguments beyond +-1 (rounding of a unit normal) are clamped */
    float a = fabsf(xx);
    if (a > 1.0f) a = 1.0f;
    float r;
    if (a < 1.0e-4f) {
        r = a;
    } else {
        float z, x;
        int flag = a > 0.5f;
        if (flag) { z = 0.5f * (1.0f - a); x = sqrtf(z); }
        else { x = a; z = x * x; }
        float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
        p = fmaf(p, z, 4.5470025998e-2f);
        p = fmaf(p, z, 7.4953002686e-2f);
        p = fmaf(p, z, 1.6666752422e-1f);
        r = fmaf(p * z, x, x);
        if (flag) r = 1.5707963267948966f - (r + r);
    }
    return xx < 0.0f ? -r : r;
}

/* ------------------------------------------------------------------ intersection */
typedef struct {
    float t;
    int prim;      /* -1 = miss */
    v3 o_obj, d_obj; /* object-space ray of the winning primitive */
    v3 motion;     /* moving sphere: translation at ray time */
    v3 xcenter;    /* moving sphere: object-space centre at gather time */
} hit_t;

static inline int is_volume(int type) { return type == RTW_PRIM_VOLUME_BOX || type == RTW_PRIM_VOLUME_SPHERE; }

/* geometry/movingSphere.cu:33-39 */
static v3 moving_center(const rtw_prim* pr, float time) {
    v3 c0 = ld3(&pr->p[0]);
    float t0 = pr->p[7], t1 = pr->p[8];
    if (t0 == t1) return c0;
    v3 c1 = ld3(&pr->p[4]);
    float u = (time - t0) / (t1 - t0);
    return vfma(vsub(c1, c0), u, c0);
}

/* first root inside (tmin,tmax) of the sphere quadratic — geometry/sphere.cu:52-60,93-95.
 * OptiX accepts the nearer root first; the farther one is then beyond the shrunk tmax. */
static int sphere_roots(v3 o, v3 d, v3 c, float r, float tmin, float tmax, float* t_out) {
    v3 oc = vsub(o, c);
    float a = dot3(d, d);
    float b = dot3(oc, d);
    float cc = fmaf(-r, r, dot3(oc, oc));
    float disc = fmaf(b, b, -(a * cc));
    if (disc < 0.0f) return 0;
    float sq = sqrtf(disc);
    float t = (-b - sq) / a;
    if (t < tmax && t > tmin) { *t_out = t; return 1; }
    t = (-b + sq) / a;
    if (t < tmax && t > tmin) { *t_out = t; return 1; }
    return 0;
}

/* geometry/volumeBox.cu:29-52 */
static int box_boundary(float temp1, float temp2, float tMin, float tMax, float* rec) {
    if (temp1 > temp2) return 0;
    if (temp1 < tMax && temp1 > tMin) { *rec = temp1; return 1; }
    if (temp2 < tMax && temp2 > tMin) { *rec = temp2; return 1; }
    return 0;
}

#define RTW_FLT_MAX 3.402823466e+38f

/* One primitive against the ray. Returns 1 and *t_out when the intersection program would report
 * a hit inside (tmin, tmax_cur). Volumes draw from the path's generator (quirk Q9). */
static int prim_intersect(const scene_t* sc, int pi, v3 o, v3 d, float tmin, float tmax_cur, float ray_time,
                          float gather_time, rng_t* g, float* t_out, hit_t* scratch) {
    const rtw_prim* pr = &sc->prims[pi];
    v3 oo = o, dd = d;
    if (pr->xform != 0) {
        const rtw_xform* xf = &sc->xforms[pr->xform];
        oo = xf_point(xf->inv, o);
        dd = xf_vector(xf->inv, d);
    }
    scratch->motion = V(0.f, 0.f, 0.f);
    scratch->xcenter = V(0.f, 0.f, 0.f);
    switch (pr->type) {
    case RTW_PRIM_SPHERE: {
        scratch->o_obj = oo; scratch->d_obj = dd;
        return sphere_roots(oo, dd, ld3(&pr->p[0]), pr->p[3], tmin, tmax_cur, t_out);
    }
    case RTW_PRIM_MOVING_SPHERE: {
        /* matrix-motion transform translate(lerp(C0,C1,rayTime)) — ioMovingSphere.h:161-203 */
        v3 c0 = ld3(&pr->p[0]), c1 = ld3(&pr->p[4]);
        v3 mt = vfma(vsub(c1, c0), ray_time, c0);
        oo = vsub(oo, mt);
        v3 xc = moving_center(pr, gather_time); /* movingSphere.cu:66 */
        scratch->o_obj = oo; scratch->d_obj = dd; scratch->motion = mt; scratch->xcenter = xc;
        return sphere_roots(oo, dd, xc, pr->p[3], tmin, tmax_cur, t_out);
    }
    case RTW_PRIM_RECT_X:
    case RTW_PRIM_RECT_Y:
    case RTW_PRIM_RECT_Z: {
        /* shaders/aarectx.cu:8-22, aarecty.cu:8-22, aarectz.cu:9-23 */
        float ok, dk, oa, da, ob, db;
        if (pr->type == RTW_PRIM_RECT_X) { ok = oo.x; dk = dd.x; oa = oo.y; da = dd.y; ob = oo.z; db = dd.z; }
        else if (pr->type == RTW_PRIM_RECT_Y) { ok = oo.y; dk = dd.y; oa = oo.x; da = dd.x; ob = oo.z; db = dd.z; }
        else { ok = oo.z; dk = dd.z; oa = oo.x; da = dd.x; ob = oo.y; db = dd.y; }
        float inv = 1.0f / dk;
        float t = (pr->p[4] - ok) * inv;
        if (!(t >= tmin && t < tmax_cur)) return 0;
        float a = fmaf(t, da, oa);
        float b = fmaf(t, db, ob);
        if (!(a >= pr->p[0] && a <= pr->p[1] && b >= pr->p[2] && b <= pr->p[3])) return 0;
        scratch->o_obj = oo; scratch->d_obj = dd;
        *t_out = t;
        return 1;
    }
    case RTW_PRIM_VOLUME_BOX: {
        /* geometry/volumeBox.cu:55-113 */
        v3 inv = V(1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z);
        v3 t0 = vmul(vsub(ld3(&pr->p[0]), oo), inv);
        v3 t1 = vmul(vsub(ld3(&pr->p[3]), oo), inv);
        float temp1 = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
        float temp2 = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
        float h1, h2;
        if (!box_boundary(temp1, temp2, -RTW_FLT_MAX, RTW_FLT_MAX, &h1)) return 0;
        if (!box_boundary(temp1, temp2, h1 /* + (int)0.0001f == 0, quirk Q8 */, RTW_FLT_MAX, &h2)) return 0;
        if (h1 < tmin) h1 = tmin;
        if (h2 > tmax_cur) h2 = tmax_cur;
        if (h1 >= h2) return 0;
        if (h1 < 0.f) h1 = 0.f;
        float len = length3(dd);
        float hit_distance = -(1.0f / pr->p[6]) * rtwo_logf(rng_randf(g));
        float t = h1 + hit_distance / len;
        if (!(t >= tmin && t < tmax_cur)) return 0; /* optixReportIntersection interval test */
        if (sc->bounded_media && !(t < h2)) return 0; /* corrected: the free flight left the medium */
        scratch->o_obj = oo; scratch->d_obj = dd;
        *t_out = t;
        return 1;
    }
    case RTW_PRIM_VOLUME_SPHERE: {
        /* geometry/volumeSphere.cu:67-127 */
        v3 c = ld3(&pr->p[0]);
        float h1, h2;
        if (!sphere_roots(oo, dd, c, pr->p[3], -RTW_FLT_MAX, RTW_FLT_MAX, &h1)) return 0;
        if (!sphere_roots(oo, dd, c, pr->p[3], h1, RTW_FLT_MAX, &h2)) return 0;
        if (h1 < tmin) h1 = tmin;
        if (h2 > tmax_cur) h2 = tmax_cur;
        if (h1 >= h2) return 0;
        if (h1 < 0.f) h1 = 0.f;
        float len = length3(dd);
        float hit_distance = -(1.0f / pr->p[4]) * rtwo_logf(rng_next(g, 1));
        float t = h1 + hit_distance / len;
        if (!(t >= tmin && t < tmax_cur)) return 0;
        if (sc->bounded_media && !(t < h2)) return 0;
        scratch->o_obj = oo; scratch->d_obj = dd;
        *t_out = t;
        return 1;
    }
    default:
        return 0;
    }
}

/* Closest hit (optixTraverse of raygen.cu:41-54). Canonical candidate order: volume primitives in
 * index order first (their RNG draws make order observable, Q9), then the rest in index order;
 * on equal t the earlier candidate wins. any_hit!=0: terminate on first accepted hit
 * (traceOcclusion, closehit.cu:16-42). */
static void traverse(const scene_t* sc, v3 o, v3 d, float tmin, float tmax, float ray_time, float gather_time,
                     rng_t* g, int any_hit, int skip_volumes, hit_t* best) {
    best->t = tmax;
    best->prim = -1;
    int n = (int)sc->h->n_prims;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++) {
            int vol = is_volume(sc->prims[i].type);
            if ((pass == 0) != (vol != 0)) continue;
            if (vol && skip_volumes) continue;
            hit_t tmp;
            float t;
            if (prim_intersect(sc, i, o, d, tmin, best->t, ray_time, gather_time, g, &t, &tmp)) {
                best->t = t;
                best->prim = i;
                best->o_obj = tmp.o_obj; best->d_obj = tmp.d_obj;
                best->motion = tmp.motion; best->xcenter = tmp.xcenter;
                if (any_hit) return;
            }
        }
    }
}

/* Attributes of the committed hit: world point, shading normal (what the IS programs pass through
 * optixReportIntersection registers 0..7), and the texture coordinates u, v of the same programs.
 * The world point is the world ray evaluated at t — the same real point as the reference's
 * optixTransformPointFromObjectToWorldSpace(o_obj + t*d_obj), without the round trip through object space. */
/* geometry/sphere.cu:24-30 get_sphere_uv, applied to the (unnormalised) shading normal */
static void sphere_uv(v3 n, float* u, float* v) {
    float phi = rtwo_atan2f(n.z, n.x);
    float theta = rtwo_asinf(n.y);
    *u = 1.0f - (phi + 3.14159265358979323846f) / 6.28318530717958647692f;
    *v = (theta + 1.57079632679489661923f) / 3.14159265358979323846f;
}

static void hit_attributes(const scene_t* sc, const hit_t* h, v3 o, v3 d, v3* point, v3* normal, float* tu, float* tv) {
    const rtw_prim* pr = &sc->prims[h->prim];
    const rtw_xform* xf = &sc->xforms[pr->xform];
    v3 pw = vfma(d, h->t, o);
    *point = pw;
    *tu = 0.0f; *tv = 0.0f;
    switch (pr->type) {
    case RTW_PRIM_SPHERE: {
        /* sphere.cu:63-67: normal from the WORLD point and the OBJECT-space centre (quirk Q13) */
        v3 n = vscale(vsub(pw, ld3(&pr->p[0])), 1.0f / pr->p[3]);
        if (pr->xform != 0) n = xf_normal(xf->inv, n);
        *normal = n;
        sphere_uv(n, tu, tv);
        break;
    }
    case RTW_PRIM_MOVING_SPHERE: {
        /* movingSphere.cu:83-85: world point includes the motion translation, centre does not */
        v3 n = vscale(vsub(pw, h->xcenter), 1.0f / pr->p[3]);
        if (pr->xform != 0) n = xf_normal(xf->inv, n);
        *normal = n;
        sphere_uv(n, tu, tv);
        break;
    }
    case RTW_PRIM_RECT_X:
    case RTW_PRIM_RECT_Y:
    case RTW_PRIM_RECT_Z: {
        v3 n = (pr->type == RTW_PRIM_RECT_X) ? V(1.f, 0.f, 0.f) : (pr->type == RTW_PRIM_RECT_Y) ? V(0.f, 1.f, 0.f) : V(0.f, 0.f, 1.f);
        if (pr->flip) n = vneg(n);
        *normal = (pr->xform != 0) ? normalize3(xf_normal(xf->inv, n)) : n; /* normalize of an exact unit axis is the identity */
        /* aarectx.cu:33-34 (y,z), aarecty.cu (x,z), aarectz.cu (x,y): the in-plane coordinates of the object-space hit */
        float oa, da, ob, db;
        if (pr->type == RTW_PRIM_RECT_X) { oa = h->o_obj.y; da = h->d_obj.y; ob = h->o_obj.z; db = h->d_obj.z; }
        else if (pr->type == RTW_PRIM_RECT_Y) { oa = h->o_obj.x; da = h->d_obj.x; ob = h->o_obj.z; db = h->d_obj.z; }
        else { oa = h->o_obj.x; da = h->d_obj.x; ob = h->o_obj.y; db = h->d_obj.y; }
        float a = fmaf(h->t, da, oa), b = fmaf(h->t, db, ob);
        *tu = (a - pr->p[0]) / (pr->p[1] - pr->p[0]);
        *tv = (b - pr->p[2]) / (pr->p[3] - pr->p[2]);
        break;
    }
    default: { /* volumes: volumeBox.cu:86-93, volumeSphere.cu:97-105 (u = v = 0) */
        v3 n = V(1.f, 0.f, 0.f);
        *normal = (pr->xform != 0) ? normalize3(xf_normal(xf->inv, n)) : n;
        break;
    }
    }
}

/* ---- texture callables: texture/constantTexture.cu:5-10, nullTexture.cu:7-12, checkeredTexture.cu:8-19,
 * noiseTexture.cu:20-78, imageTexture.cu:11-17 ---- */
static float perlin_noise(const uint32_t* tab, v3 p) {
    const float* ranvec = (const float*)tab;
    const int32_t* px = (const int32_t*)(tab + 768);
    const int32_t* py = px + 256;
    const int32_t* pz = py + 256;
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    /* perlin_interp, noiseTexture.cu:20-36 */
    float uu = (u * u) * (3.0f - 2.0f * u);
    float vv = (v * v) * (3.0f - 2.0f * v);
    float ww = (w * w) * (3.0f - 2.0f * w);
    float accum = 0.0f;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                int idx = (px[(i + di) & 255] ^ py[(j + dj) & 255] ^ pz[(k + dk) & 255]) & 255;
                v3 c = V(ranvec[3 * idx], ranvec[3 * idx + 1], ranvec[3 * idx + 2]);
                v3 wv = V(u - (float)di, v - (float)dj, w - (float)dk);
                float wi = di ? uu : 1.0f - uu, wj = dj ? vv : 1.0f - vv, wk = dk ? ww : 1.0f - ww;
                accum = accum + ((wi * wj) * wk) * dot3(c, wv);
            }
    return accum;
}

static float perlin_turb(const uint32_t* tab, v3 p) { /* noiseTexture.cu:54-67, 7 octaves */
    float accum = 0.0f, weight = 1.0f;
    v3 tp = p;
    for (int i = 0; i < 7; i++) {
        accum = accum + weight * perlin_noise(tab, tp);
        weight = weight * 0.5f;
        tp = vscale(tp, 2.0f);
    }
    return fabsf(accum);
}

static v3 image_fetch(const uint32_t* img, float u, float v) {
    /* tex2D, normalised coordinates, clamp addressing, linear filter, 8-bit texels read as x/255
     * (ioTexture.h:264-283). The hardware filter rounds its weights to 8 fractional bits; this one keeps fp32. */
    const uint32_t W = img[0], H = img[1];
    const uint32_t* tex = img + 2;
    if (!(u == u)) u = 0.0f;
    if (!(v == v)) v = 0.0f;
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    x = fminf(fmaxf(x, -1.0f), (float)W);
    y = fminf(fmaxf(y, -1.0f), (float)H);
    float x0 = floorf(x), y0 = floorf(y);
    float a = x - x0, b = y - y0;
    int ix = (int)x0, iy = (int)y0;
    int ix0 = ix < 0 ? 0 : (ix > (int)W - 1 ? (int)W - 1 : ix), ix1 = ix + 1 < 0 ? 0 : (ix + 1 > (int)W - 1 ? (int)W - 1 : ix + 1);
    int iy0 = iy < 0 ? 0 : (iy > (int)H - 1 ? (int)H - 1 : iy), iy1 = iy + 1 < 0 ? 0 : (iy + 1 > (int)H - 1 ? (int)H - 1 : iy + 1);
    uint32_t t00 = tex[(size_t)iy0 * W + ix0], t10 = tex[(size_t)iy0 * W + ix1], t01 = tex[(size_t)iy1 * W + ix0], t11 = tex[(size_t)iy1 * W + ix1];
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float rgb[3];
    for (int ch = 0; ch < 3; ch++) {
        float c00 = (float)((t00 >> (8 * ch)) & 255u) / 255.0f, c10 = (float)((t10 >> (8 * ch)) & 255u) / 255.0f;
        float c01 = (float)((t01 >> (8 * ch)) & 255u) / 255.0f, c11 = (float)((t11 >> (8 * ch)) & 255u) / 255.0f;
        rgb[ch] = ((w00 * c00 + w10 * c10) + w01 * c01) + w11 * c11;
    }
    return V(rgb[0], rgb[1], rgb[2]);
}

static v3 texture_eval(const scene_t* sc, int ti, float u, float v, v3 p) {
    if (ti < 0) return V(0.f, 0.f, 0.f);
    const rtw_texture* t = &sc->texs[ti];
    if (t->type == RTW_TEX_CHECKER) {
        /* checkeredTexture.cu:9: note "10.f - p.y" (kept) */
        float sines = (rtwo_sinf(10.0f * p.x) * rtwo_sinf(10.0f - p.y)) * rtwo_sinf(10.0f * p.z);
        t = &sc->texs[sines < 0.0f ? t->odd : t->even]; /* children are never checkers (scene_open) */
    }
    switch (t->type) {
    case RTW_TEX_CONSTANT: return ld3(t->color);
    case RTW_TEX_NOISE: {
        /* noiseTexture.cu:77 */
        const uint32_t* tab = sc->texdata + t->data;
        float tb = perlin_turb(tab, vscale(p, t->scale));
        float s = rtwo_sinf(t->scale * p.z + 5.0f * tb);
        float g = 0.5f * (1.0f + s);
        return V(g, g, g);
    }
    case RTW_TEX_IMAGE: return image_fetch(sc->texdata + t->data, u, v);
    default: return V(0.f, 0.f, 0.f);
    }
}

static v3 texture_value(const scene_t* sc, const rtw_material* m, float u, float v, v3 p) {
    return texture_eval(sc, m->texture, u, v, p);
}

/* lib/sampling.cuh:25-34 */
static v3 random_in_unit_sphere(rng_t* g) {
    v3 p;
    do {
        float a = rng_next(g, 1);
        float b = rng_next(g, 1);
        float c = rng_next(g, 1);
        p = V(fmaf(2.0f, a, -1.0f), fmaf(2.0f, b, -1.0f), fmaf(2.0f, c, -1.0f));
    } while (dot3(p, p) >= 1.0f);
    return p;
}

/* sutil reflect(i,n) = i - 2 n dot(n,i) */
static v3 reflect3(v3 i, v3 n) {
    float k = -2.0f * dot3(n, i);
    return vfma(n, k, i);
}

typedef struct {
    uint64_t segments, shadow_rays;
} counters_t;

enum { EV_MISS = 0, EV_HIT = 1, EV_FINISH = 2, EV_CANCEL = 3 };

/* One camera path: raygen.cu:123-147 (one sample) + color :89-105 + rayColor :28-87. */
/* RTW_EST_CORRECTED. A light definition describes an emitting rectangle of the scene (Director.cpp:527-530 fills the
 * list by hand next to the geometry), but not always exactly: the Cornell box's light primitive sits at y = 554.9, its
 * definition at y = 554. The corrected estimator samples the geometry that emits: definition i is matched
 * to the untransformed emitting rectangle with the same normal axis and in-plane extent whose plane lies within 1 % of
 * the longer edge, and moved onto that plane; emitter hits on matched primitives are what light samples stand in for. */
static int corrected_lights_build(scene_t* sc) {
    const uint32_t nl = sc->h->n_lights, np = sc->h->n_prims;
    sc->clights = (rtw_light*)malloc((nl ? nl : 1) * sizeof(rtw_light));
    sc->listed = (uint8_t*)calloc(np ? np : 1, 1);
    if (!sc->clights || !sc->listed) return RTW_ERR_OOM;
    for (uint32_t i = 0; i < nl; i++) {
        rtw_light lt = sc->lights[i];
        for (uint32_t j = 0; j < np; j++) {
            const rtw_prim* pr = &sc->prims[j];
            if (pr->type < RTW_PRIM_RECT_X || pr->type > RTW_PRIM_RECT_Z || pr->xform != 0) continue;
            if (sc->mats[pr->material].type != RTW_MAT_DIFFUSE_LIGHT) continue;
            const int ax = pr->type - RTW_PRIM_RECT_X;                 /* normal axis */
            const int aa = ax == 0 ? 1 : 0, ab = ax == 2 ? 1 : 2;      /* in-plane axes a, b */
            const float ea = pr->p[1] - pr->p[0], eb = pr->p[3] - pr->p[2];
            float u[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
            u[aa] = ea; v[ab] = eb;
            int same = lt.position[aa] == pr->p[0] && lt.position[ab] == pr->p[2];
            for (int k = 0; k < 3; k++) if (lt.vec_u[k] != u[k] || lt.vec_v[k] != v[k]) same = 0;
            if (!same || !(fabsf(lt.position[ax] - pr->p[4]) <= 0.01f * fmaxf(ea, eb))) continue;
            lt.position[ax] = pr->p[4];
            sc->listed[j] = 1;
            break;
        }
        sc->clights[i] = lt;
    }
    return RTW_OK;
}
static void corrected_lights_free(scene_t* sc) {
    free(sc->clights); free(sc->listed);
    sc->clights = NULL; sc->listed = NULL;
}

/* RTW_EST_MIXTURE: the solid-angle density, seen from `so`, of "pick one of the nl listed lights uniformly, then a point
 * on its parallelogram uniformly" in the unit direction w: the sum over the lights whose parallelogram the ray
 * (so, w) meets of dist^2 / (area |cos|), over nl. What the reference's rect_*_value callables were meant to return
 * (pdf/rectPdf.cu:75-122 are stubs that return 1e-7 / 1e-8) - "The Rest of Your Life", hittable pdf. */
static float light_list_pdf(const rtw_light* lights, int nl, v3 so, v3 w) {
    float sum = 0.0f;
    for (int i = 0; i < nl; i++) {
        const rtw_light* lt = &lights[i];
        const v3 n = ld3(lt->normal), pos = ld3(lt->position), eu = ld3(lt->vec_u), evv = ld3(lt->vec_v);
        const float denom = dot3(w, n);
        const float dn = dot3(vsub(pos, so), n);
        if (denom == 0.0f) continue;
        const float t = dn / denom;
        if (!(t > 1.0e-6f)) continue;
        const v3 rel = vsub(vfma(w, t, so), pos);
        const float a = dot3(rel, eu) / dot3(eu, eu);
        const float b = dot3(rel, evv) / dot3(evv, evv);
        if (!(a >= -1.0e-4f && a <= 1.0001f && b >= -1.0e-4f && b <= 1.0001f)) continue;
        sum += (t * t) / (lt->area * fabsf(denom));
    }
    return sum / (float)nl;
}

static v3 trace_path(const scene_t* sc, const rtw_params* P, int px, int py, int sample, counters_t* cn) {
    const int est = P->estimator;
    int nee_prev = 0; /* RTW_EST_CORRECTED: a light sample was taken at the previous vertex */
    /* Ray epsilons. The reference starts scattered rays at t = 1e-6 from a hit point that is itself only accurate to
     * ~2e-5 in a 555-unit scene: about half of the rays leaving a surface re-hit it at once (the surface then
     * scatters twice). The corrected estimators start rays and end shadow probes 1e-3 away. */
    const float ray_tmin = est ? 1.0e-3f : 1e-6f;
    const float probe_eps = est ? 1.0e-3f : 500 * 1.0e-7f;
    const rtw_scene_header* H = sc->h;
    const rtw_camera* cam = &H->camera;
    rng_t g;
    memset(&g, 0, sizeof g);
    g.kind = P->rng_kind;
    uint32_t pixel = (uint32_t)P->width * (uint32_t)py + (uint32_t)px;
    g.key0 = P->seed; g.pixel = pixel; g.sample = (uint32_t)sample;
    if (g.kind == RTW_RNG_TEA_LCG) g.lcg[0] = rtwo_tea(64, pixel, (uint32_t)sample); /* raygen.cu:129 */

    /* raygen.cu:134-135 */
    float s = ((float)px + rng_next(&g, 0)) / (float)P->width;
    float t = ((float)py + rng_next(&g, 0)) / (float)P->height;
    /* shaders/camera.cu:11-19 + sampling.cuh:15-22 : the perspective camera draws twice even when the lens radius is 0;
     * the environment and orthographic cameras (scene/camera.cuh:35-56) take no seed and draw nothing */
    float la = 0.0f, lb = 0.0f;
    if (H->camera_type == RTW_CAM_PERSPECTIVE) {
        la = rng_next(&g, 0);
        lb = rng_next(&g, 0);
    }
    v3 origin = ld3(cam->origin);
    if (cam->lens_radius != 0.0f) {
        float sn, cs;
        rtwo_sincos2pi(la, &sn, &cs);
        float sq = sqrtf(lb);
        float rx = cam->lens_radius * (sn * sq);
        float ry = cam->lens_radius * (cs * sq);
        v3 off = vfma(ld3(cam->v), ry, vscale(ld3(cam->u), rx));
        origin = vadd(origin, off);
    }
    v3 dir = vfma(ld3(cam->horizontal), s, ld3(cam->lower_left));
    dir = vfma(ld3(cam->vertical), t, dir);
    if (H->camera_type == RTW_CAM_ENVIRONMENT) {
        /* scene/camera.cuh:35-47: sin / cos of 2 pi s and of pi t = 2 pi (t / 2), by the spec's sincos2pi */
        float sx, cx, sy, cy;
        rtwo_sincos2pi(s, &sx, &cx);
        rtwo_sincos2pi(t * 0.5f, &sy, &cy);
        v3 a = V(cx * sy, -cy, sx * sy);
        origin = ld3(cam->origin);
        dir = normalize3(vfma(ld3(cam->w), a.z, vfma(ld3(cam->v), a.y, vscale(ld3(cam->u), a.x))));
    } else if (H->camera_type == RTW_CAM_ORTHOGRAPHIC) {
        /* scene/camera.cuh:49-54 */
        origin = vadd(dir, ld3(cam->origin));
        dir = vneg(normalize3(ld3(cam->w)));
    } else {
        dir = vsub(dir, origin);
    }

    /* color(): raygen.cu:92-95 */
    if (g.kind == RTW_RNG_TEA_LCG) { g.lcg[1] = g.lcg[0]; g.lcg[2] = g.lcg[0]; }
    /* raygen.cu:95. Philox: the fifth raygen draw is cut from the low bytes of block 0 of stream 0 (whose four
     * 24-bit uniforms were the jitter and lens draws), so that a camera path costs one raygen block */
    float r_gather;
    if (g.kind == RTW_RNG_TEA_LCG) {
        r_gather = rng_next(&g, 0);
    } else {
        uint32_t ctr[4] = {g.pixel, g.sample, 0u, 0u};
        uint32_t key[2] = {g.key0, 0u};
        uint32_t out[4];
        rtwo_philox4x32_10(ctr, key, out);
        r_gather = (float)(((out[0] & 0xffu) << 16) | ((out[1] & 0xffu) << 8) | (out[2] & 0xffu)) * (1.0f / 16777216.0f);
    }
    float gather_time = fmaf(r_gather, cam->time1 - cam->time0, cam->time0);

    v3 T = V(1.f, 1.f, 1.f), L = V(0.f, 0.f, 0.f);
    int depth = 0;
    while (depth < P->max_depth) {
        float ray_time = rng_next(&g, 2); /* raygen.cu:48 */
        hit_t h;
        traverse(sc, origin, dir, ray_tmin, 1.e27f, ray_time, gather_time, &g, 0, 0, &h);
        cn->segments++;
        v3 radiance = V(0.f, 0.f, 0.f);
        int ev;
        v3 att = V(0.f, 0.f, 0.f), so = origin, sd = dir;
        if (h.prim < 0) {
            /* miss/miss.cu:8-30 */
            if (H->sky_light) {
                v3 u = normalize3(dir);
                float tt = 0.5f * (u.y + 1.0f);
                float w = 1.0f - tt;
                radiance = V(fmaf(tt, 0.5f, w), fmaf(tt, 0.7f, w), fmaf(tt, 1.0f, w));
            }
            ev = EV_MISS;
        } else {
            /* shaders/closehit.cu:45-121 */
            /* Philox stream 1: the closest-hit program of every segment starts at a fresh 4-word block, so that all
             * lanes of a GPU wave refill their generator at the same call sites (unused words are skipped) */
            if (g.kind == RTW_RNG_PHILOX) { g.draw[1] = (g.draw[1] + 3u) & ~3u; g.seg_base = g.draw[1]; }
            v3 hp, hn;
            float tu, tv;
            hit_attributes(sc, &h, origin, dir, &hp, &hn, &tu, &tv);
            const rtw_material* m = &sc->mats[sc->prims[h.prim].material];
            int specular = 0;
            switch (m->type) {
            case RTW_MAT_LAMBERTIAN: {
                /* material/lambertianMaterial.cu:41-71, lib/onb.cuh:20-32, sampling.cuh:49-60 (Q1) */
                v3 w = normalize3(hn);
                v3 a = (w.x > 0.9f || w.x < -0.9f) ? V(0.f, 1.f, 0.f) : V(1.f, 0.f, 0.f);
                v3 v = normalize3(cross3(w, a));
                v3 u = cross3(w, v);
                float r1 = rng_next(&g, 1);
                float r2 = rng_next(&g, 1);
                float sn, cs;
                rtwo_sincos2pi(r1, &sn, &cs);
                float sq = sqrtf(r2);
                float lx = est ? cs * sq : (cs * 2.0f) * sq; /* corrected: cosine-weighted, without the stray 2 (Q1) */
                float ly = est ? sn * sq : (sn * 2.0f) * sq;
                float lz = sqrtf(1.0f - r2);
                float pdf = lz * RTW_1_PI_F;
                v3 sdir = V(fmaf(lz, w.x, fmaf(ly, v.x, lx * u.x)),
                            fmaf(lz, w.y, fmaf(ly, v.y, lx * u.y)),
                            fmaf(lz, w.z, fmaf(ly, v.z, lx * u.z)));
                sdir = normalize3(sdir);
                so = hp; sd = sdir;
                float cosine = dot3(hn, sdir);
                ev = EV_HIT;
                if (cosine <= 0.0f || pdf <= 0.0f) { ev = EV_CANCEL; break; }
                att = texture_value(sc, m, tu, tv, hp);
                break;
            }
            case RTW_MAT_DIFFUSE_LIGHT: {
                /* material/diffuseLight.cu:48-69 */
                if (dot3(hn, dir) < 0.0f) radiance = texture_value(sc, m, tu, tv, hp);
                /* corrected: the light sample of the previous vertex already accounted for this emitter */
                if (est == RTW_EST_CORRECTED && nee_prev && sc->listed[h.prim]) radiance = V(0.f, 0.f, 0.f);
                ev = EV_CANCEL;
                break;
            }
            case RTW_MAT_METAL: {
                /* material/metalMaterial.cu:32-64 (Q5: direction is not normalised) */
                specular = 1;
                v3 refl = reflect3(est ? normalize3(dir) : dir, hn); /* corrected: unit incoming direction (Q5) */
                v3 ball = random_in_unit_sphere(&g);
                v3 sdir = normalize3(vfma(ball, m->fuzz_or_eta, refl));
                so = hp; sd = sdir;
                att = texture_value(sc, m, tu, tv, hp);
                ev = (dot3(sdir, hn) <= 0.0f) ? EV_CANCEL : EV_HIT;
                break;
            }
            case RTW_MAT_DIELECTRIC: {
                /* material/dielectricMaterial.cu:37-114 */
                specular = 1;
                v3 unit = normalize3(dir);
                v3 ln;
                float eta_i, eta_t;
                if (dot3(dir, hn) < 0.0f) { ln = hn; eta_i = 1.0f; eta_t = m->fuzz_or_eta; }
                else { ln = vneg(hn); eta_i = m->fuzz_or_eta; eta_t = 1.0f; }
                float cos_i = fminf(dot3(vneg(unit), ln), 1.0f);
                float sin_i = sqrtf(fmaf(-cos_i, cos_i, 1.0f));
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
                    float refl_prob = fmaf(1.0f - r0, p5, r0);
                    if (rng_next(&g, 1) < refl_prob) {
                        sdir = reflect3(unit, ln);
                    } else {
                        float sin_t = fminf(ratio * sin_i, 1.0f);
                        float cos_t = sqrtf(fmaf(-sin_t, sin_t, 1.0f));
                        v3 a = vscale(vfma(ln, cos_i, unit), ratio);
                        sdir = vfma(ln, -cos_t, a);
                    }
                }
                so = hp; sd = sdir;
                att = V(1.f, 1.f, 1.f);
                ev = EV_HIT;
                break;
            }
            case RTW_MAT_ISOTROPIC: {
                /* material/isotropicMaterial.cu:30-51 (Q14) */
                specular = 1;
                sd = random_in_unit_sphere(&g);
                so = hp;
                att = texture_value(sc, m, tu, tv, hp);
                ev = EV_HIT;
                break;
            }
            default: {
                /* material/normalMaterial.cu:21-31 */
                specular = 1;
                att = vfma(hn, 0.5f, V(0.5f, 0.5f, 0.5f));
                ev = EV_FINISH;
                break;
            }
            }
            /* next-event estimation: closehit.cu:70-118 */
            int nl = (int)H->n_lights;
            nee_prev = 0;
            if (est == RTW_EST_CORRECTED && ev == EV_HIT && !specular && nl > 0) {
                /* each listed light over its own parallelogram, area-measure estimator, no heuristic weight */
                nee_prev = 1;
                int il = 0;
                if (nl > 1) {
                    il = (int)floorf(rng_next(&g, 1) * (float)nl);
                    if (il < 0) il = 0;
                    if (il > nl - 1) il = nl - 1;
                }
                const rtw_light* lt = &sc->clights[il];
                float ra = rng_next(&g, 1);
                float rb = rng_next(&g, 1);
                v3 rp = vfma(ld3(lt->vec_v), rb, vfma(ld3(lt->vec_u), ra, ld3(lt->position)));
                v3 ldir = vsub(rp, so);
                float ldist = length3(ldir);
                if (ldist > 1.0e-6f && m->bsdf_eval == 0) {
                    ldir = vscale(ldir, 1.0f / ldist);
                    float costa = dot3(vneg(ldir), ld3(lt->normal));
                    float ndl = dot3(ldir, hn);
                    v3 f = vscale(att, RTW_1_PI_F);
                    if (costa > 1.0e-6f && ndl > 0.0f && (f.x != 0.0f || f.y != 0.0f || f.z != 0.0f)) {
                        const float eps = probe_eps;
                        hit_t sh;
                        traverse(sc, so, ldir, eps, ldist - eps, 0.0f, gather_time, &g, 1, 0, &sh);
                        cn->shadow_rays++;
                        if (sh.prim < 0) {
                            float lpdf = (ldist * ldist) / (lt->area * costa);
                            float k = ndl / lpdf;
                            v3 lem = vscale(ld3(lt->emission), (float)nl);
                            radiance = vadd(radiance, vscale(vmul(f, lem), k));
                        }
                    }
                }
            } else if (est == RTW_EST_MIXTURE && ev == EV_HIT && !specular && nl > 0 && m->bsdf_eval == 0) {
                /* The book's estimator ("The Rest of Your Life", mixture_pdf: what pdf/mixturePdf.cu set out to be, Q3 / Q4):
                 * the scattered direction comes from the light list or from the cosine lobe with probability 1/2 each, the
                 * throughput carries f cos / pdf = albedo * p_cos / (p_cos / 2 + p_light / 2), no shadow probe is traced and
                 * every emitter hit counts: one-sample multiple importance sampling with the balance heuristic. */
                int il = 0;
                if (nl > 1) {
                    il = (int)floorf(rng_next(&g, 1) * (float)nl);
                    if (il < 0) il = 0;
                    if (il > nl - 1) il = nl - 1;
                }
                const float u0 = rng_next(&g, 1);
                const float ra = rng_next(&g, 1);
                const float rb = rng_next(&g, 1);
                if (u0 < 0.5f) {
                    const rtw_light* lt = &sc->clights[il];
                    v3 rp = vfma(ld3(lt->vec_v), rb, vfma(ld3(lt->vec_u), ra, ld3(lt->position)));
                    v3 ldir = vsub(rp, so);
                    float ldist = length3(ldir);
                    if (ldist > 1.0e-6f) sd = vscale(ldir, 1.0f / ldist);
                }
                const float ndl = dot3(sd, hn);
                const float pb = fmaxf(0.0f, ndl) * RTW_1_PI_F;
                const float pl = light_list_pdf(sc->clights, nl, so, sd);
                const float pm = 0.5f * (pb + pl);
                if (pb > 0.0f && pm > 0.0f) att = vscale(att, pb / pm);
                else ev = EV_CANCEL;
            } else if (est == RTW_EST_REFERENCE && ev == EV_HIT && !specular && nl > 0) {
                int il = 0;
                if (nl > 1) {
                    il = (int)floorf(rng_next(&g, 1) * (float)nl);
                    if (il < 0) il = 0;
                    if (il > nl - 1) il = nl - 1;
                }
                const rtw_light* lt = &sc->lights[il];
                /* pdf/mixturePdf.cu:25-38 -> pdf/rectPdf.cu:124-193 (always child p1, Q4) */
                int gen = H->pdf.gen;
                if (gen == RTW_PDF_MIXTURE || gen == RTW_PDF_MIXTURE_BIAS) gen = H->pdf.p1_gen;
                float lpdf = 0.0f, ldist = 0.0f;
                v3 ldir = V(0.f, 0.f, 0.f), lem = V(0.f, 0.f, 0.f);
                if (gen == RTW_PDF_RECT_X || gen == RTW_PDF_RECT_Y || gen == RTW_PDF_RECT_Z) {
                    const float* rc = H->pdf.rect;
                    float ra = rng_next(&g, 1);
                    float rb = rng_next(&g, 1);
                    float pa = fmaf(ra, rc[1] - rc[0], rc[0]);
                    float pb = fmaf(rb, rc[3] - rc[2], rc[2]);
                    v3 rp = (gen == RTW_PDF_RECT_X) ? V(rc[4], pa, pb) : (gen == RTW_PDF_RECT_Y) ? V(pa, rc[4], pb) : V(pa, pb, rc[4]);
                    ldir = vsub(rp, so);
                    ldist = length3(ldir);
                    if (ldist > 1.0e-6f) {
                        ldir = vscale(ldir, 1.0f / ldist);
                        float costa = dot3(vneg(ldir), ld3(lt->normal));
                        if (costa > 1.0e-6f) {
                            lem = vscale(ld3(lt->emission), (float)nl);
                            lpdf = (ldist * ldist) / (lt->area * costa);
                        }
                    }
                }
                if (lpdf > 0.0f && m->bsdf_eval == 0) {
                    /* lambertianMaterial.cu:74-81 */
                    v3 f = vscale(att, RTW_1_PI_F);
                    float ndl = dot3(ldir, hn);
                    float bpdf = fmaxf(0.0f, ndl * RTW_1_PI_F);
                    if (0.0f < bpdf && (f.x != 0.0f || f.y != 0.0f || f.z != 0.0f)) {
                        const float eps = 500 * 1.0e-7f;
                        hit_t sh;
                        traverse(sc, so, ldir, eps, ldist - eps, 0.0f, gather_time, &g, 1, 0, &sh);
                        cn->shadow_rays++;
                        if (sh.prim < 0) {
                            float a2 = lpdf * lpdf;
                            float weight = a2 / fmaf(bpdf, bpdf, a2); /* raydata.cuh:167-171 */
                            float k = (weight * ndl) / lpdf;
                            radiance = vadd(radiance, vscale(vmul(f, lem), k));
                        }
                    }
                }
            }
        }
        L = vadd(L, vmul(radiance, T)); /* raygen.cu:60 — product, then sum: the GPU may hold the product of a light sample
                                         * until its shadow probe is back and add it then, with the same two roundings */
        if (ev != EV_HIT) break;
        origin = so; dir = sd;
        T = vmul(T, att);
        if (2 <= depth) {
            /* raygen.cu:74-82 */
            float p = fmaxf(fmaxf(T.x, T.y), T.z);
            /* the mixture estimator's weights reach 2, so its throughput can exceed 1: a survival probability is at most 1
             * (the reference's own throughput never exceeds 1, so its roulette needs no such cap) */
            if (est == RTW_EST_MIXTURE) p = fminf(p, 1.0f);
            float xi = rng_rr(&g);
            if (!g_debug_no_roulette) {  /* (the draw is consumed either way, so both settings walk the same streams) */
                if (p < xi) break;
                T = vscale(T, 1.0f / p);
            }
        }
        depth++;
    }
    /* raygen.cu:17-24 */
    if (!(L.x == L.x)) L.x = 0.f;
    if (!(L.y == L.y)) L.y = 0.f;
    if (!(L.z == L.z)) L.z = 0.f;
    return L;
}

static int row_stride_of(const rtw_params* P) { return P->row_stride > 1 ? P->row_stride : 1; }
static int local_rows_of(const rtw_params* P) { int k = row_stride_of(P); return (P->row1 - P->row0 + k - 1) / k; }

/* local rows [l0,l1) of the shard: local row l is image row row0 + l*row_stride */
static void render_rows(const scene_t* sc, const rtw_params* P, int l0, int l1, float* out, counters_t* cn) {
    int W = P->width;
    int k = row_stride_of(P);
    for (int l = l0; l < l1; l++) {
        int y = P->row0 + l * k;
        for (int x = 0; x < W; x++) {
            /* summation order of include/rtw.h (RTW_SUM_BLOCK, RTW_SUM_UNIT_BLOCKS): samples in order inside aligned blocks, the
             * block sums in order inside aligned units of blocks, the unit sums in order */
            v3 sum = V(0.f, 0.f, 0.f);
            const int unit = RTW_SUM_BLOCK * RTW_SUM_UNIT_BLOCKS;
            for (int u0 = 0; u0 < P->spp; u0 += unit) {
                v3 usum = V(0.f, 0.f, 0.f);
                const int u1 = u0 + unit < P->spp ? u0 + unit : P->spp;
                for (int s0 = u0; s0 < u1; s0 += RTW_SUM_BLOCK) {
                    v3 bsum = V(0.f, 0.f, 0.f);
                    const int s1 = s0 + RTW_SUM_BLOCK < u1 ? s0 + RTW_SUM_BLOCK : u1;
                    for (int s = s0; s < s1; s++) {
                        v3 L = trace_path(sc, P, x, y, P->sample_offset + s, cn);
                        bsum = vadd(bsum, L);
                    }
                    usum = vadd(usum, bsum);
                }
                sum = vadd(sum, usum);
            }
            float n = (float)P->spp;
            float* o = out + 4 * ((size_t)l * (size_t)W + (size_t)x);
            o[0] = sum.x / n; o[1] = sum.y / n; o[2] = sum.z / n; o[3] = 1.0f;
        }
    }
}

typedef struct {
    const scene_t* sc;
    const rtw_params* P;
    int r0, r1;
    float* out;
    counters_t cn;
} job_t;

static void* job_main(void* arg) {
    job_t* j = (job_t*)arg;
    render_rows(j->sc, j->P, j->r0, j->r1, j->out, &j->cn);
    return NULL;
}

static int check_params(const rtw_params* P) {
    if (!P || P->width <= 0 || P->height <= 0 || P->spp <= 0 || P->max_depth < 0) return RTW_ERR_INVALID_ARG;
    if (P->row0 < 0 || P->row1 > P->height || P->row0 > P->row1 || P->row_stride < 0) return RTW_ERR_INVALID_ARG;
    if (P->rng_kind != RTW_RNG_PHILOX && P->rng_kind != RTW_RNG_TEA_LCG) return RTW_ERR_INVALID_ARG;
    if (P->estimator < RTW_EST_REFERENCE || P->estimator > RTW_EST_MIXTURE) return RTW_ERR_INVALID_ARG;
    return RTW_OK;
}

/* Oracle render: same inputs and output layout as rtw_render. threads<=1: scalar single thread. */
int rtwo_render(const void* blob, size_t bytes, const rtw_params* P, float* rgba_out, rtw_stats* st, int threads) {
    scene_t sc;
    int rc = scene_open(&sc, blob, bytes);
    if (rc) return rc;
    rc = check_params(P);
    if (rc) return rc;
    if (!rgba_out) return RTW_ERR_INVALID_ARG;
    if ((P->estimator == RTW_EST_CORRECTED || P->estimator == RTW_EST_MIXTURE) && (rc = corrected_lights_build(&sc)) != RTW_OK) { corrected_lights_free(&sc); return rc; }
    sc.bounded_media = P->estimator != RTW_EST_REFERENCE;
    int rows = local_rows_of(P);
    if (threads < 1) threads = 1;
    if (threads > rows) threads = rows > 0 ? rows : 1;
    if (threads > 256) threads = 256;
    job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); corrected_lights_free(&sc); return RTW_ERR_OOM; }
    for (int i = 0; i < threads; i++) {
        jobs[i].sc = &sc; jobs[i].P = P; jobs[i].out = rgba_out;
        jobs[i].r0 = (int)(((int64_t)rows * i) / threads);
        jobs[i].r1 = (int)(((int64_t)rows * (i + 1)) / threads);
    }
    if (threads == 1) {
        job_main(&jobs[0]);
    } else {
        for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, job_main, &jobs[i]);
        for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    }
    if (st) {
        memset(st, 0, sizeof *st);
        for (int i = 0; i < threads; i++) { st->segments += jobs[i].cn.segments; st->shadow_rays += jobs[i].cn.shadow_rays; }
        st->samples = (uint64_t)rows * (uint64_t)P->width * (uint64_t)P->spp;
        st->algorithmic_bytes = 128u * st->segments + 32u * st->samples;
    }
    free(jobs); free(th);
    corrected_lights_free(&sc);
    return RTW_OK;
}

/* Brute-force closest hit over non-volume primitives, same I/O as rtw_debug_intersect. */
int rtwo_intersect(const void* blob, size_t bytes, const float* rays, const float* ray_time, const float* gather_time,
                   int n, float* out_t, int32_t* out_prim) {
    scene_t sc;
    int rc = scene_open(&sc, blob, bytes);
    if (rc) return rc;
    rng_t g;
    memset(&g, 0, sizeof g);
    for (int i = 0; i < n; i++) {
        const float* r = rays + 8 * (size_t)i;
        hit_t h;
        traverse(&sc, V(r[0], r[1], r[2]), V(r[3], r[4], r[5]), r[6], r[7], ray_time ? ray_time[i] : 0.f,
                 gather_time ? gather_time[i] : 0.f, &g, 0, 1, &h);
        out_t[i] = h.t;
        out_prim[i] = h.prim;
    }
    return RTW_OK;
}

/* One path's radiance (debugging aid for the parity tests). */
int rtwo_trace_pixel(const void* blob, size_t bytes, const rtw_params* P, int px, int py, int sample, float* rgb_out) {
    scene_t sc;
    int rc = scene_open(&sc, blob, bytes);
    if (rc) return rc;
    counters_t cn = {0, 0};
    if ((P->estimator == RTW_EST_CORRECTED || P->estimator == RTW_EST_MIXTURE) && (rc = corrected_lights_build(&sc)) != RTW_OK) { corrected_lights_free(&sc); return rc; }
    sc.bounded_media = P->estimator != RTW_EST_REFERENCE;
    v3 L = trace_path(&sc, P, px, py, sample, &cn);
    corrected_lights_free(&sc);
    rgb_out[0] = L.x; rgb_out[1] = L.y; rgb_out[2] = L.z;
    return RTW_OK;
}

/* CPU restatement of rtw_denoise (include/rtw.h): the same taps, weights and order of operations. */
int rtwo_denoise(const float* rgba_in, float* rgba_out, int width, int height, int iterations, float sigma) {
    if (!rgba_in || !rgba_out || rgba_in == rgba_out || width <= 0 || height <= 0 || iterations < 1 || iterations > 8 || !(sigma > 0.f))
        return RTW_ERR_INVALID_ARG;
    const size_t n = (size_t)width * height;
    float* buf[2] = {(float*)malloc(n * 16), (float*)malloc(n * 16)};
    if (!buf[0] || !buf[1]) { free(buf[0]); free(buf[1]); return RTW_ERR_OOM; }
    memcpy(buf[0], rgba_in, n * 16);
    const float kern[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
    int cur = 0;
    float s_i = sigma;
    for (int it = 0; it < iterations; it++) {
        const float* in = buf[cur];
        float* out = buf[cur ^ 1];
        const int step = 1 << it;
        const float inv_sigma2 = 1.0f / (s_i * s_i);
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                const float* c = in + 4 * ((size_t)y * width + x);
                float sr = 0.f, sg = 0.f, sb = 0.f, sw = 0.f;
                for (int dy = -2; dy <= 2; dy++) {
                    int yy = y + dy * step;
                    yy = yy < 0 ? 0 : (yy > height - 1 ? height - 1 : yy);
                    for (int dx = -2; dx <= 2; dx++) {
                        int xx = x + dx * step;
                        xx = xx < 0 ? 0 : (xx > width - 1 ? width - 1 : xx);
                        const float* q = in + 4 * ((size_t)yy * width + xx);
                        const float dr = c[0] - q[0], dg = c[1] - q[1], db = c[2] - q[2];
                        const float d2 = (dr * dr + dg * dg) + db * db;
                        const float w = (kern[dy + 2] * kern[dx + 2]) / (1.0f + d2 * inv_sigma2);
                        sr = sr + w * q[0]; sg = sg + w * q[1]; sb = sb + w * q[2]; sw = sw + w;
                    }
                }
                float* o = out + 4 * ((size_t)y * width + x);
                o[0] = sr / sw; o[1] = sg / sw; o[2] = sb / sw; o[3] = c[3];
            }
        cur ^= 1;
        s_i = s_i * 0.5f;
    }
    memcpy(rgba_out, buf[cur], n * 16);
    free(buf[0]); free(buf[1]);
    return RTW_OK;
}
