/* rtw.h — C ABI of the MI355X wavefront path tracer (librtw_hip.so).
 *
 * This is the drop-in boundary for the reference's one device launch:
 *
 *   Director::renderFrame  ->  optixLaunch(pipeline, stream, d_params,
 *                                          sizeof(SysParamter), &m_sbt, Nx, Ny, 1)
 *                              (reference RestOfLife/Director.cpp:971-1008, launch at :982-984)
 *
 * and for the marshalling that feeds it:
 *
 *   Director::createSBT          (Director.cpp:628-885)  geometry records  -> rtw_prim[]
 *   Director::initLaunchParams   (Director.cpp:483-553)  camera, materials, textures,
 *                                                        lights, pdf tree  -> rtw_scene_header + arrays
 *   SysParamter                  (shaders/sysparameter.h:32-60)            -> rtw_scene_header + rtw_params
 *   HitGroupData / hitRectData / hitVolumeBoxdata (lib/raydata.cuh:79-115) -> rtw_prim
 *   OptixInstance transform[12] / instanceId / sbtOffset
 *                                (geometry/ioGeometryInstance.h:20-26)     -> rtw_xform + rtw_prim.material
 *   MaterialParams / textureParam (sysparameter.h:5-16, raydata.cuh:127-138)-> rtw_material / rtw_texture
 *   LightDefinition              (raydata.cuh:31-48)                       -> rtw_light
 *   pdfCallfun                   (sysparameter.h:18-30)                    -> rtw_pdf
 *
 * Plain pointers and sizes only. No exceptions and no exit() cross this ABI:
 * every call returns 0 on success or a negative rtw_status; the message is
 * available from rtw_last_error().  Nothing is retained from caller pointers
 * after a call returns.
 */
#ifndef RTW_H
#define RTW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTW_ABI_VERSION 2   /* entry points and the structs they take (rtw_params, rtw_stats) */
#define RTW_SCENE_VERSION 1 /* layout of the scene blob (rtw_scene_header.version) */
#define RTW_SCENE_MAGIC 0x57545221u /* "!RTW" */
/* Summation order of a pixel's samples (part of the arithmetic contract). Three levels, all counted from
 * sample_offset: the samples of a render call are summed in ascending order inside aligned BLOCKS of RTW_SUM_BLOCK samples;
 * the block sums in ascending order inside aligned UNITS of RTW_SUM_UNIT_BLOCKS blocks (128 samples); the unit sums in
 * ascending order; the mean is that sum divided by spp. The reference renders one sample per launch (raygen.cu:123-159) and
 * so defines no order. Blocks let a lane own a run of a pixel's samples in registers, and a small block lets the last few
 * percent of a render be handed out in short pieces (a launch ends when its slowest piece ends); units are what a lane of a
 * long render owns at a time, so that one 16-byte sum per 128 samples reaches memory instead of one per 16 (round 3: the
 * metric frame's sums take 1.3 GB instead of 8.5 GB, BASELINE config 5's shard 2.6 GB instead of 17 GB in two passes).
 * Renders of at most 128 spp are unchanged by the unit level (0 + x = x). */
#define RTW_SUM_BLOCK 16
#define RTW_SUM_UNIT_BLOCKS 8

typedef enum rtw_status {
    RTW_OK = 0,
    RTW_ERR_INVALID_ARG = -1,
    RTW_ERR_BAD_SCENE = -2,
    RTW_ERR_NO_SCENE = -3,
    RTW_ERR_DEVICE = -4,
    RTW_ERR_OOM = -5,
    RTW_ERR_UNSUPPORTED = -6
} rtw_status;

/* Primitive kinds. One per reference intersection program
 * (Director.h ProgramIdentifier PROGRAM_ID_IS_*; geometry/sphere.cu, movingSphere.cu,
 * shaders/aarect{x,y,z}.cu, geometry/volumeBox.cu, volumeSphere.cu). */
typedef enum rtw_prim_type {
    RTW_PRIM_SPHERE = 0,        /* p[0..2] centre, p[3] radius                                   */
    RTW_PRIM_MOVING_SPHERE = 1, /* p[0..2] centre0, p[3] radius, p[4..6] centre1, p[7] t0, p[8] t1;
                                   carries the reference's matrix-motion transform translate(C0)->translate(C1)
                                   (geometry/ioMovingSphere.h:161-203) on top of its xform             */
    RTW_PRIM_RECT_X = 2,        /* p[0] a0 p[1] a1 p[2] b0 p[3] b1 p[4] k ; a=y b=z (aarectx.cu)  */
    RTW_PRIM_RECT_Y = 3,        /*                                          a=x b=z (aarecty.cu)  */
    RTW_PRIM_RECT_Z = 4,        /*                                          a=x b=y (aarectz.cu)  */
    RTW_PRIM_VOLUME_BOX = 5,    /* p[0..2] boxMin, p[3..5] boxMax, p[6] density                   */
    RTW_PRIM_VOLUME_SPHERE = 6  /* p[0..2] centre, p[3] radius, p[4] density                      */
} rtw_prim_type;

/* Same numbering as MaterialType, lib/raydata.cuh:22-29. */
typedef enum rtw_material_type {
    RTW_MAT_LAMBERTIAN = 0,
    RTW_MAT_DIFFUSE_LIGHT = 1,
    RTW_MAT_METAL = 2,
    RTW_MAT_DIELECTRIC = 3,
    RTW_MAT_ISOTROPIC = 4,
    RTW_MAT_NORMAL = 5
} rtw_material_type;

/* Same numbering as TexCallFunction, shaders/FunctionIdx.h:8-15. */
typedef enum rtw_texture_type {
    RTW_TEX_CHECKER = 0,
    RTW_TEX_CONSTANT = 1,
    RTW_TEX_IMAGE = 2,
    RTW_TEX_NOISE = 3,
    RTW_TEX_NULL = 4
} rtw_texture_type;

/* Same numbering as PDFCallFunction generate ids, shaders/FunctionIdx.h:27-33. */
typedef enum rtw_pdf_gen {
    RTW_PDF_COSINE = 0,
    RTW_PDF_MIXTURE_BIAS = 1,
    RTW_PDF_MIXTURE = 2,
    RTW_PDF_RECT_X = 3,
    RTW_PDF_RECT_Y = 4,
    RTW_PDF_RECT_Z = 5
} rtw_pdf_gen;

typedef enum rtw_rng_kind {
    RTW_RNG_PHILOX = 0,  /* Philox4x32-10, key=(seed,0), counter=(pixel, sample, block, stream)        */
    RTW_RNG_TEA_LCG = 1  /* the reference's own: tea<64>(pixel, sample) + 24-bit LCG rnd() + xorshift
                            randf() (lib/random.cuh:7-38, raygen/raygen.cu:129)                       */
} rtw_rng_kind;

typedef struct rtw_prim {
    int32_t type;     /* rtw_prim_type                                                             */
    int32_t material; /* index into materials[] (== OptixInstance.instanceId, closehit.cu:50,63)  */
    int32_t xform;    /* index into xforms[]; 0 is the identity                                   */
    int32_t flip;     /* rects: hitRectData.flip                                                   */
    float p[12];
} rtw_prim; /* 64 B */

/* Object->world 3x4 row-major (OptixInstance.transform) and its inverse (world->object). */
typedef struct rtw_xform {
    float m[12];
    float inv[12];
} rtw_xform; /* 96 B */

typedef struct rtw_material {
    int32_t type;       /* rtw_material_type                        */
    int32_t texture;    /* index into textures[] or -1              */
    float fuzz_or_eta;  /* MaterialParams union{fuzz,eta}           */
    int32_t bsdf_eval;  /* MaterialParams.lightreflectIdx - CALLABLE_ID_LIGHT_SAMPLE_PDF: 0 diffuse, 1 dielectric, 2 metal, -1 none */
} rtw_material; /* 16 B */

/* textureParam (lib/raydata.cuh:127-138) without its device pointers: tables and texels live in the blob's texture
 * data section (rtw_scene_header.off_texdata), addressed in 4-byte words from the start of that section.
 *   RTW_TEX_CONSTANT  color                                  (texture/constantTexture.cu)
 *   RTW_TEX_CHECKER   odd / even = indices into textures[]   (texture/checkeredTexture.cu; the reference stores the
 *                     children's callable ids and so never shows a checker - here the children are evaluated)
 *   RTW_TEX_NOISE     scale; data -> float ranvec[256][3], int32 perm_x[256], perm_y[256], perm_z[256]
 *                     (texture/noiseTexture.cu, ioTexture.h:118-222)
 *   RTW_TEX_IMAGE     data -> uint32 width, height, then width*height texels r | g<<8 | b<<16 | a<<24, row 0 at
 *                     v = 0 (texture/imageTexture.cu; ioTexture.h:225-262 flips the file's rows the same way);
 *                     sampled bilinearly, clamped, texel centres at (i + 0.5) / width                        */
typedef struct rtw_texture {
    int32_t type; /* rtw_texture_type */
    float color[3];
    int32_t odd, even;
    float scale;
    uint32_t data; /* word offset into the texture data section (noise, image), else 0 */
} rtw_texture; /* 32 B */

typedef struct rtw_light {
    float position[3];
    float vec_u[3];
    float vec_v[3];
    float normal[3];
    float area;
    float emission[3];
} rtw_light; /* 64 B */

typedef struct rtw_pdf {
    int32_t gen;     /* top-level generate id (rtw_pdf_gen), e.g. RTW_PDF_MIXTURE */
    int32_t p0_gen;  /* mixture child 0 (cosine) or -1                          */
    int32_t p1_gen;  /* mixture child 1 (rect x/y/z) or -1                      */
    int32_t flip;
    float rect[5];   /* p1's hitRectData a0,a1,b0,b1,k                          */
    float bias;
} rtw_pdf; /* 40 B */

/* Camera kinds: the reference's scene/camera.cuh:35-56 `cameraType` values (its OptiX-7 path only ever builds type 0,
 * shaders/camera.cu:11-19; the other two are defined by scene/ioCamera.h:118-179 and never instantiated).
 *   RTW_CAM_PERSPECTIVE   origin (+ lens offset), direction = lower_left + s*horizontal + t*vertical - origin
 *   RTW_CAM_ENVIRONMENT   origin; a = (cos(2 pi s) sin(pi t), -cos(pi t), sin(2 pi s) sin(pi t)); direction = normalize(a.x u + a.y v + a.z w)
 *   RTW_CAM_ORTHOGRAPHIC  origin = lower_left + s*horizontal + t*vertical + camera origin (as camera.cuh:52 states it: the origin
 *                         enters twice when lower_left is built the way ioOrthographicCamera builds it); direction = -normalize(w)
 * Only the perspective camera draws a lens sample (two draws, consumed even at lens radius 0: camera.cu:11-19); the other
 * two take no seed in the reference and draw nothing (visible in the TEA+LCG stream; Philox raygen draws are positional). */
typedef enum rtw_camera_type { RTW_CAM_PERSPECTIVE = 0, RTW_CAM_ENVIRONMENT = 1, RTW_CAM_ORTHOGRAPHIC = 2 } rtw_camera_type;

typedef struct rtw_camera {
    float origin[3];
    float u[3], v[3], w[3];
    float lower_left[3];
    float horizontal[3];
    float vertical[3];
    float lens_radius; /* SysParamter.cameraLensRadius (never set by the reference: 0) */
    float time0, time1;
} rtw_camera; /* 96 B */

/* The scene blob is this header followed by the arrays at the given byte offsets
 * (all offsets relative to the start of the header, 16-byte aligned). */
typedef struct rtw_scene_header {
    uint32_t magic;   /* RTW_SCENE_MAGIC */
    uint32_t version; /* RTW_SCENE_VERSION */
    uint32_t total_bytes;
    uint32_t n_prims, n_xforms, n_materials, n_textures, n_lights;
    uint32_t off_prims, off_xforms, off_materials, off_textures, off_lights;
    int32_t sky_light; /* SysParamter.skyLight */
    uint32_t off_texdata;   /* byte offset of the texture data section (0 = none)  */
    uint32_t texdata_bytes; /* its size; rtw_texture.data counts 4-byte words in it */
    rtw_camera camera;
    rtw_pdf pdf;
    int32_t camera_type; /* rtw_camera_type (these 8 bytes were rtw_pdf.reserved, always 0, before the cameras of SURVEY 8f rank 4) */
    uint32_t reserved;
} rtw_scene_header;

/* Estimators. The default reproduces the reference, including what makes its images
 * physically off: the stray factor 2 in the cosine sampler, light samples weighted by the power heuristic
 * while emitter hits are counted in full (Q3), the pdf rectangle of the scene instead of the chosen light's own (Q12,
 * and every light but the first), an un-normalised incoming direction in the metal reflection (Q5).
 * RTW_EST_CORRECTED fixes those: cosine-weighted scattering, each listed light sampled over its own parallelogram with
 * the plain area-measure estimator, emitter hits of listed lights counted only where no light sample stood in for them,
 * media that scatter only inside their extent (Q9), rays started 1e-3 (not 1e-6) away from the hit point.
 * RTW_EST_CORRECTED_NO_NEE is the same integrand estimated without light sampling (every emitter hit counts): slow to
 * converge, but an independent check - both converge to the same image.
 * RTW_EST_MIXTURE is the same integrand again, estimated the way the reference's pdf/ callables set out to ("The Rest of
 * Your Life", mixture_pdf): at a diffuse vertex the scattered direction is drawn from the light list or
 * from the cosine lobe with probability 1/2 each, the throughput carries albedo * p_cos / (p_cos / 2 + p_light / 2) with
 * p_light the true solid-angle density of the light list (the reference's rect_*_value stubs return constants), no
 * shadow probe is traced and every emitter hit counts - one-sample multiple importance sampling, balance heuristic. */
typedef enum rtw_estimator {
    RTW_EST_REFERENCE = 0,
    RTW_EST_CORRECTED = 1,
    RTW_EST_CORRECTED_NO_NEE = 2,
    RTW_EST_MIXTURE = 3
} rtw_estimator;

typedef struct rtw_params {
    int32_t width, height;     /* full image (launch dimensions of the reference's optixLaunch) */
    int32_t spp;               /* samples per pixel rendered by this call                       */
    int32_t max_depth;         /* SysParamter.maxRayDepth                                       */
    uint32_t seed;
    int32_t row0, row1;        /* rows [row0,row1) of the full image are rendered (row shard of one GPU)      */
    int32_t rng_kind;          /* rtw_rng_kind                                                  */
    int32_t sample_offset;     /* first sample index (progressive / resumed renders)            */
    int32_t samples_per_pass;  /* paths kept in flight = rows*width*samples_per_pass; 0 = auto  */
    int32_t row_stride;        /* 0 or 1: every row of [row0,row1). k > 1: rows row0, row0+k, row0+2k ... < row1
                                  (interleaved shard: rank g of N uses row0=g, row1=height, row_stride=N, which
                                  balances the ranks); the output holds those rows consecutively               */
    int32_t estimator;         /* rtw_estimator: 0 = the reference's estimator, quirks and all (the parity mode)  */
} rtw_params;

/* kernels of the wavefront loop, index into the per-kernel arrays of rtw_stats */
enum { RTW_K_FIRST = 0, RTW_K_SHADE = 1, RTW_K_TRACE = 2, RTW_K_BOUNCE = 3, RTW_K_PATH = 4, RTW_K_COUNT = 5 };

typedef struct rtw_stats {
    uint64_t samples;           /* camera paths started                                              */
    uint64_t segments;          /* radiance ray segments traced (one optixTraverse of raygen.cu:41) */
    uint64_t shadow_rays;       /* occlusion probes traced (closehit.cu:95-101)                     */
    uint64_t algorithmic_bytes; /* 128*segments + 32*samples                 */
    uint64_t bounce_launches;   /* launches of the wavefront-loop kernels (all four kinds)          */
    uint64_t reserved;
    double seconds;             /* device time of the whole render call (events on the stream)     */
    double bounce_seconds;      /* device time inside the wavefront loops (first launch to last)   */
    /* per kernel kind, measured with HIP events recorded on the launch stream around every launch */
    double kernel_seconds[RTW_K_COUNT];
    uint64_t kernel_launches[RTW_K_COUNT];
    uint64_t kernel_segments[RTW_K_COUNT]; /* radiance segments shaded by that kernel (k_trace: path slots traced = a radiance ray and / or its queued probe) */
} rtw_stats;

typedef struct rtw_ctx rtw_ctx;

int rtw_abi_version(void);

/* Replaces Director::initContext (Director.cpp:106-122).
 * n_devices == 1: one context on device_ids[0] (NULL: device 0).
 * n_devices  > 1: a group. rtw_upload_scene copies the scene to every device; rtw_render / rtw_render_device split the
 * rows of the call into n_devices interleaved shards (shard g: every n_devices-th row of the call's rows, starting at its
 * g-th), render shard g on device_ids[g] from that device's own host thread (created here, alive until rtw_destroy: no
 * thread is created per call), every device pushes its float4 shard to device_ids[0] with one hipMemcpyPeerAsync on its
 * own stream as soon as it is done (n concurrent xGMI transfers) and the rows are interleaved there: the caller sees one
 * frame, bit-identical to the single-device render. Entries of device_ids may repeat (two shards on one GPU). The caller
 * stays single-threaded; a worker that fails or throws reports through the call's return code. */
int rtw_create(rtw_ctx** out, int n_devices, const int* device_ids);

/* Replaces createSBT + initLaunchParams + the per-primitive optixAccelBuild calls
 * (Director.cpp:628-885, 483-553; geometry/io*.h init()). Copies the blob, builds the BVH. */
int rtw_upload_scene(rtw_ctx* ctx, const void* scene_blob, size_t bytes);

/* Replaces optixLaunch + the D2H copy (Director.cpp:982-984, 999-1000).
 * rgba_out: host, rows*width float4 (rows = the rows of [row0,row1) the shard owns: all of them, or every
 * row_stride-th), LINEAR mean radiance of samples [sample_offset, sample_offset+spp), alpha 1; row r of the output is
 * image row row0 + r*max(row_stride,1); image row 0 is the bottom row, like the reference's frame buffer. */
int rtw_render(rtw_ctx* ctx, const rtw_params* params, float* rgba_out, rtw_stats* stats);

/* Same render, result left in device memory (d_rgba: device pointer on device_ids[0], same layout). The work is ordered
 * on hip_stream (a hipStream_t passed as void*): the render starts after what that stream holds and the final frame is
 * written on it. NULL (which is also HIP's legacy default stream handle) selects the context's own non-blocking stream,
 * which is NOT ordered with the default stream: pass a stream of your own (or hipStreamLegacy / hipStreamPerThread) when
 * d_rgba has pending work. Returns when done. rtw_stats.kernel_seconds is filled from HIP events recorded on the launch
 * streams around every kernel (only when stats != NULL; RTW_KERNEL_TIMING=0 turns the events off). */
int rtw_render_device(rtw_ctx* ctx, const rtw_params* params, void* d_rgba, void* hip_stream,
                      rtw_stats* stats);

/* Replaces Director::destroy (Director.cpp:66-104). */
int rtw_destroy(rtw_ctx* ctx);

/* Message of the last failing call on this context (what OPTIX_CHECK / CUDA_CHECK would have thrown,
 * sutil/Exception.h); valid until the next call on the context. */
const char* rtw_last_error(rtw_ctx* ctx);

/* Stand-in for the reference's output stage, the OptiX AI denoiser (Director::initDenoiser, Director.cpp:887-949,
 * invoked at :986-997 on the beauty layer alone, LDR model, no albedo / normal guides). The AI model is closed; this is
 * an edge-avoiding a-trous wavelet filter on the same input (Dammertz et al. 2010, colour edge-stopping only):
 * `iterations` passes with a 5x5 B3-spline kernel at hole sizes 1, 2, 4 ..., each tap weighted by
 * 1 / (1 + |c_p - c_q|^2 / sigma_i^2), sigma_i = sigma * 2^-i. rgba_in / rgba_out: host, width*height float4, may not
 * alias; alpha is copied. Meant for display-encoded values in [0, 1] (the reference's LDR model sees sqrt(colour));
 * not part of rtw_render: 4096-spp frames need none. iterations in 1..8, sigma > 0. */
int rtw_denoise(rtw_ctx* ctx, const float* rgba_in, float* rgba_out, int32_t width, int32_t height, int32_t iterations, float sigma);

/* Test hooks (no reference counterpart): one closest-hit query per ray on the GPU accel structure,
 * used by the parity tests to compare BVH traversal with the oracle's brute force.
 * rays: n*8 floats (ox,oy,oz,dx,dy,dz,tmin,tmax); ray_time: n floats or NULL;
 * out_t: n floats (tmax if miss); out_prim: n int32 (-1 if miss). All host pointers. */
int rtw_debug_intersect(rtw_ctx* ctx, const float* rays, const float* ray_time, const float* gather_time,
                        int n, float* out_t, int32_t* out_prim);

#ifdef __cplusplus
}
#endif
#endif /* RTW_H */
