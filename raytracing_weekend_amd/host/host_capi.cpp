// host_capi.cpp — C entry points over the C++ host scene description, so that the Python
// plumbing (bench.py, tests) can obtain the same scene blobs the Director uploads.
// No GPU dependency: librtw_host.so loads on CPU-only machines.
#include <cstring>
#include <new>

#include "SceneMarshal.h"

// No exception crosses these entry points (the scene description allocates: std::vector, std::string, texture decode)
#define RTW_HOST_GUARD_BEGIN try {
#define RTW_HOST_GUARD_END                                   \
    } catch (const std::bad_alloc&) { return RTW_ERR_OOM; }  \
    catch (...) { return RTW_ERR_BAD_SCENE; }

extern "C" {

// Builds reference scene `scene` (0 Cornell box, 1 moving spheres, 3 Cornell box with volumes) for an
// Nx x Ny image and writes the blob to buf (capacity cap). *needed receives the blob size.
// Returns 0, RTW_ERR_INVALID_ARG for an unknown scene (ioScene::init -> 1), or RTW_ERR_OOM when cap is too small.
int rtw_host_build_scene(int scene, int nx, int ny, void* buf, size_t cap, size_t* needed) {
    RTW_HOST_GUARD_BEGIN
    if (nx <= 0 || ny <= 0) return RTW_ERR_INVALID_ARG;
    // scene + 100 * kind selects a camera kind (rtw_camera_type) for the scene: 100..104 environment, 200..204 orthographic
    const int cam_kind = scene / 100;
    scene %= 100;
    if (cam_kind < 0 || cam_kind > RTW_CAM_ORTHOGRAPHIC) return RTW_ERR_INVALID_ARG;
    rtwhost::ioScene sc;
    if (sc.init(nx, ny, 1, 20, scene)) return RTW_ERR_INVALID_ARG;
    sc.setCameraKind(cam_kind);
    std::vector<uint8_t> blob = rtwhost::marshalScene(sc);
    if (needed) *needed = blob.size();
    if (!buf || cap < blob.size()) return RTW_ERR_OOM;
    memcpy(buf, blob.data(), blob.size());
    return RTW_OK;
    RTW_HOST_GUARD_END
}

// Test hook for the image-texture decoder (JpegDecode.h): decodes a baseline JPEG held in memory into RGB8, rows top to
// bottom. Returns 0, RTW_ERR_BAD_SCENE when the file is refused (message in err, if given), RTW_ERR_OOM when cap is too small.
int rtw_host_decode_jpeg(const void* data, size_t size, int* width, int* height, void* rgb, size_t cap, char* err, size_t err_cap) {
    RTW_HOST_GUARD_BEGIN
    std::vector<uint8_t> out;
    std::string msg;
    int w = 0, h = 0;
    if (!data || !rtwhost::decodeJpeg(static_cast<const uint8_t*>(data), size, w, h, out, msg)) {
        if (err && err_cap) { strncpy(err, msg.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
        return RTW_ERR_BAD_SCENE;
    }
    if (width) *width = w;
    if (height) *height = h;
    if (!rgb || cap < out.size()) return RTW_ERR_OOM;
    memcpy(rgb, out.data(), out.size());
    return RTW_OK;
    RTW_HOST_GUARD_END
}
}
