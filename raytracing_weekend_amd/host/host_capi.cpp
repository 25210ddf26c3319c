// host_capi.cpp — C entry points over the C++ host scene description, so that the Python
// plumbing (bench.py, tests) can obtain the same scene blobs the Director uploads.
// No GPU dependency: librtw_host.so loads on CPU-only machines.
#include <cstring>

#include "SceneMarshal.h"

extern "C" {

// Builds reference scene `scene` (0 Cornell box, 1 moving spheres, 3 Cornell box with volumes) for an
// Nx x Ny image and writes the blob to buf (capacity cap). *needed receives the blob size.
// Returns 0, RTW_ERR_INVALID_ARG for an unknown scene (ioScene::init -> 1), or RTW_ERR_OOM when cap is too small.
int rtw_host_build_scene(int scene, int nx, int ny, void* buf, size_t cap, size_t* needed) {
    if (nx <= 0 || ny <= 0) return RTW_ERR_INVALID_ARG;
    rtwhost::ioScene sc;
    if (sc.init(nx, ny, 1, 20, scene)) return RTW_ERR_INVALID_ARG;
    std::vector<uint8_t> blob = rtwhost::marshalScene(sc);
    if (needed) *needed = blob.size();
    if (!buf || cap < blob.size()) return RTW_ERR_OOM;
    memcpy(buf, blob.data(), blob.size());
    return RTW_OK;
}
}
