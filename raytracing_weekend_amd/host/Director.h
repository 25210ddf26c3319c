// Director.h — host driver with the reference's public surface (RestOfLife/Director.h:115-126):
//   init / createScene / renderFrame / printPPM / destroy.
// Everything OptiX inside the reference's Director (context, 26 modules, 36 program groups, pipeline,
// SBT, launch params, denoiser) is replaced by calls into the C ABI of include/rtw.h.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rtw.h"
#include "ioScene.h"

class Director {
public:
    Director(bool verbose, bool debug) : _verbose(verbose), _debug(debug) {}

    void init(unsigned int width, unsigned int height, unsigned int samples);
    void destroy();

    void createScene(unsigned int sceneNumber);
    void renderFrame();
    void printPPM();
    // additions: an 8K frame is ~400 MB as ASCII P3; P6 is 100 MB, PFM keeps the linear floats
    bool writeBinaryPPM(const std::string& path) const;  // P6, same sqrt + 255.99 quantisation as printPPM
    bool writePFM(const std::string& path) const;        // "PF", little-endian, linear radiance, rows bottom-up
    bool writePNG(const std::string& path) const;        // 8-bit RGB, same quantisation; stored (uncompressed) deflate blocks

    // additions over the reference (its depth is hard-wired to 20 at Director.cpp:42, its RNG to tea+lcg)
    void setMaxDepth(int depth) { m_maxRayDepth = depth; }
    void setSeed(uint32_t seed) { m_seed = seed; }
    void setRngKind(int kind) { m_rngKind = kind; }
    void setDevice(int device) { m_devices.assign(1, device); }
    // n GPUs of this node, devices first .. first + n - 1: the frame is split into n interleaved row shards inside the library
    // (rtw_create with n_devices = n) and gathered on the first device; the image does not depend on n
    void setDevices(int n, int first = 0, bool same = false) { m_devices.clear(); for (int i = 0; i < n; i++) m_devices.push_back(same ? first : first + i); }
    void setEstimator(int estimator) { m_estimator = estimator; }  // rtw_estimator
    void setCameraKind(int kind) { m_cameraKind = kind; }          // rtw_camera_type (the reference only ever builds the perspective one)
    // the reference's renderFrame ends with the OptiX AI denoiser (Director.cpp:986-997); iterations > 0 runs the
    // a-trous stand-in (rtw_denoise) on the frame instead
    void setDenoise(int iterations, float sigma) { m_denoiseIterations = iterations; m_denoiseSigma = sigma; }
    const rtw_stats& stats() const { return m_stats; }
    const std::vector<float>& hostBuffer() const { return m_hostBuffer; }  // linear RGBA, row 0 = bottom row

private:
    void marshalAndUpload();  // createSBT + initLaunchParams of the reference

    int m_Nx = 0, m_Ny = 0, m_Ns = 0;
    int m_maxRayDepth = 20;
    uint32_t m_seed = 0x6314759u;
    int m_rngKind = RTW_RNG_PHILOX;
    std::vector<int> m_devices{0};
    int m_estimator = RTW_EST_REFERENCE;
    int m_cameraKind = RTW_CAM_PERSPECTIVE;
    int m_denoiseIterations = 0;
    float m_denoiseSigma = 0.5f;
    rtw_ctx* m_ctx = nullptr;
    rtwhost::ioScene m_scene;
    std::vector<float> m_hostBuffer;
    rtw_stats m_stats{};
    bool _verbose = false;
    bool _debug = false;
};
