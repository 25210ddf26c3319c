// Director.cpp — see Director.h. Reference call sites replaced:
//   Director::init          Director.cpp:33-64    -> rtw_create
//   Director::createScene   Director.cpp:951-969  -> ioScene::init + marshalScene + rtw_upload_scene
//   Director::renderFrame   Director.cpp:971-1008 -> rtw_render (optixLaunch + D2H copy; no AI denoiser)
//   Director::printPPM      Director.cpp:1010-1031
//   Director::destroy       Director.cpp:66-104   -> rtw_destroy
#include "Director.h"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "SceneMarshal.h"

namespace {
[[noreturn]] void die(rtw_ctx* ctx, const char* what, int rc) {
    std::cerr << "ERROR: " << what << " failed (" << rc << "): " << (ctx ? rtw_last_error(ctx) : "") << std::endl;
    std::exit(EXIT_FAILURE);  // the reference lets OPTIX_CHECK/CUDA_CHECK exceptions terminate the process
}
}  // namespace

void Director::init(unsigned int width, unsigned int height, unsigned int samples) {
    m_Nx = static_cast<int>(width);
    m_Ny = static_cast<int>(height);
    m_Ns = static_cast<int>(samples);
    int dev = m_device;
    int rc = rtw_create(&m_ctx, 1, &dev);
    if (rc != RTW_OK) die(nullptr, "rtw_create", rc);
    m_hostBuffer.assign(static_cast<size_t>(m_Nx) * m_Ny * 4, 0.f);
}

void Director::destroy() {
    m_scene.destroy();
    if (m_ctx) rtw_destroy(m_ctx);
    m_ctx = nullptr;
    m_hostBuffer.clear();
}

void Director::createScene(unsigned int sceneNumber) {
    int error = m_scene.init(m_Nx, m_Ny, m_Ns, m_maxRayDepth, static_cast<int>(sceneNumber));
    if (error) std::exit(EXIT_FAILURE);  // Director.cpp:954-958
    marshalAndUpload();
    if (_verbose) std::cerr << "INFO: Scene description: " << m_scene.getDescription() << std::endl;
}

void Director::marshalAndUpload() {
    std::vector<uint8_t> blob = rtwhost::marshalScene(m_scene);
    int rc = rtw_upload_scene(m_ctx, blob.data(), blob.size());
    if (rc != RTW_OK) die(m_ctx, "rtw_upload_scene", rc);
}

void Director::renderFrame() {
    rtw_params p{};
    p.width = m_Nx;
    p.height = m_Ny;
    p.spp = m_Ns;
    p.max_depth = m_maxRayDepth;
    p.seed = m_seed;
    p.row0 = 0;
    p.row1 = m_Ny;
    p.rng_kind = m_rngKind;
    int rc = rtw_render(m_ctx, &p, m_hostBuffer.data(), &m_stats);
    if (rc != RTW_OK) die(m_ctx, "rtw_render", rc);
    if (_verbose) {
        const double s = m_stats.seconds > 0 ? m_stats.seconds : 1e-9;
        std::cerr << "INFO: " << m_stats.samples << " samples, " << m_stats.segments << " segments, " << m_stats.shadow_rays
                  << " shadow rays in " << m_stats.seconds << " s on the GPU = " << m_stats.samples / s / 1e6 << " Msamples/s, "
                  << m_stats.algorithmic_bytes / s / 1e9 << " GB/s algorithmic" << std::endl;
    }
}

// P3 ASCII PPM on stdout, rows top to bottom, gamma 2 then int(255.99*clamp) — Director.cpp:1010-1031.
// The reference applies sqrt on the device (raygen.cu:151-155); rtw_render returns linear radiance,
// so the square root is taken here.
void Director::printPPM() {
    std::cout << "P3\n" << m_Nx << " " << m_Ny << "\n255\n";
    auto enc = [](float c) {
        float g = std::sqrt(c);
        g = g < 0.f ? 0.f : (g > 1.f ? 1.f : g);
        if (!(g == g)) g = 0.f;
        return static_cast<int>(255.99f * g);
    };
    std::string line;
    for (int j = m_Ny - 1; j >= 0; j--) {
        line.clear();
        for (int i = 0; i < m_Nx; i++) {
            const float* px = &m_hostBuffer[(static_cast<size_t>(m_Nx) * j + i) * 4];
            line += std::to_string(enc(px[0]));
            line += ' ';
            line += std::to_string(enc(px[1]));
            line += ' ';
            line += std::to_string(enc(px[2]));
            line += '\n';
        }
        std::cout << line;
    }
}

bool Director::writeBinaryPPM(const std::string& path) const {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "P6\n" << m_Nx << " " << m_Ny << "\n255\n";
    auto enc = [](float c) {
        float g = std::sqrt(c);
        g = g < 0.f ? 0.f : (g > 1.f ? 1.f : g);
        if (!(g == g)) g = 0.f;
        return static_cast<unsigned char>(static_cast<int>(255.99f * g));
    };
    std::vector<unsigned char> row(static_cast<size_t>(m_Nx) * 3);
    for (int j = m_Ny - 1; j >= 0; j--) {  // the buffer is bottom-up like the reference's, image files are top-down
        for (int i = 0; i < m_Nx; i++) {
            const float* px = &m_hostBuffer[(static_cast<size_t>(m_Nx) * j + i) * 4];
            row[3 * i] = enc(px[0]); row[3 * i + 1] = enc(px[1]); row[3 * i + 2] = enc(px[2]);
        }
        f.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size()));
    }
    return static_cast<bool>(f);
}

bool Director::writePFM(const std::string& path) const {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "PF\n" << m_Nx << " " << m_Ny << "\n-1.0\n";  // negative scale = little-endian; PFM rows run bottom-up
    std::vector<float> row(static_cast<size_t>(m_Nx) * 3);
    for (int j = 0; j < m_Ny; j++) {
        for (int i = 0; i < m_Nx; i++) {
            const float* px = &m_hostBuffer[(static_cast<size_t>(m_Nx) * j + i) * 4];
            row[3 * i] = px[0]; row[3 * i + 1] = px[1]; row[3 * i + 2] = px[2];
        }
        f.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size() * sizeof(float)));
    }
    return static_cast<bool>(f);
}
