// Director.cpp — see Director.h. Reference call sites replaced:
//   Director::init          Director.cpp:33-64    -> rtw_create
//   Director::createScene   Director.cpp:951-969  -> ioScene::init + marshalScene + rtw_upload_scene
//   Director::renderFrame   Director.cpp:971-1008 -> rtw_render (optixLaunch + D2H copy; no AI denoiser)
//   Director::printPPM      Director.cpp:1010-1031
//   Director::destroy       Director.cpp:66-104   -> rtw_destroy
#include "Director.h"

#include <cmath>
#include <algorithm>
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
    if (m_devices.empty()) m_devices.assign(1, 0);
    int rc = rtw_create(&m_ctx, static_cast<int>(m_devices.size()), m_devices.data());
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
    m_scene.setCameraKind(m_cameraKind);
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
    p.estimator = m_estimator;
    int rc = rtw_render(m_ctx, &p, m_hostBuffer.data(), &m_stats);
    if (rc != RTW_OK) die(m_ctx, "rtw_render", rc);
    if (m_denoiseIterations > 0) {
        // Director.cpp:986-997: the denoiser pass closes the frame. The reference feeds its LDR model the display-encoded
        // image (raygen.cu:151-155 writes sqrt(colour)); the stand-in filters the same encoding, clamped to [0, 1], and
        // the buffer goes back to linear for the writers.
        std::vector<float> enc(m_hostBuffer.size()), filtered(m_hostBuffer.size());
        for (size_t i = 0; i < enc.size(); i++) {
            float c = m_hostBuffer[i];
            if ((i & 3) != 3) { c = !(c == c) || c < 0.f ? 0.f : (c > 1.f ? 1.f : c); c = std::sqrt(c); }
            enc[i] = c;
        }
        rc = rtw_denoise(m_ctx, enc.data(), filtered.data(), m_Nx, m_Ny, m_denoiseIterations, m_denoiseSigma);
        if (rc != RTW_OK) die(m_ctx, "rtw_denoise", rc);
        for (size_t i = 0; i < enc.size(); i++) m_hostBuffer[i] = (i & 3) != 3 ? filtered[i] * filtered[i] : filtered[i];
    }
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

// PNG without a compression library: zlib stream of stored deflate blocks (the reference vendors stb_image_write.h for
// this and never calls it, main.cpp:9). CRC-32 and Adler-32 as in RFC 1950 / the PNG specification.
bool Director::writePNG(const std::string& path) const {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    uint32_t crcTable[256];
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crcTable[n] = c;
    }
    auto be32 = [](std::vector<unsigned char>& v, uint32_t x) {
        v.push_back(static_cast<unsigned char>(x >> 24)); v.push_back(static_cast<unsigned char>(x >> 16));
        v.push_back(static_cast<unsigned char>(x >> 8)); v.push_back(static_cast<unsigned char>(x));
    };
    auto chunk = [&](const char* type, const std::vector<unsigned char>& data) {
        std::vector<unsigned char> c;
        be32(c, static_cast<uint32_t>(data.size()));
        c.insert(c.end(), type, type + 4);
        c.insert(c.end(), data.begin(), data.end());
        uint32_t crc = 0xffffffffu;
        for (size_t i = 4; i < c.size(); i++) crc = crcTable[(crc ^ c[i]) & 0xffu] ^ (crc >> 8);
        be32(c, crc ^ 0xffffffffu);
        f.write(reinterpret_cast<const char*>(c.data()), static_cast<std::streamsize>(c.size()));
    };
    auto enc = [](float c) {
        float g = std::sqrt(c);
        g = g < 0.f ? 0.f : (g > 1.f ? 1.f : g);
        if (!(g == g)) g = 0.f;
        return static_cast<unsigned char>(static_cast<int>(255.99f * g));
    };
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    f.write(reinterpret_cast<const char*>(sig), 8);
    std::vector<unsigned char> ihdr;
    be32(ihdr, static_cast<uint32_t>(m_Nx)); be32(ihdr, static_cast<uint32_t>(m_Ny));
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8 bit, RGB, no interlace
    chunk("IHDR", ihdr);
    // raw scanlines (filter byte 0 + RGB), top row first
    std::vector<unsigned char> raw;
    raw.reserve((static_cast<size_t>(m_Nx) * 3 + 1) * m_Ny);
    for (int j = m_Ny - 1; j >= 0; j--) {
        raw.push_back(0);
        for (int i = 0; i < m_Nx; i++) {
            const float* px = &m_hostBuffer[(static_cast<size_t>(m_Nx) * j + i) * 4];
            raw.push_back(enc(px[0])); raw.push_back(enc(px[1])); raw.push_back(enc(px[2]));
        }
    }
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
        z.push_back(static_cast<unsigned char>(n)); z.push_back(static_cast<unsigned char>(n >> 8));
        z.push_back(static_cast<unsigned char>(~n)); z.push_back(static_cast<unsigned char>((~n) >> 8));
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        z.insert(z.end(), raw.begin() + static_cast<std::ptrdiff_t>(pos), raw.begin() + static_cast<std::ptrdiff_t>(pos + n));
        pos += n;
        if (n == 0) break;
    }
    be32(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});
    return static_cast<bool>(f);
}
