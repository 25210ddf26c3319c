// ioGeometry.h — host scene-description data model: primitives, textures, materials, instances.
// Same roles and names as the reference's geometry/io*.h, material/io*.h, texture/ioTexture.h,
// but every class only fills the pointer-free records of include/rtw.h; there is no OptiX
// accel build here (the library builds its own BVH in rtw_upload_scene).
#pragma once
#include <iterator>

#include "JpegDecode.h"
#include <cctype>
#include <cmath>
#include <cstring>
#include <fstream>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtw.h"
#include "ioTransform.h"

namespace rtwhost {

enum Axis { X_AXIS, Y_AXIS, Z_AXIS };

// ---------------------------------------------------------------- geometry
class ioGeometry {
public:
    virtual ~ioGeometry() {}
    // geometry record, the counterpart of HitGroupData filled in Director::createSBT (Director.cpp:670-841)
    virtual rtw_prim record() const = 0;
};

// geometry/ioSphere.h:16-23
class ioSphere : public ioGeometry {
public:
    ioSphere(float x, float y, float z, float r) : cx(x), cy(y), cz(z), r(r) {}
    rtw_prim record() const override {
        rtw_prim p{};
        p.type = RTW_PRIM_SPHERE;
        p.p[0] = cx; p.p[1] = cy; p.p[2] = cz; p.p[3] = r;
        return p;
    }
    float cx, cy, cz, r;
};

// geometry/ioMovingSphere.h:30-44 ; the motion transform of :161-203 is implied by the type
class ioMovingSphere : public ioGeometry {
public:
    ioMovingSphere(float x0, float y0, float z0, float x1, float y1, float z1, float r, float t0 = 0.f, float t1 = 1.f)
        : x0(x0), y0(y0), z0(z0), x1(x1), y1(y1), z1(z1), r(r), t0(t0), t1(t1) {}
    rtw_prim record() const override {
        rtw_prim p{};
        p.type = RTW_PRIM_MOVING_SPHERE;
        p.p[0] = x0; p.p[1] = y0; p.p[2] = z0; p.p[3] = r;
        p.p[4] = x1; p.p[5] = y1; p.p[6] = z1; p.p[7] = t0; p.p[8] = t1;
        return p;
    }
    float x0, y0, z0, x1, y1, z1, r, t0, t1;
};

// geometry/ioAARect.h:17-26
class ioAARect : public ioGeometry {
public:
    ioAARect(float a0, float a1, float b0, float b1, float k, bool flip, Axis kind)
        : a0(a0), a1(a1), b0(b0), b1(b1), k(k), flip(flip), kind(kind) {}
    rtw_prim record() const override {
        rtw_prim p{};
        p.type = kind == X_AXIS ? RTW_PRIM_RECT_X : kind == Y_AXIS ? RTW_PRIM_RECT_Y : RTW_PRIM_RECT_Z;
        p.flip = flip ? 1 : 0;
        p.p[0] = a0; p.p[1] = a1; p.p[2] = b0; p.p[3] = b1; p.p[4] = k;
        return p;
    }
    float a0, a1, b0, b1, k;
    bool flip;
    Axis kind;
};

// geometry/ioVolumeBox.h:20-24
class ioVolumeBox : public ioGeometry {
public:
    ioVolumeBox(const Float3& p0, const Float3& p1, float density) : mn(p0), mx(p1), density(density) {}
    rtw_prim record() const override {
        rtw_prim p{};
        p.type = RTW_PRIM_VOLUME_BOX;
        p.p[0] = mn.x; p.p[1] = mn.y; p.p[2] = mn.z;
        p.p[3] = mx.x; p.p[4] = mx.y; p.p[5] = mx.z;
        p.p[6] = density;
        return p;
    }
    Float3 mn, mx;
    float density;
};

// geometry/ioVolumeSphere.h:23-24
class ioVolumeSphere : public ioGeometry {
public:
    ioVolumeSphere(float x, float y, float z, float r, float density) : cx(x), cy(y), cz(z), r(r), density(density) {}
    rtw_prim record() const override {
        rtw_prim p{};
        p.type = RTW_PRIM_VOLUME_SPHERE;
        p.p[0] = cx; p.p[1] = cy; p.p[2] = cz; p.p[3] = r; p.p[4] = density;
        return p;
    }
    float cx, cy, cz, r, density;
};

// ---------------------------------------------------------------- textures (texture/ioTexture.h:28-338)
// What getTexRec() hands to the device in the reference (a textureParam with device pointers) becomes a record in
// textures[] plus, for noise and image textures, words in the blob's texture data section.
struct TexEmit {
    std::vector<rtw_texture> texs;
    std::vector<uint32_t> data;
    // ioTexture.h:21-26 localRnd(): one mt19937(0) shared by every noise texture of the scene, drawn in the order
    // the material list reaches them. std::uniform_real_distribution<float> is not specified bit for bit across
    // standard libraries: this is libstdc++'s, the reference was only ever built with MSVC's (tables unpinned).
    std::mt19937 gen{0};
    std::uniform_real_distribution<float> dis{0.f, 1.f};
    float localRnd() { return dis(gen); }
};

struct ioTexture {
    virtual ~ioTexture() {}
    // appends this texture's record (and whatever it refers to) and returns the record's index in e.texs
    virtual int32_t emit(TexEmit& e) const = 0;
};
struct ioNullTexture : ioTexture {
    int32_t emit(TexEmit& e) const override {
        rtw_texture t{};
        t.type = RTW_TEX_NULL;
        e.texs.push_back(t);
        return static_cast<int32_t>(e.texs.size()) - 1;
    }
};
struct ioConstantTexture : ioTexture {
    explicit ioConstantTexture(const Float3& c) : color(c) {}
    int32_t emit(TexEmit& e) const override {
        rtw_texture t{};
        t.type = RTW_TEX_CONSTANT;
        t.color[0] = color.x; t.color[1] = color.y; t.color[2] = color.z;
        e.texs.push_back(t);
        return static_cast<int32_t>(e.texs.size()) - 1;
    }
    Float3 color;
};
// ioTexture.h:88-116. The reference stores the children's callable ids (par.odd = odd->texIdx), so its checker
// only ever evaluates constant colours of an empty record; here the children are emitted and evaluated.
struct ioCheckerTexture : ioTexture {
    ioCheckerTexture(const ioTexture* o, const ioTexture* e) : odd(o), even(e) {}
    int32_t emit(TexEmit& e) const override {
        rtw_texture t{};
        t.type = RTW_TEX_CHECKER;
        t.odd = odd->emit(e);
        t.even = even->emit(e);
        e.texs.push_back(t);
        return static_cast<int32_t>(e.texs.size()) - 1;
    }
    const ioTexture* odd;
    const ioTexture* even;
};
// ioTexture.h:118-222: Perlin gradient table and three permutations, generated when the record is requested
struct ioNoiseTexture : ioTexture {
    explicit ioNoiseTexture(float s) : scale(s) {}
    int32_t emit(TexEmit& e) const override {
        rtw_texture t{};
        t.type = RTW_TEX_NOISE;
        t.scale = scale;
        t.data = static_cast<uint32_t>(e.data.size());
        auto putf = [&](float f) { uint32_t u; memcpy(&u, &f, 4); e.data.push_back(u); };
        for (int i = 0; i < 256; i++) {
            // unit_float3(-1 + 2*localRnd(), -1 + 2*localRnd(), -1 + 2*localRnd()): argument order as MSVC (Q6)
            float z = -1 + 2 * e.localRnd();
            float y = -1 + 2 * e.localRnd();
            float x = -1 + 2 * e.localRnd();
            float l = sqrtf(x * x + y * y + z * z);
            putf(x / l); putf(y / l); putf(z / l);
        }
        for (int k = 0; k < 3; k++) {  // perm_x, perm_y, perm_z (ioTexture.h:128-149)
            int p[256];
            for (int i = 0; i < 256; i++) p[i] = i;
            for (int i = 256 - 1; i > 0; i--) {
                int target = int(e.localRnd() * (i + 1));
                int tmp = p[i]; p[i] = p[target]; p[target] = tmp;
            }
            for (int i = 0; i < 256; i++) e.data.push_back(static_cast<uint32_t>(p[i]));
        }
        e.texs.push_back(t);
        return static_cast<int32_t>(e.texs.size()) - 1;
    }
    float scale;
};
// ioTexture.h:225-338. The reference decodes assets/earthmap.jpg with stb_image; this host reads baseline JPEG with its
// own decoder (JpegDecode.h) and binary or ASCII PPM (P6 / P3, 8 bit). Rows are flipped like ioTexture.h:247-250, alpha is 255.
struct ioImageTexture : ioTexture {
    explicit ioImageTexture(const std::string& fileName) { load(fileName); }
    int32_t emit(TexEmit& e) const override {
        rtw_texture t{};
        t.type = RTW_TEX_IMAGE;
        t.data = static_cast<uint32_t>(e.data.size());
        e.data.push_back(static_cast<uint32_t>(nx));
        e.data.push_back(static_cast<uint32_t>(ny));
        e.data.insert(e.data.end(), texels.begin(), texels.end());
        e.texs.push_back(t);
        return static_cast<int32_t>(e.texs.size()) - 1;
    }
    int nx = 0, ny = 0;
    std::vector<uint32_t> texels;

private:
    void load(const std::string& fileName) {
        std::ifstream f(fileName, std::ios::binary);
        if (!f) throw std::runtime_error("image texture: cannot open " + fileName);
        std::vector<unsigned char> rgb;
        if (f.peek() == 0xFF) {  // JPEG (SOI = FF D8)
            std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            std::string err;
            if (!decodeJpeg(file.data(), file.size(), nx, ny, rgb, err)) throw std::runtime_error("image texture: " + fileName + ": " + err);
            pack(rgb);
            return;
        }
        auto token = [&]() {  // PNM header token, '#' comments skipped
            std::string tok;
            int ch;
            while ((ch = f.get()) != EOF) {
                if (ch == '#') { while ((ch = f.get()) != EOF && ch != '\n') {} continue; }
                if (isspace(ch)) { if (!tok.empty()) break; continue; }
                tok.push_back(static_cast<char>(ch));
            }
            return tok;
        };
        const std::string magic = token();
        if (magic != "P6" && magic != "P3") throw std::runtime_error("image texture: " + fileName + " is not a P6/P3 PPM");
        int maxv = 0;
        try { nx = std::stoi(token()); ny = std::stoi(token()); maxv = std::stoi(token()); }
        catch (const std::exception&) { throw std::runtime_error("image texture: bad PPM header in " + fileName); }
        if (nx <= 0 || ny <= 0 || nx > 32768 || ny > 32768 || maxv != 255) throw std::runtime_error("image texture: unsupported PPM (size or maxval) " + fileName);
        rgb.resize(static_cast<size_t>(nx) * ny * 3);
        if (magic == "P6") {
            f.read(reinterpret_cast<char*>(rgb.data()), static_cast<std::streamsize>(rgb.size()));
            if (static_cast<size_t>(f.gcount()) != rgb.size()) throw std::runtime_error("image texture: truncated PPM " + fileName);
        } else {
            for (size_t i = 0; i < rgb.size(); i++) {
                const std::string t = token();
                if (t.empty()) throw std::runtime_error("image texture: truncated PPM " + fileName);
                rgb[i] = static_cast<unsigned char>(std::stoi(t));
            }
        }
        pack(rgb);
    }
    void pack(const std::vector<unsigned char>& rgb) {  // ioTexture.h:244-262
        texels.resize(static_cast<size_t>(nx) * ny);
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < ny; ++j) {
                const size_t bindex = static_cast<size_t>(j) * nx + i;
                const size_t iindex = (static_cast<size_t>(ny - j - 1) * nx + i) * 3;
                texels[bindex] = rgb[iindex] | (uint32_t(rgb[iindex + 1]) << 8) | (uint32_t(rgb[iindex + 2]) << 16) | (255u << 24);
            }
    }
};

// ---------------------------------------------------------------- materials (material/io*Material.h)
class ioMaterial {
public:
    virtual ~ioMaterial() {}
    // fills the material record; returns the texture to attach (or nullptr)
    virtual const ioTexture* assignTo(rtw_material& m) const = 0;
};
class ioLambertianMaterial : public ioMaterial {
public:
    explicit ioLambertianMaterial(const ioTexture* t) : texture(t) {}
    const ioTexture* assignTo(rtw_material& m) const override {
        m.type = RTW_MAT_LAMBERTIAN; m.bsdf_eval = 0; m.fuzz_or_eta = 0.f;
        return texture;
    }
private:
    const ioTexture* texture;
};
class ioMetalMaterial : public ioMaterial {
public:
    ioMetalMaterial(const ioTexture* t, float fuzz) : texture(t), fuzz(fuzz) {}
    const ioTexture* assignTo(rtw_material& m) const override {
        m.type = RTW_MAT_METAL; m.bsdf_eval = 2;
        m.fuzz_or_eta = fuzz < 1.f ? fuzz : 1.f;  // ioMetalMaterial.h:34-38
        return texture;
    }
private:
    const ioTexture* texture;
    float fuzz;
};
class ioDielectricMaterial : public ioMaterial {
public:
    explicit ioDielectricMaterial(float eta) : eta(eta) {}
    const ioTexture* assignTo(rtw_material& m) const override {
        m.type = RTW_MAT_DIELECTRIC; m.bsdf_eval = 1; m.fuzz_or_eta = eta;
        return nullptr;
    }
private:
    float eta;
};
class ioDiffuseLightMaterial : public ioMaterial {
public:
    explicit ioDiffuseLightMaterial(const ioTexture* t) : texture(t) {}
    const ioTexture* assignTo(rtw_material& m) const override {
        m.type = RTW_MAT_DIFFUSE_LIGHT; m.bsdf_eval = -1; m.fuzz_or_eta = 0.f;
        return texture;
    }
private:
    const ioTexture* texture;
};
class ioIsotropicMaterial : public ioMaterial {
public:
    explicit ioIsotropicMaterial(const ioTexture* t) : texture(t) {}
    const ioTexture* assignTo(rtw_material& m) const override {
        m.type = RTW_MAT_ISOTROPIC; m.bsdf_eval = -1; m.fuzz_or_eta = 0.f;
        return texture;
    }
private:
    const ioTexture* texture;
};
class ioNormalMaterial : public ioMaterial {
public:
    const ioTexture* assignTo(rtw_material& m) const override {
        m.type = RTW_MAT_NORMAL; m.bsdf_eval = -1; m.fuzz_or_eta = 0.f;
        return nullptr;
    }
};

// ---------------------------------------------------------------- instances (geometry/ioGeometryInstance.h:20-26)
// instanceId selects the material (closehit.cu:50,63), sbtOffset selects the geometry record.
struct ioGeometryInstance {
    unsigned int instanceId = 0;
    unsigned int sbtOffset = 0;
    std::array<float, 12> transform{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
    bool identity = true;
    void init(unsigned int id, unsigned int sbtidx) { instanceId = id; sbtOffset = sbtidx; }
    void setTransform(const Mat4& m) { transform = m.rows3x4(); identity = false; inv = m.inverse3x4(); }
    std::array<float, 12> inv{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
};

// geometry/ioGeometryGroup.h:27-40 — a box is six axis-aligned rectangles
struct ioGeometryGroup {
    static void createBox(const Float3& p0, const Float3& p1, std::vector<std::unique_ptr<ioGeometry>>& out) {
        out.emplace_back(new ioAARect(p0.x, p1.x, p0.y, p1.y, p0.z, true, Z_AXIS));
        out.emplace_back(new ioAARect(p0.x, p1.x, p0.y, p1.y, p1.z, false, Z_AXIS));
        out.emplace_back(new ioAARect(p0.x, p1.x, p0.z, p1.z, p0.y, true, Y_AXIS));
        out.emplace_back(new ioAARect(p0.x, p1.x, p0.z, p1.z, p1.y, false, Y_AXIS));
        out.emplace_back(new ioAARect(p0.y, p1.y, p0.z, p1.z, p0.x, true, X_AXIS));
        out.emplace_back(new ioAARect(p0.y, p1.y, p0.z, p1.z, p1.x, false, X_AXIS));
    }
};

}  // namespace rtwhost
