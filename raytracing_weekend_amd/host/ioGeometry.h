// ioGeometry.h — host scene-description data model: primitives, textures, materials, instances.
// Same roles and names as the reference's geometry/io*.h, material/io*.h, texture/ioTexture.h,
// but every class only fills the pointer-free records of include/rtw.h; there is no OptiX
// accel build here (the library builds its own BVH in rtw_upload_scene).
#pragma once
#include <memory>
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

// ---------------------------------------------------------------- textures (texture/ioTexture.h:28-86)
struct ioTexture {
    virtual ~ioTexture() {}
    virtual rtw_texture getTexRec() const = 0;
};
struct ioNullTexture : ioTexture {
    rtw_texture getTexRec() const override {
        rtw_texture t{};
        t.type = RTW_TEX_NULL;
        return t;
    }
};
struct ioConstantTexture : ioTexture {
    explicit ioConstantTexture(const Float3& c) : color(c) {}
    rtw_texture getTexRec() const override {
        rtw_texture t{};
        t.type = RTW_TEX_CONSTANT;
        t.color[0] = color.x; t.color[1] = color.y; t.color[2] = color.z;
        return t;
    }
    Float3 color;
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
