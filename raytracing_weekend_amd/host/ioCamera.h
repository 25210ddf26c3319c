// ioCamera.h — host camera description. Mirrors scene/ioCamera.h:10-179 of the reference: ioCamera base,
// ioPerspectiveCamera (the only one the reference instantiates), ioEnvironmentCamera and ioOrthographicCamera
// (defined there, never built; their ray generation is scene/camera.cuh:35-56). All math in float, like the reference.
#pragma once
#include <cmath>

#include "../../include/rtw.h"
#include "ioTransform.h"

namespace rtwhost {

inline Float3 operator-(const Float3& a, const Float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Float3 operator+(const Float3& a, const Float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Float3 operator*(float s, const Float3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline float dot(const Float3& a, const Float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float3 cross(const Float3& a, const Float3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline Float3 normalize(const Float3& v) {
    float inv = 1.0f / sqrtf(dot(v, v));
    return inv * v;
}
inline float length(const Float3& v) { return sqrtf(dot(v, v)); }

class ioCamera {
public:
    ioCamera(float fromX, float fromY, float fromZ, float toX, float toY, float toZ, float upX, float upY, float upZ,
             float t0 = 0.f, float t1 = 0.f) {
        m_origin = make_float3(fromX, fromY, fromZ);
        m_w = normalize(make_float3(fromX - toX, fromY - toY, fromZ - toZ));  // lookfrom - lookat
        m_u = normalize(cross(make_float3(upX, upY, upZ), m_w));
        m_v = cross(m_w, m_u);
        m_time0 = t0;
        m_time1 = t1;
    }
    virtual ~ioCamera() {}
    // what Director::initLaunchParams would put into SysParamter (sysparameter.h:44-54) + the camera kind
    virtual int cameraType() const = 0;
    virtual void getfrustum(Float3& pos, Float3& u, Float3& v, Float3& w, Float3& leftCorner, Float3& horizontal, Float3& vertical) const = 0;

    Float3 m_origin, m_u, m_v, m_w;
    float m_time0, m_time1;
    float m_lensRadius = 0.f;
};

class ioPerspectiveCamera : public ioCamera {
public:
    ioPerspectiveCamera(float fromX, float fromY, float fromZ, float toX, float toY, float toZ, float upX, float upY,
                        float upZ, float vFov, float aspect, float aperture, float focus_dist, float t0 = 0.f,
                        float t1 = 0.f)
        : ioCamera(fromX, fromY, fromZ, toX, toY, toZ, upX, upY, upZ, t0, t1) {
        m_lensRadius = aperture / 2.0f;
        float theta = vFov * kPiF / 180.0f;  // vFov is top to bottom in degrees
        float halfHeight = tanf(theta / 2.0f);
        float halfWidth = aspect * halfHeight;
        m_lowerLeftCorner = m_origin - (halfWidth * focus_dist) * m_u - (halfHeight * focus_dist) * m_v - focus_dist * m_w;
        m_horizontal = (2.0f * halfWidth * focus_dist) * m_u;
        m_vertical = (2.0f * halfHeight * focus_dist) * m_v;
    }

    int cameraType() const override { return RTW_CAM_PERSPECTIVE; }
    void getfrustum(Float3& pos, Float3& u, Float3& v, Float3& w, Float3& leftCorner, Float3& horizontal,
                    Float3& vertical) const override {
        pos = m_origin; u = m_u; v = m_v; w = m_w;
        leftCorner = m_lowerLeftCorner; horizontal = m_horizontal; vertical = m_vertical;
    }

private:
    Float3 m_lowerLeftCorner, m_horizontal, m_vertical;
};

// scene/ioCamera.h:118-139: position and frame only; rays cover the whole sphere of directions (camera.cuh:35-47)
class ioEnvironmentCamera : public ioCamera {
public:
    ioEnvironmentCamera(float fromX, float fromY, float fromZ, float toX, float toY, float toZ, float upX, float upY, float upZ,
                        float t0 = 0.f, float t1 = 0.f)
        : ioCamera(fromX, fromY, fromZ, toX, toY, toZ, upX, upY, upZ, t0, t1) {}
    int cameraType() const override { return RTW_CAM_ENVIRONMENT; }
    void getfrustum(Float3& pos, Float3& u, Float3& v, Float3& w, Float3& leftCorner, Float3& horizontal, Float3& vertical) const override {
        pos = m_origin; u = m_u; v = m_v; w = m_w;
        leftCorner = make_float3(0.f, 0.f, 0.f); horizontal = leftCorner; vertical = leftCorner;
    }
};

// scene/ioCamera.h:141-178: a height x width window through the camera position; parallel rays along -w (camera.cuh:49-54)
class ioOrthographicCamera : public ioCamera {
public:
    ioOrthographicCamera(float fromX, float fromY, float fromZ, float toX, float toY, float toZ, float upX, float upY, float upZ,
                         float height, float width, float t0 = 0.f, float t1 = 0.f)
        : ioCamera(fromX, fromY, fromZ, toX, toY, toZ, upX, upY, upZ, t0, t1) {
        float halfHeight = height / 2.0f;
        float halfWidth = width / 2.0f;
        m_lowerLeftCorner = m_origin - halfWidth * m_u - halfHeight * m_v - m_w;
        m_horizontal = (2.0f * halfWidth) * m_u;
        m_vertical = (2.0f * halfHeight) * m_v;
    }
    int cameraType() const override { return RTW_CAM_ORTHOGRAPHIC; }
    void getfrustum(Float3& pos, Float3& u, Float3& v, Float3& w, Float3& leftCorner, Float3& horizontal, Float3& vertical) const override {
        pos = m_origin; u = m_u; v = m_v; w = m_w;
        leftCorner = m_lowerLeftCorner; horizontal = m_horizontal; vertical = m_vertical;
    }
    Float3 m_lowerLeftCorner, m_horizontal, m_vertical;
};

}  // namespace rtwhost
