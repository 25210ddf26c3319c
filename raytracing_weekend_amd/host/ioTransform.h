// ioTransform.h — host-side 4x4 row-major transforms for scene description.
// Mirrors the reference's ioTransform utility surface (geometry/ioTransform.h:15-131:
// translate / rotateX / rotateY / rotateZ / scale, angles in degrees) without sutil::Matrix4x4.
#pragma once
#include <array>
#include <cmath>

namespace rtwhost {

constexpr float kPiF = 3.14159265358979323846f;

struct Mat4 {
    float a[16];

    static Mat4 identity() {
        Mat4 r{};
        r.a[0] = r.a[5] = r.a[10] = r.a[15] = 1.0f;
        return r;
    }
    const float* getData() const { return a; }

    // float arithmetic, row times column, like sutil::Matrix<4,4>::operator*
    Mat4 operator*(const Mat4& b) const {
        Mat4 r{};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0.0f;
                for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b.a[4 * k + j];
                r.a[4 * i + j] = s;
            }
        return r;
    }
    Mat4& operator*=(const Mat4& b) { *this = *this * b; return *this; }

    // first three rows (OptixInstance::transform layout, ioGeometryInstance.h:84-88)
    std::array<float, 12> rows3x4() const {
        std::array<float, 12> r{};
        for (int i = 0; i < 12; ++i) r[i] = a[i];
        return r;
    }

    // affine inverse (rotation/scale + translation), computed in double then rounded once.
    // OptiX derives the world->object matrix itself (closed); this is the build's definition.
    std::array<float, 12> inverse3x4() const {
        double m[3][3], t[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) m[i][j] = a[4 * i + j];
            t[i] = a[4 * i + 3];
        }
        double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
        double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
        double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
        double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
        double id = 1.0 / det;
        double inv[3][3];
        inv[0][0] = c00 * id;
        inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
        inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
        inv[1][0] = c01 * id;
        inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
        inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
        inv[2][0] = c02 * id;
        inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
        inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
        std::array<float, 12> r{};
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r[4 * i + j] = static_cast<float>(inv[i][j]);
            r[4 * i + 3] = static_cast<float>(-(inv[i][0] * t[0] + inv[i][1] * t[1] + inv[i][2] * t[2]));
        }
        return r;
    }
};

struct Float3 {
    float x, y, z;
};
inline Float3 make_float3(float x, float y, float z) { return Float3{x, y, z}; }
inline Float3 make_float3(float v) { return Float3{v, v, v}; }

class ioTransform {
public:
    static Mat4 translate(const Float3& o) {
        Mat4 m = Mat4::identity();
        m.a[3] = o.x; m.a[7] = o.y; m.a[11] = o.z;
        return m;
    }
    static Mat4 rotateX(float deg) {
        float r = deg * kPiF / 180.f, c = cosf(r), s = sinf(r);
        Mat4 m = Mat4::identity();
        m.a[5] = c; m.a[6] = -s; m.a[9] = s; m.a[10] = c;
        return m;
    }
    static Mat4 rotateY(float deg) {
        float r = deg * kPiF / 180.f, c = cosf(r), s = sinf(r);
        Mat4 m = Mat4::identity();
        m.a[0] = c; m.a[2] = s; m.a[8] = -s; m.a[10] = c;
        return m;
    }
    static Mat4 rotateZ(float deg) {
        float r = deg * kPiF / 180.f, c = cosf(r), s = sinf(r);
        Mat4 m = Mat4::identity();
        m.a[0] = c; m.a[1] = -s; m.a[4] = s; m.a[5] = c;
        return m;
    }
    static Mat4 scale(const Float3& k) {
        Mat4 m = Mat4::identity();
        m.a[0] = k.x; m.a[5] = k.y; m.a[10] = k.z;
        return m;
    }
};

}  // namespace rtwhost
