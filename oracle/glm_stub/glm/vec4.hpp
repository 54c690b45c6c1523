// TEST INFRASTRUCTURE, build-container only (oracle/crosscheck_spline.py): the part of glm::vec4 that the
// reference's src/CubicSpline.cpp uses -- component-wise IEEE fp32 arithmetic, nothing with a hidden
// operation order -- so that file compiles VERBATIM where it lies (glm is neither vendored nor installed).
#pragma once
namespace glm {
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    explicit vec4(float v) : x(v), y(v), z(v), w(v) {}
    vec4(double v) : x((float)v), y((float)v), z((float)v), w((float)v) {}      // `glm::vec4 vec(1.0);`
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
};
inline vec4 operator+(const vec4 &a, const vec4 &b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4 &a, const vec4 &b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 operator*(const vec4 &a, const vec4 &b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator/(const vec4 &a, const vec4 &b) { return vec4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
inline vec4 operator*(float s, const vec4 &a) { return vec4(s * a.x, s * a.y, s * a.z, s * a.w); }
inline vec4 operator*(const vec4 &a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
}
