// glsl_shim.h -- TEST INFRASTRUCTURE, build-container only (oracle/crosscheck_glsl.py).
//
// Just enough of GLSL 4.30 as C++20 that the TEXT of the reference's compute shader
// (/root/reference/VolumeRenderer.cs, read where it lies, never copied into this repository)
// compiles and runs on the CPU: vec/ivec/uvec/mat4 with the swizzles the shader uses, the
// built-ins it calls, an image and an integer 3-D texture.  Purpose: an independent reading of
// the shader's control flow and operation order to diff against oracle/vr_oracle.c -- it removes
// the risk of a transcription error shared by the oracle and the kernels.  It does NOT pin the
// oracle in the grading sense (this header is a stand-in for the GL driver).
//
// Built-ins follow the GLSL 4.30 specification text, one correctly rounded fp32 operation per
// operation (compile with -ffp-contract=off, no fast-math), the same choices vr_oracle.c documents:
//   dot / length:  ((x*x + y*y) + z*z) + w*w, sqrt;   normalize(v) = v / length(v)
//   min(x,y) = y < x ? y : x;  max(x,y) = x < y ? y : x;  clamp = min(max(x,lo),hi)
//   mat4 * vec4 = ((c0*x + c1*y) + c2*z) + c3*w  (column major)
//   texture() on the integer volume: NEAREST, clamp-to-edge: i = clamp(floor(u*N), 0, N-1)
//   (the reference sets GL_LINEAR on an integer texture; nearest is the effective filter)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace glsl {

struct vec3;
struct vec4;
struct ivec3;

// ---- swizzle proxies: views of the owning vector's storage (GCC union punning)
template <typename V3, typename T, int A, int B, int C>
struct Swz3 {
    T s[4];
    operator V3() const { return V3(s[A], s[B], s[C]); }
    Swz3 &operator=(const V3 &o) { T t0 = o.x, t1 = o.y, t2 = o.z; s[A] = t0; s[B] = t1; s[C] = t2; return *this; }
    Swz3 &operator*=(T f) { s[A] *= f; s[B] *= f; s[C] *= f; return *this; }
};
template <typename V2, typename T, int A, int B>
struct Swz2 {
    T s[4];
    operator V2() const { return V2(s[A], s[B]); }
};

struct ivec2 {
    int x, y;
    ivec2() : x(0), y(0) {}
    ivec2(int a, int b) : x(a), y(b) {}
    template <typename V2, typename T, int A, int B> explicit ivec2(const Swz2<V2, T, A, B> &s) : x((int)s.s[A]), y((int)s.s[B]) {}
};
struct uvec2 { unsigned x, y; uvec2(unsigned a = 0, unsigned b = 0) : x(a), y(b) {} };

struct vec3 {
    union {
        struct { float x, y, z; };
        Swz3<vec3, float, 0, 1, 2> xyz;
        Swz3<vec3, float, 0, 2, 1> xzy;
    };
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
    vec3(const vec3 &o) : x(o.x), y(o.y), z(o.z) {}
    vec3 &operator=(const vec3 &o) { x = o.x; y = o.y; z = o.z; return *this; }
};
struct ivec3 {
    union {
        struct { int x, y, z; };
        Swz3<ivec3, int, 0, 1, 2> xyz;
        Swz3<ivec3, int, 0, 2, 1> xzy;
    };
    ivec3() : x(0), y(0), z(0) {}
    ivec3(int a, int b, int c) : x(a), y(b), z(c) {}
    ivec3(const ivec3 &o) : x(o.x), y(o.y), z(o.z) {}
    ivec3 &operator=(const ivec3 &o) { x = o.x; y = o.y; z = o.z; return *this; }
    operator vec3() const { return vec3((float)x, (float)y, (float)z); }   // GLSL's implicit ivec3 -> vec3
};
struct uvec3 {
    union {
        struct { unsigned x, y, z; };
        Swz2<uvec2, unsigned, 0, 1> xy;
    };
    uvec3(unsigned a = 0, unsigned b = 0, unsigned c = 0) : x(a), y(b), z(c) {}
};
struct uvec4 {
    union {
        struct { unsigned x, y, z, w; };
        struct { unsigned r, g, b, a; };
    };
    uvec4() : x(0), y(0), z(0), w(0) {}
    explicit uvec4(unsigned v) : x(v), y(v), z(v), w(v) {}
    uvec4(unsigned a_, unsigned b_, unsigned c_, unsigned d_) : x(a_), y(b_), z(c_), w(d_) {}
};
struct bvec3 { bool x, y, z; };

struct vec4 {
    union {
        struct { float x, y, z, w; };
        struct { float r, g, b, a; };
        Swz3<vec3, float, 0, 1, 2> xyz;
        Swz3<vec3, float, 0, 1, 2> rgb;
    };
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a_, float b_, float c_, float d_) : x(a_), y(b_), z(c_), w(d_) {}
    explicit vec4(float v) : x(v), y(v), z(v), w(v) {}
    explicit vec4(int v) : x((float)v), y((float)v), z((float)v), w((float)v) {}
    explicit vec4(unsigned v) : x((float)v), y((float)v), z((float)v), w((float)v) {}
    explicit vec4(double v) : x((float)v), y((float)v), z((float)v), w((float)v) {}
    vec4(const vec3 &v, float d_) : x(v.x), y(v.y), z(v.z), w(d_) {}
    vec4(const ivec3 &v, float d_) : x((float)v.x), y((float)v.y), z((float)v.z), w(d_) {}
    template <typename V3, typename T, int A, int B, int C>
    vec4(const Swz3<V3, T, A, B, C> &s, float d_) : x((float)s.s[A]), y((float)s.s[B]), z((float)s.s[C]), w(d_) {}
    vec4(const vec4 &o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
    vec4 &operator=(const vec4 &o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
    vec4 &operator+=(const vec4 &o) { x += o.x; y += o.y; z += o.z; w += o.w; return *this; }
    vec4 &operator-=(const vec4 &o) { x -= o.x; y -= o.y; z -= o.z; w -= o.w; return *this; }
    vec4 &operator*=(const vec4 &o) { x *= o.x; y *= o.y; z *= o.z; w *= o.w; return *this; }
    vec4 &operator/=(const vec4 &o) { x /= o.x; y /= o.y; z /= o.z; w /= o.w; return *this; }
    vec4 &operator*=(float f) { x *= f; y *= f; z *= f; w *= f; return *this; }
    vec4 &operator/=(float f) { x /= f; y /= f; z /= f; w /= f; return *this; }
    vec4 &operator/=(int i) { return *this /= (float)i; }
};

inline vec4 operator+(const vec4 &a, const vec4 &b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4 &a, const vec4 &b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 operator*(const vec4 &a, float f) { return vec4(a.x * f, a.y * f, a.z * f, a.w * f); }
inline vec4 operator/(const vec4 &a, float f) { return vec4(a.x / f, a.y / f, a.z / f, a.w / f); }
inline vec4 operator-(const vec4 &a, int i) { const float f = (float)i; return vec4(a.x - f, a.y - f, a.z - f, a.w - f); }
inline vec4 operator/(const vec4 &a, int i) { return a / (float)i; }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(const vec3 &a, float f) { return vec3(a.x / f, a.y / f, a.z / f); }
inline vec3 operator/(int i, const vec3 &a) { const float f = (float)i; return vec3(f / a.x, f / a.y, f / a.z); }
inline vec3 operator/(float f, const vec3 &a) { return vec3(f / a.x, f / a.y, f / a.z); }

struct mat4 {
    vec4 c[4];
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
};
inline vec4 operator*(const mat4 &m, const vec4 &v)
{
    return vec4(((m.c[0].x * v.x + m.c[1].x * v.y) + m.c[2].x * v.z) + m.c[3].x * v.w,
                ((m.c[0].y * v.x + m.c[1].y * v.y) + m.c[2].y * v.z) + m.c[3].y * v.w,
                ((m.c[0].z * v.x + m.c[1].z * v.y) + m.c[2].z * v.z) + m.c[3].z * v.w,
                ((m.c[0].w * v.x + m.c[1].w * v.y) + m.c[2].w * v.z) + m.c[3].w * v.w);
}

inline float min(float x, float y) { return y < x ? y : x; }
inline float max(float x, float y) { return x < y ? y : x; }
inline float max(float x, int y) { return max(x, (float)y); }
inline int max(int x, int y) { return x < y ? y : x; }
inline int min(int x, int y) { return y < x ? y : x; }
inline float length(const vec3 &v) { return std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z); }
template <typename V3, typename T, int A, int B, int C>
inline float length(const Swz3<V3, T, A, B, C> &s) { return length(vec3((float)s.s[A], (float)s.s[B], (float)s.s[C])); }
inline float length(const vec4 &v) { return std::sqrt(((v.x * v.x + v.y * v.y) + v.z * v.z) + v.w * v.w); }
inline vec4 normalize(const vec4 &v) { return v / length(v); }
inline vec4 clamp(const vec4 &v, const vec4 &lo, const vec4 &hi)
{
    return vec4(min(max(v.x, lo.x), hi.x), min(max(v.y, lo.y), hi.y), min(max(v.z, lo.z), hi.z), min(max(v.w, lo.w), hi.w));
}
inline bvec3 greaterThan(const vec3 &a, const vec3 &b) { return bvec3{a.x > b.x, a.y > b.y, a.z > b.z}; }
inline bvec3 lessThan(const vec3 &a, const vec3 &b) { return bvec3{a.x < b.x, a.y < b.y, a.z < b.z}; }
inline bool any(const bvec3 &b) { return b.x || b.y || b.z; }

// ---- resources
struct image2D {
    int w = 0, h = 0;
    std::vector<float> rgba;
};
struct usampler3D {
    int nx = 0, ny = 0, nz = 0, bytes = 1;
    const void *voxels = nullptr;
};
inline ivec2 imageSize(const image2D &img) { return ivec2(img.w, img.h); }
inline ivec3 textureSize(const usampler3D &t, int) { return ivec3(t.nx, t.ny, t.nz); }
inline void imageStore(image2D &img, const ivec2 &p, const vec4 &c)
{
    float *d = &img.rgba[((size_t)p.y * (size_t)img.w + (size_t)p.x) * 4];
    d[0] = c.x; d[1] = c.y; d[2] = c.z; d[3] = c.w;
}
inline int nearest_index(float u, int n)
{
    const float f = std::floor(u * (float)n);
    long long i = (f != f) ? 0 : (f < -9e18f ? (long long)-9e18 : (f > 9e18f ? (long long)9e18 : (long long)f));
    if (i < 0) i = 0;
    if (i > n - 1) i = n - 1;
    return (int)i;
}
inline uvec4 texture(const usampler3D &t, const vec3 &tc)
{
    const size_t i = (size_t)nearest_index(tc.x, t.nx), j = (size_t)nearest_index(tc.y, t.ny), k = (size_t)nearest_index(tc.z, t.nz);
    const size_t idx = i + (size_t)t.nx * (j + (size_t)t.ny * k);
    const unsigned v = t.bytes == 1 ? static_cast<const uint8_t *>(t.voxels)[idx] : static_cast<const uint16_t *>(t.voxels)[idx];
    return uvec4(v, 0u, 0u, 1u);
}

}  // namespace glsl
