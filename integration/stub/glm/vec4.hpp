// stand-in for glm/vec4.hpp: what the reference's widget headers (include/CubicSpline.h, include/ControlPoints.h) declare with
#pragma once
namespace glm {
struct vec4 {
    float x, y, z, w;
    vec4(float a = 0, float b = 0, float c = 0, float d = 0) : x(a), y(b), z(c), w(d) {}
    float &operator[](int i) { return (&x)[i]; }
};
}
