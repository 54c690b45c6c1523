#pragma once
namespace glm {
struct vec3 {
    float x, y, z;
    vec3(float a = 0, float b = 0, float c = 0) : x(a), y(b), z(c) {}
    float &operator[](int i) { return (&x)[i]; }
    bool operator!=(const vec3 &o) const { return x != o.x || y != o.y || z != o.z; }
};
struct ivec3 {
    int x, y, z;
    ivec3(int a = 0, int b = 0, int c = 0) : x(a), y(b), z(c) {}
    int &operator[](int i) { return (&x)[i]; }
    bool operator!=(const ivec3 &o) const { return x != o.x || y != o.y || z != o.z; }
};
}
