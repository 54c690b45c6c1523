#pragma once
namespace glm { struct vec3 { float x, y, z; vec3(float a = 0, float b = 0, float c = 0) : x(a), y(b), z(c) {} }; }
