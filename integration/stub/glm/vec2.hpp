// stand-in for the two glm types the adapter's public fields use (glm itself is not vendored by the
// reference and not installed here): only what integration/gui_touchpoints.cpp touches
#pragma once
namespace glm {
struct vec2 { float x, y; vec2(float a = 0, float b = 0) : x(a), y(b) {} };
struct ivec2 {
    int x, y;
    ivec2(int a = 0, int b = 0) : x(a), y(b) {}
    ivec2(const vec2 &v) : x((int)v.x), y((int)v.y) {}          // volren.window_size = glm::vec2(...)  (RendererGUI.cpp:38-39)
};
}
