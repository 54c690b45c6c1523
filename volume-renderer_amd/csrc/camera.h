// camera.h -- orbit camera producing the 21-float block the ray-march kernel consumes.
// Host-side mirror of the reference's Camera (include/Camera.h:9-33,
// src/Camera.cpp:16-151) without glm: same member names and call semantics
// (resetCamera / setOrientation / setViewMatrix / setUBO), own small vector type.
#pragma once
#include <array>
#include <vector>

namespace vr {

struct Vec4 {
    float x = 0, y = 0, z = 0, w = 0;
};

class Camera {
public:
    Camera();
    explicit Camera(float y_FOV, float rot_speed = 0.7f, float mov_speed = 0.3f);

    void resetCamera();                                              // src/Camera.cpp:30-44
    void setOrientation(float zoom, float zenith, float azimuth);    // src/Camera.cpp:83-151
    void setViewMatrix(Vec4 eye, Vec4 side, Vec4 up, Vec4 look_at);  // src/Camera.cpp:46-57
    void setUBO(std::vector<float> &cam_data);                       // src/Camera.cpp:59-80
    // explicit block for deterministic tests / benches (no reference equivalent)
    void setBlock(const float block21[21]);

    bool is_changed = true;
    Vec4 look_at, side, up, eye;

private:
    float view_plane_dist = 0, y_FOV = 0, rotation_speed = 0;
    [[maybe_unused]] float mov_speed = 0;
    float zenith = 0, azimuth = 0, radius = 0;
    std::array<float, 16> view2world_mat{};   // column-major
    void storeColumns(const Vec4 &c0, const Vec4 &c1, const Vec4 &c2, const Vec4 &c3);
};

}  // namespace vr
