// camera.cpp -- see camera.h.  fp32 throughout, compiled with -ffp-contract=off so
// the block is reproducible against oracle/vr_oracle.c's independent restatement.
#include "camera.h"

#include <cmath>

namespace vr {

namespace {
const float kPi = 3.14159265358979323846264338327950288f;

inline Vec4 scaled(const Vec4 &v, float s) { return Vec4{v.x * s, v.y * s, v.z * s, v.w * s}; }
inline Vec4 negated(const Vec4 &v) { return Vec4{-v.x, -v.y, -v.z, -v.w}; }
// glm::normalize: v * inversesqrt(dot(v, v))
inline Vec4 unit(const Vec4 &v)
{
    const float inv = 1.0f / std::sqrt(((v.x * v.x + v.y * v.y) + v.z * v.z) + v.w * v.w);
    return scaled(v, inv);
}
// glm::cross on the xyz parts
inline Vec4 cross3(const Vec4 &a, const Vec4 &b)
{
    return Vec4{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y, 0.0f};
}
}  // namespace

Camera::Camera() {}

Camera::Camera(float fov, float rot_speed, float move_speed)
    : y_FOV(fov), rotation_speed(rot_speed), mov_speed(move_speed)
{
    view_plane_dist = 1 / std::tan(y_FOV * kPi / 360);
    is_changed = true;
    resetCamera();
}

void Camera::storeColumns(const Vec4 &c0, const Vec4 &c1, const Vec4 &c2, const Vec4 &c3)
{
    const Vec4 *cols[4] = {&c0, &c1, &c2, &c3};
    for (int c = 0; c < 4; c++) {
        view2world_mat[4 * c + 0] = cols[c]->x;
        view2world_mat[4 * c + 1] = cols[c]->y;
        view2world_mat[4 * c + 2] = cols[c]->z;
        view2world_mat[4 * c + 3] = cols[c]->w;
    }
}

void Camera::resetCamera()
{
    setViewMatrix(Vec4{0, 0, 3, 1}, Vec4{1, 0, 0, 0}, Vec4{0, 1, 0, 0}, Vec4{0, 0, -1, 0});
    zenith = kPi / 2.0f;
    azimuth = 0;
    radius = 3;
}

void Camera::setViewMatrix(Vec4 eye_in, Vec4 side_in, Vec4 up_in, Vec4 look_in)
{
    eye = eye_in;
    side = unit(side_in);
    up = unit(up_in);
    look_at = unit(look_in);
    // the reference builds the matrix from the (un-normalised) arguments, which
    // shadow the members (src/Camera.cpp:56)
    storeColumns(side_in, up_in, negated(look_in), eye_in);
}

void Camera::setUBO(std::vector<float> &cam_data)
{
    for (float f : view2world_mat) cam_data.push_back(f);
    cam_data.push_back(eye.x);
    cam_data.push_back(eye.y);
    cam_data.push_back(eye.z);
    cam_data.push_back(1);
    cam_data.push_back(view_plane_dist);
    is_changed = false;
}

void Camera::setBlock(const float b[21])
{
    for (int i = 0; i < 16; i++) view2world_mat[i] = b[i];
    side = Vec4{b[0], b[1], b[2], b[3]};
    up = Vec4{b[4], b[5], b[6], b[7]};
    look_at = Vec4{-b[8], -b[9], -b[10], -b[11]};
    eye = Vec4{b[16], b[17], b[18], 1.0f};
    view_plane_dist = b[20];
    radius = std::sqrt((eye.x * eye.x + eye.y * eye.y) + eye.z * eye.z);
    is_changed = true;
}

void Camera::setOrientation(float zoom, float d_zenith, float d_azimuth)
{
    if (d_zenith == 0 && d_azimuth == 0) {
        // dolly one unit along the view direction (src/Camera.cpp:85-94)
        const float sgn = zoom > 0 ? 1.0f : -1.0f;
        eye = Vec4{eye.x + sgn * look_at.x, eye.y + sgn * look_at.y, eye.z + sgn * look_at.z,
                   eye.w + sgn * look_at.w};
        radius = std::sqrt((eye.x * eye.x + eye.y * eye.y) + eye.z * eye.z);
        view2world_mat[12] = eye.x; view2world_mat[13] = eye.y;
        view2world_mat[14] = eye.z; view2world_mat[15] = eye.w;
        is_changed = true;
        return;
    }
    const float two_pi = kPi * 2;
    float new_zenith = zenith + d_zenith * rotation_speed;
    new_zenith = std::fmin(std::fmax(new_zenith, 0.0f), kPi);
    float new_azimuth = azimuth + d_azimuth * rotation_speed;
    // Q13: the reference wraps a negative azimuth to 2pi - a, not 2pi + a
    if (new_azimuth < 0) new_azimuth = two_pi - new_azimuth;
    else if (new_azimuth > two_pi) new_azimuth = new_azimuth - two_pi;
    if (new_zenith == zenith && new_azimuth == azimuth) return;
    zenith = new_zenith;
    azimuth = new_azimuth;

    eye.x = radius * std::sin(zenith) * std::sin(azimuth);
    eye.y = radius * std::cos(zenith);
    eye.z = radius * std::sin(zenith) * std::cos(azimuth);
    eye.w = 1;

    look_at = negated(eye);
    look_at.w = 0;
    look_at = unit(look_at);

    if (zenith == 0 || zenith == kPi) {
        // (1,0,0) rotated about +y by the azimuth: first column of glm::rotate
        const float ca = std::cos(azimuth), sa = std::sin(azimuth);
        side = Vec4{ca + (1.0f - ca) * 0.0f * 0.0f, (1.0f - ca) * 0.0f * 1.0f + sa * 0.0f,
                    (1.0f - ca) * 0.0f * 0.0f - sa * 1.0f, 0.0f};
    } else {
        side = cross3(look_at, Vec4{0, 1, 0, 0});
    }
    up = cross3(side, look_at);
    side = unit(side);
    up = unit(up);
    storeColumns(side, up, negated(look_at), eye);
    is_changed = true;
}

}  // namespace vr
