// vr_handle.h -- the opaque handle of the C ABI (include/vr_core.h: vr_handle), shared by the
// translation units that implement it (vr_capi.cpp, vr_group.cpp)
#pragma once
#include "renderer_core.h"

struct vr_renderer {
    vr::RendererCore core;
    explicit vr_renderer(int device) : core(device) {}
};
