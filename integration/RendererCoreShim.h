// RendererCoreShim.h -- drop-in for the reference's include/RendererCore.h: the same class name, the same
// public and friend-visible members, implemented over the C ABI of libvr_core.so (include/vr_core.h).
// A maintainer of gallickgunner/Volume-Renderer replaces `#include "RendererCore.h"` by this header,
// drops src/RendererCore.cpp, src/Camera.cpp and src/ddsbase.cpp from the build and links -lvr_core;
// RendererGUI.cpp, GlfwManager.cpp, ImGui and the widgets compile unchanged.
//
// Every member RendererGUI touches as a friend (src/RendererGUI.cpp:38-46,58-60,90-101,124-138,196-222,
// 336-363,382-385,423-429; include/RendererCore.h:18) is here under the reference's name.
// tests/test_integration_shim.py compiles this header together with integration/gui_touchpoints.cpp (one
// statement per touch-point, in a class named RendererGUI) and runs it.
//
// VR_DEVICES="0,1,2,3,4,5,6,7" in the environment makes the same object drive every listed GPU through
// vr_group_* (image rows sharded, RCCL gather to the first device): RendererGUI::run() then uses the whole node
// by calling render() exactly as before (src/RendererGUI.cpp:100-101).
// VR_CHOICES_FILE=<path> (round 6): the measured launch choices are loaded from that file when the object is created and written
// back when it is destroyed (vr_import_choices / vr_export_choices), so the second run of the application dispatches one settled
// kernel per frame from its first frame, as the reference's render() does (src/RendererCore.cpp:138-163).
#ifndef RENDERERCORE_H
#define RENDERERCORE_H

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "glm/vec2.hpp"
#include "glm/vec3.hpp"
#include "vr_core.h"

class RendererCore
{
    public:
        RendererCore()
        {
            std::vector<int> devices;
            if (const char *e = std::getenv("VR_DEVICES")) {
                for (const char *p = e; *p;) { char *end; const long d = std::strtol(p, &end, 10); if (end == p) break; devices.push_back((int)d); p = *end == ',' ? end + 1 : end; }
            }
            if (devices.size() > 1) {
                if (vr_group_create(&group, devices.data(), (int)devices.size()) != VR_OK) throw std::runtime_error("vr_group_create failed");
                for (int r = 0; r < vr_group_size(group); r++) handles.push_back(vr_group_member(group, r));
            } else {
                vr_handle h = nullptr;
                const int device = devices.empty() ? 0 : devices[0];          // VR_DEVICES=-1: host-only handle (no GPU calls succeed)
                if (vr_create(&h, device) != VR_OK) throw std::runtime_error(std::string("vr_create: ") + vr_last_error(nullptr));
                handles.push_back(h);
            }
            if (const char *f = std::getenv("VR_CHOICES_FILE")) {            // what an earlier run measured (a foreign / stale file takes over nothing)
                if (std::FILE *fp = std::fopen(f, "rb")) {
                    std::vector<char> blob(1 << 20);
                    blob.resize(std::fread(blob.data(), 1, blob.size(), fp));
                    std::fclose(fp);
                    for (vr_handle h : handles) (void)vr_import_choices(h, blob.data(), blob.size(), nullptr);
                }
            }
            main_cam.owner = this;
            alpha_scale = 1; kerneltime_sum = 0;
            workgroups_x = workgroups_y = 0; datasize_bytes = -1;
            min_val = max_val = max_dataset_val = min_dataset_val = 0;
            use_mip = rotate_to_bottom = rotate_to_top = false;
            voxel_size = glm::vec3(1, 1, 1); tex3D_dim = glm::ivec3(0, 0, 0);
            histogram.assign(256, 0.0f);
        }
        ~RendererCore()
        {
            if (const char *f = std::getenv("VR_CHOICES_FILE")) {            // the first device's settled choices, for the next run
                size_t n = 0;
                if (!handles.empty() && vr_export_choices(handles[0], nullptr, 0, &n) == VR_OK && n > 0) {
                    std::vector<char> blob(n);
                    if (vr_export_choices(handles[0], blob.data(), blob.size(), &n) == VR_OK)
                        if (std::FILE *fp = std::fopen(f, "wb")) { std::fwrite(blob.data(), 1, n, fp); std::fclose(fp); }
                }
            }
            if (group) vr_group_destroy(group); else if (!handles.empty()) vr_destroy(handles[0]);
        }
        RendererCore(const RendererCore &) = delete;
        RendererCore &operator=(const RendererCore &) = delete;

        void setup()                                            // src/RendererCore.cpp:34-44
        {
            const int rc = group ? vr_group_setup(group, window_size.x, window_size.y, framebuffer_size.x, framebuffer_size.y, 0, 16)
                                 : vr_setup(handles[0], window_size.x, window_size.y, framebuffer_size.x, framebuffer_size.y);
            if (rc != VR_OK)                                    // the reference throws from setupFBO (:202-218), main.cpp:14-17 catches
                throw std::runtime_error(group ? vr_group_last_error(group) : vr_last_error(handles[0]));
#ifndef VR_SHIM_NO_GL
            glGenTextures(1, &blit_tex);
            glGenFramebuffers(1, &blit_fbo);
#endif
        }
        void render()                                           // src/RendererCore.cpp:138-163
        {
            const int rc = group ? vr_group_render(group) : vr_render(handles[0]);
            if (rc != VR_OK) { title = "Error!"; msg = group ? vr_group_last_error(group) : vr_last_error(handles[0]); return; }
            kerneltime_sum += group ? vr_group_kernel_ms_take(group) : vr_kernel_ms_take(handles[0]);   // RendererGUI.cpp:58,60 reads / zeroes it
            // presentation: RGBA8 (the precision of the back buffer the reference blits to) through two pinned host frames;
            // the copy of this frame runs under the next frame's kernel, the frame shown is the previous one
            frame_valid = false;
            presented = nullptr;
            if ((group ? vr_group_present_rgba8(group, &presented) : vr_present_rgba8(handles[0], &presented)) != VR_OK) presented = nullptr;
#ifndef VR_SHIM_NO_GL
            glBindTexture(GL_TEXTURE_2D, blit_tex);                                       // :158-162: blit to the back buffer
            if (presented) glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, framebuffer_size.x, framebuffer_size.y, 0, GL_RGBA, GL_UNSIGNED_BYTE, presented);
            glBindFramebuffer(GL_READ_FRAMEBUFFER, blit_fbo);
            glFramebufferTexture2D(GL_READ_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, blit_tex, 0);
            glBindFramebuffer(GL_DRAW_FRAMEBUFFER, 0);
            glBlitFramebuffer(0, 0, framebuffer_size.x, framebuffer_size.y, 0, 0, window_size.x, window_size.y, GL_COLOR_BUFFER_BIT, GL_LINEAR);
#endif
        }
        // the frame just rendered as RGBA32F, row 0 = bottom (GL): read back on demand (screenshots of a group, tests)
        const std::vector<float> &lastFrame()
        {
            if (!frame_valid) {
                frame.resize((size_t)framebuffer_size.x * (size_t)framebuffer_size.y * 4);
                if (group) vr_group_read_pixels(group, frame.data(), frame.size());
                else vr_read_pixels(handles[0], frame.data(), frame.size());
                frame_valid = true;
            }
            return frame;
        }
        const unsigned char *presentedFrame() const { return presented; }                 // RGBA8 of the PREVIOUS render() (one frame of display latency)

    private:
        friend class RendererGUI;
        template <typename F> void each(F &&f) { for (vr_handle h : handles) f(h); }
        void setAlpha() { each([&](vr_handle h) { vr_set_alpha(h, alpha_scale); }); }                                 // :56-60
        void setMinVal() { each([&](vr_handle h) { vr_set_window(h, min_val, max_val); }); }                          // :62-71 (the +1000 of :66-67 happens inside)
        void setMaxVal() { each([&](vr_handle h) { vr_set_window(h, min_val, max_val); }); }                          // :73-82
        void setMIP() { each([&](vr_handle h) { vr_set_mip(h, use_mip ? 1 : 0); }); }                                 // :84-88
        void setInitialCameraRotation() { each([&](vr_handle h) { vr_set_view(h, rotate_to_top ? 1 : 0, rotate_to_bottom ? 1 : 0); }); }   // :90-98 (resets the camera too)
        void setUniforms() { setAlpha(); setMinVal(); setMIP(); setInitialCameraRotation(); }                        // :100-110
        void setupFBO() {}                                      // :184-219: the target lives inside the handle (vr_setup)
        void setupUBO(bool = false) {}                          // :221-240: the camera block lives inside the handle
        bool checkRawInfFile(std::string fn) { return vr_check_raw_inf_file(handles[0], fn.c_str()) != 0; }           // :46-54
        void readVolumeData(std::string fn)                     // :242-447
        {
            std::string failure;
            each([&](vr_handle h) {
                vr_set_dims(h, tex3D_dim.x, tex3D_dim.y, tex3D_dim.z);                    // the raw-inf panel's values (RendererGUI.cpp:423-424)
                vr_set_spacing(h, voxel_size.x, voxel_size.y, voxel_size.z);
                if (vr_read_volume_file(h, fn.c_str(), datasize_bytes) != VR_OK && failure.empty()) failure = vr_last_error(h);
            });
            pollMessage();                                                                // "File Loaded!" or the reference's error text
            if (!failure.empty() && (title.empty() || title == "File Loaded!")) { title = "Error!"; msg = failure; }   // e.g. no HIP device
            int d[3]; float s[3];
            vr_get_dims(handles[0], d, s, nullptr);
            tex3D_dim = glm::ivec3(d[0], d[1], d[2]); voxel_size = glm::vec3(s[0], s[1], s[2]);
            vr_get_window(handles[0], &min_val, &max_val);                                // the defaults of :360-384
            vr_get_dataset_range(handles[0], &min_dataset_val, &max_dataset_val);
            if (title == "File Loaded!") vr_histogram(handles[0], histogram.data());      // :386-405
            loaded_dataset = vr_loaded_dataset(handles[0]);
        }
        bool saveImage(std::string fn, std::string ext)         // :165-182
        {
            if (!group) return vr_save_image(handles[0], fn.c_str(), ext.c_str()) == VR_OK;
            std::vector<unsigned char> rgb((size_t)framebuffer_size.x * framebuffer_size.y * 3);
            (void)lastFrame();
            for (int y = 0; y < framebuffer_size.y; y++)                                   // top row first (stbi_flip_vertically_on_write, :172)
                for (int x = 0; x < framebuffer_size.x; x++)
                    for (int c = 0; c < 3; c++) {
                        float v = frame[((size_t)(framebuffer_size.y - 1 - y) * framebuffer_size.x + x) * 4 + c];
                        v = v != v ? 0.0f : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
                        rgb[((size_t)y * framebuffer_size.x + x) * 3 + c] = (unsigned char)(int)(v * 255.0f + 0.5f);
                    }
            return vr_write_image_rgb8(fn.c_str(), ext.c_str(), framebuffer_size.x, framebuffer_size.y, rgb.data(), framebuffer_size.x * 3) == VR_OK;
        }
        bool loadShader(std::string fn, bool reload)            // :112-136
        {
            bool ok = true;
            each([&](vr_handle h) { ok = (vr_load_shader(h, fn.c_str(), reload ? 1 : 0) == VR_OK) && ok; });
            vr_workgroups(handles[0], &workgroups_x, &workgroups_y);
            loaded_shader = vr_loaded_shader(handles[0]);
            pollMessage();
            return ok;
        }
        bool createShader(std::string, bool) { return true; }   // :449-502: the kernels are compiled into libvr_core.so
        bool createShaderProgram() { return true; }             // :504-538
        void pollMessage()
        {
            char t[256], m[1024];
            bool first = true;
            each([&](vr_handle h) { if (vr_take_message(h, t, sizeof t, m, sizeof m) && first) { title = t; msg = m; first = false; } });
        }

        // main_cam.setOrientation is bound as the GLFW camera callback (RendererGUI.cpp:42-46), resetCamera is the
        // "Reset Camera" button (:363)
        struct CameraProxy {
            RendererCore *owner = nullptr;
            void setOrientation(float zoom, float zenith, float azimuth) { owner->each([&](vr_handle h) { vr_camera_orient(h, zoom, zenith, azimuth); }); }
            void resetCamera() { owner->each([&](vr_handle h) { vr_camera_reset(h); }); }
        } main_cam;

        std::vector<float> histogram;
        std::string loaded_dataset, loaded_shader, msg, title;
        float alpha_scale, kerneltime_sum;
        int workgroups_x, workgroups_y, datasize_bytes, min_val, max_val, max_dataset_val, min_dataset_val;
        bool use_mip, rotate_to_bottom, rotate_to_top;
        glm::vec3 voxel_size;
        glm::ivec3 tex3D_dim;
        glm::ivec2 window_size, framebuffer_size;

        std::vector<vr_handle> handles;                         // one per device (VR_DEVICES), or a single handle
        vr_group_handle group = nullptr;
        std::vector<float> frame;
        bool frame_valid = false;
        const unsigned char *presented = nullptr;
        unsigned blit_tex = 0, blit_fbo = 0;
};

#endif // RENDERERCORE_H
