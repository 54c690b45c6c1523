// gui_touchpoints.cpp -- compile-and-run check of integration/RendererCoreShim.h.
//
// A class named RendererGUI (the friend of include/RendererCore.h:18) performs every access the
// reference's RendererGUI makes to its RendererCore member `volren`, one block per touch-point, in the
// order of a session: start-up, load shader, load a .raw with and without sidecar, tools panel,
// windowing panel, frames, profiler read-outs, screenshot.  The statements restate the accesses (types,
// member names, call shapes) of src/RendererGUI.cpp at the cited lines; ImGui / GLFW are not involved.
//   gui_touchpoints <raw file> <out.png>          (VR_DEVICES selects the device(s); -1 = host only)
#include <cstdio>
#include <cstring>
#include <functional>

#define VR_SHIM_NO_GL 1
#include "RendererCoreShim.h"

class RendererGUI
{
    public:
        RendererCore volren;
        std::function<void(float, float, float)> camera_callback;
        int run(const char *raw_file, const char *png_file, bool have_device)
        {
            // :38-40
            volren.window_size = glm::vec2(320, 200);
            volren.framebuffer_size = glm::vec2(320, 200);
            volren.setup();
            // :42-46  glfw_manager.setCameraUpdateCallback(std::bind(&Camera::setOrientation, &volren.main_cam, _1, _2, _3))
            camera_callback = std::bind(&decltype(volren.main_cam)::setOrientation, &volren.main_cam, std::placeholders::_1,
                                        std::placeholders::_2, std::placeholders::_3);
            // :51
            volren.loadShader("VolumeRenderer.cs", false);
            if (!popup("Shader Loaded!")) return fail("no 'Shader Loaded!' message");
            // :143-144 "Reload Shader" is enabled by !loaded_shader.empty()
            if (volren.loaded_shader.empty()) return fail("loaded_shader empty");
            volren.loadShader("", true);
            popup(nullptr);
            // :276,281 profiler panel
            std::printf("workgroups %d %d\n", volren.workgroups_x, volren.workgroups_y);
            // :124-138 File > Load PVM/RAW > UINT8
            volren.datasize_bytes = 1;
            // :196-197: a .raw with a sidecar loads at once ...
            std::string fn = raw_file;
            std::string ext = fn.substr(fn.length() - 3, 3);
            bool open_inf_panel = false;
            if (ext == "pvm" || volren.checkRawInfFile(fn)) volren.readVolumeData(fn);
            else open_inf_panel = true;
            if (open_inf_panel) {
                // :423-429 ... one without goes through the raw-inf panel
                int *dims = &volren.tex3D_dim[0];
                float *spacing = &volren.voxel_size[0];
                dims[0] = 32; dims[1] = 24; dims[2] = 16;
                spacing[0] = 1.0f; spacing[1] = 1.0f; spacing[2] = 1.5f;
                if (volren.tex3D_dim != glm::ivec3(0, 0, 0) && volren.voxel_size != glm::vec3(0, 0, 0)) volren.readVolumeData(fn);   // :209
            }
            if (!have_device) {
                // without a GPU the load fails loudly; the GUI shows the message (:90-95)
                if (volren.title.empty() || volren.msg.empty()) return fail("no error message without a device");
                std::printf("host-only: %s / %s\n", volren.title.c_str(), volren.msg.c_str());
                return 0;
            }
            if (!popup("File Loaded!")) return fail("no 'File Loaded!' message");
            // :150 "Start" is enabled by both names being set; :289,:297 show them
            if (volren.loaded_shader.empty() || volren.loaded_dataset.empty()) return fail("names not set");
            std::printf("dataset %s shader %s dims %d %d %d window [%d,%d] range [%d,%d]\n", volren.loaded_dataset.c_str(), volren.loaded_shader.c_str(),
                        volren.tex3D_dim.x, volren.tex3D_dim.y, volren.tex3D_dim.z, volren.min_val, volren.max_val, volren.min_dataset_val, volren.max_dataset_val);
            // :336-337 alpha slider
            volren.alpha_scale = 0.25f;
            volren.setAlpha();
            // :342-343 MIP checkbox (on, a frame, off)
            volren.use_mip = true;
            volren.setMIP();
            volren.render();                                    // :100-101
            volren.use_mip = false;
            volren.setMIP();
            // :347-358 view top / bottom
            volren.rotate_to_top = true; volren.rotate_to_bottom = false;
            volren.setInitialCameraRotation();
            volren.render();
            volren.rotate_to_top = false;
            volren.setInitialCameraRotation();
            // :363 Reset Camera, then the mouse callback orbits
            volren.main_cam.resetCamera();
            camera_callback(0.0f, 0.06f, 0.06f);
            camera_callback(1.0f, 0.0f, 0.0f);
            // :382-385 windowing panel (DragIntRange2 writes min_val / max_val, limits come from the dataset range)
            const int lo_limit = (volren.datasize_bytes == 2) ? volren.min_dataset_val - 1000 : volren.min_dataset_val;
            volren.min_val = lo_limit + 10; volren.max_val = volren.max_dataset_val - 20;
            volren.setMinVal();
            volren.setMaxVal();
            // :52-101 the frame loop
            for (int frame = 0; frame < 5; frame++) {
                if (!volren.title.empty() && !volren.msg.empty()) return fail(volren.msg.c_str());    // :90-95
                volren.render();
            }
            // :58-60 ms per kernel, then the sum is zeroed
            const float mspk = (float)volren.kerneltime_sum / 7;
            volren.kerneltime_sum = 0;
            std::printf("mspk %.5f\n", mspk);
            if (!(mspk > 0.0f)) return fail("kerneltime_sum did not advance");
            // :222 Save Image
            const bool show_error = !(volren.saveImage(png_file, ".png"));
            if (show_error) return fail("saveImage");
            double sum = 0.0;
            for (float v : volren.lastFrame()) sum += v;
            std::printf("frame sum %.6f centre alpha %.8f\n", sum, volren.lastFrame()[((size_t)100 * 320 + 160) * 4 + 3]);
            return 0;
        }

    private:
        bool popup(const char *expect_title)                    // :90-95
        {
            if (volren.title.empty() || volren.msg.empty()) return false;
            const bool ok = !expect_title || volren.title == expect_title;
            if (!ok) std::fprintf(stderr, "message: %s / %s\n", volren.title.c_str(), volren.msg.c_str());
            volren.title.clear();
            volren.msg.clear();
            return ok;
        }
        int fail(const char *what) { std::fprintf(stderr, "FAIL %s\n", what); return 1; }
};

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const char *dev = std::getenv("VR_DEVICES");
    const bool have_device = !(dev && std::strcmp(dev, "-1") == 0);
    try {                                                       // main.cpp:10-17
        RendererGUI gui;
        return gui.run(argv[1], argv[2], have_device);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
}
