/* Plain-C client of include/vr_core.h: proves the header is valid C (no C++/torch types in
 * the ABI) and that libvr_core.so links and runs from C.
 *   client host   -> host-only handle: camera, shader bookkeeping, transfer function, errors
 *   client gpu    -> render config 0 on device 0 and print a checksum of the frame        */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vr_core.h"

static int fail(const char *what, vr_handle h)
{
    fprintf(stderr, "FAIL %s: %s\n", what, vr_last_error(h));
    return 1;
}

int main(int argc, char **argv)
{
    const int gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    vr_handle h = NULL;
    if (vr_create(&h, gpu ? 0 : -1) != VR_OK) return fail("vr_create", NULL);
    if (vr_setup(h, 256, 256, 256, 256) != VR_OK) return fail("vr_setup", h);
    if (vr_load_shader(h, "VolumeRenderer.cs", 0) != VR_OK) return fail("vr_load_shader", h);
    char title[64], msg[256];
    if (!vr_take_message(h, title, sizeof title, msg, sizeof msg) || strcmp(title, "Shader Loaded!") != 0) return fail("message", h);
    int wx = 0, wy = 0;
    vr_workgroups(h, &wx, &wy);
    float cam[21];
    if (vr_camera_orient(h, 0.0f, 0.06f, 0.06f) != VR_OK || vr_camera_get_block(h, cam) != VR_OK) return fail("camera", h);
    const int32_t iso[4] = {0, 141, 149, 255};
    const float rgba[16] = {0, 0, 0, 0, 0, 0, 0, 0.759f, 0, 0, 0, 0.45f, 0, 0, 0, 1};
    static float lut[1024];
    if (vr_set_transfer_function(h, iso, rgba, 4) != VR_OK || vr_get_transfer_lut(h, lut) != VR_OK) return fail("tf", h);
    if (vr_set_transfer_function(h, NULL, NULL, 0) != VR_OK) return fail("tf reset", h);
    printf("workgroups %d %d eye %.6f %.6f %.6f lut141 %.6f\n", wx, wy, cam[16], cam[17], cam[18], lut[141 * 4 + 3]);
    if (!gpu) {
        if (vr_render(h) != VR_E_NO_DEVICE) return fail("render must fail without a device", h);
        printf("render without device: %s\n", vr_last_error(h));
    } else {
        if (vr_camera_reset(h) != VR_OK) return fail("reset", h);
        if (vr_generate_synthetic(h, VR_SYNTH_SPHERE_U8, 64, 64, 64, 1, 28) != VR_OK) return fail("generate", h);
        if (vr_render(h) != VR_OK) return fail("render", h);
        static float frame[256 * 256 * 4];
        if (vr_read_pixels(h, frame, 256 * 256 * 4) != VR_OK) return fail("read_pixels", h);
        uint64_t total = 0;
        if (vr_count_samples(h, &total, NULL, 0) != VR_OK) return fail("count", h);
        double sum = 0.0;
        for (int i = 0; i < 256 * 256 * 4; i++) sum += frame[i];
        printf("kernel %s ms %.4f samples %llu sum %.6f centre_alpha %.8f\n", vr_last_kernel_name(h), vr_kernel_ms_take(h),
               (unsigned long long)total, sum, frame[(128 * 256 + 128) * 4 + 3]);
    }
    vr_destroy(h);
    return 0;
}
