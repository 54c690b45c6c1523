/* Plain-C client of include/vr_core.h: proves the header is valid C (no C++/torch types in
 * the ABI) and that libvr_core.so links and runs from C.
 *   client host   -> host-only handle: camera, shader bookkeeping, transfer function, errors
 *   client gpu    -> render config 0 on device 0 and print a checksum of the frame
 *   client group N -> the same frame through vr_group_* with N members (all on device 0
 *                    when the box has fewer than N GPUs)                                   */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vr_core.h"

static int fail(const char *what, vr_handle h)
{
    fprintf(stderr, "FAIL %s: %s\n", what, vr_last_error(h));
    return 1;
}

static int group_mode(int n)
{
    int devices[64];
    vr_group_handle g = NULL;
    static float frame[256 * 256 * 4];
    double sum = 0.0;
    int r, i;
    if (n < 1 || n > 64) return 1;
    for (r = 0; r < n; r++) devices[r] = 0;            /* a one-GPU box: every member on device 0 */
    if (vr_group_create(&g, devices, n) != VR_OK) { fprintf(stderr, "FAIL vr_group_create\n"); return 1; }
    if (vr_group_setup(g, 256, 256, 256, 256, 0, 16) != VR_OK) { fprintf(stderr, "FAIL vr_group_setup: %s\n", vr_group_last_error(g)); return 1; }
    for (r = 0; r < vr_group_size(g); r++) {
        vr_handle h = vr_group_member(g, r);
        if (vr_load_shader(h, "VolumeRenderer.cs", 0) != VR_OK) return fail("member load_shader", h);
        if (vr_generate_synthetic(h, VR_SYNTH_SPHERE_U8, 64, 64, 64, 1, 28) != VR_OK) return fail("member generate", h);
    }
    if (vr_group_render(g) != VR_OK) { fprintf(stderr, "FAIL vr_group_render: %s\n", vr_group_last_error(g)); return 1; }
    if (vr_group_read_pixels(g, frame, 256 * 256 * 4) != VR_OK) { fprintf(stderr, "FAIL vr_group_read_pixels: %s\n", vr_group_last_error(g)); return 1; }
    for (i = 0; i < 256 * 256 * 4; i++) sum += frame[i];
    printf("group %d transport [%s] ms %.4f sum %.6f centre_alpha %.8f\n", vr_group_size(g), vr_group_transport(g),
           vr_group_kernel_ms_take(g), sum, frame[(128 * 256 + 128) * 4 + 3]);
    vr_group_destroy(g);
    return 0;
}

int main(int argc, char **argv)
{
    const int gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    if (argc > 2 && strcmp(argv[1], "group") == 0) return group_mode(atoi(argv[2]));
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
    {   /* round 5: the copy budget and what is resident, from plain C */
        uint64_t budget = 0, vol_b = 1, copies_b = 1, other_b = 1;
        if (vr_get_copy_budget(h, &budget) != VR_OK || budget != VR_COPY_BUDGET_AUTO) return fail("copy budget default", h);
        if (vr_set_copy_budget(h, (uint64_t)1 << 30) != VR_OK || vr_get_copy_budget(h, &budget) != VR_OK || budget != ((uint64_t)1 << 30)) return fail("copy budget", h);
        if (vr_set_copy_budget(h, VR_COPY_BUDGET_AUTO) != VR_OK) return fail("copy budget auto", h);
        if (vr_get_resident_bytes(h, &vol_b, &copies_b, &other_b) != VR_OK || vol_b != 0 || copies_b != 0) return fail("resident bytes before a volume", h);
    }
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
