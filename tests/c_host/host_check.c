/* host_check.c — TEST INFRASTRUCTURE: a plain-C99 caller of include/rpf.h, the way a cgo / JNI / Rust `extern "C"` binding
 * sees the boundary.  Built with `gcc -std=c99 -pedantic -Wall -Wextra -Werror` by tests/test_abi.py; with the argument
 * "gpu" it also creates a processor and develops a small Bayer frame (run on the GPU box), without it only the parts of
 * the ABI that need no device (operator state, PCHIP, PPM header, error reporting). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rpf.h"

#define CHECK(cond)                                                                                         \
    do {                                                                                                    \
        if (!(cond)) {                                                                                      \
            fprintf(stderr, "host_check: %s failed at line %d (last error: %s)\n", #cond, __LINE__, rpf_last_error()); \
            return 1;                                                                                       \
        }                                                                                                   \
    } while (0)

int main(int argc, char** argv)
{
    rpf_editor* ed = NULL;
    int32_t xs[3] = {0, 30000, 65535}, ys[3] = {0, 40000, 65535}, bad = -1;
    int32_t* curve = (int32_t*)malloc(65536 * sizeof(int32_t));
    uint32_t w = 0, h = 0, n = 0;
    rpf_gpu_edit_parameters gp;
    CHECK(curve != NULL);
    CHECK(rpf_version() != NULL && strlen(rpf_version()) > 0);

    /* operator state without a device (lib.rs:255-505): clamps, curve rules, mask bookkeeping, error codes */
    CHECK(rpf_editor_create_headless(8, 4, &ed) == RPF_OK && ed != NULL);
    CHECK(rpf_width(ed, &w) == RPF_OK && rpf_height(ed, &h) == RPF_OK && w == 8 && h == 4);
    CHECK(rpf_set_tone(ed, 20.0f, 150, -150, 0, 0, 0, NULL) == RPF_OK);           /* clamped to 10 / 100 / -100 */
    CHECK(rpf_editor_get_gpu_params(ed, NULL, &gp) == RPF_OK && gp.exposure == 10.0f && gp.contrast == 1.0f && gp.shadow == -1.0f);
    CHECK(rpf_set_whitebalance(ed, 1, 1, "no such mask") == RPF_ERR_MASK_NOT_FOUND);
    CHECK(strstr(rpf_last_error(), "no such mask") != NULL);
    CHECK(rpf_set_brightness_tone_curve(ed, NULL, 0, xs, 3, ys, 3, NULL) == RPF_OK);
    CHECK(rpf_editor_get_curve(ed, RPF_CURVE_BRIGHTNESS, NULL, curve) == RPF_OK && curve[0] == 0 && curve[65535] == 65535);
    CHECK(curve[30000] >= 39999 && curve[30000] <= 40000);                         /* f32 PCHIP, truncated (interpolation.rs:122) */
    CHECK(rpf_set_brightness_tone_curve(ed, curve, 100, NULL, 0, NULL, 0, NULL) == RPF_ERR_INTERP_INVALID_CURVE_LENGTH);
    CHECK(rpf_set_oklch_hue_curve(ed, NULL, 0, NULL, 0, NULL, 0, NULL) == RPF_ERR_INTERP_MISSING_CURVE_OR_CONTROL_POINTS);
    CHECK(rpf_num_masks(ed, &n) == RPF_OK && n == 1);
    CHECK(rpf_remove_mask(ed, "main") == RPF_OK && rpf_num_masks(ed, &n) == RPF_OK && n == 1);   /* "main" stays (lib.rs:501-505) */
    CHECK(rpf_apply_adjustments(ed) == RPF_ERR_GPU_COMPUTE);                       /* headless: nothing to compute on */
    rpf_editor_destroy(ed);

    /* stateless helpers */
    {
        int32_t xe[2] = {15000, 70000}, ye[2] = {0, 0}, dup[3] = {0, 5, 5};
        CHECK(rpf_pchip_interpolate_i32(xs, 3, ys, 3, xe, 2, ye, &bad) == RPF_OK && ye[0] > 15000 && ye[1] == 65535);
        CHECK(rpf_pchip_interpolate_i32(dup, 3, ys, 3, xe, 2, ye, &bad) == RPF_ERR_INTERP_NOT_STRICTLY_INCREASING && bad == 1);
        CHECK(rpf_pchip_interpolate_i32(xs, 1, ys, 1, xe, 2, ye, &bad) == RPF_ERR_INTERP_NOT_ENOUGH_POINTS);
    }
    {
        const char ppm[] = "P6\n2 1\n65535\n\x12\x34\x56\x78\x9a\xbc\xde\xf0\x11\x22\x33\x44";
        size_t off = 0;
        CHECK(rpf_ppm16_parse_header((const uint8_t*)ppm, sizeof ppm - 1, &w, &h, &off) == RPF_OK && w == 2 && h == 1 && off == 13);
        CHECK(rpf_ppm16_parse_header((const uint8_t*)"P5\n", 3, &w, &h, &off) == RPF_ERR_FAILED_TO_READ_IMAGE);
    }
    CHECK(rpf_preview_size(6000, 4000, 1280, &w, &h) == RPF_OK && w == 1280 && h == 853);

    if (argc > 1 && strcmp(argv[1], "gpu") == 0) {   /* the device path from C: processor, Bayer editor, apply, readback */
        enum { W = 64, H = 32 };
        rpf_processor* p = NULL;
        rpf_raw_params raw;
        uint16_t* cfa = (uint16_t*)malloc(W * H * sizeof(uint16_t));
        float* rgba = (float*)malloc(W * H * 4 * sizeof(float));
        uint8_t* rgb8 = (uint8_t*)malloc(W * H * 3);
        int i, cus = 0;
        CHECK(cfa && rgba && rgb8);
        for (i = 0; i < W * H; ++i) cfa[i] = (uint16_t)(512 + (i * 2654435761u >> 18) % 15000);
        memset(&raw, 0, sizeof raw);
        raw.cfa_pattern = 0; raw.white = 16383.0f;
        for (i = 0; i < 4; ++i) { raw.black[i] = 512.0f; raw.wb[i] = 1.0f; }
        raw.wb[0] = 2.0f; raw.wb[2] = 1.5f;
        raw.cam2rgb[0] = raw.cam2rgb[4] = raw.cam2rgb[8] = 1.0f;
        CHECK(rpf_processor_create(0, &p) == RPF_OK && rpf_processor_compute_units(p, &cus) == RPF_OK && cus > 0);
        CHECK(rpf_editor_create_from_bayer_u16(p, cfa, W, H, &raw, &ed) == RPF_OK);
        rpf_processor_destroy(p);                                  /* the editor keeps the processor alive (Arc semantics) */
        CHECK(rpf_set_tone(ed, 0.5f, 10, 0, 0, 0, 0, NULL) == RPF_OK);
        CHECK(rpf_apply_adjustments(ed) == RPF_OK && rpf_get_rgba_f32(ed, rgba) == RPF_OK && rpf_get_rgb_u8(ed, rgb8) == RPF_OK);
        for (i = 0; i < W * H; ++i) {
            CHECK(rgba[4 * i + 3] == 1.0f && rgba[4 * i] >= 0.0f && rgba[4 * i] <= 1.0f);
            CHECK(rgb8[3 * i] == (uint8_t)(rgba[4 * i] * 255.0f));                 /* image.rs:375-383 */
        }
        rpf_editor_destroy(ed);
        free(cfa); free(rgba); free(rgb8);
    }
    free(curve);
    printf("host_check ok\n");
    return 0;
}
