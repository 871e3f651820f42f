/* shard_check.c — TEST INFRASTRUCTURE: the row-tile sharding entry points of include/rpf.h from plain C, the way a Rust / cgo
 * host would drive them — two rpf_shards (each its own RCCL communicator and thread; the rank is its own neighbour above and
 * below, so one GPU suffices), two streams of different priority, two band and output buffers, frames alternating through
 * rpf_shard_develop_async.  The check is self-contained: after a self-exchange a band's halo rows hold its own first / last
 * two rows, so every output must equal, bit for bit, what rpf_dev_develop makes of the band with its halos filled that way.
 * (Parity with the oracle is tests/test_shard.py's business; this file is about the C boundary.)  HIP's C API supplies the
 * device buffers and streams, as it would in such a host. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rpf.h"

#define CHECK(cond)                                                                                          \
    do {                                                                                                     \
        if (!(cond)) {                                                                                       \
            fprintf(stderr, "shard_check: %s failed at line %d (last error: %s)\n", #cond, __LINE__, rpf_last_error()); \
            return 1;                                                                                        \
        }                                                                                                    \
    } while (0)

enum { W = 384, H = 400, Y0 = 100, ROWS = 96, STEPS = 10 };

int main(void)
{
    rpf_processor* p = NULL;
    rpf_shard* shard[2] = {NULL, NULL};
    rpf_editor* ed[2] = {NULL, NULL};
    rpf_editor* ref_ed = NULL;
    hipStream_t stream[2];
    uint16_t *h_band[2], *d_band[2], *d_ref_band;
    float *d_out[2], *d_ref_out, *h_out, *h_ref;
    rpf_raw_params raw;
    rpf_tile tile;
    const size_t band_px = (size_t)(ROWS + 4) * W, out_px = (size_t)ROWS * W;
    int lo = 0, hi = 0, l, k, rank, world, up, down;
    uint32_t launches = 0;
    double ms = 0.0;
    char bus[64];
    size_t i;

    memset(&raw, 0, sizeof raw);
    raw.cfa_pattern = 0; raw.white = 16383.0f;
    for (k = 0; k < 4; ++k) { raw.black[k] = 512.0f; raw.wb[k] = 1.0f; }
    raw.wb[0] = 2.0f; raw.wb[2] = 1.5f;
    raw.cam2rgb[0] = raw.cam2rgb[4] = raw.cam2rgb[8] = 1.0f;
    tile.frame_width = W; tile.frame_height = H; tile.y0 = Y0; tile.rows = ROWS;

    CHECK(rpf_processor_create(0, &p) == RPF_OK);
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess);
    h_out = (float*)malloc(out_px * 16); h_ref = (float*)malloc(out_px * 16);
    CHECK(h_out && h_ref);
    CHECK(hipMalloc((void**)&d_ref_band, band_px * 2) == hipSuccess && hipMalloc((void**)&d_ref_out, out_px * 16) == hipSuccess);
    CHECK(rpf_editor_create_for_tile(p, &tile, &ref_ed) == RPF_OK && rpf_set_tone(ref_ed, 0.5f, 10, 5, -5, 0, 0, NULL) == RPF_OK);
    for (l = 0; l < 2; ++l) {
        CHECK(rpf_shard_create(p, NULL, 0, 1, &shard[l]) == RPF_OK && rpf_shard_set_neighbours(shard[l], 0, 0) == RPF_OK);
        CHECK(rpf_shard_info(shard[l], &rank, &world, &up, &down, bus, sizeof bus) == RPF_OK && rank == 0 && world == 1 && up == 0 &&
              down == 0 && strchr(bus, ':') != NULL);
        CHECK(rpf_editor_create_for_tile(p, &tile, &ed[l]) == RPF_OK && rpf_set_tone(ed[l], 0.5f, 10, 5, -5, 0, 0, NULL) == RPF_OK);
        CHECK(hipStreamCreateWithPriority(&stream[l], hipStreamNonBlocking, l == 0 ? hi : lo) == hipSuccess);   /* different priorities */
        h_band[l] = (uint16_t*)malloc(band_px * 2);
        CHECK(h_band[l] != NULL);
        for (i = 0; i < band_px; ++i) h_band[l][i] = (uint16_t)(512 + ((i + 7919u * (size_t)l) * 2654435761u >> 18) % 15000);
        for (i = 0; i < 2u * W; ++i) h_band[l][i] = h_band[l][band_px - 1 - i] = (uint16_t)(12345 + l);   /* garbage in the halo rows */
        CHECK(hipMalloc((void**)&d_band[l], band_px * 2) == hipSuccess && hipMalloc((void**)&d_out[l], out_px * 16) == hipSuccess);
        CHECK(hipMemcpy(d_band[l], h_band[l], band_px * 2, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemset(d_out[l], 0xff, out_px * 16) == hipSuccess);
    }
    rpf_processor_destroy(p);   /* shards and editors keep the processor alive */

    /* argument errors first: unknown flags are refused */
    CHECK(rpf_shard_develop_async(shard[0], ed[0], d_band[0], 0, 0, 0, &raw, d_out[0], RPF_OUT_RGBA_F32, stream[0], 2u) ==
          RPF_ERR_INVALID_ARGUMENT);
    CHECK(rpf_shard_develop_async(shard[0], ed[0], NULL, 0, 0, 0, &raw, d_out[0], RPF_OUT_RGBA_F32, stream[0], 0) == RPF_ERR_INVALID_ARGUMENT);

    for (k = 0; k < STEPS; ++k) {   /* frames alternate between the two pipelines; every 2nd launch is timed */
        l = k & 1;
        CHECK(rpf_shard_develop_async(shard[l], ed[l], d_band[l], 0, 0, 0, &raw, d_out[l], RPF_OUT_RGBA_F32, stream[l],
                                      (k % 4 < 2) ? RPF_SHARD_TIMED : 0u) == RPF_OK);
    }
    CHECK(rpf_shard_flush(shard[0]) == RPF_OK && rpf_shard_flush(shard[1]) == RPF_OK);   /* enqueued; now the streams mean something */
    CHECK(hipStreamSynchronize(stream[0]) == hipSuccess && hipStreamSynchronize(stream[1]) == hipSuccess);

    for (l = 0; l < 2; ++l) {
        /* what the exchange must have produced: the band's own first / last two rows in its halo rows */
        memcpy(h_band[l], h_band[l] + 2 * W, 2 * W * 2);
        memcpy(h_band[l] + (size_t)(ROWS + 2) * W, h_band[l] + (size_t)ROWS * W, 2 * W * 2);
        CHECK(hipMemcpy(d_ref_band, h_band[l], band_px * 2, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(rpf_dev_develop(ref_ed, d_ref_band, &raw, d_ref_out, RPF_OUT_RGBA_F32, NULL) == RPF_OK);
        CHECK(hipDeviceSynchronize() == hipSuccess);
        CHECK(hipMemcpy(h_ref, d_ref_out, out_px * 16, hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(hipMemcpy(h_out, d_out[l], out_px * 16, hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(memcmp(h_out, h_ref, out_px * 16) == 0);
        CHECK(h_ref[3] == 1.0f && h_ref[0] >= 0.0f && h_ref[0] <= 1.0f);
        CHECK(rpf_shard_timing(shard[l], &launches, &ms) == RPF_OK && launches == (l == 0 ? 3u : 3u) && ms > 0.0);
    }
    for (l = 0; l < 2; ++l) {
        rpf_shard_destroy(shard[l]);
        rpf_editor_destroy(ed[l]);
        CHECK(hipStreamDestroy(stream[l]) == hipSuccess);
        hipFree(d_band[l]); hipFree(d_out[l]); free(h_band[l]);
    }
    rpf_editor_destroy(ref_ed);
    hipFree(d_ref_band); hipFree(d_ref_out); free(h_out); free(h_ref);
    printf("shard_check ok\n");
    return 0;
}
