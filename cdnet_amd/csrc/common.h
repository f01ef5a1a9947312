// Shared host-side helpers for the C ABI (error reporting, launch checks).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cdnet_hip.h"

namespace cdnet {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return CDNET_E_LAUNCH;
    }
    return CDNET_OK;
}

#define CDNET_REQUIRE(cond, ...)                \
    do {                                        \
        if (!(cond)) {                          \
            cdnet::set_error(__VA_ARGS__);      \
            return CDNET_E_ARG;                 \
        }                                       \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int WAVE = 64;

// postproc.hip: 8-connected labelling with raster-order ids (shared with the CDM generator)
int label8_raster(const uint8_t *mask, int N, int H, int W, int *L, int *aux, int *chunk, int32_t *labels, int32_t *counts,
                  hipStream_t st);
// postproc_tile.hip: the same labelling of tiles (W % 64 == 0, at most 65 536 pixels) in one launch; false: shape not served, nothing queued
bool label8_tile(const uint8_t *mask, int N, int H, int W, int32_t *labels, int32_t *counts, hipStream_t st, int *rc);

}  // namespace cdnet
