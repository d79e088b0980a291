// Shared helpers for libacgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/acgan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void acg_set_error(const char *fmt, ...);
void acg_note_kernel(const char *fmt, ...);
extern int g_acg_conv_impl;
// A/B switches of the kernel dispatchers (ACG_NO_*: take the previous kernel instead of the current one, for interleaved
// timings on one GPU box).  Development aids, not configuration: they are honoured only when ACG_DEBUG_SWITCHES is set, so a
// stray variable in a production environment cannot change which kernels run.
bool acg_debug_switch(const char *name);
void acg_record_mid_event(hipStream_t st);   // acg_debug_mid_event: bench.py times a main kernel and its reduction apart

#define ACG_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            acg_set_error(__VA_ARGS__);        \
            return ACG_ERR_INVALID;            \
        }                                      \
    } while (0)

#define ACG_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            acg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return ACG_ERR_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

static inline int acg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t acg_round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

__device__ __forceinline__ float acg_apply_act(float v, int act)
{
    switch (act) {
    case ACG_ACT_RELU: return v > 0.f ? v : 0.f;
    case ACG_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case ACG_ACT_TANH: return tanhf(v);
    default: return v;
    }
}
// d(act)/d(pre) expressed through the OUTPUT y (what the backward pass has at hand)
__device__ __forceinline__ float acg_act_grad_from_y(float y, int act)
{
    switch (act) {
    case ACG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ACG_ACT_LRELU: return y > 0.f ? 1.f : 0.2f;
    case ACG_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
    }
}
