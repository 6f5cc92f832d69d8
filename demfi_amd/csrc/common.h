// Shared device/host helpers of libdemfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "demfi_hip.h"

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f16x_t __attribute__((ext_vector_type(16)));
typedef float f4_t __attribute__((ext_vector_type(4)));

// host-side error plumbing (abi.cpp)
int demfi_set_error(int code, const char* fmt, ...);

struct demfi_conv;
bool demfi_persist_eligible(const demfi_conv* h);   // conv.hip: the descriptor belongs to the persistent 64-channel 3x3 kernel
bool demfi_ws2_eligible(const demfi_conv* h);       // wsconv.hip: 3x3 s1 / 4x4 s2 over 32-channel units, 64-cout blocks (streamed weights, helper-wave DMA)
int demfi_ws2_launch(const demfi_conv* h, const demfi_conv* dev, void* stream);
#define DEMFI_HIP_CHECK(expr)                                                                   \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return demfi_set_error(DEMFI_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// Pointers that arrive inside descriptors / views are generic to the compiler; accessing them as such emits FLAT
// instructions (slower issue, tie up lgkmcnt as well as vmcnt, and force conservative waits around LDS traffic).
// Every buffer of this library lives in global memory: say so.
#define DEMFI_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const DEMFI_GLOBAL T* gcp(const void* p) { return (const DEMFI_GLOBAL T*)p; }
template <typename T> __device__ __forceinline__ DEMFI_GLOBAL T* gp(void* p) { return (DEMFI_GLOBAL T*)p; }
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
// 16-byte global load / store (HIP's uint4 is a class and cannot be accessed through an address-space pointer)
__device__ __forceinline__ uint4 ld_global16(const void* p) { return __builtin_bit_cast(uint4, *gcp<u4_t>(p)); }
__device__ __forceinline__ void st_global16(void* p, const uint4& v) { *gp<u4_t>(p) = __builtin_bit_cast(u4_t, v); }
// streaming store (nt): for outputs nobody on this XCD re-reads soon -- keeps the L2 for the gathered inputs
__device__ __forceinline__ void st_global16_nt(void* p, const uint4& v) { __builtin_nontemporal_store(__builtin_bit_cast(u4_t, v), gp<u4_t>(p)); }

// ---- element access through demfi_view (device) --------------------------------------------------
__device__ __forceinline__ float view_load(const demfi_view& v, int64_t off)
{
    return v.is_f32 ? gcp<float>(v.ptr)[off] : (float)gcp<half_t>(v.ptr)[off];
}
__device__ __forceinline__ void view_store(const demfi_view& v, int64_t off, float x)
{
    if (v.is_f32) gp<float>(v.ptr)[off] = x;
    else gp<half_t>(v.ptr)[off] = (half_t)x;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Epilogue activations on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each): 5-6 instructions instead
// of the ~30 of expf + IEEE division.  |error| <= ~2e-7 absolute, far inside the fp32 parity tolerance; the GRU gate
// convolutions spent 60 % of their VALU time in libm sigmoid/tanh.
__device__ __forceinline__ float fast_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x)
{
    // tanh(x) = 1 - 2 / (1 + e^(2x)); e^(2x) -> inf gives 1, -> 0 gives -1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// fp32 round trip of the reference's coordinate handling:
//   g = 2*p/den - 1        (bwarp: den = max(size-1,1), DeMFInet.py:753-754; FGAC: den = size-1, 503-504)
//   i = ((g + 1) / 2) * (size - 1)        (ATen grid_sampler_unnormalize, align_corners=True)
// Each step is ONE fp32 rounding; this translation unit is built with -ffp-contract=off.
__device__ __forceinline__ float unnormalized_coord(float p, float den, float sizem1)
{
    float g = (2.0f * p) / den;
    g = g - 1.0f;
    float i = (g + 1.0f) / 2.0f;
    return i * sizem1;
}

struct SampleMap {          // one zero-padded bilinear sample position
    int x0, y0;             // floor indices
    float w[4];             // nw, ne, sw, se with out-of-bounds corners zeroed
    int inb;                // bit k: corner k in bounds
};

__device__ __forceinline__ SampleMap make_sample_map(float ix, float iy, int H, int W)
{
    SampleMap m;
    float fx0 = floorf(ix), fy0 = floorf(iy);
    float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    float nw = (fx1 - ix) * (fy1 - iy);
    float ne = (ix - fx0) * (fy1 - iy);
    float sw = (fx1 - ix) * (iy - fy0);
    float se = (ix - fx0) * (iy - fy0);
    // clamp before the int conversion so that huge / non-finite coordinates are simply out of bounds
    float cx = fminf(fmaxf(fx0, -4.0f), (float)W + 4.0f);
    float cy = fminf(fmaxf(fy0, -4.0f), (float)H + 4.0f);
    if (!(fx0 == fx0)) cx = -4.0f;
    if (!(fy0 == fy0)) cy = -4.0f;
    m.x0 = (int)cx;
    m.y0 = (int)cy;
    bool x0in = m.x0 >= 0 && m.x0 <= W - 1, x1in = m.x0 + 1 >= 0 && m.x0 + 1 <= W - 1;
    bool y0in = m.y0 >= 0 && m.y0 <= H - 1, y1in = m.y0 + 1 >= 0 && m.y0 + 1 <= H - 1;
    m.inb = (x0in && y0in ? 1 : 0) | (x1in && y0in ? 2 : 0) | (x0in && y1in ? 4 : 0) | (x1in && y1in ? 8 : 0);
    m.w[0] = (m.inb & 1) ? nw : 0.0f;
    m.w[1] = (m.inb & 2) ? ne : 0.0f;
    m.w[2] = (m.inb & 4) ? sw : 0.0f;
    m.w[3] = (m.inb & 8) ? se : 0.0f;
    return m;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per (function, device): remember which pairs were set -- thread-safe and
// correct when one process drives several GPUs (a function-local `static bool` is neither).
int demfi_ensure_lds_attr(const void* fn, int bytes);
#define DEMFI_LDS_ATTR(fn)                                                                    \
    do {                                                                                      \
        const int st__ = demfi_ensure_lds_attr((const void*)(fn), 160 * 1024);                \
        if (st__ < 0) return st__;                                                            \
    } while (0)

