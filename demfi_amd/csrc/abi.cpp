// Host-side pieces of the C ABI: error state, device query, weight repack, hipGraph helpers.
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <vector>
#include <mutex>
#include <set>
#include <utility>

static thread_local char g_err[512] = "";

int demfi_set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int demfi_abi_version(void) { return DEMFI_ABI_VERSION; }
extern "C" const char* demfi_last_error(void) { return g_err; }

extern "C" int demfi_device_info(char* name, int len, int* n_cu, int64_t* hbm_bytes)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return demfi_set_error(DEMFI_ERR_NODEV, "no HIP device visible");
    int dev = 0;
    DEMFI_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    DEMFI_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    if (name && len > 0) { strncpy(name, p.gcnArchName, len - 1); name[len - 1] = 0; }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return demfi_set_error(DEMFI_ERR_NODEV, "device is %s, this library is built for gfx950 only", p.gcnArchName);
    return DEMFI_OK;
}

static inline uint16_t f32_to_f16_bits(float f)
{
    _Float16 h = (_Float16)f;      // round-to-nearest-even, host compiler conversion
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}

extern "C" int demfi_pack_conv_weights(const float* w, int cout, int cin, int kh, int kw, const int32_t* cin_map, int n_k,
                                       const int32_t* chunk_nks, int n_chunks, const int32_t* cout_map, int cout_pad,
                                       int nco, int dtype, void* out, int64_t* out_bytes)
{
    if (!w || !cin_map || !chunk_nks || !cout_map || !out_bytes || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_conv_weights: null/empty argument");
    if (dtype != DEMFI_F16 && dtype != DEMFI_F32) return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_conv_weights: dtype");
    if (nco < 1 || nco > 5 || cout_pad % (32 * nco)) return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_conv_weights: cout_pad/nco");
    const int cpk = dtype == DEMFI_F16 ? 16 : 8;        // channels per k-step
    const int half = cpk / 2;                           // channels per lane
    int64_t tot_ks = 0;
    for (int c = 0; c < n_chunks; ++c) tot_ks += chunk_nks[c];
    if (tot_ks * cpk != n_k) return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_conv_weights: n_k=%d != %lld k-steps * %d", n_k, (long long)tot_ks, cpk);
    const int taps = kh * kw;
    const int nblk = cout_pad / (32 * nco);
    const int64_t vecs_per_blk = tot_ks * taps * nco * 64;          // 16-byte vectors
    *out_bytes = vecs_per_blk * nblk * 16;
    if (!out) return DEMFI_OK;
    for (int i = 0; i < n_k; ++i)
        if (cin_map[i] >= cin) return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_conv_weights: cin_map[%d]=%d >= cin", i, cin_map[i]);
    for (int i = 0; i < cout_pad; ++i)
        if (cout_map[i] >= cout) return demfi_set_error(DEMFI_ERR_ARG, "demfi_pack_conv_weights: cout_map[%d]=%d >= cout", i, cout_map[i]);
    char* o = (char*)out;
    for (int blk = 0; blk < nblk; ++blk) {
        int64_t kbase = 0;       // packed channel index of the chunk's first channel
        int64_t vbase = (int64_t)blk * vecs_per_blk;
        for (int c = 0; c < n_chunks; ++c) {
            const int nks = chunk_nks[c];
            for (int tap = 0; tap < taps; ++tap)
                for (int ks = 0; ks < nks; ++ks)
                    for (int s = 0; s < nco; ++s)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co_p = (blk * nco + s) * 32 + (lane & 31);
                            const int co = cout_map[co_p];
                            char* dst = o + (vbase + (((int64_t)tap * nks + ks) * nco + s) * 64 + lane) * 16;
                            for (int j = 0; j < half; ++j) {
                                const int64_t kp = kbase + (int64_t)ks * cpk + (lane >> 5) * half + j;
                                const int ci = cin_map[kp];
                                const float v = (co >= 0 && ci >= 0) ? w[((int64_t)co * cin + ci) * taps + tap] : 0.0f;
                                if (dtype == DEMFI_F16) { uint16_t b = f32_to_f16_bits(v); memcpy(dst + j * 2, &b, 2); }
                                else memcpy(dst + j * 4, &v, 4);
                            }
                        }
            vbase += (int64_t)nks * taps * nco * 64;
            kbase += (int64_t)nks * cpk;
        }
    }
    return DEMFI_OK;
}

// ---- hipGraph capture -------------------------------------------------------------------------------
extern "C" int demfi_graph_begin(void* stream)
{
    DEMFI_HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return DEMFI_OK;
}

extern "C" int demfi_graph_end(void* stream, void** graph_exec_out)
{
    if (!graph_exec_out) return demfi_set_error(DEMFI_ERR_ARG, "demfi_graph_end: null out");
    hipGraph_t g = nullptr;
    DEMFI_HIP_CHECK(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return demfi_set_error(DEMFI_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *graph_exec_out = (void*)ge;
    return DEMFI_OK;
}

extern "C" int demfi_graph_launch(void* graph_exec, void* stream)
{
    if (!graph_exec) return demfi_set_error(DEMFI_ERR_ARG, "demfi_graph_launch: null graph");
    DEMFI_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return DEMFI_OK;
}

extern "C" int demfi_graph_destroy(void* graph_exec)
{
    if (graph_exec) DEMFI_HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return DEMFI_OK;
}

// ---- per-(function, device) dynamic-LDS attribute ----------------------------------------------------
int demfi_ensure_lds_attr(const void* fn, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    DEMFI_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return DEMFI_OK;
    DEMFI_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({fn, dev});
    return DEMFI_OK;
}

