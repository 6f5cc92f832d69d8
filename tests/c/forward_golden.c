/* A complete DeMFI-Net_rb forward from plain C through the C ABI of libdemfi_hip.so (include/demfi_hip.h):
 * no Python, no torch -- the host side is this file, hipMalloc and the context API (demfi_ctx_*).
 *
 *   forward_golden <weights.bin> <case.bin>
 *
 * weights.bin : int32 n; n x { int32 name_len; char name[name_len]; int32 ndim; int64 dims[ndim]; float data[prod] }
 *               (the 260 state_dict tensors, written by tests/test_gpu_cabi.py from demfi_amd.weights)
 * case.bin    : int32 H, W, N; float t; float x[3*4*H*W]; float St[3*H*W]; float flows[4*H*W]
 *               (inputs + the reference's outputs frozen in tests/golden/e2e_*.npz)
 * Checks (fp32 path): |PSNR(St, B0) - PSNR(St_ref, B0)| <= 1e-3 dB (the north-star tolerance), median |St - St_ref|
 * < 2e-5, median |flow - flow_ref| < 2e-4; then the batched plan (demfi_forward_tb over two per-t contexts) bit-identical to
 * demfi_forward_t.  Exit code 0 and "C-ABI forward OK" on success.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "demfi_hip.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int st__ = (call);                                                                  \
        if (st__ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, st__, demfi_last_error()); return 1; } \
    } while (0)
#define HIPCHECK(call)                                                                      \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e__)); return 1; } \
    } while (0)

static int cmp_float(const void* a, const void* b)
{
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

static double median_abs_diff(const float* a, const float* b, size_t n)
{
    float* d = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) d[i] = fabsf(a[i] - b[i]);
    qsort(d, n, sizeof(float), cmp_float);
    const double m = d[n / 2];
    free(d);
    return m;
}

/* psnr (utils.py:652-660) of np.around(denorm255_np(.)) frames (utils.py:718-721, main.py:763) */
static double psnr255(const float* a, const float* b, size_t n)
{
    double mse = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double x = ((double)a[i] + 1.0) / 2.0, y = ((double)b[i] + 1.0) / 2.0;
        x = x < 0 ? 0 : (x > 1 ? 1 : x);
        y = y < 0 ? 0 : (y > 1 ? 1 : y);
        const double d = nearbyint(x * 255.0) - nearbyint(y * 255.0);
        mse += d * d;
    }
    mse /= (double)n;
    return mse == 0.0 ? INFINITY : 20.0 * log10(255.0 / sqrt(mse));
}

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s weights.bin case.bin\n", argv[0]); return 2; }
    char name[64];
    int ncu = 0;
    int64_t hbm = 0;
    CHECK(demfi_device_info(name, sizeof(name), &ncu, &hbm));
    if (demfi_abi_version() != DEMFI_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }

    FILE* fc = fopen(argv[2], "rb");
    if (!fc) { perror(argv[2]); return 1; }
    int32_t H, W, N;
    float t;
    if (fread(&H, 4, 1, fc) != 1 || fread(&W, 4, 1, fc) != 1 || fread(&N, 4, 1, fc) != 1 || fread(&t, 4, 1, fc) != 1) return 1;
    const size_t hw = (size_t)H * W;
    float* x = (float*)malloc(12 * hw * 4);
    float* st_ref = (float*)malloc(3 * hw * 4);
    float* fl_ref = (float*)malloc(4 * hw * 4);
    if (fread(x, 4, 12 * hw, fc) != 12 * hw || fread(st_ref, 4, 3 * hw, fc) != 3 * hw || fread(fl_ref, 4, 4 * hw, fc) != 4 * hw) {
        fprintf(stderr, "case file truncated\n");
        return 1;
    }
    fclose(fc);

    /* ---- context: create, load the state_dict, size + allocate + bind the workspace ------------------------- */
    demfi_ctx* ctx = NULL;
    CHECK(demfi_ctx_create(H, W, N, DEMFI_F32, NULL, 1, 2, &ctx));     /* two per-t contexts: the batched plan is checked below */
    FILE* fw = fopen(argv[1], "rb");
    if (!fw) { perror(argv[1]); return 1; }
    int32_t nt = 0;
    if (fread(&nt, 4, 1, fw) != 1) return 1;
    for (int i = 0; i < nt; ++i) {
        int32_t nl = 0, nd = 0;
        char key[256];
        int64_t dims[5];
        if (fread(&nl, 4, 1, fw) != 1 || nl <= 0 || nl >= (int)sizeof(key) || fread(key, 1, nl, fw) != (size_t)nl) return 1;
        key[nl] = 0;
        if (fread(&nd, 4, 1, fw) != 1 || nd < 1 || nd > 5 || fread(dims, 8, nd, fw) != (size_t)nd) return 1;
        size_t n = 1;
        for (int k = 0; k < nd; ++k) n *= (size_t)dims[k];
        float* w = (float*)malloc(n * 4);
        if (fread(w, 4, n, fw) != n) return 1;
        CHECK(demfi_load_weight(ctx, key, w, dims, nd));
        free(w);
    }
    fclose(fw);
    const int64_t ws_bytes = demfi_ctx_workspace_bytes(ctx);
    if (ws_bytes != demfi_workspace_bytes(H, W, N, DEMFI_F32, 1, 2)) { fprintf(stderr, "workspace size mismatch\n"); return 1; }
    char* ws = NULL;
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));
    HIPCHECK(hipMalloc((void**)&ws, (size_t)ws_bytes));
    HIPCHECK(hipMemset(ws, 0, (size_t)ws_bytes));
    CHECK(demfi_ctx_bind(ctx, ws, ws_bytes, 0, stream));

    /* ---- one forward: input window + t into the context's buffers, trunk, per-t segment ---------------------- */
    int64_t off_x, off_t, off_fin, off_delta;
    int32_t kind, dims[4];
    CHECK(demfi_ctx_buffer(ctx, 0, -1, "x", &off_x, &kind, dims));
    CHECK(demfi_ctx_buffer(ctx, 0, 0, "t", &off_t, &kind, dims));
    CHECK(demfi_ctx_buffer(ctx, 0, 0, "finals", &off_fin, &kind, dims));
    CHECK(demfi_ctx_buffer(ctx, 0, 0, "delta", &off_delta, &kind, dims));
    HIPCHECK(hipMemcpyAsync(ws + off_x, x, 12 * hw * 4, hipMemcpyHostToDevice, stream));
    HIPCHECK(hipMemcpyAsync(ws + off_t, &t, 4, hipMemcpyHostToDevice, stream));
    CHECK(demfi_forward_trunk(ctx, 0, NULL, stream));
    CHECK(demfi_forward_t(ctx, 0, 0, N, stream));
    float* st = (float*)malloc(3 * hw * 4);
    float* fl = (float*)malloc(4 * hw * 4);
    /* finals: [N][3 frames][3][H][W]; St of the last iteration = frame 2.  delta: [N+1][5][H][W], flows = channels 0..3 */
    HIPCHECK(hipMemcpyAsync(st, ws + off_fin + ((size_t)(N - 1) * 9 + 6) * hw * 4, 3 * hw * 4, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipMemcpyAsync(fl, ws + off_delta + (size_t)N * 5 * hw * 4, 4 * hw * 4, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));

    /* ---- checks ---------------------------------------------------------------------------------------------- */
    float* gt = (float*)malloc(3 * hw * 4);                    /* pseudo ground truth: frame B0 = planes c*4 + 0 of x */
    for (int c = 0; c < 3; ++c) memcpy(gt + c * hw, x + (size_t)(c * 4) * hw, hw * 4);
    const double p_got = psnr255(st, gt, 3 * hw), p_ref = psnr255(st_ref, gt, 3 * hw);
    const double med_st = median_abs_diff(st, st_ref, 3 * hw), med_fl = median_abs_diff(fl, fl_ref, 4 * hw);
    printf("%s (%d CUs): %dx%d N=%d t=%.3f  PSNR(St,B0) %.4f dB vs reference %.4f dB, median|dSt| %.2e, median|dflow| %.2e\n",
           name, ncu, H, W, N, t, p_got, p_ref, med_st, med_fl);
    int bad = 0;
    if (!(fabs(p_got - p_ref) <= 1e-3)) { fprintf(stderr, "PSNR differs by more than 1e-3 dB\n"); bad = 1; }
    if (!(med_st < 2e-5)) { fprintf(stderr, "St differs\n"); bad = 1; }
    if (!(med_fl < 2e-4)) { fprintf(stderr, "flows differ\n"); bad = 1; }
    if (demfi_forward_t(ctx, 0, 0, N + 1, stream) != DEMFI_ERR_ARG) { fprintf(stderr, "num_update > N was not rejected\n"); bad = 1; }
    /* ---- the batched plan: both per-t contexts in ONE launch sequence (every convolution over batch x 2) must leave in each
     * of them exactly what demfi_forward_t left in context 0 */
    {
        int64_t off_t1, off_fin1;
        CHECK(demfi_ctx_buffer(ctx, 0, 1, "t", &off_t1, &kind, dims));
        CHECK(demfi_ctx_buffer(ctx, 0, 1, "finals", &off_fin1, &kind, dims));
        const size_t fin_bytes = (size_t)N * 9 * hw * 4;
        if ((size_t)(off_fin1 - off_fin) != fin_bytes) { fprintf(stderr, "per-t copies of a buffer are not contiguous\n"); bad = 1; }
        HIPCHECK(hipMemsetAsync(ws + off_fin, 0, 2 * fin_bytes, stream));
        HIPCHECK(hipMemcpyAsync(ws + off_t1, &t, 4, hipMemcpyHostToDevice, stream));
        CHECK(demfi_forward_tb(ctx, 0, N, stream));
        float* st2 = (float*)malloc(3 * hw * 4);
        for (int c = 0; c < 2; ++c) {
            HIPCHECK(hipMemcpyAsync(st2, ws + (c ? off_fin1 : off_fin) + ((size_t)(N - 1) * 9 + 6) * hw * 4, 3 * hw * 4, hipMemcpyDeviceToHost, stream));
            HIPCHECK(hipStreamSynchronize(stream));
            if (memcmp(st2, st, 3 * hw * 4) != 0) { fprintf(stderr, "demfi_forward_tb: context %d differs from demfi_forward_t\n", c); bad = 1; }
        }
        free(st2);
    }
    CHECK(demfi_ctx_destroy(ctx));
    HIPCHECK(hipFree(ws));
    if (bad) return 1;
    printf("C-ABI forward OK\n");
    return 0;
}
