// On-GPU evaluation of a predicted frame against its target: psnr (utils.py:652-660) and the
// MATLAB-style SSIM (ssim_matlab_func, utils.py:663-683; ssim, 686-705) exactly as test() applies them (main.py:762-770):
//   pred -> np.around(denorm255_np(pred))          (float64, [0,255], rounded: rint = round-half-even like np.around)
//   gt   -> denorm255_np(gt)                       (float64, not rounded unless round_gt)
//   psnr = 20 log10(255 / sqrt(mean((gt - pred)^2)))
//   ssim = mean over the "valid" interior [5:-5, 5:-5] and all 3 channels of the 11x11 Gaussian (sigma 1.5) SSIM map
// Everything is fp64 like the reference; the Gaussian is applied separably (the 2-D window is the outer product of the 1-D
// kernel, utils.py:669-670), which changes the summation order only (|diff| ~1e-13).
// One workgroup = one 16x32 tile of SSIM-map pixels of one channel: the (16+10)x(32+10) halo of both images goes to LDS,
// the five horizontal sums (x, y, xx, yy, xy) to a second LDS array, the vertical pass produces the map value; block
// partial sums (fixed-order tree) land in a workspace and a one-workgroup kernel adds them in index order: deterministic.
#include "common.h"

namespace {

constexpr int MT_H = 16, MT_W = 32, MR = 5, MK = 11;
constexpr int MNT = 256;

struct MetricsArgs {
    const float* pred; int64_t p_sy, p_sc;     // planar fp32 [3,.,.], element strides of y and channel
    const float* gt;   int64_t g_sy, g_sc;
    int h, w, round_gt;
};

__device__ __forceinline__ double denorm255d(float v)
{
    double o = ((double)v + 1.0) / 2.0;
    o = o < 0.0 ? 0.0 : (o > 1.0 ? 1.0 : o);
    return o * 255.0;
}

__device__ __forceinline__ double block_sum(double v, double* red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = MNT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(MNT) void metrics_tile_kernel(MetricsArgs a, double* __restrict__ partial)
{
    __shared__ double ta[(MT_H + 2 * MR) * (MT_W + 2 * MR)], tb[(MT_H + 2 * MR) * (MT_W + 2 * MR)];
    __shared__ double hs[5][(MT_H + 2 * MR) * MT_W];
    __shared__ double red[MNT];
    __shared__ double gk[MK];
    const int tiles_x = (a.w + MT_W - 1) / MT_W, tiles_y = (a.h + MT_H - 1) / MT_H;
    const int tile = blockIdx.x % (tiles_x * tiles_y), ch = blockIdx.x / (tiles_x * tiles_y);
    const int ty0 = (tile / tiles_x) * MT_H, tx0 = (tile % tiles_x) * MT_W;
    if (threadIdx.x < MK) {
        double s = 0.0, mine = 0.0;
        for (int i = 0; i < MK; ++i) {                         // cv2.getGaussianKernel(11, 1.5): normalised exp(-(i-5)^2 / (2 sigma^2))
            const double d = (double)(i - MR);
            const double e = exp(-(d * d) / (2.0 * 1.5 * 1.5));
            s += e;
            if (i == (int)threadIdx.x) mine = e;
        }
        gk[threadIdx.x] = mine / s;
    }
    // ---- squared error of the tile proper (every pixel of the image belongs to exactly one tile) + halo tile to LDS ----
    constexpr int LW = MT_W + 2 * MR, LH = MT_H + 2 * MR;
    double se = 0.0;
    for (int i = threadIdx.x; i < LW * LH; i += MNT) {
        const int ly = i / LW, lx = i - ly * LW;
        const int y = ty0 - MR + ly, x = tx0 - MR + lx;
        double p = 0.0, g = 0.0;
        if (y >= 0 && y < a.h && x >= 0 && x < a.w) {
            p = rint(denorm255d(a.pred[ch * a.p_sc + (int64_t)y * a.p_sy + x]));
            g = denorm255d(a.gt[ch * a.g_sc + (int64_t)y * a.g_sy + x]);
            if (a.round_gt) g = rint(g);
            if (ly >= MR && ly < MR + MT_H && lx >= MR && lx < MR + MT_W) se += (g - p) * (g - p);
        }
        ta[i] = g;                                             // img1 = target, img2 = prediction (main.py:769-770)
        tb[i] = p;
    }
    __syncthreads();
    // ---- horizontal pass: 5 running sums for every row of the halo tile, MT_W columns ----
    for (int i = threadIdx.x; i < LH * MT_W; i += MNT) {
        const int ly = i / MT_W, c = i - ly * MT_W;
        double s1 = 0, s2 = 0, s11 = 0, s22 = 0, s12 = 0;
#pragma unroll
        for (int k = 0; k < MK; ++k) {
            const double u = ta[ly * LW + c + k], v = tb[ly * LW + c + k], wk = gk[k];
            s1 += wk * u; s2 += wk * v; s11 += wk * (u * u); s22 += wk * (v * v); s12 += wk * (u * v);
        }
        hs[0][i] = s1; hs[1][i] = s2; hs[2][i] = s11; hs[3][i] = s22; hs[4][i] = s12;
    }
    __syncthreads();
    // ---- vertical pass + SSIM map value; valid map pixels: 5 <= y < h-5, 5 <= x < w-5 ----
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    double ss = 0.0;
    for (int i = threadIdx.x; i < MT_H * MT_W; i += MNT) {
        const int r = i / MT_W, c = i - r * MT_W;
        const int y = ty0 + r, x = tx0 + c;
        if (y < MR || y >= a.h - MR || x < MR || x >= a.w - MR) continue;
        double m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
        for (int k = 0; k < MK; ++k) {
            const int j = (r + k) * MT_W + c;
            const double wk = gk[k];
            m1 += wk * hs[0][j]; m2 += wk * hs[1][j]; e11 += wk * hs[2][j]; e22 += wk * hs[3][j]; e12 += wk * hs[4][j];
        }
        const double m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
        const double v1 = e11 - m11, v2 = e22 - m22, cv = e12 - m12;
        ss += ((2 * m12 + C1) * (2 * cv + C2)) / ((m11 + m22 + C1) * (v1 + v2 + C2));
    }
    const double tse = block_sum(se, red);
    const double tss = block_sum(ss, red);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = tse; partial[2 * blockIdx.x + 1] = tss; }
}

__global__ void metrics_finish_kernel(const double* __restrict__ partial, int nblk, int h, int w, double* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double se = 0.0, ss = 0.0;
    for (int i = 0; i < nblk; ++i) { se += partial[2 * i]; ss += partial[2 * i + 1]; }     // index order: deterministic
    const double mse = se / (3.0 * h * w);
    out[0] = mse == 0.0 ? __builtin_huge_val() : 20.0 * log10(255.0 / sqrt(mse));
    const double nv = 3.0 * (double)(h - 2 * MR) * (double)(w - 2 * MR);
    out[1] = nv > 0 ? ss / nv : 0.0;
    out[2] = mse;
}

}  // namespace

extern "C" int64_t demfi_eval_workspace_bytes(int h, int w)
{
    if (h <= 0 || w <= 0) return 0;
    return (int64_t)3 * ((w + MT_W - 1) / MT_W) * ((h + MT_H - 1) / MT_H) * 2 * 8;
}

extern "C" int demfi_eval_frame(const float* pred, int64_t pred_row_stride, int64_t pred_ch_stride, const float* gt,
                                int64_t gt_row_stride, int64_t gt_ch_stride, int h, int w, int round_gt, double* workspace,
                                double* out3, void* stream)
{
    if (!pred || !gt || !workspace || !out3 || h < 2 * MR + 1 || w < 2 * MR + 1)
        return demfi_set_error(DEMFI_ERR_ARG, "demfi_eval_frame: bad arguments (frames must be at least 11x11)");
    MetricsArgs a = {pred, pred_row_stride, pred_ch_stride, gt, gt_row_stride, gt_ch_stride, h, w, round_gt};
    const int nblk = 3 * ((w + MT_W - 1) / MT_W) * ((h + MT_H - 1) / MT_H);
    hipLaunchKernelGGL(metrics_tile_kernel, dim3(nblk), dim3(MNT), 0, (hipStream_t)stream, a, workspace);
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, nblk, h, w, out3);
    DEMFI_HIP_CHECK(hipGetLastError());
    return DEMFI_OK;
}
