// Parzen-window (KDE) marginal "PDF" of the reference's NMI loss as two HIP kernels:
//   pdf[n][k] = (1/h) * mean_i K((s[n][i] - x[n][k]) / h),   K(u) = exp(-u^2 / 2) / (2 pi)        (ref:utils.py:18-37;
// the 1/(2 pi) instead of 1/sqrt(2 pi) is the reference's), and its backward wrt the samples.  The reference evaluates
// this by materialising the [N, S, bins] difference tensor (8 GB for its own 3-D setting of 8 patches x 100^3 samples
// x 256 bins - SURVEY Q5: it cannot run there); here a block keeps a chunk of samples in LDS and thread k owns bin k.
// Everything downstream of the PDFs (normalisation, entropies, NMI, |NMI - 1|) is tiny and stays in torch.
// Bound: VALU (one exp per (sample, bin) pair: S * bins per PDF); bytes are negligible (4 B per sample).
#include "trx_common.h"

namespace trx {

// v_exp_f32 itself: exp2f() wraps it in a denormal-range guard (compare, two selects, add, multiply - as many issue slots as the
// exponential).  Arguments here are <= 0 and the terms are summed with O(1) neighbours, so a result flushed below 2^-126 is exact enough.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr int kKdeChunk = 4096;   // samples per block (16 KB of LDS)

// partial[n][chunk][k] = sum over the chunk of exp(-((s - x_k)/h)^2 / 2)
__global__ __launch_bounds__(256) void kde_pdf_partial_kernel(const float *__restrict__ sig, const float *__restrict__ xis, long S, int bins, float inv_h,
                                                              double *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float s[kKdeChunk];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const long i0 = (long)chunk * kKdeChunk;
    const int cnt = (int)min((long)kKdeChunk, S - i0);
    const float *__restrict__ src = sig + (long)n * S + i0;
    for (int i = tid; i < kKdeChunk; i += 256) s[i] = (i < cnt) ? src[i] : 0.f;
    __syncthreads();
    const float c2 = -0.5f * 1.4426950408889634f * inv_h * inv_h;
    for (int k = tid; k < bins; k += 256) {
        const float x = xis[(long)n * bins + k];
        // fp32 sums of 16 terms, carried in fp64 (fp64 adds are full rate on this part): the NMI downstream amplifies the
        // relative error of these sums by ~10^4 (|NMI - 1| of nearly flat PDFs), a plain fp32 running sum is not enough
        double acc = 0.0;
        const int c16 = cnt & ~15;
        for (int i = 0; i < c16; i += 16) {
            f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(&s[i + j]);   // same address in every lane: LDS broadcast
                // exp(-((s - x)/h)^2 / 2) = exp2(c (s - x)^2), c = -log2(e) / (2 h^2): 3 packed ops + 2 exp2 per two samples
                f2 d01 = {v.x - x, v.y - x}, d23 = {v.z - x, v.w - x};
                d01 = d01 * d01 * c2; d23 = d23 * d23 * c2;
                a01 += f2{fast_exp2(d01.x), fast_exp2(d01.y)}; a23 += f2{fast_exp2(d23.x), fast_exp2(d23.y)};
            }
            const float a0 = a01.x, a1 = a01.y, a2 = a23.x, a3 = a23.y;
            acc += (double)((a0 + a1) + (a2 + a3));
        }
        {
            float a0 = 0.f;
            for (int i = c16; i < cnt; i++) {
                const float d = s[i] - x;
                a0 += fast_exp2(d * d * c2);
            }
            acc += (double)a0;
        }
        partial[((long)n * gridDim.x + chunk) * bins + k] = acc;
    }
}

// pdf[n][k] = scale * sum over chunks (fixed order, fp64)
// 64 bins x 16 chunk groups per block: thread (g, k) adds chunks g, g + 16, ... (8 loads in flight), the groups are then added in
// fixed order.  (One thread per bin walking all ~250-500 chunks on 8 CUs took 78 us for 8 rows - latency, not bytes.)
__global__ __launch_bounds__(1024) void kde_pdf_finalize_kernel(const double *__restrict__ partial, int nchunk, int bins, double scale, float *__restrict__ pdf)
{
    __shared__ double acc[16][64];
    const int n = blockIdx.y, kl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl, kc = min(k, bins - 1);
    const double *__restrict__ row = partial + (long)n * nchunk * bins + kc;
    double a = 0.0;
    for (int c0 = g; c0 < nchunk; c0 += 8 * 16) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int c = c0 + i * 16;
            const double x = row[(long)min(c, nchunk - 1) * bins];   // clamped + select keeps the eight loads in flight together
            v[i] = (c < nchunk) ? x : 0.0;
        }
        a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    acc[g][kl] = a;
    __syncthreads();
    if (g != 0 || k >= bins) return;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) t += acc[i][kl];
    pdf[(long)n * bins + k] = (float)(t * scale);
}

// grad_s[n][i] = sum_k g[n][k] * d pdf[n][k] / d s[n][i] = scale * sum_k g_k * exp(-u^2/2) * (-u) / h,  u = (s_i - x_k) / h
__global__ __launch_bounds__(256) void kde_pdf_backward_kernel(const float *__restrict__ sig, const float *__restrict__ xis, const float *__restrict__ gpdf, long S,
                                                               int bins, float inv_h, float scale, float *__restrict__ gsig)
{
    __shared__ __attribute__((aligned(16))) float xs[1024], gs[1024];
    const int n = blockIdx.y, tid = threadIdx.x;
    for (int k = tid; k < bins; k += 256) { xs[k] = xis[(long)n * bins + k]; gs[k] = gpdf[(long)n * bins + k]; }
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + tid;
    if (i >= S) return;
    const float v = sig[(long)n * S + i];
    const float c2 = -0.5f * 1.4426950408889634f * inv_h * inv_h;
    // sum_k g_k (s - x_k) exp2(c (s - x_k)^2); the common factor -scale / h^2 is applied once at the end
    // four bins per step: two 16-byte LDS broadcasts, 5 packed ops + 2 exp2 per two (sample, bin) pairs
    f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    const f2 vv = {v, v};
    int k = 0;
    for (; k + 3 < bins; k += 4) {
        const float4 x = *reinterpret_cast<const float4 *>(&xs[k]), g = *reinterpret_cast<const float4 *>(&gs[k]);
        const f2 d01 = vv - f2{x.x, x.y}, d23 = vv - f2{x.z, x.w};
        const f2 t01 = d01 * d01 * c2, t23 = d23 * d23 * c2;
        const f2 e01 = {fast_exp2(t01.x), fast_exp2(t01.y)}, e23 = {fast_exp2(t23.x), fast_exp2(t23.y)};
        a01 += (f2{g.x, g.y} * d01) * e01;
        a23 += (f2{g.z, g.w} * d23) * e23;
    }
    float a0 = (a01.x + a01.y), a1 = (a23.x + a23.y);
    for (; k < bins; k++) {
        const float d0 = v - xs[k];
        a0 = fmaf(gs[k] * d0, fast_exp2(d0 * d0 * c2), a0);
    }
    gsig[(long)n * S + i] = -scale * inv_h * inv_h * (a0 + a1);
}

static int kde_nchunk(long S) { return (int)((S + kKdeChunk - 1) / kKdeChunk); }

}  // namespace trx

using namespace trx;

extern "C" size_t trx_kde_workspace_bytes(int N, long S, int bins)
{
    if (N < 1 || S < 1 || bins < 1 || bins > 1024) return 0;
    return (size_t)N * kde_nchunk(S) * bins * sizeof(double);
}

extern "C" int trx_kde_pdf(const float *signals, const float *xis, int N, long S, int bins, float h, float *pdf, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    if (!signals || !xis || !pdf || !workspace) return TRX_ERR_ARG;
    if (N < 1 || N > 65535 || S < 1 || bins < 1 || bins > 1024 || !(h > 0.f)) return TRX_ERR_ARG;
    if (workspace_bytes < trx_kde_workspace_bytes(N, S, bins)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = kde_nchunk(S);
    hipLaunchKernelGGL(kde_pdf_partial_kernel, dim3(nchunk, N), dim3(256), 0, s, signals, xis, S, bins, 1.0f / h, (double *)workspace);
    TRX_CHECK_LAUNCH();
    const double scale = 1.0 / ((double)h * (double)S * 6.283185307179586);   // (1/h) * (1/S) * 1/(2 pi)
    hipLaunchKernelGGL(kde_pdf_finalize_kernel, dim3((bins + 63) / 64, N), dim3(1024), 0, s, (const double *)workspace, nchunk, bins, scale, pdf);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_kde_pdf_backward(const float *signals, const float *xis, const float *grad_pdf, int N, long S, int bins, float h,
                                    float *grad_signals, void *stream)
{
    if (!signals || !xis || !grad_pdf || !grad_signals) return TRX_ERR_ARG;
    if (N < 1 || N > 65535 || S < 1 || bins < 1 || bins > 1024 || !(h > 0.f)) return TRX_ERR_ARG;
    const float scale = (float)(1.0 / ((double)h * (double)S * 6.283185307179586));
    hipLaunchKernelGGL(kde_pdf_backward_kernel, dim3((unsigned)((S + 255) / 256), N), dim3(256), 0, (hipStream_t)stream, signals, xis, grad_pdf, S, bins,
                       1.0f / h, scale, grad_signals);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}
