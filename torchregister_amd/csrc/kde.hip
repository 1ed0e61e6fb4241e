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
static int kde_ser_nchunk(long S);

// ------------------------------------------------------------------------------------------------------------------------------
// Series form.  When the window is wide against the data - |s - x| <= h for every (sample, bin) pair, which is the NMI loss's own
// setting (bandwidth 3) on intensities normalised to [0, 1] - the Gaussian is its Taylor series in u = d^2 / (2 h^2) <= 1/2, thirteen
// terms to 2e-14:  sum_i exp(-(s_i - x)^2 / 2h^2) = sum_j a_j sum_i (t_i - y)^{2j},  a_j = (-1 / 2h^2)^j / j!,  t = s - c, y = x - c,
// and sum_i (t_i - y)^{2j} = sum_m C(2j, m) (-y)^{2j-m} p_m with the power sums p_m = sum_i t_i^m.  The S x bins exponentials become
// 25 power sums of the samples (fp64) and a 13 x 25 polynomial per bin: O(25 S + 170 bins) instead of O(S bins).  All terms of the
// binomial sum are bounded by R^{2j} against a result of order (R/2)^{2j}, so fp64 keeps >= 8 digits in the worst (j = 12) term,
// which itself weighs < 1e-12 of the sum.  The backward is the same algebra: a degree-25 polynomial in t_i whose coefficients
// come from the moments sum_k g_k (-y_k)^r of the incoming gradient.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int kSerChunk = 8192;         // samples per block of the power-sum pass (32 per thread: the 25 wave reductions are amortised; 8 x 10^6 samples = 984 blocks)
constexpr int kSerN = 12;               // highest series term
constexpr int kSerP = 2 * kSerN + 1;    // power sums p_0 .. p_24
constexpr int kSerQ = 2 * kSerN + 2;    // backward polynomial coefficients q_0 .. q_25

// C(n, k) for n <= 25, exact in fp64 (Pascal's triangle at compile time: the kernels used to rebuild the coefficients with an fp64
// division per term, which was most of their run time)
struct BinomTable {
    double v[kSerQ][kSerQ];
    constexpr BinomTable() : v{}
    {
        for (int n = 0; n < kSerQ; n++) {
            v[n][0] = 1.0;
            for (int k = 1; k <= n; k++) v[n][k] = v[n - 1][k - 1] + (k <= n - 1 ? v[n - 1][k] : 0.0);
        }
    }
};
__device__ const BinomTable kBinom = BinomTable();

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // lane 0
}

// partial[n][chunk][m] = sum over the chunk of (s - c)^m
__global__ __launch_bounds__(256) void kde_powsum_kernel(const float *__restrict__ sig, long S, double center, double *__restrict__ partial)
{
    __shared__ double red[4][kSerP];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const long i0 = (long)chunk * kSerChunk;
    const float *__restrict__ src = sig + (long)n * S + i0;
    const int cnt = (int)min((long)kSerChunk, S - i0);
    double acc[kSerP];
#pragma unroll
    for (int m = 0; m < kSerP; m++) acc[m] = 0.0;
    for (int i = tid; i < cnt; i += 256) {
        const double t = (double)src[i] - center, t2 = t * t;
        double pe = 1.0, po = t;     // even and odd powers as two independent chains (half the dependent multiplies)
        acc[0] += 1.0;
#pragma unroll
        for (int m = 1; m < kSerP; m += 2) {
            acc[m] += po;
            pe *= t2;
            acc[m + 1] += pe;
            po *= t2;
        }
    }
#pragma unroll
    for (int m = 0; m < kSerP; m++) {
        const double w = wave_sum(acc[m]);
        if ((tid & 63) == 0) red[tid >> 6][m] = w;
    }
    __syncthreads();
    if (tid < kSerP) partial[((long)n * gridDim.x + chunk) * kSerP + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// pdf[n][k] = scale * sum_j a_j sum_m C(2j, m) (-y_k)^{2j-m} p_m
__global__ __launch_bounds__(1024) void kde_series_pdf_kernel(const double *__restrict__ partial, int nchunk, const float *__restrict__ xis, int bins,
                                                              double center, double inv_2h2, double scale, float *__restrict__ pdf, int xis_stride)
{
    __shared__ double acc[32][kSerP], p[kSerP];
    const int n = blockIdx.x, tid = threadIdx.x, m = tid & 31, g = tid >> 5;
    if (m < kSerP) {   // fixed-order reduction of the chunk partials: 32 groups, then the groups in order
        double a = 0.0;
        for (int c = g; c < nchunk; c += 32) a += partial[((long)n * nchunk + c) * kSerP + m];
        acc[g][m] = a;
    }
    __syncthreads();
    if (tid < kSerP) {
        double a = 0.0;
        for (int i = 0; i < 32; i++) a += acc[i][tid];
        p[tid] = a;
    }
    __syncthreads();
    for (int k = tid; k < bins; k += 1024) {
        const double ny = center - (double)xis[(long)n * xis_stride + k];   // -y
        double npw[kSerP];
        npw[0] = 1.0;
#pragma unroll
        for (int r = 1; r < kSerP; r++) npw[r] = npw[r - 1] * ny;
        double res = 0.0, aj = 1.0;
#pragma unroll
        for (int j = 0; j <= kSerN; j++) {   // fully unrolled: npw[] stays in registers, the binomials are constant-memory operands
            double sj = 0.0;
#pragma unroll
            for (int mm = 0; mm <= 2 * j; mm++) sj += (kBinom.v[2 * j][mm] * npw[2 * j - mm]) * p[mm];
            res += aj * sj;
            aj *= -inv_2h2 / (double)(j + 1);
        }
        pdf[(long)n * bins + k] = (float)(scale * res);
    }
}

// q[n][m], m = 0 .. 25: d/ds of sum_k g_k pdf_k = -(scale / h^2) * sum_m q_m t^m,
//   q_m = sum_{j : 2j+1 >= m} a_j C(2j+1, m) G_{2j+1-m},  G_r = sum_k g_k (-y_k)^r
__global__ __launch_bounds__(256) void kde_series_coef_kernel(const float *__restrict__ xis, const float *__restrict__ gpdf, int bins, double center,
                                                              double inv_2h2, double *__restrict__ q)
{
    __shared__ double red[4][kSerQ], G[kSerQ];
    const int n = blockIdx.x, tid = threadIdx.x;
    double acc[kSerQ];
#pragma unroll
    for (int r = 0; r < kSerQ; r++) acc[r] = 0.0;
    for (int k = tid; k < bins; k += 256) {
        const double ny = center - (double)xis[(long)n * bins + k], gk = (double)gpdf[(long)n * bins + k];
        double pw = gk;
#pragma unroll
        for (int r = 0; r < kSerQ; r++) { acc[r] += pw; pw *= ny; }
    }
#pragma unroll
    for (int r = 0; r < kSerQ; r++) {
        const double w = wave_sum(acc[r]);
        if ((tid & 63) == 0) red[tid >> 6][r] = w;
    }
    __syncthreads();
    if (tid < kSerQ) G[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __syncthreads();
    if (tid < kSerQ) {
        const int m = tid;
        double qm = 0.0, aj = 1.0;
        for (int j = 0; j <= kSerN; j++) {
            const int nn = 2 * j + 1;
            if (nn >= m) qm += aj * kBinom.v[nn][m] * G[nn - m];
            aj *= -inv_2h2 / (double)(j + 1);
        }
        q[(long)n * kSerQ + m] = qm;
    }
}

__global__ __launch_bounds__(256) void kde_series_backward_kernel(const float *__restrict__ sig, long S, double center, const double *__restrict__ q,
                                                                  double factor, float *__restrict__ gsig)
{
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const double *__restrict__ qn = q + (long)n * kSerQ;   // uniform: scalar loads
    const double t = (double)sig[(long)n * S + i] - center;
    double a = qn[kSerQ - 1];
#pragma unroll
    for (int m = kSerQ - 2; m >= 0; m--) a = fma(a, t, qn[m]);
    gsig[(long)n * S + i] = (float)(factor * a);
}

static int kde_ser_nchunk(long S) { return (int)((S + kSerChunk - 1) / kSerChunk); }

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

// Series form of the same PDFs (see the block comment above kde_powsum_kernel): valid when |s - x| <= h for every (sample, bin)
// pair of the call - the caller checks that (torchregister_amd.utils.PDF_xis does) and passes the centre of the data range.
extern "C" size_t trx_kde_series_workspace_bytes(int N, long S, int bins)
{
    if (N < 1 || S < 1 || bins < 1 || bins > 1024) return 0;
    const size_t fwd = (size_t)N * kde_ser_nchunk(S) * kSerP * sizeof(double), bwd = (size_t)N * kSerQ * sizeof(double);
    return fwd > bwd ? fwd : bwd;
}

extern "C" int trx_kde_pdf_series(const float *signals, const float *xis, int N, long S, int bins, float h, double center, float *pdf, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    if (!signals || !xis || !pdf || !workspace) return TRX_ERR_ARG;
    if (N < 1 || N > 65535 || S < 1 || bins < 1 || bins > 1024 || !(h > 0.f)) return TRX_ERR_ARG;
    if (workspace_bytes < trx_kde_series_workspace_bytes(N, S, bins)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = kde_ser_nchunk(S);
    hipLaunchKernelGGL(kde_powsum_kernel, dim3(nchunk, N), dim3(256), 0, s, signals, S, center, (double *)workspace);
    TRX_CHECK_LAUNCH();
    const double scale = 1.0 / ((double)h * (double)S * 6.283185307179586);
    hipLaunchKernelGGL(kde_series_pdf_kernel, dim3(N), dim3(1024), 0, s, (const double *)workspace, nchunk, xis, bins, center, 0.5 / ((double)h * (double)h),
                       scale, pdf, bins);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_kde_pdf_series_cached(const void *sums, const float *xis, int xis_stride, int N, long S, int bins, float h, double center, float *pdf,
                                         void *stream)
{
    if (!sums || !xis || !pdf || xis_stride < bins) return TRX_ERR_ARG;
    if (N < 1 || N > 65535 || S < 1 || bins < 1 || bins > 1024 || !(h > 0.f)) return TRX_ERR_ARG;
    const double scale = 1.0 / ((double)h * (double)S * 6.283185307179586);
    hipLaunchKernelGGL(kde_series_pdf_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, (const double *)sums, kde_ser_nchunk(S), xis, bins, center,
                       0.5 / ((double)h * (double)h), scale, pdf, xis_stride);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_kde_pdf_series_backward(const float *signals, const float *xis, const float *grad_pdf, int N, long S, int bins, float h, double center,
                                           float *grad_signals, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!signals || !xis || !grad_pdf || !grad_signals || !workspace) return TRX_ERR_ARG;
    if (N < 1 || N > 65535 || S < 1 || bins < 1 || bins > 1024 || !(h > 0.f)) return TRX_ERR_ARG;
    if (workspace_bytes < trx_kde_series_workspace_bytes(N, S, bins)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(kde_series_coef_kernel, dim3(N), dim3(256), 0, s, xis, grad_pdf, bins, center, 0.5 / ((double)h * (double)h), (double *)workspace);
    TRX_CHECK_LAUNCH();
    const double scale = 1.0 / ((double)h * (double)S * 6.283185307179586);
    hipLaunchKernelGGL(kde_series_backward_kernel, dim3((unsigned)((S + 255) / 256), N), dim3(256), 0, s, signals, S, center, (const double *)workspace,
                       -scale / ((double)h * (double)h), grad_signals);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

namespace trx {
// ---------------------------------------------------------------------------------------------------------------------------------
// The 256-bin algebra of the NMI loss behind the three PDFs (ref:utils.py:53-79, 224-259) as ONE kernel per evaluation, value AND
// gradient: normalise the three "histograms", Shannon terms E = sum p log2(p + EPSILON) (the reference's sign convention),
// MI = E1 + E2 - Ej, NMI = 2 MI / (E1 + E2), loss = alpha * mean_n |NMI_n - 1|, and d loss / d hist (closed form):
//   dE/dh_m = (L_m - sum_k p_k L_k) / s,  L_k = log2(p_k + eps) + p_k / ((p_k + eps) ln 2),  s = sum_k h_k,
//   dNMI/dE1 = dNMI/dE2 = 2 Ej / (E1 + E2)^2,  dNMI/dEj = -2 / (E1 + E2),  dloss/dNMI_n = alpha sign(NMI_n - 1) / N.
// In torch this is ~25 element-wise / reduction launches forward and ~40 backward on [N, 256] tensors (N = 4 or 8 patches): at 128^3
// the default criterion was host-bound on them.  One block per patch, thread k owns bin k, fp64 throughout (|NMI - 1| of nearly flat
// PDFs amplifies relative errors by ~1e4).
// `hjb` != nullptr: the pooled histogram is 0.5 (hj + hjb) (the NMI loss pools warped and target samples: the warped image's half
// arrives as the second half of its own 2 x bins evaluation, the target's half from its cached power sums) and d loss / d hj is
// scaled by `gj_scale` (0.5: the gradient wrt the warped half).  s2 / sj / sg2 / sgj: row strides of h2, hj, g2, gj in floats.
__global__ __launch_bounds__(256) void nmi_algebra_kernel(const float *__restrict__ h1, const float *__restrict__ h2, const float *__restrict__ hj, int N, int bins,
                                                          double alpha, double eps, float *__restrict__ nmi_out, float *__restrict__ mi_out,
                                                          float *__restrict__ loss_terms, float *__restrict__ g1, float *__restrict__ g2, float *__restrict__ gj,
                                                          const float *__restrict__ hjb = nullptr, int s2 = 0, int sj = 0, int sg2 = 0, int sgj = 0,
                                                          double gj_scale = 1.0)
{
    __shared__ double red[3][4];
    __shared__ double tot[3];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (s2 == 0) s2 = bins;
    if (sj == 0) sj = bins;
    if (sg2 == 0) sg2 = bins;
    if (sgj == 0) sgj = bins;
    const float *hs[3] = {h1 + (size_t)n * bins, h2 + (size_t)n * s2, hj + (size_t)n * sj};
    const float *hb = hjb ? hjb + (size_t)n * bins : nullptr;
    auto hval = [&](int q, int k) -> double {   // fp32 pooling like the torch composition: 0.5 * (a + b)
        return (q == 2 && hb) ? (double)(0.5f * (hs[2][k] + hb[k])) : (double)hs[q][k];
    };
    // fixed-order block reduction of three values per thread (bins <= 1024: a thread owns bins tid, tid + 256, ...)
    auto reduce3 = [&](double a, double b, double c, double *out) {
        double v[3] = {a, b, c};
#pragma unroll
        for (int q = 0; q < 3; q++) {
            for (int off = 32; off >= 1; off >>= 1) v[q] += __shfl_down(v[q], off);
            if (lane == 0) red[q][wave] = v[q];
        }
        __syncthreads();
        if (tid < 3) tot[tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
        __syncthreads();
        out[0] = tot[0]; out[1] = tot[1]; out[2] = tot[2];
        __syncthreads();
    };
    double s[3], e[3], pl[3];
    {
        double a[3] = {0, 0, 0};
        for (int k = tid; k < bins; k += 256)
#pragma unroll
            for (int q = 0; q < 3; q++) a[q] += hval(q, k);
        reduce3(a[0], a[1], a[2], s);
    }
    const double inv_ln2 = 1.4426950408889634;
    {
        double a[3] = {0, 0, 0}, bsum[3] = {0, 0, 0};
        for (int k = tid; k < bins; k += 256)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const double p = hval(q, k) / s[q];
                const double lg = log2(p + eps);
                a[q] += p * lg;
                bsum[q] += p * (lg + p / (p + eps) * inv_ln2);
            }
        reduce3(a[0], a[1], a[2], e);
        reduce3(bsum[0], bsum[1], bsum[2], pl);
    }
    const double e12 = e[0] + e[1];
    const double mi = e12 - e[2];
    const double nmi = 2.0 * mi / e12;
    const double dl = alpha * ((nmi > 1.0) ? 1.0 : ((nmi < 1.0) ? -1.0 : 0.0)) / (double)N;   // d loss / d NMI_n (torch.abs: 0 at 0)
    const double dn12 = 2.0 * e[2] / (e12 * e12), dnj = -2.0 / e12;
    const double w[3] = {dl * dn12 / s[0], dl * dn12 / s[1], dl * dnj / s[2]};
    float *gs[3] = {g1 ? g1 + (size_t)n * bins : nullptr, g2 ? g2 + (size_t)n * sg2 : nullptr, gj ? gj + (size_t)n * sgj : nullptr};
    const double gsc[3] = {1.0, 1.0, gj_scale};
    for (int k = tid; k < bins; k += 256)
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (gs[q]) {
                const double p = hval(q, k) / s[q];
                const double L = log2(p + eps) + p / (p + eps) * inv_ln2;
                gs[q][k] = (float)(gsc[q] * (double)(float)(w[q] * (L - pl[q])));
            }
    if (tid == 0) {
        if (nmi_out) nmi_out[n] = (float)nmi;
        if (mi_out) mi_out[n] = (float)mi;
        if (loss_terms) loss_terms[n] = (float)(alpha * fabs(nmi - 1.0) / (double)N);   // the loss is the sum of these
    }
}

}  // namespace trx

extern "C" int trx_nmi_from_pdfs_pooled(const float *h1, const float *pdf_w, const float *pdf_t, int N, int bins, float alpha, float *nmi, float *mi,
                                        float *loss_terms, float *grad_w, void *stream)
{
    if (!h1 || !pdf_w || !pdf_t || N < 1 || bins < 1 || bins > 1024) return TRX_ERR_ARG;
    if (!nmi && !mi && !loss_terms && !grad_w) return TRX_ERR_ARG;
    hipLaunchKernelGGL(trx::nmi_algebra_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, h1, pdf_w, pdf_w + bins, N, bins, (double)alpha, 1e-10, nmi, mi,
                       loss_terms, (float *)nullptr, grad_w, grad_w ? grad_w + bins : nullptr, pdf_t, 2 * bins, 2 * bins, 2 * bins, 2 * bins, 0.5);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_nmi_from_pdfs(const float *h1, const float *h2, const float *hj, int N, int bins, float alpha, float *nmi, float *mi,
                                 float *loss_terms, float *grad_h1, float *grad_h2, float *grad_hj, void *stream)
{
    if (!h1 || !h2 || !hj || N < 1 || bins < 1 || bins > 1024) return TRX_ERR_ARG;
    if (!nmi && !mi && !loss_terms && !grad_h1 && !grad_h2 && !grad_hj) return TRX_ERR_ARG;
    hipLaunchKernelGGL(trx::nmi_algebra_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, h1, h2, hj, N, bins, (double)alpha, 1e-10, nmi, mi, loss_terms,
                       grad_h1, grad_h2, grad_hj);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}
