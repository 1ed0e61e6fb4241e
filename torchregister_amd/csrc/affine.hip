// Affine / rigid hot path for gfx950: fused forward warp + loss moments + analytic backward
// accumulation (F1, one pass over moving and target), device-side finalise (loss, dL/dtheta,
// Theta chain, SGD/Adam, best-theta tracking), forward warp and generic warp backward.
//
// Replaces, per iteration, the ATen chain behind ref:warpings.py:67-93 / :138-159:
//   affine_grid_generator -> grid_sampler_{2,3}d -> mse/NCC reductions -> grid_sampler backward
//   -> affine_grid backward -> SGD.step -> .item()
// with ONE streaming kernel (8 algorithmic bytes / voxel) + one tiny finalise kernel.
//
// Math: with J_p = (d out/d g_c)_p * (xn, yn, zn, 1)_p and dL/dw_p affine in
// (y_p, w_p) given the global moments,  dL/dw_p = cy*y_p + cw*w_p + c0, hence
//   dL/dtheta = cy * sum(y J) + cw * sum(w J) + c0 * sum(J).
// The streaming kernel accumulates {Sy, Sw, Syy, Sww, Syw} and sum(q J) for q in {1, y, w};
// the coefficients only exist after the pass, in the finalise kernel.
#include "trx_common.h"

namespace trx {

struct AffineGeom {
    int TX, TY, logTX, RPT, nxseg, nychunk, nblk;  // nblk = blocks per pair
};

static AffineGeom affine_geom(const trx_volumes &v, int target_blocks_total)
{
    AffineGeom g;
    int tx = 16, l = 4;
    while (tx < v.W && tx < TRX_BLOCK) { tx <<= 1; l++; }
    g.TX = tx; g.logTX = l; g.TY = TRX_BLOCK / tx;
    g.nxseg = (v.W + tx - 1) / tx;
    long rows = (long)v.D * v.H * g.nxseg * v.B;
    long rpt = rows / ((long)g.TY * target_blocks_total);
    if (rpt < 1) rpt = 1;
    if (rpt > 64) rpt = 64;
    g.RPT = (int)rpt;
    int rows_per_blk = g.TY * g.RPT;
    g.nychunk = (v.H + rows_per_blk - 1) / rows_per_blk;
    g.nblk = v.D * g.nychunk * g.nxseg;
    return g;
}

constexpr int np_full(int nd) { return 5 + 3 * nd * (nd + 1); }

// MODE 0: moments + sum(qJ), q in {1,y,w}   (the optimiser step)
// MODE 1: moments only                       (loss evaluation)
// MODE 2: sum(go*J) over channels            (generic warp backward; `tgt` = grad_out)
template <int ND, int MODE>
__global__ __launch_bounds__(TRX_BLOCK) void affine_accum_kernel(trx_volumes vol, const float *__restrict__ theta,
                                                                 AffineGeom g, int channels, size_t chan_stride,
                                                                 float *__restrict__ partials)
{
    constexpr int NC = ND;                 // gradient components (x, y[, z])
    constexpr int NQ = (MODE == 0) ? 3 : (MODE == 2 ? 1 : 0);
    constexpr int NP = (MODE == 0) ? np_full(ND) : (MODE == 1 ? 5 : ND * (ND + 1));
    const int b = blockIdx.y;
    int id = blockIdx.x;
    const int xs = id % g.nxseg; id /= g.nxseg;
    const int yc = id % g.nychunk;
    const int z = id / g.nychunk;
    const int tid = threadIdx.x;
    const int lx = tid & (g.TX - 1), ly = tid >> g.logTX;
    const int x = xs * g.TX + lx;
    const int D = vol.D, H = vol.H, W = vol.W;

    const float *__restrict__ th = theta + (size_t)b * TRX_PSTRIDE;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    const float *__restrict__ tgt = vol.target + (size_t)b * vol.target_stride;

    float A[NQ > 0 ? NQ : 1][NC], Bq[NQ > 0 ? NQ : 1][NC], m[5];
#pragma unroll
    for (int q = 0; q < (NQ > 0 ? NQ : 1); q++)
#pragma unroll
        for (int c = 0; c < NC; c++) A[q][c] = Bq[q][c] = 0.f;
#pragma unroll
    for (int i = 0; i < 5; i++) m[i] = 0.f;

    float xn = 0.f, zn = 0.f;
    if (x < W) {
        xn = base_coord(vol.xn, x, W);
        // hoisted: everything that does not depend on the row
        float bx, by, bz = 0.f, t_x1, t_y1, t_z1 = 0.f;
        if constexpr (ND == 3) {
            zn = base_coord(vol.zn, z, D);
            bx = fmaf(th[0], xn, fmaf(th[2], zn, th[3]));
            by = fmaf(th[4], xn, fmaf(th[6], zn, th[7]));
            bz = fmaf(th[8], xn, fmaf(th[10], zn, th[11]));
            t_x1 = th[1]; t_y1 = th[5]; t_z1 = th[9];
        } else {
            bx = fmaf(th[0], xn, th[2]);
            by = fmaf(th[3], xn, th[5]);
            t_x1 = th[1]; t_y1 = th[4];
        }
        const float hW = 0.5f * W, hH = 0.5f * H, hD = 0.5f * D;
        const float oW = 0.5f * (W - 1), oH = 0.5f * (H - 1), oD = 0.5f * (D - 1);
        const int y0 = yc * g.TY * g.RPT + ly;
        for (int j = 0; j < g.RPT; j++) {
            const int y = y0 + j * g.TY;
            if (y >= H) break;
            const float yn = base_coord(vol.yn, y, H);
            const float ix = fmaf(fmaf(t_x1, yn, bx), hW, oW);
            const float iy = fmaf(fmaf(t_y1, yn, by), hH, oH);
            const size_t vox = ((size_t)z * H + y) * W + x;
            float gq[NC];
            float w;
            if constexpr (MODE == 2) {
#pragma unroll
                for (int c = 0; c < NC; c++) gq[c] = 0.f;
                for (int ch = 0; ch < channels; ch++) {
                    const float go = tgt[ch * chan_stride + vox];
                    if constexpr (ND == 3) {
                        const float iz = fmaf(fmaf(t_z1, yn, bz), hD, oD);
                        Samp3 s = sample3(mov + ch * chan_stride, D, H, W, ix, iy, iz);
                        gq[0] = fmaf(go, s.dx, gq[0]); gq[1] = fmaf(go, s.dy, gq[1]); gq[2] = fmaf(go, s.dz, gq[2]);
                    } else {
                        Samp2 s = sample2(mov + ch * chan_stride, H, W, ix, iy);
                        gq[0] = fmaf(go, s.dx, gq[0]); gq[1] = fmaf(go, s.dy, gq[1]);
                    }
                }
                w = 0.f;
            } else {
                if constexpr (ND == 3) {
                    const float iz = fmaf(fmaf(t_z1, yn, bz), hD, oD);
                    Samp3 s = sample3(mov, D, H, W, ix, iy, iz);
                    w = s.v; gq[0] = s.dx; gq[1] = s.dy; gq[2] = s.dz;
                } else {
                    Samp2 s = sample2(mov, H, W, ix, iy);
                    w = s.v; gq[0] = s.dx; gq[1] = s.dy;
                }
            }
            if constexpr (MODE != 2) {
                const float yv = tgt[vox];
                m[0] += yv; m[1] += w;
                m[2] = fmaf(yv, yv, m[2]); m[3] = fmaf(w, w, m[3]); m[4] = fmaf(yv, w, m[4]);
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const float u = yn * gq[c];
                        A[0][c] += gq[c];            Bq[0][c] += u;
                        A[1][c] = fmaf(yv, gq[c], A[1][c]); Bq[1][c] = fmaf(yv, u, Bq[1][c]);
                        A[2][c] = fmaf(w, gq[c], A[2][c]);  Bq[2][c] = fmaf(w, u, Bq[2][c]);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NC; c++) { A[0][c] += gq[c]; Bq[0][c] = fmaf(yn, gq[c], Bq[0][c]); }
            }
        }
    }

    // thread-level fold: x and z are fixed per thread, so the xn / zn columns are A scaled
    float vals[NP];
    int o = 0;
    if constexpr (MODE != 2) {
#pragma unroll
        for (int i = 0; i < 5; i++) vals[o++] = m[i];
    }
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            vals[o++] = xn * A[q][c];
            vals[o++] = Bq[q][c];
            if constexpr (ND == 3) vals[o++] = zn * A[q][c];
            vals[o++] = A[q][c];
        }
    block_reduce_store<NP>(vals, partials + ((size_t)b * g.nblk + blockIdx.x) * NP);
}

// ------------------------------------------------------------------------------------------
// Finalise: reduce the per-block partials in fp64, then (MODE 0) loss, gradient, optimiser and
// best tracking for one pair; one 1024-thread block per pair.
// ------------------------------------------------------------------------------------------
struct LossCoef {
    double total, mse, ncc, ssd, cy, cw, c0;
};

__device__ __forceinline__ LossCoef loss_from_moments(const double *S, double n, const trx_loss_cfg &lc)
{
    const double Sy = S[0], Sw = S[1], Syy = S[2], Sww = S[3], Syw = S[4];
    const double my = Sy / n, mw = Sw / n;
    const double Saa = Syy - Sy * my, Sbb = Sww - Sw * mw, Sab = Syw - Sy * mw;
    const double s = sqrt(Saa * Sbb + 1e-10);  // EPSILON, ref:utils.py:15,201
    const double alpha = lc.ncc_alpha;
    const double sq = Syy - 2.0 * Syw + Sww;
    LossCoef r;
    r.mse = sq / n;
    r.ncc = alpha * (1.0 - Sab / s);
    r.ssd = (double)lc.ssd_alpha * sq;
    r.total = (double)lc.w_mse * r.mse + (double)lc.w_ncc * r.ncc + (double)lc.w_ssd * r.ssd;
    // dNCCloss/dw_p = -alpha*(a_p/s - Sab*Saa*b_p/s^3); a = y - my, b = w - mw
    const double k1 = -alpha / s, k2 = alpha * Sab * Saa / (s * s * s);
    const double q = (double)lc.w_mse * 2.0 / n + (double)lc.w_ssd * (double)lc.ssd_alpha * 2.0;
    r.cy = (double)lc.w_ncc * k1 - q;
    r.cw = (double)lc.w_ncc * k2 + q;
    r.c0 = (double)lc.w_ncc * (-k1 * my - k2 * mw);
    return r;
}

template <int ND>
__device__ void theta_from_pose(const float *p, double *th)
{
    if constexpr (ND == 3) {
        const double cps = cos((double)p[0]), sps = sin((double)p[0]);
        const double cth = cos((double)p[1]), sth = sin((double)p[1]);
        const double cph = cos((double)p[2]), sph = sin((double)p[2]);
        th[0] = cps * cth; th[1] = sph * sps * cth - cph * sth; th[2] = cph * sps * cth + sph * sth; th[3] = 0.25 * tanh((double)p[3]);
        th[4] = cps * sth; th[5] = sph * sps * sth + cph * cth; th[6] = cph * sps * sth - sph * cth; th[7] = 0.25 * tanh((double)p[4]);
        th[8] = -sps;      th[9] = sph * cps;                   th[10] = cph * cps;                  th[11] = 0.25 * tanh((double)p[5]);
    } else {
        const double c = cos((double)p[0]), s = sin((double)p[0]);
        th[0] = c; th[1] = -s; th[2] = p[1];
        th[3] = s; th[4] = c;  th[5] = p[2];
    }
}

template <int ND>
__device__ void pose_vjp(const float *p, const double *g, double *dx)
{
    if constexpr (ND == 3) {
        const double cps = cos((double)p[0]), sps = sin((double)p[0]);
        const double cth = cos((double)p[1]), sth = sin((double)p[1]);
        const double cph = cos((double)p[2]), sph = sin((double)p[2]);
        dx[0] = g[0] * (-sps * cth) + g[1] * (sph * cps * cth) + g[2] * (cph * cps * cth) + g[4] * (-sps * sth) +
                g[5] * (sph * cps * sth) + g[6] * (cph * cps * sth) + g[8] * (-cps) + g[9] * (-sph * sps) + g[10] * (-cph * sps);
        dx[1] = g[0] * (-cps * sth) + g[1] * (-sph * sps * sth - cph * cth) + g[2] * (-cph * sps * sth + sph * cth) +
                g[4] * (cps * cth) + g[5] * (sph * sps * cth - cph * sth) + g[6] * (cph * sps * cth + sph * sth);
        dx[2] = g[1] * (cph * sps * cth + sph * sth) + g[2] * (-sph * sps * cth + cph * sth) +
                g[5] * (cph * sps * sth - sph * cth) + g[6] * (-sph * sps * sth - cph * cth) + g[9] * (cph * cps) + g[10] * (-sph * cps);
        for (int i = 0; i < 3; i++) {
            const double t = tanh((double)p[3 + i]);
            dx[3 + i] = g[3 + 4 * i] * 0.25 * (1.0 - t * t);
        }
    } else {
        const double c = cos((double)p[0]), s = sin((double)p[0]);
        dx[0] = g[0] * (-s) + g[1] * (-c) + g[3] * c + g[4] * (-s);
        dx[1] = g[2];
        dx[2] = g[5];
    }
}

#define TRX_FIN_THREADS 1024

template <int NP>
__device__ __forceinline__ void reduce_partials(const float *__restrict__ part, int nblk, double *S /*shared [64]*/)
{
    __shared__ double acc[TRX_FIN_THREADS / 64][64];
    const int tid = threadIdx.x, k = tid & 63, grp = tid >> 6;
    constexpr int NG = TRX_FIN_THREADS / 64;
    double s = 0.0;
    if (k < NP) {
        int blk = grp;
        for (; blk + 3 * NG < nblk; blk += 4 * NG) {
            const float a0 = part[(size_t)blk * NP + k], a1 = part[(size_t)(blk + NG) * NP + k];
            const float a2 = part[(size_t)(blk + 2 * NG) * NP + k], a3 = part[(size_t)(blk + 3 * NG) * NP + k];
            s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        }
        for (; blk < nblk; blk += NG) s += (double)part[(size_t)blk * NP + k];
    }
    acc[grp][k] = s;
    __syncthreads();
    if (tid < 64) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < NG; i++) t += acc[i][tid];
        S[tid] = t;
    }
    __syncthreads();
}

template <int ND>
__global__ __launch_bounds__(TRX_FIN_THREADS) void affine_finalize_kernel(const float *__restrict__ partials, int nblk,
                                                                          double nvox, int D, int H, int W,
                                                                          trx_loss_cfg lc, trx_opt_cfg oc,
                                                                          trx_affine_state st)
{
    constexpr int NP = np_full(ND);
    constexpr int NT = ND * (ND + 1);
    constexpr int NPOSE = (ND == 3) ? 6 : 3;
    __shared__ double S[64];
    const int b = blockIdx.x;
    reduce_partials<NP>(partials + (size_t)b * nblk * NP, nblk, S);
    if (threadIdx.x != 0) return;

    const LossCoef L = loss_from_moments(S, nvox, lc);
    const double scale[3] = {0.5 * W, 0.5 * H, 0.5 * D};
    double dth[NT];
    for (int c = 0; c < ND; c++)
        for (int k = 0; k <= ND; k++) {
            const int i = c * (ND + 1) + k;
            dth[i] = scale[c] * (L.c0 * S[5 + i] + L.cy * S[5 + NT + i] + L.cw * S[5 + 2 * NT + i]);
        }

    float *param = st.param + (size_t)b * TRX_PSTRIDE;
    float *theta = st.theta + (size_t)b * TRX_PSTRIDE;
    const int t = st.step[b];
    const float lossf = (float)L.total;
    if (st.losses && t < st.losses_capacity) st.losses[(size_t)b * st.losses_capacity + t] = lossf;
    // best = first strict minimum, theta of THIS forward (ref:warpings.py:85-93)
    if (t == 0 || lossf < st.best_loss[b]) {
        st.best_loss[b] = lossf;
        st.best_idx[b] = t;
        for (int i = 0; i < NT; i++) st.best_theta[(size_t)b * TRX_PSTRIDE + i] = theta[i];
    }

    double g[NT];
    int np;
    if (st.mode == TRX_PARAM_RIGID) {
        pose_vjp<ND>(param, dth, g);
        np = NPOSE;
    } else {
        for (int i = 0; i < NT; i++) g[i] = dth[i];
        np = NT;
    }
    for (int i = 0; i < np; i++) {
        const float gf = (float)g[i];
        if (st.grad) st.grad[(size_t)b * TRX_PSTRIDE + i] = gf;
        float p = param[i];
        if (oc.kind == TRX_OPT_ADAM) {
            float *mm = st.adam_m + (size_t)b * TRX_PSTRIDE, *vv = st.adam_v + (size_t)b * TRX_PSTRIDE;
            const float mi = mm[i] + (gf - mm[i]) * (1.0f - oc.beta1);
            const float vi = oc.beta2 * vv[i] + (1.0f - oc.beta2) * gf * gf;
            mm[i] = mi; vv[i] = vi;
            const double bc1 = 1.0 - pow((double)oc.beta1, (double)(t + 1));
            const double bc2 = 1.0 - pow((double)oc.beta2, (double)(t + 1));
            const float denom = (float)(sqrt((double)vi) / sqrt(bc2)) + oc.eps;
            p = p - (float)((double)oc.lr / bc1) * (mi / denom);
        } else {
            p = p - oc.lr * gf;
        }
        param[i] = p;
    }
    if (st.mode == TRX_PARAM_RIGID) {
        double thd[NT];
        theta_from_pose<ND>(param, thd);
        for (int i = 0; i < NT; i++) theta[i] = (float)thd[i];
    } else {
        for (int i = 0; i < NT; i++) theta[i] = param[i];
    }
    st.step[b] = t + 1;
}

__global__ __launch_bounds__(TRX_FIN_THREADS) void affine_loss_finalize_kernel(const float *__restrict__ partials, int nblk,
                                                                               double nvox, trx_loss_cfg lc,
                                                                               float *__restrict__ terms)
{
    __shared__ double S[64];
    const int b = blockIdx.x;
    reduce_partials<5>(partials + (size_t)b * nblk * 5, nblk, S);
    if (threadIdx.x != 0) return;
    const LossCoef L = loss_from_moments(S, nvox, lc);
    terms[b * 4 + 0] = (float)L.total; terms[b * 4 + 1] = (float)L.mse;
    terms[b * 4 + 2] = (float)L.ncc;   terms[b * 4 + 3] = (float)L.ssd;
}

template <int ND>
__global__ __launch_bounds__(TRX_FIN_THREADS) void affine_bwd_finalize_kernel(const float *__restrict__ partials, int nblk,
                                                                              int D, int H, int W, float *__restrict__ dtheta)
{
    constexpr int NT = ND * (ND + 1);
    __shared__ double S[64];
    const int b = blockIdx.x;
    reduce_partials<NT>(partials + (size_t)b * nblk * NT, nblk, S);
    if (threadIdx.x >= NT) return;
    const double scale[3] = {0.5 * W, 0.5 * H, 0.5 * D};
    dtheta[(size_t)b * TRX_PSTRIDE + threadIdx.x] = (float)(scale[threadIdx.x / (ND + 1)] * S[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// Forward warp (Register.__call__): one thread per output voxel, channels share coordinates.
// ------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(TRX_BLOCK) void affine_warp_kernel(trx_volumes vol, const float *__restrict__ theta,
                                                                int channels, size_t chan_stride,
                                                                float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int D = vol.D, H = vol.H, W = vol.W;
    const size_t nvox = (size_t)D * H * W;
    const float *__restrict__ th = theta + (size_t)b * TRX_PSTRIDE;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    float *__restrict__ o = out + (size_t)b * channels * nvox;
    const float hW = 0.5f * W, hH = 0.5f * H, hD = 0.5f * D;
    const float oW = 0.5f * (W - 1), oH = 0.5f * (H - 1), oD = 0.5f * (D - 1);
    for (size_t i = (size_t)blockIdx.x * TRX_BLOCK + threadIdx.x; i < nvox; i += (size_t)gridDim.x * TRX_BLOCK) {
        const int x = (int)(i % W);
        const size_t r = i / W;
        const int y = (int)(r % H);
        const int z = (int)(r / H);
        const float xn = base_coord(vol.xn, x, W), yn = base_coord(vol.yn, y, H);
        if constexpr (ND == 3) {
            const float zn = base_coord(vol.zn, z, D);
            const float ix = fmaf(fmaf(th[1], yn, fmaf(th[0], xn, fmaf(th[2], zn, th[3]))), hW, oW);
            const float iy = fmaf(fmaf(th[5], yn, fmaf(th[4], xn, fmaf(th[6], zn, th[7]))), hH, oH);
            const float iz = fmaf(fmaf(th[9], yn, fmaf(th[8], xn, fmaf(th[10], zn, th[11]))), hD, oD);
            for (int ch = 0; ch < channels; ch++) o[ch * nvox + i] = sample3(mov + ch * chan_stride, D, H, W, ix, iy, iz).v;
        } else {
            const float ix = fmaf(fmaf(th[1], yn, fmaf(th[0], xn, th[2])), hW, oW);
            const float iy = fmaf(fmaf(th[4], yn, fmaf(th[3], xn, th[5])), hH, oH);
            for (int ch = 0; ch < channels; ch++) o[ch * nvox + i] = sample2(mov + ch * chan_stride, H, W, ix, iy).v;
        }
    }
}

static int check_vol(const trx_volumes *v, bool need_target)
{
    if (!v || !v->moving || (need_target && !v->target)) return TRX_ERR_ARG;
    if (v->ndim != 2 && v->ndim != 3) return TRX_ERR_NDIM;
    if (v->B < 1 || v->D < 1 || v->H < 1 || v->W < 1) return TRX_ERR_ARG;
    if (v->ndim == 2 && v->D != 1) return TRX_ERR_NDIM;
    if (v->B > 65535) return TRX_ERR_ARG;
    return TRX_OK;
}

constexpr int kTargetBlocks = 2048;

}  // namespace trx

using namespace trx;

extern "C" size_t trx_affine_workspace_bytes(const trx_volumes *vol)
{
    if (check_vol(vol, false) != TRX_OK) return 0;
    AffineGeom g = affine_geom(*vol, kTargetBlocks);
    return (size_t)vol->B * g.nblk * np_full(3) * sizeof(float) + 256;
}

template <int MODE>
static int launch_accum(const trx_volumes *vol, const float *theta, const AffineGeom &g, int channels, size_t chan_stride,
                        float *partials, hipStream_t s)
{
    dim3 grid(g.nblk, vol->B), block(TRX_BLOCK);
    if (vol->ndim == 3)
        hipLaunchKernelGGL((affine_accum_kernel<3, MODE>), grid, block, 0, s, *vol, theta, g, channels, chan_stride, partials);
    else
        hipLaunchKernelGGL((affine_accum_kernel<2, MODE>), grid, block, 0, s, *vol, theta, g, channels, chan_stride, partials);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_affine_step(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                               const trx_affine_state *st, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_vol(vol, true);
    if (rc) return rc;
    if (!loss || !opt || !st || !workspace) return TRX_ERR_ARG;
    if (!st->param || !st->theta || !st->best_theta || !st->best_loss || !st->best_idx || !st->step) return TRX_ERR_ARG;
    if (opt->kind != TRX_OPT_SGD && opt->kind != TRX_OPT_ADAM) return TRX_ERR_ARG;
    if (opt->kind == TRX_OPT_ADAM && (!st->adam_m || !st->adam_v)) return TRX_ERR_ARG;
    if (st->mode != TRX_PARAM_AFFINE && st->mode != TRX_PARAM_RIGID) return TRX_ERR_ARG;
    if (workspace_bytes < trx_affine_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    AffineGeom g = affine_geom(*vol, kTargetBlocks);
    float *partials = (float *)workspace;
    rc = launch_accum<0>(vol, st->theta, g, 1, 0, partials, s);
    if (rc) return rc;
    const double nvox = (double)vol->D * vol->H * vol->W;
    if (vol->ndim == 3)
        hipLaunchKernelGGL((affine_finalize_kernel<3>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, g.nblk, nvox,
                           vol->D, vol->H, vol->W, *loss, *opt, *st);
    else
        hipLaunchKernelGGL((affine_finalize_kernel<2>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, g.nblk, nvox,
                           vol->D, vol->H, vol->W, *loss, *opt, *st);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_affine_run(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                              const trx_affine_state *st, int iters, void *workspace, size_t workspace_bytes, void *stream)
{
    if (iters < 0) return TRX_ERR_ARG;
    for (int i = 0; i < iters; i++) {
        int rc = trx_affine_step(vol, loss, opt, st, workspace, workspace_bytes, stream);
        if (rc) return rc;
    }
    return TRX_OK;
}

extern "C" int trx_affine_loss(const trx_volumes *vol, const trx_loss_cfg *loss, const float *theta, float *terms,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_vol(vol, true);
    if (rc) return rc;
    if (!loss || !theta || !terms || !workspace) return TRX_ERR_ARG;
    if (workspace_bytes < trx_affine_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    AffineGeom g = affine_geom(*vol, kTargetBlocks);
    float *partials = (float *)workspace;
    rc = launch_accum<1>(vol, theta, g, 1, 0, partials, s);
    if (rc) return rc;
    const double nvox = (double)vol->D * vol->H * vol->W;
    hipLaunchKernelGGL(affine_loss_finalize_kernel, dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, g.nblk, nvox, *loss, terms);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_affine_warp(const trx_volumes *vol, const float *theta, int channels, float *out, void *stream)
{
    int rc = check_vol(vol, false);
    if (rc) return rc;
    if (!theta || !out || channels < 1) return TRX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t nvox = (size_t)vol->D * vol->H * vol->W;
    size_t nb = (nvox + TRX_BLOCK - 1) / TRX_BLOCK;
    if (nb > 8192) nb = 8192;
    dim3 grid((unsigned)nb, vol->B), block(TRX_BLOCK);
    if (vol->ndim == 3)
        hipLaunchKernelGGL((affine_warp_kernel<3>), grid, block, 0, s, *vol, theta, channels, nvox, out);
    else
        hipLaunchKernelGGL((affine_warp_kernel<2>), grid, block, 0, s, *vol, theta, channels, nvox, out);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_affine_warp_backward(const trx_volumes *vol, const float *theta, int channels, const float *grad_out,
                                        float *dtheta, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_vol(vol, false);
    if (rc) return rc;
    if (!theta || !grad_out || !dtheta || !workspace || channels < 1) return TRX_ERR_ARG;
    if (workspace_bytes < trx_affine_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    AffineGeom g = affine_geom(*vol, kTargetBlocks);
    float *partials = (float *)workspace;
    const size_t nvox = (size_t)vol->D * vol->H * vol->W;
    trx_volumes v = *vol;
    v.target = grad_out;
    v.target_stride = (size_t)channels * nvox;
    rc = launch_accum<2>(&v, theta, g, channels, nvox, partials, s);
    if (rc) return rc;
    if (vol->ndim == 3)
        hipLaunchKernelGGL((affine_bwd_finalize_kernel<3>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, g.nblk, vol->D, vol->H, vol->W, dtheta);
    else
        hipLaunchKernelGGL((affine_bwd_finalize_kernel<2>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, g.nblk, vol->D, vol->H, vol->W, dtheta);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}
