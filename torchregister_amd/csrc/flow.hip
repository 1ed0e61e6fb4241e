// Dense flow-field hot path for gfx950 (SpatialTransformer semantics, ref:utils.py:339-365):
// sample moving at voxel coordinate p + flow(p) (the reference's normalise + align_corners=True
// un-normalise cancel exactly), zeros outside.  d out / d flow_c is the plain trilinear
// derivative along spatial dim c.
//
// One optimiser iteration (F2) = two streaming passes + one tiny kernel:
//   pass A  flow_moments_kernel : sample, accumulate {Sy,Sw,Syy,Sww,Syw} (+ smoothness sums)
//   coef    flow_coef_kernel    : fp64 reduce -> loss, NCC/MSE/SSD gradient coefficients, Adam scalars
//   pass B  flow_update_kernel  : re-sample + derivative, dL/dflow, SGD/Adam update in place
// replacing SpatialTransformer.forward -> criterion -> backward -> optimizer.step of
// ref:warpings.py:208-220 when the parameter is the flow itself.
#include "trx_common.h"

namespace trx {

struct FlowCoef {   // per pair, written by flow_coef_kernel, read by pass B (wave-uniform)
    float k1, k2, my, mw, q;   // dL/dw_p = k1*(y-my) + k2*(w-mw) + q*(w-y)
    float step_size, inv_sqrt_bc2, sm[3];  // Adam scalars; smoothness gradient scale per dim
    int mode;                  // kUpd*: what the update kernel does for this pair in this iteration (early stop, see flow_coef_kernel)
};
// Early stop (ref:warpings.py:231-233) without a host sync.  stopped[b]: 0 running; 1 the loss of the last executed iteration was <= stop_crit
// (its update has been / is being applied, as in the reference, which steps before it tests) and a double-buffered flow still differs
// between its two buffers; 2 stopped and settled.  The host keeps enqueueing iterations and swapping its buffer pointers; for a stopped
// pair they are no-ops:
//   kUpdNormal      the usual update
//   kUpdHit         the usual update, and the flow it starts from is kept in flow_last (the flow of the last forward)
//   kUpdCopy        flow_out = flow (first iteration after a hit, double-buffered flows only: both buffers hold the final flow afterwards)
//   kUpdSkip        nothing
//   kUpdTransition  lagged regulariser (the hit is seen one coefficient kernel late, the update of the hit iteration is already in
//                   `flow`, the flow it started from is still in `flow_out`): flow_last = flow_out, then flow_out = flow
constexpr int kUpdNormal = 0, kUpdHit = 1, kUpdCopy = 2, kUpdSkip = 3, kUpdTransition = 4;

// Z-slab partition of ONE volume (BASELINE config 5): target / flow / optimiser state of a rank hold planes
// [zoff, zoff + D) of a Dm-deep volume; `moving` is the whole volume (replicated: it is constant, so no halo
// of it is ever exchanged).  {0, D} = no partition.
struct Slab {
    int zoff, Dm;
    // neighbour ranks' flow planes adjacent to this slab ([ndim][H][W] each, flow BEFORE this iteration's update),
    // NULL at the ends of the volume or when the smoothness regulariser is off
    const float *halo_lo, *halo_hi;
};

// sample at p + flow; returns value and derivative in flow-channel order (dim0, dim1[, dim2])
template <int ND>
__device__ __forceinline__ float flow_sample(const float *__restrict__ mov, const float *__restrict__ fl, size_t nvox,
                                             size_t i, int D, int H, int W, int z, int y, int x, float *d)
{
    if constexpr (ND == 3) {
        // D is the depth of the volume SAMPLED (the full moving volume), z the absolute plane index
        const float iz = (float)z + fl[i], iy = (float)y + fl[nvox + i], ix = (float)x + fl[2 * nvox + i];
        Samp3 s = sample3(mov, D, H, W, ix, iy, iz);
        d[0] = s.dz; d[1] = s.dy; d[2] = s.dx;
        return s.v;
    } else {
        const float iy = (float)y + fl[i], ix = (float)x + fl[nvox + i];
        Samp2 s = sample2(mov, H, W, ix, iy);
        d[0] = s.dy; d[1] = s.dx;
        return s.v;
    }
}

// same, with the flow of this voxel already in registers (f[0..ND-1] in channel order)
template <int ND>
__device__ __forceinline__ float flow_sample_v(const float *__restrict__ mov, const float *f, int D, int H, int W, int z, int y, int x, float *d)
{
    if constexpr (ND == 3) {
        Samp3 s = sample3(mov, D, H, W, (float)x + f[2], (float)y + f[1], (float)z + f[0]);
        d[0] = s.dz; d[1] = s.dy; d[2] = s.dx;
        return s.v;
    } else {
        Samp2 s = sample2(mov, H, W, (float)x + f[1], (float)y + f[0]);
        d[0] = s.dy; d[1] = s.dx;
        return s.v;
    }
}

constexpr int kFlowNP = 8;  // 5 moments + up to 3 smoothness sums
constexpr int kLagRecord = 1, kLagPatch = 2, kLagFlush = 4;

template <int ND, bool SMOOTH>
__global__ __launch_bounds__(TRX_BLOCK) void flow_moments_kernel(trx_volumes vol, const float *__restrict__ flow,
                                                                 float *__restrict__ partials, Slab slab)
{
    const int b = blockIdx.y;
    const int D = vol.D, H = vol.H, W = vol.W;
    const size_t nvox = (size_t)D * H * W;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    const float *__restrict__ tgt = vol.target + (size_t)b * vol.target_stride;
    const float *__restrict__ fl = flow + (size_t)b * ND * nvox;
    float vals[kFlowNP] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t dstride[3] = {ND == 3 ? (size_t)H * W : (size_t)W, ND == 3 ? (size_t)W : 1, 1};
    // The flow and target of voxel k+1 are loaded before the gather of voxel k is consumed: the gather addresses depend on the
    // flow, so without this each iteration is two dependent memory round trips with only four loads in flight in the first.
    VoxelWalk vw(blockIdx.x * TRX_BLOCK + threadIdx.x, gridDim.x * TRX_BLOCK, H, W);
    float fc[3] = {0.f, 0.f, 0.f}, tc = 0.f;
    if (vw.i < nvox) {
#pragma unroll
        for (int c = 0; c < ND; c++) fc[c] = fl[c * nvox + vw.i];
        tc = tgt[vw.i];
    }
    while (vw.i < nvox) {
        const size_t i = vw.i;
        const int z = vw.z, y = vw.y, x = vw.x;
        vw.next(H, W);
        const size_t in = vw.i < nvox ? (size_t)vw.i : i;   // clamped: the loads stay unconditional
        float fn[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < ND; c++) fn[c] = fl[c * nvox + in];
        const float tn = tgt[in];
        float d[3];
        const float w = flow_sample_v<ND>(mov, fc, slab.Dm, H, W, z + slab.zoff, y, x, d);
        const float yv = tc;
#pragma unroll
        for (int c = 0; c < ND; c++) fc[c] = fn[c];
        tc = tn;
        vals[0] += yv; vals[1] += w;
        vals[2] = fmaf(yv, yv, vals[2]); vals[3] = fmaf(w, w, vals[3]); vals[4] = fmaf(yv, w, vals[4]);
        if constexpr (SMOOTH) {
            const int pos[3] = {ND == 3 ? z : y, ND == 3 ? y : x, x};
            const int ext[3] = {ND == 3 ? D : H, ND == 3 ? H : W, W};
#pragma unroll
            for (int dd = 0; dd < ND; dd++) {
                // forward difference; at the far face the neighbour index is the voxel itself (difference 0, no branch)
                const bool inner = pos[dd] + 1 < ext[dd];
                const size_t nb = i + (inner ? dstride[dd] : 0);
                const bool halo = (ND == 3 && dd == 0 && !inner && slab.halo_hi != nullptr);   // across the slab boundary: the lower slab counts it
#pragma unroll
                for (int c = 0; c < ND; c++) {
                    const float f0 = fl[c * nvox + i];
                    float fnb = fl[c * nvox + nb];
                    if (halo) fnb = slab.halo_hi[c * ((size_t)H * W) + (size_t)y * W + x];
                    const float df = fnb - f0;
                    vals[5 + dd] = fmaf(df, df, vals[5 + dd]);
                }
            }
        }
    }
    block_reduce_store<kFlowNP, 8>(vals, partials + ((size_t)b * gridDim.x + blockIdx.x) * kFlowNP);
}

__global__ __launch_bounds__(1024) void flow_coef_kernel(const float *__restrict__ partials, int nblk, int ndim, int D, int H,
                                                         int W, trx_loss_cfg lc, trx_opt_cfg oc, float smooth_weight,
                                                         float *__restrict__ losses, int losses_capacity, int *__restrict__ step,
                                                         float *__restrict__ terms, FlowCoef *__restrict__ coef,
                                                         double *__restrict__ mom_out, const double *__restrict__ mom_in, int D_full,
                                                         int lag = 0, double *__restrict__ stash = nullptr, float stop_crit = 0.f,
                                                         int *__restrict__ stopped = nullptr, int double_buffered = 0,
                                                         const float *__restrict__ ext_loss = nullptr)
{
    // ext_loss != NULL (the local-NCC loop, trx_flow_lncc_run): the data term of pair b is ext_loss[b] - a criterion evaluated by other
    // kernels, whose dL/dwarped the update reads from a buffer - instead of the closed forms of the five global moments; the smoothness
    // term, the optimiser scalars, the loss curve and the early stop are handled as for them.
    // lag (fused steps with the smoothness term, trx_flow_run): the data moments S[0..4] describe THIS iteration's flow, the
    // smoothness sums S[5..7] the PREVIOUS one (they are collected by the update kernel, which is where the neighbours of a flow are
    // read).  The gradient step needs only the former; the recorded loss of iteration t - 1 gets its regulariser term one
    // coefficient kernel later, from the data part stashed in fp64 - the same additions as the two-pass step, in the same order.
    //   kLagRecord: record the data part of this iteration (provisional) and stash it;  kLagPatch: complete losses[t - 1];
    //   kLagFlush: only that (after the last iteration of a run).
    __shared__ double acc[16][8];
    const int b = blockIdx.x, tid = threadIdx.x, k = tid & 7, grp = tid >> 3;  // 128 groups of 8
    double s = 0.0;
    // batches of 8 independent loads (all in flight together), fixed summation order
    for (int blk0 = grp; blk0 < nblk; blk0 += 8 * 128) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int blk = blk0 + i * 128;
            const float v = partials[((size_t)b * nblk + min(blk, nblk - 1)) * kFlowNP + k];   // clamped + select: a predicated load
            a[i] = (blk < nblk) ? v : 0.f;                                                     // compiles to a branch and a wait per load
        }
        s += (((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3])) + (((double)a[4] + (double)a[5]) + ((double)a[6] + (double)a[7]));
    }
    // reduce the 128 groups: lanes k, k+8, ... hold the same component
    for (int off = 32; off >= 8; off >>= 1) s += __shfl_down(s, off);
    if ((tid & 63) < 8) acc[tid >> 6][k] = s;
    __syncthreads();
    if (tid != 0) return;
    const int t_step = step ? step[b] : 0;   // issued early: its latency hides under the fp64 arithmetic below
    const int stp = stopped ? stopped[b] : 0;
    double S[8];
    for (int j = 0; j < 8; j++) {
        double t = 0.0;
        for (int wv = 0; wv < 16; wv++) t += acc[wv][j];
        S[j] = t;
    }
    if (mom_out) {               // slab mode, pass A: publish this rank's raw sums (the caller all-reduces them)
        for (int j = 0; j < 8; j++) mom_out[b * 8 + j] = S[j];
        return;
    }
    if (mom_in)                  // slab mode, pass B: sums of the WHOLE volume
        for (int j = 0; j < 8; j++) S[j] = mom_in[b * 8 + j];
    if (stp != 0) {              // this pair has stopped: nothing is recorded any more, the update kernel only settles the buffers
        if (lag & kLagFlush) return;
        coef[b].mode = (stp == 1 && double_buffered) ? kUpdCopy : kUpdSkip;
        stopped[b] = 2;
        return;
    }
    D = D_full;
    const double n = (double)D * H * W;
    const double Sy = S[0], Sw = S[1], Syy = S[2], Sww = S[3], Syw = S[4];
    const double my = Sy / n, mw = Sw / n;
    const double Saa = Syy - Sy * my, Sbb = Sww - Sw * mw, Sab = Syw - Sy * mw;
    const double sd = sqrt(Saa * Sbb + 1e-10);
    const double alpha = lc.ncc_alpha, sq = Syy - 2.0 * Syw + Sww;
    const double mse = sq / n, ncc = alpha * (1.0 - Sab / sd), ssd = (double)lc.ssd_alpha * sq;
    double total = ext_loss ? (double)ext_loss[b] : (double)lc.w_mse * mse + (double)lc.w_ncc * ncc + (double)lc.w_ssd * ssd;
    // smoothness (extension): lambda/ndim * sum_d mean_{c,p}(forward difference along d)^2
    const int ext[3] = {ndim == 3 ? D : H, ndim == 3 ? H : W, W};
    FlowCoef c;
    c.mode = kUpdNormal;
    c.sm[0] = c.sm[1] = c.sm[2] = 0.f;
    if (smooth_weight != 0.f) {
        double reg = 0.0;
        for (int dd = 0; dd < ndim; dd++) {
            const double cnt = (double)ndim * n / ext[dd] * (ext[dd] - 1);
            if (cnt > 0) {
                reg += S[5 + dd] / cnt;
                c.sm[dd] = (float)((double)smooth_weight / ndim * 2.0 / cnt);
            }
        }
        if (lag == 0) {
            total += (double)smooth_weight / ndim * reg;
        } else {
            const int tp = t_step - 1;
            if ((lag & kLagPatch) && tp >= 0) {
                const float done = (float)(stash[b] + (double)smooth_weight / ndim * reg);   // the complete loss of iteration t - 1
                if (losses && tp < losses_capacity) losses[(size_t)b * losses_capacity + tp] = done;
                if (stopped && done <= stop_crit) {   // the reference would have left its loop after the update of iteration t - 1: that is now
                    if (lag & kLagFlush) { stopped[b] = 1; return; }   // (the run ends here anyway; the next call settles the buffers)
                    coef[b].mode = kUpdTransition;
                    stopped[b] = 2;
                    return;
                }
            }
            if (lag & kLagFlush) return;
            stash[b] = total;
        }
    }
    c.k1 = (float)((double)lc.w_ncc * (-alpha / sd));
    c.k2 = (float)((double)lc.w_ncc * (alpha * Sab * Saa / (sd * sd * sd)));
    c.my = (float)my; c.mw = (float)mw;
    c.q = (float)((double)lc.w_mse * 2.0 / n + (double)lc.w_ssd * (double)lc.ssd_alpha * 2.0);
    int t = t_step;
    if (oc.kind == TRX_OPT_ADAM) {
        const double bc1 = 1.0 - ipow((double)oc.beta1, t + 1), bc2 = 1.0 - ipow((double)oc.beta2, t + 1);
        c.step_size = (float)((double)oc.lr / bc1);
        c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    } else {
        c.step_size = oc.lr;
        c.inv_sqrt_bc2 = 1.f;
    }
    // early stop on the complete loss of THIS iteration (a lagged regulariser term is tested one coefficient kernel later, above)
    if (stopped && lag == 0 && (float)total <= stop_crit) {
        c.mode = kUpdHit;
        stopped[b] = 1;
    }
    coef[b] = c;
    if (losses && t < losses_capacity) losses[(size_t)b * losses_capacity + t] = (float)total;
    if (step) step[b] = t + 1;
    if (terms) {
        terms[b * 4 + 0] = (float)total; terms[b * 4 + 1] = (float)mse;
        terms[b * 4 + 2] = (float)ncc;   terms[b * 4 + 3] = (float)ssd;
    }
}

// MODE 0: optimiser update (SGD/Adam) written to flow_out;  MODE 1: write the gradient to flow_out
// PIPE: read voxel k+1's flow / target ahead of voxel k's gather (plain SGD without the regulariser: -9 %; the Adam and smoothness
// variants hold more state per voxel and lose 3-4 % to it, so they load at the point of use).
// NEXT: pass A of the FOLLOWING iteration rides along - the voxel is sampled once more at its updated flow and the five moments go
// to `next_partials` in exactly the layout, voxel order and arithmetic of flow_moments_kernel, so the next iteration starts at its
// coefficient kernel.  One gather more, one 20 B/voxel pass less.  With the smoothness term the three regulariser sums in
// `next_partials` are those of the flow this step STARTS from (its neighbours are read here anyway); see flow_coef_kernel's lag mode.
template <int ND, int MODE, bool SMOOTH, bool PIPE = false, bool NEXT = false>
__global__ __launch_bounds__(TRX_BLOCK) void flow_update_kernel(trx_volumes vol, const float *flow,
                                                                float *flow_out, float *__restrict__ adam_m,
                                                                float *__restrict__ adam_v, const FlowCoef *__restrict__ coef,
                                                                trx_opt_cfg oc, Slab slab, float *__restrict__ next_partials = nullptr,
                                                                float *__restrict__ flow_last = nullptr, int save_last = 0)
{
    static_assert(!NEXT || MODE == 0, "the fused next-iteration moments ride on the update");
    float nv[kFlowNP] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int b = blockIdx.y;
    const int D = vol.D, H = vol.H, W = vol.W;
    const size_t nvox = (size_t)D * H * W;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    const float *__restrict__ tgt = vol.target + (size_t)b * vol.target_stride;
    const float *fl = flow + (size_t)b * ND * nvox;   // may alias fo (in-place update): no restrict
    float *fo = flow_out + (size_t)b * ND * nvox;
    const FlowCoef c = coef[b];
    float *__restrict__ fkeep = nullptr;   // != nullptr: this update also keeps the flow it starts from (the flow of the last forward)
    if constexpr (MODE == 0) {
        if (c.mode >= kUpdCopy) {          // a pair that has stopped early (block-uniform): settle the buffers, no arithmetic
            if (c.mode != kUpdSkip && fo != fl) {
                float *__restrict__ keep = (c.mode == kUpdTransition && flow_last) ? flow_last + (size_t)b * ND * nvox : nullptr;
                for (size_t i = (size_t)blockIdx.x * TRX_BLOCK + threadIdx.x; i < (size_t)ND * nvox; i += (size_t)gridDim.x * TRX_BLOCK) {
                    if (keep) keep[i] = fo[i];
                    fo[i] = fl[i];
                }
            }
            return;   // (a fused update leaves next_partials as they are: the coefficient kernel of a stopped pair does not read them)
        }
        if (flow_last && (save_last || c.mode == kUpdHit)) fkeep = flow_last + (size_t)b * ND * nvox;
    }
    const size_t dstride[3] = {ND == 3 ? (size_t)H * W : (size_t)W, ND == 3 ? (size_t)W : 1, 1};
    // As in pass A: the flow / target of voxel k+1 are requested before the gather of voxel k is
    // consumed (each thread owns its voxels, so reading ahead of the in-place update is safe).
    const bool adam = (MODE != 1) && (oc.kind == TRX_OPT_ADAM);
    float *__restrict__ am = adam_m + (size_t)b * ND * nvox, *__restrict__ av = adam_v + (size_t)b * ND * nvox;
    VoxelWalk vw(blockIdx.x * TRX_BLOCK + threadIdx.x, gridDim.x * TRX_BLOCK, H, W);
    float fc[3] = {0.f, 0.f, 0.f}, tc = 0.f;
    if (PIPE && vw.i < nvox) {
#pragma unroll
        for (int ch = 0; ch < ND; ch++) fc[ch] = fl[ch * nvox + vw.i];
        tc = tgt[vw.i];
    }
    while (vw.i < nvox) {
        const size_t i = vw.i;
        const int z = vw.z, y = vw.y, x = vw.x;
        vw.next(H, W);
        const size_t in = !PIPE ? i : (vw.i < nvox ? (size_t)vw.i : i);   // PIPE: the next voxel (clamped); else this one
        float fnx[3] = {0.f, 0.f, 0.f}, m0[3] = {0.f, 0.f, 0.f}, v0[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < ND; ch++) fnx[ch] = fl[ch * nvox + in];
        const float tn = tgt[in];
        if constexpr (!PIPE) {
#pragma unroll
            for (int ch = 0; ch < ND; ch++) fc[ch] = fnx[ch];
            tc = tn;
        }
        float d[3];
        const float w = flow_sample_v<ND>(mov, fc, slab.Dm, H, W, z + slab.zoff, y, x, d);
        const float yv = tc;
        const float fcur[3] = {fc[0], fc[1], fc[2]};
#pragma unroll
        for (int ch = 0; ch < ND; ch++) fc[ch] = fnx[ch];
        tc = tn;
        const float go = fmaf(c.k1, yv - c.my, fmaf(c.k2, w - c.mw, c.q * (w - yv)));
        float pnew[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < ND; ch++) {
            float g = go * d[ch];
            const float f0 = fcur[ch];
            if constexpr (SMOOTH) {
                const int pos[3] = {ND == 3 ? z : y, ND == 3 ? y : x, x};
                const int ext[3] = {ND == 3 ? D : H, ND == 3 ? H : W, W};
#pragma unroll
                for (int dd = 0; dd < ND; dd++) {
                    // (f0 - f[lo]) - (f[hi] - f0); at a face the neighbour index is the voxel itself (term 0, no branch)
                    const bool has_lo = pos[dd] > 0, has_hi = pos[dd] + 1 < ext[dd];
                    float flo = fl[ch * nvox + i - (has_lo ? dstride[dd] : 0)];
                    float fhi = fl[ch * nvox + i + (has_hi ? dstride[dd] : 0)];
                    if (ND == 3 && dd == 0) {   // Z-slab partition: the neighbour plane lives on another rank
                        if (!has_lo && slab.halo_lo) flo = slab.halo_lo[ch * (size_t)H * W + (size_t)y * W + x];
                        if (!has_hi && slab.halo_hi) fhi = slab.halo_hi[ch * (size_t)H * W + (size_t)y * W + x];
                    }
                    g = fmaf(c.sm[dd], (f0 - flo) - (fhi - f0), g);
                    if constexpr (NEXT) {   // smoothness sums of the flow this step STARTS from (flow_moments_kernel's arithmetic and order)
                        const float df = fhi - f0;
                        nv[5 + dd] = fmaf(df, df, nv[5 + dd]);
                    }
                }
            }
            if constexpr (MODE == 1) {
                fo[ch * nvox + i] = g;
            } else {
                float p = f0;
                if (adam) {
                    m0[ch] = am[ch * nvox + i]; v0[ch] = av[ch * nvox + i];
                    const float mi = m0[ch] + (g - m0[ch]) * (1.0f - oc.beta1);
                    const float vi = oc.beta2 * v0[ch] + (1.0f - oc.beta2) * g * g;
                    am[ch * nvox + i] = mi; av[ch * nvox + i] = vi;
                    const float denom = sqrtf(vi) * c.inv_sqrt_bc2 + oc.eps;
                    p = p - c.step_size * (mi / denom);
                } else {
                    p = p - c.step_size * g;
                }
                fo[ch * nvox + i] = p;
                if (fkeep) fkeep[ch * nvox + i] = f0;
                if constexpr (NEXT) pnew[ch] = p;
            }
        }
        if constexpr (NEXT) {
            float d2[3];
            const float w2 = flow_sample_v<ND>(mov, pnew, slab.Dm, H, W, z + slab.zoff, y, x, d2);
            nv[0] += yv; nv[1] += w2;
            nv[2] = fmaf(yv, yv, nv[2]); nv[3] = fmaf(w2, w2, nv[3]); nv[4] = fmaf(yv, w2, nv[4]);
        }
    }
    if constexpr (NEXT) block_reduce_store<kFlowNP, 8>(nv, next_partials + ((size_t)b * gridDim.x + blockIdx.x) * kFlowNP);
}


// ---------------------------------------------------------------------------------------------------
// 3-D passes as COLUMN WALKS (round 3).  The grid-stride kernels above (still the 2-D path) spend ~490 vector instructions per
// voxel-wave, most of them 64-bit index arithmetic and selects for the 18 neighbour loads of the smoothness term, and every XCD
// re-fetches the z neighbours of its voxels from the fabric (rocprofv3: 128 M VALU instructions and 1.07 GB of L2 read misses per
// 256^3 Adam + smoothness update against 0.74 GB of unique reads).  Here a 256-thread block owns a (64 x, 4 y) column of one z
// segment - wave w = row y0 + w, lane = x - and every thread walks its (x, y) along z: the voxel index advances by one plane per trip,
// the z neighbours of the flow are the previous and the next trip's own loads (kept in registers), the y / x neighbours are the other
// waves' / lanes' lines (L1), and the XCD-aware block order keeps a column's y neighbours in its L2.  Per-voxel arithmetic, the order
// of every sum and the partial-row layout are those of the kernels above, so the coefficient kernel and the slab entry points are
// unchanged and a fused update still leaves the bits a stand-alone moments pass would.
// ---------------------------------------------------------------------------------------------------
struct ColGeom {
    int ntx, nty, nzseg, planes_per_seg, nblk;
    int colw_log2;   // a block covers 2^colw_log2 (x) by TRX_BLOCK >> colw_log2 (y) columns: 64 x 4, 128 x 2 or 256 x 1
};

// The eight corners of a trilinear sample, fetched now and interpolated later (the column walk requests the corners of trip z + 1 while
// trip z computes).  Same loads and the same lerp3 as sample3 / sample3_padded (trx_common.h): bit-identical values.
struct Corners3 {
    float v[8], tx, ty, tz;
};
__device__ __forceinline__ Corners3 load_corners3(const float *__restrict__ mov, int D, int H, int W, float ix, float iy, float iz)
{
    Corners3 c;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    c.tx = ix - fx; c.ty = iy - fy; c.tz = iz - fz;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const bool interior = ((unsigned)x0 < (unsigned)(W - 1)) & ((unsigned)y0 < (unsigned)(H - 1)) & ((unsigned)z0 < (unsigned)(D - 1));
    const size_t HW = (size_t)H * W;
    if (__all(interior)) {
        const float *p = mov + ((size_t)z0 * H + y0) * W + x0, *q = p + HW;
        c.v[0] = p[0]; c.v[1] = p[1]; c.v[2] = p[W]; c.v[3] = p[W + 1];
        c.v[4] = q[0]; c.v[5] = q[1]; c.v[6] = q[W]; c.v[7] = q[W + 1];
        return c;
    }
    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const bool bx0 = (unsigned)x0 < (unsigned)W, bx1 = (unsigned)x1 < (unsigned)W;
    const bool by0 = (unsigned)y0 < (unsigned)H, by1 = (unsigned)y1 < (unsigned)H;
    const bool bz0 = (unsigned)z0 < (unsigned)D, bz1 = (unsigned)z1 < (unsigned)D;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    const int cz0 = min(max(z0, 0), D - 1), cz1 = min(max(z1, 0), D - 1);
    const float *r00 = mov + ((size_t)cz0 * H + cy0) * W, *r01 = mov + ((size_t)cz0 * H + cy1) * W;
    const float *r10 = mov + ((size_t)cz1 * H + cy0) * W, *r11 = mov + ((size_t)cz1 * H + cy1) * W;
    c.v[0] = (bz0 & by0 & bx0) ? r00[cx0] : 0.f; c.v[1] = (bz0 & by0 & bx1) ? r00[cx1] : 0.f;
    c.v[2] = (bz0 & by1 & bx0) ? r01[cx0] : 0.f; c.v[3] = (bz0 & by1 & bx1) ? r01[cx1] : 0.f;
    c.v[4] = (bz1 & by0 & bx0) ? r10[cx0] : 0.f; c.v[5] = (bz1 & by0 & bx1) ? r10[cx1] : 0.f;
    c.v[6] = (bz1 & by1 & bx0) ? r11[cx0] : 0.f; c.v[7] = (bz1 & by1 & bx1) ? r11[cx1] : 0.f;
    return c;
}
// value and derivative in flow-channel order (z, y, x) like flow_sample_v<3>
__device__ __forceinline__ float lerp_corners3(const Corners3 &c, float *d)
{
    const Samp3 s = lerp3(c.v[0], c.v[1], c.v[2], c.v[3], c.v[4], c.v[5], c.v[6], c.v[7], c.tx, c.ty, c.tz);
    d[0] = s.dz; d[1] = s.dy; d[2] = s.dx;
    return s.v;
}
// Streams of the update pass that are touched once per iteration - the optimiser state (read, written), the updated flow (written), the
// target (read) - carry the non-temporal hint: they are 1.3 GB a pass, nothing of them survives in the caches until the next pass, and
// without the hint they evict the moving volume and the flow planes that the gathers and the regulariser's neighbours DO re-read
// (256^3, with the row-wide blocks below: Adam 310 -> 247 ... 265 us per iteration depending on the box, Adam + smoothness 338 -> 303; profiles/r04h_flow_variants.txt).
#ifndef TRX_FLOW_DBG
#define TRX_FLOW_DBG 0
#endif
#ifndef TRX_FLOW_NT
#define TRX_FLOW_NT 7      // bit 0: stores of flow / m / v, bit 1: loads of m / v, bit 2: loads of the target (0: development baseline)
#endif
template <typename T> __device__ __forceinline__ void st_stream(T *p, T v) { if constexpr (TRX_FLOW_NT & 1) __builtin_nontemporal_store(v, p); else *p = v; }
template <typename T> __device__ __forceinline__ T ld_stream(const T *p) { if constexpr (TRX_FLOW_NT & 2) return __builtin_nontemporal_load(p); else return *p; }
template <typename T> __device__ __forceinline__ T ld_stream4(const T *p) { if constexpr (TRX_FLOW_NT & 4) return __builtin_nontemporal_load(p); else return *p; }

// Block shape: as wide in x as the rows allow without idle lanes - every stream then moves 1 KB of one row per block and trip (64 x 4
// blocks moved four 256-byte pieces of four rows: 5 % slower at W = 256) - falling back to narrower blocks when W is not a multiple.
#ifndef TRX_FLOW_BLOCKS
#define TRX_FLOW_BLOCKS 2048
#endif
static ColGeom flow_col_geom(const trx_volumes &v)
{
    ColGeom g;
    g.colw_log2 = 6;
    long best = ((long)v.W + 63) / 64 * 64;
    for (int l2 = 7; l2 <= 8; l2++) {
        const long w = 1L << l2, padded = ((long)v.W + w - 1) / w * w;
        if (padded <= best && (TRX_BLOCK >> l2) >= 1) { best = padded; g.colw_log2 = l2; }
    }
    const int colw = 1 << g.colw_log2, rows = TRX_BLOCK >> g.colw_log2;
    g.ntx = (v.W + colw - 1) / colw; g.nty = (v.H + rows - 1) / rows;
    const long cols = (long)g.ntx * g.nty * v.B;
    long want = (TRX_FLOW_BLOCKS + cols - 1) / cols; // ~2048 blocks in flight (8 per CU at the kernels' 7-8 waves per SIMD)
    if (want > v.D / 8) want = v.D / 8;              // a segment start costs one extra plane of flow loads
    if (want < 1) want = 1;
    g.planes_per_seg = (int)((v.D + want - 1) / want);
    g.nzseg = (v.D + g.planes_per_seg - 1) / g.planes_per_seg;
    g.nblk = g.ntx * g.nty * g.nzseg;
    return g;
}

struct ColWalk {
    int x, y, z0, z1;
    bool active;
    unsigned i;   // voxel index of (z0, y, x)
    __device__ __forceinline__ ColWalk(const ColGeom &g, int D, int H, int W)
    {
        int l = blockIdx.x;
        if ((g.nblk & 7) == 0) l = (l & 7) * (g.nblk >> 3) + (l >> 3);   // blocks b, b + 8, ... share an XCD: give it a contiguous run of columns
        const int ncol = g.ntx * g.nty, zs = l / ncol, c = l - zs * ncol, ty = c / g.ntx, tx = c - ty * g.ntx;
        const int cl2 = g.colw_log2;
        x = (tx << cl2) + (int)(threadIdx.x & ((1u << cl2) - 1u)); y = ty * (TRX_BLOCK >> cl2) + (int)(threadIdx.x >> cl2);
        z0 = zs * g.planes_per_seg; z1 = min(z0 + g.planes_per_seg, D);
        active = (x < W) && (y < H) && (z0 < z1);
        i = active ? (unsigned)((z0 * H + y) * W + x) : 0u;
    }
};

template <bool SMOOTH>
__global__ __launch_bounds__(TRX_BLOCK) void flow_moments3_kernel(trx_volumes vol, const float *__restrict__ flow, float *__restrict__ partials, Slab slab,
                                                                  ColGeom cg, float *__restrict__ warped_out = nullptr)
{
    const int b = blockIdx.y;
    const int D = vol.D, H = vol.H, W = vol.W;
    const unsigned HW = (unsigned)(H * W), nvox = (unsigned)D * HW;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    const float *__restrict__ tgt = vol.target + (size_t)b * vol.target_stride;
    const float *__restrict__ f0p = flow + (size_t)b * 3 * nvox, *__restrict__ f1p = f0p + nvox, *__restrict__ f2p = f1p + nvox;
    float vals[kFlowNP] = {0, 0, 0, 0, 0, 0, 0, 0};
    const ColWalk cw(cg, D, H, W);
    if (cw.active) {
        const int x = cw.x, y = cw.y;
        const bool y_hi = y + 1 < H, x_hi = x + 1 < W;
        unsigned i = cw.i;
        float fc[3] = {f0p[i], f1p[i], f2p[i]}, tc = tgt[i];
        for (int z = cw.z0; z < cw.z1; z++, i += HW) {
            const bool z_hi = z + 1 < D;
            const unsigned in = i + (z_hi ? HW : 0u);                  // the next plane's voxel (this one again at the far face: difference 0)
            float fn[3] = {f0p[in], f1p[in], f2p[in]};
            const float tn = tgt[in];
            float d[3];
            const float w = flow_sample_v<3>(mov, fc, slab.Dm, H, W, z + slab.zoff, y, x, d);
            if (warped_out) warped_out[(size_t)b * nvox + i] = w;   // (the local-NCC loop: this pass is also its forward warp)
            const float yv = tc;
            vals[0] += yv; vals[1] += w;
            vals[2] = fmaf(yv, yv, vals[2]); vals[3] = fmaf(w, w, vals[3]); vals[4] = fmaf(yv, w, vals[4]);
            if constexpr (SMOOTH) {
                // forward differences; dd = 0 (z): the next plane is in registers, across a slab boundary the lower slab counts it
                float fz[3] = {fn[0], fn[1], fn[2]};
                if (!z_hi && slab.halo_hi != nullptr) {
                    const size_t hi = (size_t)y * W + x;
                    fz[0] = slab.halo_hi[hi]; fz[1] = slab.halo_hi[(size_t)HW + hi]; fz[2] = slab.halo_hi[2 * (size_t)HW + hi];
                }
                const unsigned iy = i + (y_hi ? (unsigned)W : 0u), ix = i + (x_hi ? 1u : 0u);
                const float fy[3] = {f0p[iy], f1p[iy], f2p[iy]}, fx[3] = {f0p[ix], f1p[ix], f2p[ix]};
#pragma unroll
                for (int c = 0; c < 3; c++) { const float df = fz[c] - fc[c]; vals[5] = fmaf(df, df, vals[5]); }
#pragma unroll
                for (int c = 0; c < 3; c++) { const float df = fy[c] - fc[c]; vals[6] = fmaf(df, df, vals[6]); }
#pragma unroll
                for (int c = 0; c < 3; c++) { const float df = fx[c] - fc[c]; vals[7] = fmaf(df, df, vals[7]); }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) fc[c] = fn[c];
            tc = tn;
        }
    }
    block_reduce_store<kFlowNP, 8>(vals, partials + ((size_t)b * gridDim.x + blockIdx.x) * kFlowNP);
}

// MODE 0: optimiser update (SGD / Adam) written to flow_out;  MODE 1: write the gradient to flow_out.  NEXT: see flow_update_kernel.
// Registers: the smoothness variants hold a trip of prefetched state (up to 18 values) next to the gathers' operands - four waves per
// SIMD with all of it in registers beat five with part of the prefetch undone (256^3: 367 against 407 us); the lighter variants fit 5-8.
template <int MODE, bool SMOOTH, bool NEXT, bool ADAM>
__global__ __launch_bounds__(TRX_BLOCK, SMOOTH ? 4 : 5) void flow_update3_kernel(trx_volumes vol, const float *flow, float *flow_out, float *__restrict__ adam_m,
                                                                 float *__restrict__ adam_v, const FlowCoef *__restrict__ coef, trx_opt_cfg oc, Slab slab,
                                                                 ColGeom cg, float *__restrict__ next_partials, float *__restrict__ flow_last, int save_last,
                                                                 const float *__restrict__ go_buf = nullptr)
{
    // MODE 2 (the local-NCC loop): the optimiser update of MODE 0 with dL/dwarped of every voxel read from go_buf [B][D][H][W]
    // instead of the closed form k1 (y - my) + k2 (w - mw) + q (w - y) of the global losses
    static_assert(!NEXT || MODE == 0, "the fused next-iteration moments ride on the update");
    float nv[kFlowNP] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int b = blockIdx.y;
    const int D = vol.D, H = vol.H, W = vol.W;
    const unsigned HW = (unsigned)(H * W), nvox = (unsigned)D * HW;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    const float *__restrict__ tgt = (MODE == 2) ? go_buf + (size_t)b * nvox : vol.target + (size_t)b * vol.target_stride;
    const float *fl = flow + (size_t)b * 3 * nvox;   // may alias fo (in-place update without the regulariser): no restrict
    float *fo = flow_out + (size_t)b * 3 * nvox;
    const FlowCoef c = coef[b];
    float *__restrict__ fkeep = nullptr;   // != nullptr: this update also keeps the flow it starts from (the flow of the last forward)
    if constexpr (MODE == 0 || MODE == 2) {
        if (c.mode >= kUpdCopy) {          // a pair that has stopped early (block-uniform): settle the buffers, no arithmetic
            if (c.mode != kUpdSkip && fo != fl) {
                float *__restrict__ keep = (c.mode == kUpdTransition && flow_last) ? flow_last + (size_t)b * 3 * nvox : nullptr;
                for (size_t i = (size_t)blockIdx.x * TRX_BLOCK + threadIdx.x; i < (size_t)3 * nvox; i += (size_t)gridDim.x * TRX_BLOCK) {
                    if (keep) keep[i] = fo[i];
                    fo[i] = fl[i];
                }
            }
            return;   // (a fused update leaves next_partials as they are: the coefficient kernel of a stopped pair does not read them)
        }
        if (flow_last && (save_last || c.mode == kUpdHit)) fkeep = flow_last + (size_t)b * 3 * nvox;
    }
    constexpr bool adam = (MODE != 1) && ADAM;
    float *__restrict__ am = adam_m + (size_t)b * 3 * nvox, *__restrict__ av = adam_v + (size_t)b * 3 * nvox;
    const ColWalk cw(cg, D, H, W);
    if (cw.active) {
        const int x = cw.x, y = cw.y;
        const unsigned dyl = y > 0 ? (unsigned)W : 0u, dyh = y + 1 < H ? (unsigned)W : 0u, dxl = x > 0 ? 1u : 0u, dxh = x + 1 < W ? 1u : 0u;
        const size_t hidx = (size_t)y * W + x;
        unsigned i = cw.i;
        float fc[3], fm[3] = {0.f, 0.f, 0.f}, tc = ld_stream4(tgt + i);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) fc[ch] = fl[ch * (size_t)nvox + i];
        if constexpr (SMOOTH) {   // the plane below the segment's first: the flow itself, the lower slab's plane, or (first plane of the volume) the voxel itself
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                fm[ch] = cw.z0 > 0 ? fl[ch * (size_t)nvox + i - HW] : (slab.halo_lo ? slab.halo_lo[ch * (size_t)HW + hidx] : fc[ch]);
        }
        // Everything a trip needs besides the gather has an address that is known one trip ahead (the walk is i += H W): the optimiser
        // state and the in-plane neighbours of trip z + 1 are requested during trip z, next to the flow / target of z + 1 - one memory
        // round trip per trip (the gather's) instead of three dependent ones.
        struct Side { float m[3], v[3], yl[3], yh[3], xl[3], xh[3]; };
        auto load_side = [&](unsigned iv) {
            Side sd;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                sd.m[ch] = sd.v[ch] = sd.yl[ch] = sd.yh[ch] = sd.xl[ch] = sd.xh[ch] = 0.f;
                if (adam) { sd.m[ch] = ld_stream(am + ch * (size_t)nvox + iv); sd.v[ch] = ld_stream(av + ch * (size_t)nvox + iv); }
                if constexpr (SMOOTH) {
                    const float *fch = fl + ch * (size_t)nvox;
#if TRX_FLOW_DBG & 1   // development ablation: no y-neighbour loads (wrong results: what would removing them buy?)
                    sd.yl[ch] = sd.yh[ch] = 0.5f;
#else
                    sd.yl[ch] = fch[iv - dyl]; sd.yh[ch] = fch[iv + dyh];
#endif
#if TRX_FLOW_DBG & 2   // ... no x-neighbour loads
                    sd.xl[ch] = sd.xh[ch] = 0.25f;
#else
                    sd.xl[ch] = fch[iv - dxl]; sd.xh[ch] = fch[iv + dxh];
#endif
                }
            }
            return sd;
        };
        Side cur = load_side(i);
        // ... and so has the gather of trip z + 1 once the flow of plane z + 1 has arrived (it was requested a trip earlier): its eight
        // corners are requested before trip z computes.  Only the second sample of a fused step (at the flow this trip writes) waits.
        auto corners_at = [&](const float (&f)[3], int zz) {
            return load_corners3(mov, slab.Dm, H, W, (float)x + f[2], (float)y + f[1], (float)(zz + slab.zoff) + f[0]);
        };
        Corners3 gc = corners_at(fc, cw.z0);
        Corners3 g2 = gc;   // (NEXT) corners of the previous trip's second sample, and its target value
        float y2 = 0.f;
        auto fold_next = [&](const Corners3 &g, float yq) {
            float d2[3];
            const float w2 = lerp_corners3(g, d2);
            nv[0] += yq; nv[1] += w2;
            nv[2] = fmaf(yq, yq, nv[2]); nv[3] = fmaf(w2, w2, nv[3]); nv[4] = fmaf(yq, w2, nv[4]);
        };
        for (int z = cw.z0; z < cw.z1; z++, i += HW) {
            const bool z_hi = z + 1 < D;
            const unsigned in = i + (z_hi ? HW : 0u);
            float fn[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                if constexpr ((TRX_FLOW_NT & 8) && !SMOOTH) fn[ch] = __builtin_nontemporal_load(fl + ch * (size_t)nvox + in);
                else fn[ch] = fl[ch * (size_t)nvox + in];
            }
            const float tn = ld_stream4(tgt + in);
            const Side nxt = load_side(in);   // (the last trip of the volume re-reads its own voxel: harmless, discarded)
            float d[3];
            const float w = lerp_corners3(gc, d);
            const float yv = tc;
            const float go = (MODE == 2) ? tc : fmaf(c.k1, yv - c.my, fmaf(c.k2, w - c.mw, c.q * (w - yv)));   // MODE 2: `tgt` IS go_buf (see below)
            float pnew[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                float g = go * d[ch];
                const float f0 = fc[ch];
                if constexpr (SMOOTH) {
                    // (f0 - f[lo]) - (f[hi] - f0) per axis; at a face the neighbour is the voxel itself (term 0), across a slab boundary the halo plane
                    const float zlo = fm[ch];
                    float zhi = fn[ch];
                    if (!z_hi && slab.halo_hi) zhi = slab.halo_hi[ch * (size_t)HW + hidx];
                    g = fmaf(c.sm[0], (f0 - zlo) - (zhi - f0), g);
                    g = fmaf(c.sm[1], (f0 - cur.yl[ch]) - (cur.yh[ch] - f0), g);
                    g = fmaf(c.sm[2], (f0 - cur.xl[ch]) - (cur.xh[ch] - f0), g);
                    if constexpr (NEXT) {   // smoothness sums of the flow this step STARTS from (flow_moments3_kernel's arithmetic and order)
                        const float dz = zhi - f0, dy = cur.yh[ch] - f0, dx = cur.xh[ch] - f0;
                        nv[5] = fmaf(dz, dz, nv[5]); nv[6] = fmaf(dy, dy, nv[6]); nv[7] = fmaf(dx, dx, nv[7]);
                    }
                }
                if constexpr (MODE == 1) {
                    fo[ch * (size_t)nvox + i] = g;
                } else {
                    float p = f0;
                    if (adam) {
                        const float m0 = cur.m[ch], v0 = cur.v[ch];
                        const float mi = m0 + (g - m0) * (1.0f - oc.beta1);
                        const float vi = oc.beta2 * v0 + (1.0f - oc.beta2) * g * g;
                        st_stream(am + ch * (size_t)nvox + i, mi); st_stream(av + ch * (size_t)nvox + i, vi);
                        const float denom = sqrtf(vi) * c.inv_sqrt_bc2 + oc.eps;
                        p = p - c.step_size * (mi / denom);
                    } else {
                        p = p - c.step_size * g;
                    }
                    st_stream(fo + ch * (size_t)nvox + i, p);
                    if (fkeep) fkeep[ch * (size_t)nvox + i] = f0;
                    if constexpr (NEXT) pnew[ch] = p;
                }
            }
            if constexpr (NEXT) {
                // The sample at the flow this trip has just written (the next iteration's moments) is the one gather whose address no
                // earlier trip knows: its corners are requested here and folded in ONE TRIP LATER - same values, same order of sums.
                if (z > cw.z0) fold_next(g2, y2);
                g2 = corners_at(pnew, z);
                y2 = yv;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) { fm[ch] = fc[ch]; fc[ch] = fn[ch]; }
            tc = tn;
            cur = nxt;
            if (z + 1 < cw.z1) gc = corners_at(fc, z + 1);   // fc: the flow of plane z + 1, requested at the top of this trip
        }
        if constexpr (NEXT) fold_next(g2, y2);
    }
    if constexpr (NEXT) block_reduce_store<kFlowNP, 8>(nv, next_partials + ((size_t)b * gridDim.x + blockIdx.x) * kFlowNP);
}

template <int ND>
__global__ __launch_bounds__(TRX_BLOCK) void flow_warp_kernel(trx_volumes vol, const float *__restrict__ flow, int channels,
                                                              float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int D = vol.D, H = vol.H, W = vol.W;
    const size_t nvox = (size_t)D * H * W;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    const float *__restrict__ fl = flow + (size_t)b * ND * nvox;
    float *__restrict__ o = out + (size_t)b * channels * nvox;
    VoxelWalk vw(blockIdx.x * TRX_BLOCK + threadIdx.x, gridDim.x * TRX_BLOCK, H, W);
    float fc[3] = {0.f, 0.f, 0.f};
    if (vw.i < nvox) {
#pragma unroll
        for (int c = 0; c < ND; c++) fc[c] = fl[c * nvox + vw.i];
    }
    while (vw.i < nvox) {   // the next voxel's flow is requested before this voxel's gather is consumed (see flow_moments_kernel)
        const size_t i = vw.i;
        const int z = vw.z, y = vw.y, x = vw.x;
        vw.next(H, W);
        const size_t in = vw.i < nvox ? (size_t)vw.i : i;
        float fnx[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < ND; c++) fnx[c] = fl[c * nvox + in];
        float d[3];
        for (int ch = 0; ch < channels; ch++) o[ch * nvox + i] = flow_sample_v<ND>(mov + ch * nvox, fc, D, H, W, z, y, x, d);
#pragma unroll
        for (int c = 0; c < ND; c++) fc[c] = fnx[c];
    }
}

// SpatialTransformer(mode='nearest') (ref:utils.py:339-365 hands `mode` to grid_sample; Attention_UNet's default, ref:utils.py:409-410,520):
// out = src[rint(voxel + flow)] with zeros outside - grid_sample's nearest mode rounds half to even (std::nearbyint) and tests the
// rounded index against the volume.  The reference's normalise / un-normalise round trip (ref:utils.py:354-356, align_corners=True) moves
// a coordinate by ~1e-7 of the axis length, which matters only within that distance of a half-integer; like the bilinear kernels this one
// works in voxel space.  No gradient wrt the flow exists (ATen returns zeros for it).
template <int ND>
__global__ __launch_bounds__(TRX_BLOCK) void flow_warp_nearest_kernel(trx_volumes vol, const float *__restrict__ flow, int channels, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int D = vol.D, H = vol.H, W = vol.W;
    const size_t nvox = (size_t)D * H * W;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    const float *__restrict__ fl = flow + (size_t)b * ND * nvox;
    float *__restrict__ o = out + (size_t)b * channels * nvox;
    for (VoxelWalk vw(blockIdx.x * TRX_BLOCK + threadIdx.x, gridDim.x * TRX_BLOCK, H, W); vw.i < nvox; vw.next(H, W)) {
        const size_t i = vw.i;
        float pz = 0.f, py, px;
        if constexpr (ND == 3) { pz = (float)vw.z + fl[i]; py = (float)vw.y + fl[nvox + i]; px = (float)vw.x + fl[2 * nvox + i]; }
        else { py = (float)vw.y + fl[i]; px = (float)vw.x + fl[nvox + i]; }
        const float rz = rintf(pz), ry = rintf(py), rx = rintf(px);
        const bool in = (rx >= 0.f) && (rx < (float)W) && (ry >= 0.f) && (ry < (float)H) && (rz >= 0.f) && (rz < (float)D);   // NaN compares false: zeros
        const size_t src = in ? ((size_t)(int)rz * H + (size_t)(int)ry) * W + (size_t)(int)rx : 0;
        for (int ch = 0; ch < channels; ch++) o[ch * nvox + i] = in ? mov[ch * nvox + src] : 0.f;
    }
}

template <int ND>
__global__ __launch_bounds__(TRX_BLOCK) void flow_warp_bwd_kernel(trx_volumes vol, const float *__restrict__ flow, int channels,
                                                                  const float *__restrict__ grad_out, float *__restrict__ dflow)
{
    const int b = blockIdx.y;
    const int D = vol.D, H = vol.H, W = vol.W;
    const size_t nvox = (size_t)D * H * W;
    const float *__restrict__ mov = vol.moving + (size_t)b * vol.moving_stride;
    const float *__restrict__ fl = flow + (size_t)b * ND * nvox;
    const float *__restrict__ go = grad_out + (size_t)b * channels * nvox;
    float *__restrict__ df = dflow + (size_t)b * ND * nvox;
    VoxelWalk vw(blockIdx.x * TRX_BLOCK + threadIdx.x, gridDim.x * TRX_BLOCK, H, W);
    float fc[3] = {0.f, 0.f, 0.f};
    if (vw.i < nvox) {
#pragma unroll
        for (int c = 0; c < ND; c++) fc[c] = fl[c * nvox + vw.i];
    }
    while (vw.i < nvox) {
        const size_t i = vw.i;
        const int z = vw.z, y = vw.y, x = vw.x;
        vw.next(H, W);
        const size_t in = vw.i < nvox ? (size_t)vw.i : i;
        float fnx[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < ND; c++) fnx[c] = fl[c * nvox + in];
        const float fcur[3] = {fc[0], fc[1], fc[2]};
#pragma unroll
        for (int c = 0; c < ND; c++) fc[c] = fnx[c];
        float acc[3] = {0.f, 0.f, 0.f};
        for (int ch = 0; ch < channels; ch++) {
            float d[3];
            flow_sample_v<ND>(mov + ch * nvox, fcur, D, H, W, z, y, x, d);
            const float g = go[ch * nvox + i];
#pragma unroll
            for (int c = 0; c < ND; c++) acc[c] = fmaf(g, d[c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < ND; c++) df[c * nvox + i] = acc[c];
    }
}

// Z-slab mode: the one term of pass A that needs a neighbour rank's data - the squared forward differences across the slab's upper
// face, sum_c sum_{y,x} (halo_hi[c][y][x] - flow[c][D-1][y][x])^2 - as its own small kernel, so that pass A proper (called without a halo)
// can run while the halo planes are still travelling.  One block, fixed summation order (fp64), added to moments[b][5].
__global__ __launch_bounds__(1024) void slab_boundary_smooth_kernel(const float *__restrict__ flow, const float *__restrict__ halo_hi, int D, int H, int W,
                                                                    double *__restrict__ moments)
{
    __shared__ double red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t plane = (size_t)H * W, nvox = (size_t)D * plane;
    const float *__restrict__ fl = flow + (size_t)b * 3 * nvox + (size_t)(D - 1) * plane;
    const float *__restrict__ hh = halo_hi + (size_t)b * 3 * plane;
    double acc = 0.0;
    for (int c = 0; c < 3; c++)
        for (size_t i0 = 0; i0 < plane; i0 += 1024 * 8) {
            float part = 0.f;   // 8 terms in fp32, carried in fp64
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const size_t i = i0 + (size_t)k * 1024 + tid;
                if (i < plane) {
                    const float d = hh[c * plane + i] - fl[c * nvox + i];
                    part = fmaf(d, d, part);
                }
            }
            acc += (double)part;
        }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; w++) t += red[w];
        moments[b * 8 + 5] += t;
    }
}

static int check_vol_flow(const trx_volumes *v, bool need_target)
{
    if (!v || !v->moving || (need_target && !v->target)) return TRX_ERR_ARG;
    if (v->ndim != 2 && v->ndim != 3) return TRX_ERR_NDIM;
    if (v->B < 1 || v->D < 1 || v->H < 1 || v->W < 1 || v->B > 65535) return TRX_ERR_ARG;
    if (v->ndim == 2 && v->D != 1) return TRX_ERR_NDIM;
    if ((size_t)v->D * v->H * v->W >= ((size_t)1 << 31)) return TRX_ERR_ARG;   // same limit as the affine path (include/trx.h)
    return TRX_OK;
}

static unsigned flow_grid_x(const trx_volumes &v)
{
    if (v.ndim == 3) return (unsigned)flow_col_geom(v).nblk;   // the column-walk kernels
    const size_t nvox = (size_t)v.D * v.H * v.W;
    size_t nb = (nvox + TRX_BLOCK - 1) / TRX_BLOCK;
    const size_t total = 4096;             // blocks in flight overall (measured sweep: 2048 .. 8192 within 2 %)
    size_t cap = (total + v.B - 1) / v.B;
    if (cap < 64) cap = 64;
    return (unsigned)(nb < cap ? nb : cap);
}

}  // namespace trx

using namespace trx;

extern "C" size_t trx_flow_workspace_bytes(const trx_volumes *vol)
{
    if (check_vol_flow(vol, false) != TRX_OK) return 0;
    return (size_t)vol->B * flow_grid_x(*vol) * kFlowNP * sizeof(float) + (size_t)vol->B * sizeof(FlowCoef) + 256 +
           (size_t)vol->B * sizeof(double) + 256;   // + the fp64 data part of the last recorded loss (lagged regulariser term)
}

static FlowCoef *coef_ptr(const trx_volumes *vol, void *workspace)
{
    size_t off = (size_t)vol->B * flow_grid_x(*vol) * kFlowNP * sizeof(float);
    off = (off + 255) & ~(size_t)255;
    return (FlowCoef *)((char *)workspace + off);
}

static double *stash_ptr(const trx_volumes *vol, void *workspace)
{
    size_t off = (size_t)((char *)(coef_ptr(vol, workspace) + vol->B) - (char *)workspace);
    off = (off + 255) & ~(size_t)255;
    return (double *)((char *)workspace + off);
}

static int launch_moments(const trx_volumes *vol, const float *flow, bool smooth, float *partials, hipStream_t s, Slab slab = Slab{0, -1, nullptr, nullptr})
{
    if (slab.Dm < 0) slab.Dm = vol->D;
    dim3 grid(flow_grid_x(*vol), vol->B), block(TRX_BLOCK);
    if (vol->ndim == 3) {
        const ColGeom cg = flow_col_geom(*vol);
        if (smooth) hipLaunchKernelGGL((flow_moments3_kernel<true>), grid, block, 0, s, *vol, flow, partials, slab, cg);
        else hipLaunchKernelGGL((flow_moments3_kernel<false>), grid, block, 0, s, *vol, flow, partials, slab, cg);
    } else {
        if (smooth) hipLaunchKernelGGL((flow_moments_kernel<2, true>), grid, block, 0, s, *vol, flow, partials, slab);
        else hipLaunchKernelGGL((flow_moments_kernel<2, false>), grid, block, 0, s, *vol, flow, partials, slab);
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template <int MODE>
static int launch_update(const trx_volumes *vol, const float *flow, float *flow_out, float *m, float *v, const FlowCoef *coef,
                         const trx_opt_cfg &oc, bool smooth, hipStream_t s, Slab slab = Slab{0, -1, nullptr, nullptr}, float *next_partials = nullptr,
                         float *flow_last = nullptr, int save_last = 0)
{
    if (slab.Dm < 0) slab.Dm = vol->D;
    dim3 grid(flow_grid_x(*vol), vol->B), block(TRX_BLOCK);
#define TRX_LAUNCH_UPD(...) hipLaunchKernelGGL((flow_update_kernel<__VA_ARGS__>), grid, block, 0, s, *vol, flow, flow_out, m, v, coef, oc, slab, next_partials, flow_last, save_last)
#define TRX_LAUNCH_UPD3(...) hipLaunchKernelGGL((flow_update3_kernel<__VA_ARGS__>), grid, block, 0, s, *vol, flow, flow_out, m, v, coef, oc, slab, cg, next_partials, flow_last, save_last)
    if (vol->ndim == 3) {   // column-walk kernels
        const ColGeom cg = flow_col_geom(*vol);
        const bool adam = (MODE == 0) && oc.kind == TRX_OPT_ADAM;
        if constexpr (MODE == 0) {
            if (next_partials) {   // the update + the next iteration's pass A in one kernel
                if (smooth) { if (adam) TRX_LAUNCH_UPD3(0, true, true, true); else TRX_LAUNCH_UPD3(0, true, true, false); }
                else { if (adam) TRX_LAUNCH_UPD3(0, false, true, true); else TRX_LAUNCH_UPD3(0, false, true, false); }
                TRX_CHECK_LAUNCH();
                return TRX_OK;
            }
            if (smooth) { if (adam) TRX_LAUNCH_UPD3(0, true, false, true); else TRX_LAUNCH_UPD3(0, true, false, false); }
            else { if (adam) TRX_LAUNCH_UPD3(0, false, false, true); else TRX_LAUNCH_UPD3(0, false, false, false); }
        } else {
            if (smooth) TRX_LAUNCH_UPD3(MODE, true, false, false);
            else TRX_LAUNCH_UPD3(MODE, false, false, false);
        }
        TRX_CHECK_LAUNCH();
        return TRX_OK;
    }
    if constexpr (MODE == 0) {
        if (next_partials) {
            if (smooth) TRX_LAUNCH_UPD(2, 0, true, false, true);
            else TRX_LAUNCH_UPD(2, 0, false, false, true);
            TRX_CHECK_LAUNCH();
            return TRX_OK;
        }
    }
    if (smooth) TRX_LAUNCH_UPD(2, MODE, true);
    else TRX_LAUNCH_UPD(2, MODE, false);
#undef TRX_LAUNCH_UPD3
#undef TRX_LAUNCH_UPD
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// have_moments: the partials of `cur` are already in the workspace (written by the previous iteration's fused update);
// fuse_next: let this iteration's update write the partials of the flow it produces.
static int flow_step_impl(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt, const trx_flow_state *st,
                          float *cur, float *nxt, void *workspace, hipStream_t s, bool have_moments = false, bool fuse_next = false, int lag = 0,
                          bool last_of_call = false)
{
    const bool smooth = st->smooth_weight != 0.f;
    float *partials = (float *)workspace;
    FlowCoef *coef = coef_ptr(vol, workspace);
    int rc = have_moments ? TRX_OK : launch_moments(vol, cur, smooth, partials, s);
    if (rc) return rc;
    hipLaunchKernelGGL(flow_coef_kernel, dim3(vol->B), dim3(1024), 0, s, partials, (int)flow_grid_x(*vol), vol->ndim, vol->D, vol->H,
                       vol->W, *loss, *opt, st->smooth_weight, st->losses, st->losses_capacity, st->step, (float *)nullptr, coef, (double *)nullptr, (const double *)nullptr, vol->D,
                       lag, stash_ptr(vol, workspace), st->stop_crit, st->stopped, (int)(cur != nxt));
    TRX_CHECK_LAUNCH();
    return launch_update<0>(vol, cur, nxt, st->adam_m, st->adam_v, coef, *opt, smooth, s, Slab{0, -1, nullptr, nullptr}, fuse_next ? partials : nullptr,
                            st->flow_last, last_of_call ? 1 : 0);
}

static int check_flow_args(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt, const trx_flow_state *st,
                           void *workspace, size_t workspace_bytes)
{
    int rc = check_vol_flow(vol, true);
    if (rc) return rc;
    if (!loss || !opt || !st || !workspace || !st->flow) return TRX_ERR_ARG;
    if (opt->kind != TRX_OPT_SGD && opt->kind != TRX_OPT_ADAM) return TRX_ERR_ARG;
    if (opt->kind == TRX_OPT_ADAM && (!st->adam_m || !st->adam_v)) return TRX_ERR_ARG;
    if (st->smooth_weight != 0.f && !st->flow_tmp) return TRX_ERR_ARG;
    if (workspace_bytes < trx_flow_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    return TRX_OK;
}

extern "C" int trx_flow_run(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt, const trx_flow_state *st,
                            int iters, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_flow_args(vol, loss, opt, st, workspace, workspace_bytes);
    if (rc) return rc;
    if (iters < 0) return TRX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool smooth = st->smooth_weight != 0.f;
    float *cur = st->flow, *nxt = smooth ? st->flow_tmp : st->flow;
    // Inside one call the update of iteration i also produces the moments of iteration i + 1 (no smoothness term): after the
    // first iteration every step is coefficient kernel + one streaming kernel.  TRX_FLAG_TWO_PASS_FLOW keeps the two-pass steps.
    const bool fuse = !(vol->flags & TRX_FLAG_TWO_PASS_FLOW);
    if (st->losses && iters > st->losses_capacity) return TRX_ERR_CAPACITY;
    // With the smoothness term every update of a run of >= 2 iterations is the fused one (its regulariser sums lag by one iteration,
    // see flow_coef_kernel) and one coefficient-kernel launch after the loop completes the last recorded loss.
    const bool lagged = fuse && smooth && iters >= 2;
    bool have = false;
    for (int i = 0; i < iters; i++) {
        const bool next = lagged || (fuse && !smooth && (i + 1 < iters));
        const int lag = (lagged && i >= 1) ? (kLagRecord | (i >= 2 ? kLagPatch : 0)) : 0;
        rc = flow_step_impl(vol, loss, opt, st, cur, nxt, workspace, s, have, next, lag, i + 1 == iters);
        if (rc) return rc;
        have = next;
        float *t = cur; cur = nxt; nxt = t;
    }
    if (lagged) {
        hipLaunchKernelGGL(flow_coef_kernel, dim3(vol->B), dim3(1024), 0, s, (const float *)workspace, (int)flow_grid_x(*vol), vol->ndim, vol->D, vol->H,
                           vol->W, *loss, *opt, st->smooth_weight, st->losses, st->losses_capacity, st->step, (float *)nullptr, coef_ptr(vol, workspace),
                           (double *)nullptr, (const double *)nullptr, vol->D, kLagPatch | kLagFlush, stash_ptr(vol, workspace), st->stop_crit, st->stopped, 1);
        TRX_CHECK_LAUNCH();
    }
    if (cur != st->flow) {  // odd number of double-buffered steps: result lives in flow_tmp
        const size_t bytes = (size_t)vol->B * vol->ndim * vol->D * vol->H * vol->W * sizeof(float);
        if (hipMemcpyAsync(st->flow, cur, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return TRX_ERR_HIP;
    }
    return TRX_OK;
}

// ---------------------------------------------------------------------------------------------------
// Direct flow + LOCAL-window NCC (+ smoothness) as one loop on the device (extension, SURVEY 8f.3: the VoxelMorph-style objective;
// definition and arbiter: oracle/compose.py::local_ncc_loss + smooth_regulariser under torch autograd).  Per iteration:
//   flow_moments3_kernel   warp at the current flow (written to the workspace) + the smoothness sums of that flow
//   trx_lncc_loss_grad     window sums -> loss[b], dL/dwarped (csrc/lncc.hip: fields, finalise, gradient)
//   flow_coef_kernel       total loss -> loss curve, step counter, early stop, Adam scalars, smoothness scales
//   flow_update3_kernel<2> dL/dflow = dL/dwarped * (trilinear derivative) + smoothness gradient; SGD / Adam in place
// No autograd, no host sync, no torch optimiser.
// ---------------------------------------------------------------------------------------------------
static size_t lncc_loop_offsets(const trx_volumes *vol, size_t *o_warped, size_t *o_go, size_t *o_loss, size_t *o_lncc)
{
    const size_t nvox = (size_t)vol->D * vol->H * vol->W;
    size_t off = (trx_flow_workspace_bytes(vol) + 255) & ~(size_t)255;
    *o_warped = off; off += ((size_t)vol->B * nvox * sizeof(float) + 255) & ~(size_t)255;
    *o_go = off;     off += ((size_t)vol->B * nvox * sizeof(float) + 255) & ~(size_t)255;
    *o_loss = off;   off += ((size_t)vol->B * sizeof(float) + 255) & ~(size_t)255;
    *o_lncc = off;   off += trx_lncc_workspace_bytes(vol->ndim, vol->B, vol->D, vol->H, vol->W);
    return off;
}

extern "C" size_t trx_flow_lncc_workspace_bytes(const trx_volumes *vol)
{
    if (check_vol_flow(vol, false) != TRX_OK || vol->ndim != 3) return 0;
    size_t a, b, c, d;
    return lncc_loop_offsets(vol, &a, &b, &c, &d);
}

extern "C" int trx_flow_lncc_run(const trx_volumes *vol, int window, float lncc_alpha, float lncc_eps, const trx_opt_cfg *opt, const trx_flow_state *st,
                                 int iters, void *workspace, size_t workspace_bytes, void *stream)
{
    const trx_loss_cfg none = {0.f, 0.f, 0.f, 0.f, 0.f};
    int rc = check_flow_args(vol, &none, opt, st, workspace, workspace_bytes);
    if (rc) return rc;
    if (vol->ndim != 3) return TRX_ERR_NDIM;
    const size_t nvox = (size_t)vol->D * vol->H * vol->W;
    if (iters < 0 || (window != 3 && window != 5 && window != 7 && window != 9)) return TRX_ERR_ARG;
    if (vol->B > 1 && vol->target_stride != nvox) return TRX_ERR_ARG;   // the window kernels take a dense [B][D][H][W] target
    if (workspace_bytes < trx_flow_lncc_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    if (st->losses && iters > st->losses_capacity) return TRX_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    size_t o_warped, o_go, o_loss, o_lncc;
    lncc_loop_offsets(vol, &o_warped, &o_go, &o_loss, &o_lncc);
    char *ws = (char *)workspace;
    float *partials = (float *)workspace, *warped = (float *)(ws + o_warped), *go = (float *)(ws + o_go), *lloss = (float *)(ws + o_loss);
    const size_t lncc_bytes = trx_lncc_workspace_bytes(3, vol->B, vol->D, vol->H, vol->W);
    FlowCoef *coef = coef_ptr(vol, workspace);
    const bool smooth = st->smooth_weight != 0.f, adam = opt->kind == TRX_OPT_ADAM;
    float *cur = st->flow, *nxt = smooth ? st->flow_tmp : st->flow;
    const ColGeom cg = flow_col_geom(*vol);
    const Slab slab = {0, vol->D, nullptr, nullptr};
    dim3 grid(cg.nblk, vol->B), block(TRX_BLOCK);
    for (int i = 0; i < iters; i++) {
        if (smooth) hipLaunchKernelGGL((flow_moments3_kernel<true>), grid, block, 0, s, *vol, cur, partials, slab, cg, warped);
        else hipLaunchKernelGGL((flow_moments3_kernel<false>), grid, block, 0, s, *vol, cur, partials, slab, cg, warped);
        TRX_CHECK_LAUNCH();
        rc = trx_lncc_loss_grad(vol->target, warped, 3, vol->B, vol->D, vol->H, vol->W, window, lncc_alpha, lncc_eps, lloss, go, ws + o_lncc, lncc_bytes, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(flow_coef_kernel, dim3(vol->B), dim3(1024), 0, s, partials, cg.nblk, 3, vol->D, vol->H, vol->W, none, *opt, st->smooth_weight,
                           st->losses, st->losses_capacity, st->step, (float *)nullptr, coef, (double *)nullptr, (const double *)nullptr, vol->D, 0,
                           stash_ptr(vol, workspace), st->stop_crit, st->stopped, (int)(cur != nxt), lloss);
        TRX_CHECK_LAUNCH();
        const int save_last = (i + 1 == iters) ? 1 : 0;
#define TRX_LAUNCH_L(SM, AD) hipLaunchKernelGGL((flow_update3_kernel<2, SM, false, AD>), grid, block, 0, s, *vol, cur, nxt, st->adam_m, st->adam_v, coef, *opt, slab, cg, \
                                                (float *)nullptr, st->flow_last, save_last, go)
        if (smooth) { if (adam) TRX_LAUNCH_L(true, true); else TRX_LAUNCH_L(true, false); }
        else { if (adam) TRX_LAUNCH_L(false, true); else TRX_LAUNCH_L(false, false); }
#undef TRX_LAUNCH_L
        TRX_CHECK_LAUNCH();
        float *t = cur; cur = nxt; nxt = t;
    }
    if (cur != st->flow) {  // odd number of double-buffered steps: result lives in flow_tmp
        if (hipMemcpyAsync(st->flow, cur, (size_t)vol->B * 3 * nvox * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return TRX_ERR_HIP;
    }
    return TRX_OK;
}

extern "C" int trx_flow_step(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt, const trx_flow_state *st,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    return trx_flow_run(vol, loss, opt, st, 1, workspace, workspace_bytes, stream);
}

extern "C" int trx_flow_loss_grad(const trx_volumes *vol, const trx_loss_cfg *loss, const float *flow, float *terms, float *dflow,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_vol_flow(vol, true);
    if (rc) return rc;
    if (!loss || !flow || !terms || !workspace) return TRX_ERR_ARG;
    if (workspace_bytes < trx_flow_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float *partials = (float *)workspace;
    FlowCoef *coef = coef_ptr(vol, workspace);
    rc = launch_moments(vol, flow, false, partials, s);
    if (rc) return rc;
    trx_opt_cfg oc = {TRX_OPT_SGD, 0.f, 0.f, 0.f, 0.f};
    hipLaunchKernelGGL(flow_coef_kernel, dim3(vol->B), dim3(1024), 0, s, partials, (int)flow_grid_x(*vol), vol->ndim, vol->D, vol->H,
                       vol->W, *loss, oc, 0.f, (float *)nullptr, 0, (int *)nullptr, terms, coef, (double *)nullptr, (const double *)nullptr, vol->D);
    TRX_CHECK_LAUNCH();
    if (!dflow) return TRX_OK;
    return launch_update<1>(vol, flow, dflow, nullptr, nullptr, coef, oc, false, s);
}

extern "C" int trx_flow_warp(const trx_volumes *vol, const float *flow, int channels, float *out, void *stream)
{
    int rc = check_vol_flow(vol, false);
    if (rc) return rc;
    if (!flow || !out || channels < 1) return TRX_ERR_ARG;
    dim3 grid(flow_grid_x(*vol), vol->B), block(TRX_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    if (vol->flags & TRX_FLAG_NEAREST) {
        if (vol->ndim == 3) hipLaunchKernelGGL((flow_warp_nearest_kernel<3>), grid, block, 0, s, *vol, flow, channels, out);
        else hipLaunchKernelGGL((flow_warp_nearest_kernel<2>), grid, block, 0, s, *vol, flow, channels, out);
    } else if (vol->ndim == 3) hipLaunchKernelGGL((flow_warp_kernel<3>), grid, block, 0, s, *vol, flow, channels, out);
    else hipLaunchKernelGGL((flow_warp_kernel<2>), grid, block, 0, s, *vol, flow, channels, out);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_flow_warp_backward(const trx_volumes *vol, const float *flow, int channels, const float *grad_out, float *dflow,
                                      void *stream)
{
    int rc = check_vol_flow(vol, false);
    if (rc) return rc;
    if (!flow || !grad_out || !dflow || channels < 1) return TRX_ERR_ARG;
    dim3 grid(flow_grid_x(*vol), vol->B), block(TRX_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    if (vol->ndim == 3) hipLaunchKernelGGL((flow_warp_bwd_kernel<3>), grid, block, 0, s, *vol, flow, channels, grad_out, dflow);
    else hipLaunchKernelGGL((flow_warp_bwd_kernel<2>), grid, block, 0, s, *vol, flow, channels, grad_out, dflow);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// ---------------------------------------------------------------------------------------------------
// Z-slab mode (one volume partitioned over ranks, BASELINE config 5 / SURVEY 8e): the global NCC moments are
// the only per-iteration exchange: pass A per rank -> 8 raw fp64 sums -> all-reduce by the caller (RCCL, 64
// bytes: latency-bound) -> pass B per rank with the sums of the whole volume.
// ---------------------------------------------------------------------------------------------------
static int check_slab(const trx_volumes *vol, int z_offset, int D_full)
{
    int rc = check_vol_flow(vol, true);
    if (rc) return rc;
    if (vol->ndim != 3) return TRX_ERR_NDIM;
    if (z_offset < 0 || D_full < 1 || z_offset + vol->D > D_full) return TRX_ERR_ARG;
    return TRX_OK;
}

extern "C" int trx_flow_slab_moments(const trx_volumes *vol, int z_offset, int D_full, const float *flow, int smooth,
                                     const float *halo_hi, double *moments, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_slab(vol, z_offset, D_full);
    if (rc) return rc;
    if (!flow || !moments || !workspace) return TRX_ERR_ARG;
    if (workspace_bytes < trx_flow_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float *partials = (float *)workspace;
    rc = launch_moments(vol, flow, smooth != 0, partials, s, Slab{z_offset, D_full, nullptr, smooth ? halo_hi : nullptr});
    if (rc) return rc;
    trx_loss_cfg lc = {0.f, 0.f, 0.f, 0.f, 0.f};
    trx_opt_cfg oc = {TRX_OPT_SGD, 0.f, 0.f, 0.f, 0.f};
    hipLaunchKernelGGL(flow_coef_kernel, dim3(vol->B), dim3(1024), 0, s, partials, (int)flow_grid_x(*vol), vol->ndim, vol->D, vol->H,
                       vol->W, lc, oc, 0.f, (float *)nullptr, 0, (int *)nullptr, (float *)nullptr, (FlowCoef *)nullptr, moments,
                       (const double *)nullptr, D_full);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_flow_slab_update(const trx_volumes *vol, int z_offset, int D_full, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                                    const trx_flow_state *st, const double *global_moments, const float *halo_lo, const float *halo_hi,
                                    void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_slab(vol, z_offset, D_full);
    if (rc) return rc;
    if (!global_moments) return TRX_ERR_ARG;
    rc = check_flow_args(vol, loss, opt, st, workspace, workspace_bytes);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const bool smooth = st->smooth_weight != 0.f;
    FlowCoef *coef = coef_ptr(vol, workspace);
    hipLaunchKernelGGL(flow_coef_kernel, dim3(vol->B), dim3(1024), 0, s, (const float *)workspace, 0, vol->ndim, vol->D, vol->H, vol->W,
                       *loss, *opt, st->smooth_weight, st->losses, st->losses_capacity, st->step, (float *)nullptr, coef, (double *)nullptr,
                       global_moments, D_full, 0, (double *)nullptr, st->stop_crit, st->stopped, (int)smooth);
    TRX_CHECK_LAUNCH();
    // with the regulariser the update reads neighbours of the OLD flow: it is written to flow_tmp (the caller swaps)
    return launch_update<0>(vol, st->flow, smooth ? st->flow_tmp : st->flow, st->adam_m, st->adam_v, coef, *opt, smooth, s,
                            Slab{z_offset, D_full, halo_lo, halo_hi}, nullptr, st->flow_last, (st->flow_last && (vol->flags & TRX_FLAG_SAVE_LAST)) ? 1 : 0);
}

// Slab counterpart of the fused step of trx_flow_run (no smoothness term, 3-D): the update also leaves the slab's block partials of
// the UPDATED flow in the workspace; trx_flow_slab_moments_ready then reduces them to the 8 sums without another pass over the slab.
extern "C" int trx_flow_slab_update_fused(const trx_volumes *vol, int z_offset, int D_full, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                                          const trx_flow_state *st, const double *global_moments, void *workspace, size_t workspace_bytes,
                                          void *stream)
{
    int rc = check_slab(vol, z_offset, D_full);
    if (rc) return rc;
    if (!global_moments) return TRX_ERR_ARG;
    rc = check_flow_args(vol, loss, opt, st, workspace, workspace_bytes);
    if (rc) return rc;
    if (st->smooth_weight != 0.f || vol->ndim != 3) return TRX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    FlowCoef *coef = coef_ptr(vol, workspace);
    hipLaunchKernelGGL(flow_coef_kernel, dim3(vol->B), dim3(1024), 0, s, (const float *)workspace, 0, vol->ndim, vol->D, vol->H, vol->W,
                       *loss, *opt, 0.f, st->losses, st->losses_capacity, st->step, (float *)nullptr, coef, (double *)nullptr,
                       global_moments, D_full, 0, (double *)nullptr, st->stop_crit, st->stopped, 0);
    TRX_CHECK_LAUNCH();
    return launch_update<0>(vol, st->flow, st->flow, st->adam_m, st->adam_v, coef, *opt, false, s, Slab{z_offset, D_full, nullptr, nullptr},
                            (float *)workspace, st->flow_last, (st->flow_last && (vol->flags & TRX_FLAG_SAVE_LAST)) ? 1 : 0);
}

extern "C" int trx_flow_slab_moments_ready(const trx_volumes *vol, int z_offset, int D_full, double *moments, void *workspace,
                                           size_t workspace_bytes, void *stream)
{
    int rc = check_slab(vol, z_offset, D_full);
    if (rc) return rc;
    if (!moments || !workspace) return TRX_ERR_ARG;
    if (workspace_bytes < trx_flow_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    trx_loss_cfg lc = {0.f, 0.f, 0.f, 0.f, 0.f};
    trx_opt_cfg oc = {TRX_OPT_SGD, 0.f, 0.f, 0.f, 0.f};
    hipLaunchKernelGGL(flow_coef_kernel, dim3(vol->B), dim3(1024), 0, s, (const float *)workspace, (int)flow_grid_x(*vol), vol->ndim, vol->D, vol->H,
                       vol->W, lc, oc, 0.f, (float *)nullptr, 0, (int *)nullptr, (float *)nullptr, (FlowCoef *)nullptr, moments,
                       (const double *)nullptr, D_full);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// The cross-slab part of the smoothness sums of pass A (see slab_boundary_smooth_kernel): call trx_flow_slab_moments with halo_hi = NULL,
// then this once the neighbour's plane has arrived; together they equal trx_flow_slab_moments with the halo.
extern "C" int trx_flow_slab_boundary_smooth(const trx_volumes *vol, const float *flow, const float *halo_hi, double *moments, void *stream)
{
    int rc = check_vol_flow(vol, false);
    if (rc) return rc;
    if (vol->ndim != 3) return TRX_ERR_NDIM;
    if (!flow || !halo_hi || !moments) return TRX_ERR_ARG;
    hipLaunchKernelGGL(slab_boundary_smooth_kernel, dim3(vol->B), dim3(1024), 0, (hipStream_t)stream, flow, halo_hi, vol->D, vol->H, vol->W, moments);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}
