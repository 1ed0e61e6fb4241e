// Local-window NCC (extension, SURVEY §8f.3 / north_star "NCC local-window sums"): loss and its gradient wrt the
// warped image in two streaming kernels.  Not in the reference ("parity unpinned"): the definition is the box-window
// NCC used by VoxelMorph-style registration, restated with torch ops in oracle/compose.py::local_ncc_loss,
//
//     S_X(q) = sum over the (2R+1)^nd window around q of X, zero padding outside the image, n = (2R+1)^nd
//     c = S_IJ - S_I S_J / n,  a = S_II - S_I^2 / n,  b = S_JJ - S_J^2 / n,  cc(q) = c^2 / (a b + eps)
//     loss = alpha * (1 - mean_q cc(q))
//
// and its derivative wrt J (I = target, J = warped) is again a set of box sums:
//     dloss/dJ_p = -(alpha/N) * ( I_p box(P)_p - J_p box(Q)_p + box(Q uJ - P uI)_p ),
//     P = 2c/den, Q = 2 c^2 a / den^2, den = a b + eps, uI = S_I / n, uJ = S_J / n
// (box is linear: the two fields that are not multiplied by a per-voxel value travel as one, T = Q uJ - P uI - three fields instead of
// four: 12 instead of 16 bytes per voxel between the kernels, three instead of four sets of window sums in the second).
//
// Kernel 1 (lncc_fields_kernel): per voxel the 5 window sums -> cc (loss partial) and the 3 fields P, Q, T.
// Kernel 2 (lncc_grad_kernel):   the 3 window sums of those fields, combined with I_p, J_p -> dloss/dJ.
// Both walk a (32 x, 8 y) column of the volume along z: per plane the tile (+4 halo) goes to LDS, the x window is a
// sliding sum over float4 reads (4 outputs per thread), the y window 2R+1 LDS reads per field, and the z window a
// running sum over a register ring of the last 2R+1 plane sums (re-summed exactly once per ring turn, so rounding
// does not drift along z).  Window sums are separable box filters - adds only; MFMA has nothing to contract here
// (a banded-matrix formulation would spend 40 MACs where 9 adds do).
// Algorithmic bytes: kernel 1 reads I, J (8) and writes 12; kernel 2 reads 12 + 8 and writes 4 -> 44 B / voxel.
#include "trx_common.h"

namespace trx {

#ifndef TRX_LNCC_SHARE
#define TRX_LNCC_SHARE 0
#endif
#ifndef TRX_LNCC_RCP
#define TRX_LNCC_RCP 0
#endif
#ifndef TRX_LNCC_DIRECT
#define TRX_LNCC_DIRECT 1   // round 5: ONE barrier per plane (1: in the builds where it wins - windows 7 and 9 at two blocks per CU; 2: everywhere; 0: nowhere).  The x window is summed in registers straight from the global loads (a thread of the
                            // x pass loads the 12 consecutive cells of its row as three aligned float4 per input, forms the product fields, slides
                            // the sums) and only the x sums go to LDS, double-buffered by plane parity - the raw tile, its commit and two of the
                            // three barriers of a plane are gone.  0: the round 1-4 form (tile -> LDS -> x pass -> LDS -> y pass, three barriers)
#endif
#ifndef TRX_LNCC_PREFETCH
#define TRX_LNCC_PREFETCH 1   // planes of the tile in flight ahead of the window passes (2: measured alternative - +20 registers, 8 x 256^3 w = 5 1818 -> 1932 us, w = 9 2274 -> 2251)
#endif
#ifndef TRX_LNCC_TWO_ROWS
#define TRX_LNCC_TWO_ROWS 1   // 1: a thread owns two y-adjacent outputs (32 x 16 tile per plane); 0: one output (32 x 8 tile) - measured alternative
#endif
// Registers: the two-row version takes 140 VGPRs up to w = 5 and 180 for w = 7, 9 (the z ring: 2 x 9 x 5) = two 256-thread blocks per
// CU; under __launch_bounds__(256, 3) the same code fits 160 without spilling = three blocks.  Measured on one box (256^3, w = 9, loss +
// gradient): one pair 371 us with two blocks per CU, 336 us with three (six z segments fill 768 slots); eight pairs 2.62 ms with two,
// 2.83 ms with three - so the three-wave build runs w = 7, 9 when the batch is small enough to be z-split, the default build otherwise.
// The direct form keeps 12 x NL prefetched cells per thread where the tile form keeps 3.75 x NL: with the z ring of a wide window at two blocks
// per CU (194-196 VGPRs, no scratch) it is 12 % faster (8 x 256^3, w = 9: 2269 -> 1983 us), in the three- and four-blocks-per-CU builds it
// spills and loses 40 % (profiles/r05a_lncc_direct.txt) - so it is chosen per build: windows 7 / 9 at two blocks per CU, and windows 3 / 5 (whose
// strip is 8 cells, two of them halo float2s: 102-128 VGPRs at four blocks per CU, 8 x 256^3 w = 5 1817 -> 1750 us).
template <int R, int MW>
constexpr bool lncc_direct() { return TRX_LNCC_DIRECT == 2 || (TRX_LNCC_DIRECT == 1 && ((MW == 1 && R >= 3) || R <= 2)); }
constexpr int kLO = TRX_LNCC_TWO_ROWS ? 2 : 1;      // outputs per thread: rows kLO * (tid >> 5) + o of the tile
constexpr int kLX = 32, kLY = 8 * kLO;      // output tile of a block in x, y
constexpr int kLRows = kLY + 8, kLCols = kLX + 8;   // tile + halo 4 (the largest radius)

#ifndef TRX_LNCC_XS_PITCH
#define TRX_LNCC_XS_PITCH (kLRows + 1)   // row pitch of the transposed x sums.  25: the y pass of a wave (32 columns x 2 thread rows, two apart) collides two-way on 14 of
                                         // its 32 column pairs (25 (ox' - ox) = 2 mod 64 at ox' - ox = 18); 27 has no such pair below 32 (d = 38) - measured alternative
#endif
constexpr int kXsPitch = TRX_LNCC_XS_PITCH;
constexpr int kLCells = (kLRows * kLCols + TRX_BLOCK - 1) / TRX_BLOCK;   // tile cells per thread (4: a 40 x 24 tile on 256 threads)

// The raw inputs of one plane of the block's tile (+halo), held in registers between the global loads and the LDS
// commit: the loads of plane z+1 are issued before the window passes of plane z, so their latency is hidden.
template <int NL>
struct PlaneRegs {
    float v[kLCells][NL];
};

// Issue the global loads of plane `zin` (zeros outside the volume).  fetch(z, gy, gx, out[NL]) reads one in-volume cell.
template <int NL, typename Fetch>
__device__ __forceinline__ void plane_fetch(int zin, int D, int H, int W, int X0, int Y0, Fetch fetch, PlaneRegs<NL> &r)
{
    const bool zin_ok = (zin >= 0) && (zin < D);   // uniform
#pragma unroll
    for (int c = 0; c < kLCells; c++) {
        const int rc = threadIdx.x + c * TRX_BLOCK;
        const int row = rc / kLCols, col = rc - row * kLCols;
        const int gy = Y0 - 4 + row, gx = X0 - 4 + col;
        const bool inb = zin_ok && (rc < kLRows * kLCols) && ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
#pragma unroll
        for (int f = 0; f < NL; f++) r.v[c][f] = 0.f;
        if (inb) fetch(zin, gy, gx, r.v[c]);
    }
}

// Window sums of NF fields over the x-y window of the plane held in `r` -> P[NF] of thread (ox, oy).
// expand(cell, in[NL]) writes the NF fields of one tile cell through cell[f * kLRows * kLCols].
// All threads of the block must call this together.
template <int R, int NF, int NL, typename Expand>
__device__ __forceinline__ void plane_window_sums(const PlaneRegs<NL> &r, Expand expand, float (*raw)[kLRows][kLCols],
                                                  float (*xs)[kLX][kXsPitch], float (&P)[kLO][NF])
{
    const int tid = threadIdx.x;
    __syncthreads();   // the previous plane's LDS reads are done
#pragma unroll
    for (int c = 0; c < kLCells; c++) {
        const int rc = tid + c * TRX_BLOCK;
        if (rc < kLRows * kLCols) expand(&raw[0][0][0] + rc, r.v[c]);
    }
    __syncthreads();
    // x window: thread (row, quad) produces outputs x = 4 quad .. 4 quad + 3 of its row from 12 consecutive inputs
    if (tid < kLRows * (kLX / 4)) {
        const int row = tid >> 3, q = tid & 7;
        if (row >= 4 - R && row < kLY + 4 + R) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                float v[12];
                const float4 *src = reinterpret_cast<const float4 *>(&raw[f][row][4 * q]);
                const float4 a = src[0], b = src[1], c = src[2];
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
                float s = 0.f;
#pragma unroll
                for (int k = 4 - R; k <= 4 + R; k++) s += v[k];
                xs[f][4 * q][row] = s;
#pragma unroll
                for (int i = 1; i < 4; i++) {
                    s += v[4 + R + i] - v[3 - R + i];
                    xs[f][4 * q + i][row] = s;
                }
            }
        }
    }
    __syncthreads();
    // y window: the kLO outputs of a thread are adjacent rows, so their windows share 2R of their 2R + 1 rows: 2R + kLO LDS reads for kLO
    // outputs; every output is still its own (2R + 1)-term sum in a fixed order
    const int ox = tid & (kLX - 1), oy = kLO * (tid >> 5);
#pragma unroll
    for (int f = 0; f < NF; f++) {
        float v[2 * R + kLO];
#pragma unroll
        for (int k = 0; k < 2 * R + kLO; k++) v[k] = xs[f][ox][oy + 4 - R + k];
#if TRX_LNCC_SHARE
        {   // measured alternative: the second output's window from the first's (two adds instead of 2R + 1; no longer its own fixed-order sum)
            float s = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * R; k++) s += v[k];
            P[0][f] = s;
#pragma unroll
            for (int o = 1; o < kLO; o++) { s += v[2 * R + o] - v[o - 1]; P[o][f] = s; }
        }
#else
#pragma unroll
        for (int o = 0; o < kLO; o++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * R; k++) s += v[o + k];
            P[o][f] = s;
        }
#endif
    }
}

// ---- the direct form (TRX_LNCC_DIRECT) ----
// The inputs of one plane for one thread of the x pass: 12 consecutive cells (output quad + halo 4 on both sides) of one tile row, NL inputs.
template <int NL>
struct StripRegs {
    float v[NL][12];
};
// Loads of plane `zin` for x-pass thread (row, q): cells gx0 .. gx0 + 11 of row gy from the NL arrays src[f] (zeros outside the volume).
// W % 4 == 0 (uniform): three float4 per input - the strip starts on a multiple of four, so a float4 is inside or outside as a whole.
template <int NL, int R>
__device__ __forceinline__ void strip_fetch(int zin, int D, int H, int W, int X0, int Y0, const float *const (&src)[NL], StripRegs<NL> &r)
{
    const int tid = threadIdx.x;
    const int row = tid >> 3, q = tid & 7;
    const int gy = Y0 - 4 + row, gx0 = X0 + 4 * q - 4;
    const bool row_ok = (zin >= 0) && (zin < D) && (tid < kLRows * (kLX / 4)) && ((unsigned)gy < (unsigned)H);
#pragma unroll
    for (int f = 0; f < NL; f++)
#pragma unroll
        for (int k = 0; k < 12; k++) r.v[f][k] = 0.f;
    if (!row_ok) return;
    const size_t base = ((size_t)zin * H + gy) * W;
    if ((W & 3) == 0) {
#pragma unroll
        for (int f = 0; f < NL; f++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int gx = gx0 + 4 * c;
                if (gx >= 0 && gx + 4 <= W) {
                    if (R <= 2 && c != 1) {   // windows 3 and 5 touch cells 2 .. 9 only: two of the halo quad's cells (a float2), half the registers
                        const int k0 = (c == 0) ? 2 : 8;
                        const float2 t = *reinterpret_cast<const float2 *>(src[f] + base + gx0 + k0);
                        r.v[f][k0] = t.x; r.v[f][k0 + 1] = t.y;
                    } else {
                        const float4 t = *reinterpret_cast<const float4 *>(src[f] + base + gx);
                        r.v[f][4 * c] = t.x; r.v[f][4 * c + 1] = t.y; r.v[f][4 * c + 2] = t.z; r.v[f][4 * c + 3] = t.w;
                    }
                }
            }
    } else {
#pragma unroll
        for (int f = 0; f < NL; f++)
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const int gx = gx0 + k;
                if ((unsigned)gx < (unsigned)W) r.v[f][k] = src[f][base + gx];
            }
    }
}
// x window in registers -> xs (this plane's buffer) -> ONE barrier -> y window.  field(f, in[NL]) = field f of a cell from its NL inputs.
// All threads of the block must call this together; xsb = the xs buffer of this plane's parity (the other one may still be read by threads
// that have not finished the previous plane's y pass - nobody writes it before the NEXT barrier).
template <int R, int NF, int NL, typename Field>
__device__ __forceinline__ void plane_window_sums_direct(const StripRegs<NL> &r, Field field, float (*xsb)[kLX][kXsPitch], float (&P)[kLO][NF])
{
    const int tid = threadIdx.x;
    if (tid < kLRows * (kLX / 4)) {
        const int row = tid >> 3, q = tid & 7;
        if (row >= 4 - R && row < kLY + 4 + R) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                float v[12];
#pragma unroll
                for (int k = 3 - R + 1; k <= 4 + R + 3; k++) {   // the cells this window radius touches: 4 - R .. 7 + R
                    float in[NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) in[l] = r.v[l][k];
                    v[k] = field(f, in);
                }
                float sum = 0.f;
#pragma unroll
                for (int k = 4 - R; k <= 4 + R; k++) sum += v[k];
                xsb[f][4 * q][row] = sum;
#pragma unroll
                for (int i = 1; i < 4; i++) {
                    sum += v[4 + R + i] - v[3 - R + i];
                    xsb[f][4 * q + i][row] = sum;
                }
            }
        }
    }
    __syncthreads();
    const int ox = tid & (kLX - 1), oy = kLO * (tid >> 5);
#pragma unroll
    for (int f = 0; f < NF; f++) {
        float v[2 * R + kLO];
#pragma unroll
        for (int k = 0; k < 2 * R + kLO; k++) v[k] = xsb[f][ox][oy + 4 - R + k];
#pragma unroll
        for (int o = 0; o < kLO; o++) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * R; k++) sum += v[o + k];
            P[o][f] = sum;
        }
    }
}

// Walks the column along z, keeping the z window as a running sum over a register ring; emit(z, Z) is called for
// every output plane with the full window sums Z[NF] of this thread's voxel.
// pre(z, o) is called one plane's worth of window passes BEFORE emit(z, Z, o, pre(z, o)): what emit needs from global memory about its
// own voxel (the gradient kernel: I_p, J_p) is requested there, so that its latency hides under the passes instead of stalling every plane.
template <int R, int NF, int NL, typename Fetch, typename Expand, typename Pre, typename Emit>
__device__ __forceinline__ void column_walk(int nd, int D, int H, int W, int X0, int Y0, int z0, int z1, Fetch fetch, Expand expand, Pre pre, Emit emit,
                                            float (*raw)[kLRows][kLCols], float (*xs)[kLX][kXsPitch])
{   // output planes [z0, z1) of the column (a z segment: small batches split columns so that the chip is filled)
    PlaneRegs<NL> regs;
    if (nd == 2) {   // images: the window has no z extent
        float P[kLO][NF];
        plane_fetch<NL>(0, 1, H, W, X0, Y0, fetch, regs);
        decltype(pre(0, 0)) pv[kLO];
#pragma unroll
        for (int o = 0; o < kLO; o++) pv[o] = pre(0, o);
        plane_window_sums<R, NF, NL>(regs, expand, raw, xs, P);
#pragma unroll
        for (int o = 0; o < kLO; o++) emit(0, P[o], o, pv[o]);
        return;
    }
    constexpr int WN = 2 * R + 1;
    float ring[kLO][WN][NF], Z[kLO][NF];
#pragma unroll
    for (int o = 0; o < kLO; o++) {
#pragma unroll
        for (int k = 0; k < WN; k++)
#pragma unroll
            for (int f = 0; f < NF; f++) ring[o][k][f] = 0.f;
#pragma unroll
        for (int f = 0; f < NF; f++) Z[o][f] = 0.f;
    }
    plane_fetch<NL>(z0 - R, D, H, W, X0, Y0, fetch, regs);
#if TRX_LNCC_PREFETCH == 2
    PlaneRegs<NL> regs2;
    plane_fetch<NL>(z0 - R + 1, D, H, W, X0, Y0, fetch, regs2);
#endif
    for (int base = z0 - R; base < z1 + R; base += WN) {
#pragma unroll
        for (int k = 0; k < WN; k++) {
            const int zin = base + k;
            if (zin < z1 + R) {
                float P[kLO][NF];
                decltype(pre(0, 0)) pv[kLO];
                if (zin - R >= z0) {
#pragma unroll
                    for (int o = 0; o < kLO; o++) pv[o] = pre(zin - R, o);
                }
                if (zin >= 0 && zin < D) {   // uniform
                    PlaneRegs<NL> cur = regs;
#if TRX_LNCC_PREFETCH == 2
                    regs = regs2;
                    plane_fetch<NL>(zin + 2, D, H, W, X0, Y0, fetch, regs2);   // two planes in flight during this plane's passes
#else
                    plane_fetch<NL>(zin + 1, D, H, W, X0, Y0, fetch, regs);   // next plane in flight during this plane's passes
#endif
                    plane_window_sums<R, NF, NL>(cur, expand, raw, xs, P);
                } else {                     // planes outside the volume are zero padding
#if TRX_LNCC_PREFETCH == 2
                    regs = regs2;
                    plane_fetch<NL>(zin + 2, D, H, W, X0, Y0, fetch, regs2);
#else
                    plane_fetch<NL>(zin + 1, D, H, W, X0, Y0, fetch, regs);
#endif
#pragma unroll
                    for (int o = 0; o < kLO; o++)
#pragma unroll
                        for (int f = 0; f < NF; f++) P[o][f] = 0.f;
                }
#pragma unroll
                for (int o = 0; o < kLO; o++) {
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        if (k == 0) {   // once per ring turn: exact re-summation instead of the running update
                            ring[o][0][f] = P[o][f];
                            float s = 0.f;
#pragma unroll
                            for (int j = 0; j < WN; j++) s += ring[o][j][f];
                            Z[o][f] = s;
                        } else {
                            Z[o][f] += P[o][f] - ring[o][k][f];
                            ring[o][k][f] = P[o][f];
                        }
                    }
                    if (zin - R >= z0) emit(zin - R, Z[o], o, pv[o]);
                }
            }
        }
    }
}

// The same walk over the direct form of the plane passes: src[NL] = the input arrays of this pair, field(f, in) = field f of a cell.
template <int R, int NF, int NL, typename Field, typename Pre, typename Emit>
__device__ __forceinline__ void column_walk_direct(int nd, int D, int H, int W, int X0, int Y0, int z0, int z1, const float *const (&src)[NL], Field field, Pre pre, Emit emit,
                                                   float (*xs)[NF][kLX][kXsPitch])
{
    StripRegs<NL> regs;
    int par = 0;
    if (nd == 2) {
        float P[kLO][NF];
        strip_fetch<NL, R>(0, 1, H, W, X0, Y0, src, regs);
        decltype(pre(0, 0)) pv[kLO];
#pragma unroll
        for (int o = 0; o < kLO; o++) pv[o] = pre(0, o);
        plane_window_sums_direct<R, NF, NL>(regs, field, xs[0], P);
#pragma unroll
        for (int o = 0; o < kLO; o++) emit(0, P[o], o, pv[o]);
        return;
    }
    constexpr int WN = 2 * R + 1;
    float ring[kLO][WN][NF], Z[kLO][NF];
#pragma unroll
    for (int o = 0; o < kLO; o++) {
#pragma unroll
        for (int k = 0; k < WN; k++)
#pragma unroll
            for (int f = 0; f < NF; f++) ring[o][k][f] = 0.f;
#pragma unroll
        for (int f = 0; f < NF; f++) Z[o][f] = 0.f;
    }
    strip_fetch<NL, R>(z0 - R, D, H, W, X0, Y0, src, regs);
    for (int base = z0 - R; base < z1 + R; base += WN) {
#pragma unroll
        for (int k = 0; k < WN; k++) {
            const int zin = base + k;
            if (zin < z1 + R) {
                float P[kLO][NF];
                decltype(pre(0, 0)) pv[kLO];
                if (zin - R >= z0) {
#pragma unroll
                    for (int o = 0; o < kLO; o++) pv[o] = pre(zin - R, o);
                }
                if (zin >= 0 && zin < D) {   // uniform
                    const StripRegs<NL> cur = regs;
                    strip_fetch<NL, R>(zin + 1, D, H, W, X0, Y0, src, regs);   // next plane in flight during this plane's passes
                    plane_window_sums_direct<R, NF, NL>(cur, field, xs[par], P);
                    par ^= 1;
                } else {                     // planes outside the volume are zero padding
                    strip_fetch<NL, R>(zin + 1, D, H, W, X0, Y0, src, regs);
#pragma unroll
                    for (int o = 0; o < kLO; o++)
#pragma unroll
                        for (int f = 0; f < NF; f++) P[o][f] = 0.f;
                }
#pragma unroll
                for (int o = 0; o < kLO; o++) {
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        if (k == 0) {   // once per ring turn: exact re-summation instead of the running update
                            ring[o][0][f] = P[o][f];
                            float sum = 0.f;
#pragma unroll
                            for (int j = 0; j < WN; j++) sum += ring[o][j][f];
                            Z[o][f] = sum;
                        } else {
                            Z[o][f] += P[o][f] - ring[o][k][f];
                            ring[o][k][f] = P[o][f];
                        }
                    }
                    if (zin - R >= z0) emit(zin - R, Z[o], o, pv[o]);
                }
            }
        }
    }
}

#ifndef TRX_LNCC_NT
#define TRX_LNCC_NT 7   // non-temporal hints - bit 0 stores of the fields, bit 1 store of the gradient, bit 2 loads of I, J at the emit of the gradient kernel
                        // (8 x 256^3: w = 5 1890 -> 1826 us, w = 9 2323 -> 2278; one pair within the noise; profiles/r04h_lncc_variants.txt)
#endif
template <int BIT> __device__ __forceinline__ void st_nt(float *p, float v) { if constexpr ((TRX_LNCC_NT >> BIT) & 1) __builtin_nontemporal_store(v, p); else *p = v; }
template <int BIT> __device__ __forceinline__ float ld_nt(const float *p) { if constexpr ((TRX_LNCC_NT >> BIT) & 1) return __builtin_nontemporal_load(p); else return *p; }
template <int R, int MW>
__global__ __launch_bounds__(TRX_BLOCK, MW) void lncc_fields_kernel(const float *__restrict__ tgt, const float *__restrict__ wrp, int nd, int D, int H,
                                                               int W, int zsplit, float eps, float *__restrict__ fields, float *__restrict__ partials)
{
    // direct form: x sums of I, J, I^2, J^2, I J - the plane in work / the previous one (by plane parity); else: the five fields of the tile (+halo), then their x sums
    constexpr bool kDirect = lncc_direct<R, MW>();
    constexpr int kTile = 5 * kLRows * kLCols, kXs = 5 * kLX * kXsPitch;
    __shared__ __attribute__((aligned(16))) float lds[kDirect ? 2 * kXs : kTile + kXs];
    float (*raw)[kLRows][kLCols] = reinterpret_cast<float (*)[kLRows][kLCols]>(lds);
    float (*xs)[kLX][kXsPitch] = reinterpret_cast<float (*)[kLX][kXsPitch]>(lds + (kDirect ? 0 : kTile));
    float (*xs2)[5][kLX][kXsPitch] = reinterpret_cast<float (*)[5][kLX][kXsPitch]>(lds);
    const int b = blockIdx.z / zsplit, seg = blockIdx.z - b * zsplit, X0 = blockIdx.x * kLX, Y0 = blockIdx.y * kLY;
    const int zlen = (D + zsplit - 1) / zsplit, z0 = seg * zlen, z1 = min(D, z0 + zlen);
    const size_t n = (size_t)D * H * W;
    const float *__restrict__ I = tgt + (size_t)b * n, *__restrict__ J = wrp + (size_t)b * n;
    float *__restrict__ F = fields + (size_t)b * 3 * n;
    const int tid = threadIdx.x, x = X0 + (tid & (kLX - 1)), y = Y0 + kLO * (tid >> 5);   // this thread's rows: y, ..., y + kLO - 1
    const float wn = (nd == 3) ? (float)((2 * R + 1) * (2 * R + 1) * (2 * R + 1)) : (float)((2 * R + 1) * (2 * R + 1));
    const float inv_n = 1.0f / wn;
    float lsum = 0.f;
    constexpr int FS = kLRows * kLCols;
    // the 5 product fields are formed while the tile is copied to LDS: the window passes then only add
    auto fetch = [&](int z, int gy, int gx, float (&o)[2]) {
        const size_t off = ((size_t)z * H + gy) * W + gx;
        o[0] = I[off]; o[1] = J[off];
    };
    auto expand = [&](float *cell, const float (&in)[2]) {
        const float i = in[0], j = in[1];
        cell[0] = i; cell[FS] = j; cell[2 * FS] = i * i; cell[3 * FS] = j * j; cell[4 * FS] = i * j;
    };
    auto pre = [](int, int) { return 0; };
    auto emit = [&](int z, const float (&Z)[5], int o, int) {
        if (x >= W || y + o >= H) return;
        const float Is = Z[0], Js = Z[1];
        const float c = Z[4] - Is * Js * inv_n, a = Z[2] - Is * Is * inv_n, bv = Z[3] - Js * Js * inv_n;
        const float den = a * bv + eps;
#if TRX_LNCC_RCP
        float rden = __builtin_amdgcn_rcpf(den);           // v_rcp_f32 (1 ulp) + one Newton step instead of the ten-instruction IEEE division
        rden = fmaf(fmaf(-den, rden, 1.0f), rden, rden);
#else
        const float rden = 1.0f / den;
#endif
        const float Pq = 2.0f * c * rden, Qq = Pq * c * a * rden;
        lsum += c * c * rden;
        const size_t off = ((size_t)z * H + y + o) * W + x;
        st_nt<0>(F + off, Pq); st_nt<0>(F + n + off, Qq); st_nt<0>(F + 2 * n + off, (Qq * Js - Pq * Is) * inv_n);
    };
    if constexpr (kDirect) {
        const float *const src[2] = {I, J};
        auto field = [](int f, const float (&in)[2]) { return f == 0 ? in[0] : (f == 1 ? in[1] : (f == 2 ? in[0] * in[0] : (f == 3 ? in[1] * in[1] : in[0] * in[1]))); };
        column_walk_direct<R, 5, 2>(nd, D, H, W, X0, Y0, z0, z1, src, field, pre, emit, xs2);
    } else {
        column_walk<R, 5, 2>(nd, D, H, W, X0, Y0, z0, z1, fetch, expand, pre, emit, raw, xs);
    }
    // block sum of the cc partials in a fixed order: butterfly inside each wave, then the 4 wave sums through LDS
    // (a generic block_reduce_store would add 16 KB of static LDS and halve the blocks per CU)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lsum += __shfl_xor(lsum, m, 64);
    __syncthreads();                       // the last plane's xs reads are done: reuse xs as scratch
    float *ws = &xs[0][0][0];
    if ((tid & 63) == 0) ws[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0) partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

template <int R, int MW>
__global__ __launch_bounds__(TRX_BLOCK, MW) void lncc_grad_kernel(const float *__restrict__ tgt, const float *__restrict__ wrp, int nd, int D, int H,
                                                             int W, int zsplit, float scale, const float *__restrict__ fields, float *__restrict__ grad)
{
    constexpr bool kDirect = lncc_direct<R, MW>();
    constexpr int kTile = 3 * kLRows * kLCols, kXs = 3 * kLX * kXsPitch;
    __shared__ __attribute__((aligned(16))) float lds[kDirect ? 2 * kXs : kTile + kXs];
    float (*raw)[kLRows][kLCols] = reinterpret_cast<float (*)[kLRows][kLCols]>(lds);
    float (*xs)[kLX][kXsPitch] = reinterpret_cast<float (*)[kLX][kXsPitch]>(lds + (kDirect ? 0 : kTile));
    float (*xs2)[3][kLX][kXsPitch] = reinterpret_cast<float (*)[3][kLX][kXsPitch]>(lds);
    const int b = blockIdx.z / zsplit, seg = blockIdx.z - b * zsplit, X0 = blockIdx.x * kLX, Y0 = blockIdx.y * kLY;
    const int zlen = (D + zsplit - 1) / zsplit, z0 = seg * zlen, z1 = min(D, z0 + zlen);
    const size_t n = (size_t)D * H * W;
    const float *__restrict__ I = tgt + (size_t)b * n, *__restrict__ J = wrp + (size_t)b * n;
    const float *__restrict__ F = fields + (size_t)b * 3 * n;
    float *__restrict__ G = grad + (size_t)b * n;
    const int tid = threadIdx.x, x = X0 + (tid & (kLX - 1)), y = Y0 + kLO * (tid >> 5);
    constexpr int FS = kLRows * kLCols;
    auto fetch = [&](int z, int gy, int gx, float (&o)[3]) {
        const size_t off = ((size_t)z * H + gy) * W + gx;
#pragma unroll
        for (int f = 0; f < 3; f++) o[f] = F[(size_t)f * n + off];
    };
    auto expand = [&](float *cell, const float (&in)[3]) {
#pragma unroll
        for (int f = 0; f < 3; f++) cell[f * FS] = in[f];
    };
    struct IJ { float i, j; };
    auto pre = [&](int z, int o) {
        IJ v = {0.f, 0.f};
        if (x < W && y + o < H) {
            const size_t off = ((size_t)z * H + y + o) * W + x;
            v.i = ld_nt<2>(I + off); v.j = ld_nt<2>(J + off);
        }
        return v;
    };
    auto emit = [&](int z, const float (&Z)[3], int o, const IJ &v) {
        if (x >= W || y + o >= H) return;
        const size_t off = ((size_t)z * H + y + o) * W + x;
        st_nt<1>(G + off, scale * (v.i * Z[0] - v.j * Z[1] + Z[2]));
    };
    if constexpr (kDirect) {
        const float *const src[3] = {F, F + n, F + 2 * n};
        auto field = [](int f, const float (&in)[3]) { return f == 0 ? in[0] : (f == 1 ? in[1] : in[2]); };
        column_walk_direct<R, 3, 3>(nd, D, H, W, X0, Y0, z0, z1, src, field, pre, emit, xs2);
    } else {
        column_walk<R, 3, 3>(nd, D, H, W, X0, Y0, z0, z1, fetch, expand, pre, emit, raw, xs);
    }
}

// loss[b] = alpha * (1 - sum(partials) / N), partials reduced in fp64 in a fixed order
__global__ __launch_bounds__(TRX_BLOCK) void lncc_finalize_kernel(const float *__restrict__ partials, int nblk, double nvox, float alpha, float *__restrict__ loss)
{
    __shared__ double red[TRX_BLOCK];
    const int b = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < nblk; i += TRX_BLOCK) s += (double)partials[(size_t)b * nblk + i];
    red[tid] = s;
    __syncthreads();
    for (int w = TRX_BLOCK / 2; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) loss[b] = (float)((double)alpha * (1.0 - red[0] / nvox));
}

// z segments per column: enough blocks to fill `want` block slots, each segment at least 32 planes deep (it re-reads 2R halo planes).
static int lncc_zsplit(int nd, int B, int D, int H, int W, long want)
{
    if (nd == 2) return 1;
    const long cols = (long)((W + kLX - 1) / kLX) * ((H + kLY - 1) / kLY) * B;
    long z = (want + cols - 1) / cols;
    if (z > D / 32) z = D / 32;
    return (int)(z < 1 ? 1 : z);
}
// block slots of the register builds: windows 3 and 5 fit four 256-thread blocks per CU (116 / 106 VGPRs under __launch_bounds__(256, 4),
// no scratch; 4 x 35 KB of LDS), windows 7 and 9 three (the z ring: 2 x 9 x 5 registers) or two
constexpr long kLnccSlots4 = 1024, kLnccSlots3 = TRX_LNCC_TWO_ROWS ? 768 : 1024;

template <int R, int MW>
static int launch_lncc_mw(const float *target, const float *warped, int nd, int B, int D, int H, int W, float alpha, float eps, float *loss, float *grad,
                          float *fields, float *partials, int zsplit, hipStream_t s)
{
    dim3 grid((W + kLX - 1) / kLX, (H + kLY - 1) / kLY, B * zsplit), block(TRX_BLOCK);
    hipLaunchKernelGGL((lncc_fields_kernel<R, MW>), grid, block, 0, s, target, warped, nd, D, H, W, zsplit, eps, fields, partials);
    TRX_CHECK_LAUNCH();
    const double nvox = (double)D * H * W;
    if (loss) {
        hipLaunchKernelGGL(lncc_finalize_kernel, dim3(B), block, 0, s, partials, (int)(grid.x * grid.y) * zsplit, nvox, alpha, loss);
        TRX_CHECK_LAUNCH();
    }
    if (grad) {
        hipLaunchKernelGGL((lncc_grad_kernel<R, MW>), grid, block, 0, s, target, warped, nd, D, H, W, zsplit, (float)(-(double)alpha / nvox), fields, grad);
        TRX_CHECK_LAUNCH();
    }
    return TRX_OK;
}

template <int R>
static int launch_lncc(const float *target, const float *warped, int nd, int B, int D, int H, int W, float alpha, float eps, float *loss, float *grad,
                       float *fields, float *partials, hipStream_t s)
{
    if (TRX_LNCC_TWO_ROWS && R <= 2)   // small windows: the four-blocks-per-CU build, whatever the batch (8 x 256^3: 1024 columns resident in one round)
        return launch_lncc_mw<R, 4>(target, warped, nd, B, D, H, W, alpha, eps, loss, grad, fields, partials, lncc_zsplit(nd, B, D, H, W, kLnccSlots4), s);
    const int zsplit = lncc_zsplit(nd, B, D, H, W, kLnccSlots3);
    if (TRX_LNCC_TWO_ROWS && R >= 3 && zsplit > 1)   // a small batch of a wide window: the three-blocks-per-CU build (see the note on registers above)
        return launch_lncc_mw<R, 3>(target, warped, nd, B, D, H, W, alpha, eps, loss, grad, fields, partials, zsplit, s);
    return launch_lncc_mw<R, 1>(target, warped, nd, B, D, H, W, alpha, eps, loss, grad, fields, partials, zsplit, s);
}

static size_t lncc_partials_bytes(int nd, int B, int D, int H, int W)
{
    const size_t nb = (size_t)((W + kLX - 1) / kLX) * ((H + kLY - 1) / kLY) * lncc_zsplit(nd, B, D, H, W, kLnccSlots4);   // the finest split any window uses
    return ((size_t)B * nb * sizeof(float) + 255) & ~(size_t)255;
}

}  // namespace trx

using namespace trx;

extern "C" size_t trx_lncc_workspace_bytes(int ndim, int B, int D, int H, int W)
{
    if ((ndim != 2 && ndim != 3) || B < 1 || D < 1 || H < 1 || W < 1 || (ndim == 2 && D != 1)) return 0;
    return lncc_partials_bytes(ndim, B, D, H, W) + (size_t)3 * B * D * H * W * sizeof(float);
}

extern "C" int trx_lncc_loss_grad(const float *target, const float *warped, int ndim, int B, int D, int H, int W, int window, float alpha,
                                  float eps, float *loss, float *grad, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!target || !warped || !workspace || (!loss && !grad)) return TRX_ERR_ARG;
    if ((ndim != 2 && ndim != 3) || (ndim == 2 && D != 1)) return TRX_ERR_NDIM;
    if (B < 1 || B > 65535 || D < 1 || H < 1 || W < 1 || (double)D * H * W >= 2147483648.0) return TRX_ERR_ARG;
    if (window != 3 && window != 5 && window != 7 && window != 9) return TRX_ERR_ARG;
    if (workspace_bytes < trx_lncc_workspace_bytes(ndim, B, D, H, W)) return TRX_ERR_WORKSPACE;
    float *partials = (float *)workspace;
    float *fields = (float *)((char *)workspace + lncc_partials_bytes(ndim, B, D, H, W));
    hipStream_t s = (hipStream_t)stream;
    switch (window) {
    case 3: return launch_lncc<1>(target, warped, ndim, B, D, H, W, alpha, eps, loss, grad, fields, partials, s);
    case 5: return launch_lncc<2>(target, warped, ndim, B, D, H, W, alpha, eps, loss, grad, fields, partials, s);
    case 7: return launch_lncc<3>(target, warped, ndim, B, D, H, W, alpha, eps, loss, grad, fields, partials, s);
    default: return launch_lncc<4>(target, warped, ndim, B, D, H, W, alpha, eps, loss, grad, fields, partials, s);
    }
}
