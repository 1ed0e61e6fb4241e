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
#ifdef TRX_DEV
#include <cstdlib>   // getenv: development builds only (tools/kbench.hip); the product library reads no environment variable
#endif
#include "trx_dev.h"   // development instrumentation (in-kernel stamps, LDS padding): no-ops unless tools/kbench.hip asks for them

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
// rows_used[b] (the step kernels' note to the finalise kernel): low 24 bits = partial rows the pair's kernel wrote, bits 24-27 = which body
// (1 + dual_choice for the tile kernel: 1 GeomD, 2 GeomA, 3 GeomR, 4 GeomRD, 5 z-streaming inside it; 6 = the z-streaming kernel, 7 = the
// exact-footprint kernel, 8 = the z-streaming kernel's flat tile), NEGATIVE when a kernel in front of the tile kernel took the pair.  Read back by AffineSolver.bodies().
constexpr int kRowsMask = 0xFFFFFF;
__host__ __device__ constexpr int rows_note(int rows, int body) { return rows > 0 ? (rows | (body << 24)) : 0; }
constexpr int kNpMse = 13;   // partial-row layout of the MSE / SSD-only step kernel (3-D): sum d^2, then 12 x sum(d J)

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
        const float fW = (float)W, fH = (float)H, fD = (float)D;
        const int y0 = yc * g.TY * g.RPT + ly;
        for (int j = 0; j < g.RPT; j++) {
            const int y = y0 + j * g.TY;
            if (y >= H) break;
            const float yn = base_coord(vol.yn, y, H);
            const float ix = unnorm<ND>(fmaf(t_x1, yn, bx), fW);
            const float iy = unnorm<ND>(fmaf(t_y1, yn, by), fH);
            const size_t vox = ((size_t)z * H + y) * W + x;
            float gq[NC];
            float w;
            if constexpr (MODE == 2) {
#pragma unroll
                for (int c = 0; c < NC; c++) gq[c] = 0.f;
                for (int ch = 0; ch < channels; ch++) {
                    const float go = tgt[ch * chan_stride + vox];
                    if constexpr (ND == 3) {
                        const float iz = unnorm<ND>(fmaf(t_z1, yn, bz), fD);
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
                    const float iz = unnorm<ND>(fmaf(t_z1, yn, bz), fD);
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
// LDS-tiled 3-D variant of the F1 pass (the path the headline number runs).
//
// A block owns one (32 x, 8 z) COLUMN of the volume and walks it along y in tiles of 16 rows.  The pre-image
// of a tile under the affine map is a small parallelepiped; its bounding box (x origin aligned to 4 voxels)
// is staged into LDS by LDS-DMA (global_load_lds_dwordx4, masked to the needed, in-volume float4 slots; cells
// outside the volume are zero-filled, which IS grid_sample's zero padding), then all 8-corner gathers are
// ds_read2_b32 from LDS with fixed strides and no bounds checks.  Full tiles whose box fits run in the fast
// loop; partial tiles, tiles whose box exceeds the LDS budget (large rotations / zoom-out) and the one tile per
// pair that holds the volume's last row when W % 4 != 0 run in the generic loop behind it, which also holds
// the global-gather fallback - results never depend on which path ran beyond fp32 rounding.
// Thread (x, z) is fixed for the whole column, so only sum(q grad) and sum(q grad yn) live in registers and
// the xn / zn columns of the 41 sums are one multiply at the very end.
// ------------------------------------------------------------------------------------------
// Two geometries of the tile kernel:
//  cfg 0: 512-thread blocks, tile 32 x 16 y 8 z, ONE 56.7 KB box, 2 blocks per CU (staging of one block overlaps the gather
//         of the other), 7 DMA pieces of two box planes each;
//  cfg 1: 1024-thread blocks, tile 32 x 16 y 16 z, TWO 81.3 KB boxes (all 160 KB of a CU's LDS), 1 block per CU: the DMA
//         of tile t+1 runs during the gather of tile t inside the block, one barrier per tile, and the deeper tile cuts the
//         z-halo share of the bytes a CU ingests from 13/8 to 21/16; 6 DMA pieces of four box planes each.
// LDS box (floats), kBW % 4 == 0.  One LDS-DMA piece (one global_load_lds_dwordx4 per wave) covers kPP z planes of the
// box - at most one float4 slot per thread - so piece k of a thread is its piece-0 slot shifted by k * kPP planes: one VGPR
// offset + one packed slot id per thread instead of one per piece.
template <int TX_, int TZ_, int THREADS_, int BW_, int BH_, int BD_, int PP_, int BUFS_, int TY_ = 16>
struct TileCfg {
    static constexpr int TX = TX_, TY = TY_, TZ = TZ_, Threads = THREADS_;
    static constexpr int BW = BW_, BH = BH_, BD = BD_, PP = PP_, Bufs = BUFS_;
    static constexpr int NH = Threads / (TX * TZ);          // y groups of a tile (2 halves of 8 rows, or 4 quarters of 4)
    static constexpr int Rows = TY / NH;                    // rows per thread
    static constexpr int Waves = Threads / 64;
    static constexpr int BW4 = BW / 4;
    static constexpr int PlaneSlots = BH * BW4;             // float4 slots per box plane
    static constexpr int Pieces = (BD + PP - 1) / PP;       // DMA pieces per tile
    static constexpr int PieceFloats = PP * BH * BW;        // floats of LDS per piece
    static constexpr int BoxFloats = BW * BH * BD;          // one box; lanes of the last piece past it are always masked
    static constexpr int ReduceScratch = Waves * 16 * 65 + Waves * 16;   // floats block_reduce_store_nw needs (aliases the box)
    static constexpr int BoxAlloc = (Bufs * BoxFloats > ReduceScratch) ? Bufs * BoxFloats : ReduceScratch;
    static_assert(BW % 4 == 0 && PP * PlaneSlots <= Threads && (PP == 2 || PP == 4), "one DMA piece: at most one slot per thread");
};
using GeomA = TileCfg<32, 8, 512, 44, 23, TRX_GEOMA_BD, 2, 1>;      // near-identity transforms: 32 x 16 x 8 tile, 44 x 23 x 14 box (56.7 KB)
using GeomDeep = TileCfg<32, 16, 1024, 44, 22, 21, 4, 2>;  // cfg 1 (measured alternative): 1024 threads, two 81 KB boxes
// Rotated transforms: the pre-image of a 32-wide tile grows by 31 sin(angle) rows / planes and stops fitting any box beyond
// ~0.1 rad.  A more cubic tile (16 x 16 x 8, four y-quarters of 4 rows per thread) with a 28 x 26 x 16 box (46.6 KB) fits
// every rotation about one axis up to ~0.5 rad at 55 us per 256^3 pair, whatever the angle (the global-gather fallback: 125 us).
#if TRX_GEOMR_BIG
// 28 x 27 x 26 box (78.6 KB, still two blocks per CU): the pre-image of the 16 x 16 x 8 tile under ANY rotation (span <= |(15,15,7)| = 22.3
// voxels per axis) fits, i.e. every pose the reference's rigid mode can draw (angles uniform in [0,1) rad, ref:utils.py:316-330);
// only the needed extent is fetched, so small angles cost what they cost with the 28 x 26 x 16 box.
#ifndef TRX_GEOMR_BH
#define TRX_GEOMR_BH 27
#define TRX_GEOMR_BD 26
#endif
using GeomR = TileCfg<16, 8, 512, 28, TRX_GEOMR_BH, TRX_GEOMR_BD, 2, 1>;
#else
using GeomR = TileCfg<16, 8, 512, 28, 26, 16, 2, 1>;
#endif
// Wide tile (64 x 8 x 8, one row block of 8 per thread): a box row of 64 + halo voxels touches 3.3 L2 lines for 64 voxels where the
// 32-wide tile touches 2.3 for 32, i.e. 366 instead of 444 box lines per 4096 voxels; the 76 x 13 x 14 box fits |rotation| < ~0.04 rad.
using GeomW = TileCfg<64, 8, 512, 76, 13, 14, 2, 1, 8>;
// Deep tile on 512 threads (32 x 16 x 16, sixteen rows per thread, ONE 44 x 23 x 19 box = 76.9 KB, two blocks per CU): per-tile work and the two
// barriers are paid once per 8192 voxels instead of 4096 and the z halo is 18 / 16 instead of 10 / 8; the box has one plane of slack, so it
// serves |theta - I| up to ~0.03 rad about x / y only.
using GeomD = TileCfg<32, 16, 512, 44, 23, 19, 2, 1>;
// GeomR's box under a 16 x 16 x 16 tile (eight rows per thread): per-tile work is paid once per 4096 voxels instead of 2048 and the z halo is
// 18 / 16 instead of 10 / 8; the pre-image fits the 26-plane box for rotations about z of any size GeomR serves and for general rotations up to
// ~0.4 rad per axis.
using GeomRD = TileCfg<16, 16, 512, 28, TRX_GEOMR_BH, TRX_GEOMR_BD, 2, 1>;
#if TRX_TILE_CFG == 5
using GeomP = GeomRD;
#elif TRX_TILE_CFG == 4
using GeomP = GeomD;
#elif TRX_TILE_CFG == 1
using GeomP = GeomDeep;   // primary geometry
#elif TRX_TILE_CFG == 3
using GeomP = GeomW;
#elif TRX_TILE_CFG == 2
using GeomP = GeomR;
#else
using GeomP = GeomA;
#endif

struct TileGeom {
    int ntx, nty, ntz, ntiles, blocks_per_pair, ysplit, tiles_per_seg;
};

template <class G = GeomP>
static TileGeom tile_geom(const trx_volumes &v)
{
    TileGeom t;
    t.ntx = (v.W + G::TX - 1) / G::TX; t.nty = (v.H + G::TY - 1) / G::TY; t.ntz = (v.D + G::TZ - 1) / G::TZ;
    t.ntiles = t.ntx * t.nty * t.ntz;
    // One block per (x-tile, z-tile) column walking y; columns are split into y segments where that fills the chip better
    // (512 block slots: 2 blocks on each of 256 CUs).  Measured with tools/kbench.hip (TRX_TILE_TARGET_BLOCKS sweeps):
    //  - few columns (<= 512 blocks): one round of ~512 blocks, but at least 2 tiles per block
    //    (1 x 256^3: 47 us at 512 blocks, 53 at 256 and 1024; 1 x 128^3: 12.1 us at 256 blocks x 2 tiles, 13.6 at 512 x 1);
    //  - many columns: the split (1..4) that minimises  (slot rounds * 512 / blocks) * (1 + 1.5 / tiles per block)  - the
    //    idle tail of the last round against the per-block prologue / epilogue (8 x 182^3: 181 us unsplit, 173 us split in 2).
    const int ncol = t.ntx * t.ntz;
#ifdef TRX_DEV
    static const int target = [] { const char *e = getenv("TRX_TILE_TARGET_BLOCKS"); return e ? atoi(e) : 0; }();   // development override (kbench sweeps)
#else
    constexpr int target = 0;
#endif
    const long cols = (long)v.B * ncol;
    int ys = 1;
    if (target > 0) {
        ys = (int)((target + cols - 1) / cols);
#if TRX_GEOM_MODEL
    } else if (cols * t.nty > 1024) {
        // Occupancy model of one launch, in units of "one tile on a CU that runs a single block": 512 block slots (two per CU); a block
        // costs its tiles + 1.5 of prologue / epilogue, times 1.6 when it shares its CU (the pair together: 1.25x a lone block); blocks
        // beyond the slots run in further rounds, a last round of <= 256 blocks has the CUs to itself.  Reproduces the sweeps the
        // round-1 rules were fitted to by hand (1 x 256^3: 512 blocks 47 us, 256 or 1024 blocks 53 us; 8 x 182^3: split in two 173 us,
        // unsplit 181) and fixes what they missed - a second round that is nearly empty (1 x 192^3: 576 blocks 39 us, 432 blocks 32 us;
        // 2 x 182 x 218 x 182: 552 blocks 77 us, 828 blocks 66 us).  Many columns: at most four segments, as before (every segment of
        // every geometry enlarges the dual kernel's grid, whose surplus blocks cost their dispatch).
        double best = 1e30;
        int last_tps = 0;
        const int cmax = cols > 512 ? (t.nty < 4 ? t.nty : 4) : t.nty;
        for (int c = 1; c <= cmax; c++) {
            const int tps = (t.nty + c - 1) / c, segs = (t.nty + tps - 1) / tps;
            if (tps == last_tps) continue;   // same split as the previous c
            last_tps = tps;
            const long blocks = cols * segs, full = blocks / 512, rem = blocks % 512;
            double cost = (double)full * 1.6 * (tps + 1.5);
            if (rem > 0) cost += (rem <= 256 ? 1.0 : 1.6) * (tps + 1.5);
            if (cost < best - 1e-9) { best = cost; ys = c; }
        }
#endif
    } else if (cols <= 512) {   // (with the model: launches of at most two tiles per block slot, where fixed costs decide and the model does not resolve them)
        ys = (int)((512 + cols - 1) / cols);
        const int cap = t.nty / 2 > 1 ? t.nty / 2 : 1;
        if (ys > cap && cols * cap >= 256) ys = cap;   // (tiny volumes: as many blocks as there are tiles - 1 x 64^3: 8.9 vs 12.1 us)
    } else {
        double best = 1e30;
        for (int c = 1; c <= 4 && c <= t.nty; c++) {
            const int tps = (t.nty + c - 1) / c, segs = (t.nty + tps - 1) / tps;
            const double blocks = (double)cols * segs;
            const double rounds = (double)(((long)blocks + 511) / 512);
            const double waste = rounds * 512.0 / blocks * (1.0 + 1.5 / tps);
            if (waste < best - 1e-9) { best = waste; ys = c; }
        }
    }
    if (ys < 1) ys = 1;
    if (ys > t.nty) ys = t.nty;
    t.tiles_per_seg = (t.nty + ys - 1) / ys;
    t.ysplit = (t.nty + t.tiles_per_seg - 1) / t.tiles_per_seg;
    t.blocks_per_pair = ncol * t.ysplit;
    return t;
}

// packed running sums of the tile kernel: AB[q][c] = (sum q*g_c, sum q*g_c*yn), M01 = (Sy, Sw), M23 = (Syy, Sww)
struct F1Acc {
    f2 AB[3][3], M01, M23;
    float M4;
};

template <int MODE>
__device__ __forceinline__ void f1_accumulate_pk(const Samp3 &sm, float yv, float yn, F1Acc &a)
{
    if constexpr (MODE == 2) {   // generic warp backward: yv = grad_out of this voxel, only sum(go * grad) and sum(go * grad * yn)
        const float gq[3] = {sm.dx, sm.dy, sm.dz};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const f2 gu = {gq[c], yn * gq[c]};
            a.AB[0][c] = gu * yv + a.AB[0][c];
        }
        return;
    }
    if constexpr (MODE == 4) {   // MSE / SSD only (no NCC term): d = w - y carries everything - sum d^2 and sum(d * grad), sum(d * grad * yn)
        const float d = sm.v - yv;
        a.M4 = fmaf(d, d, a.M4);
        const float gq[3] = {sm.dx, sm.dy, sm.dz};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const f2 gu = {gq[c], yn * gq[c]};
            a.AB[0][c] = gu * d + a.AB[0][c];
        }
        return;
    }
    const f2 yw = {yv, sm.v};
    a.M01 += yw;
    a.M23 = yw * yw + a.M23;
    a.M4 = fmaf(yv, sm.v, a.M4);
    if constexpr (MODE == 0) {
        const float gq[3] = {sm.dx, sm.dy, sm.dz};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const f2 gu = {gq[c], yn * gq[c]};
            a.AB[0][c] += gu;
            a.AB[1][c] = gu * yv + a.AB[1][c];
            a.AB[2][c] = gu * sm.v + a.AB[2][c];
        }
    }
}

constexpr int kTileThreads = GeomP::Threads;   // block size of the primary kernel (the dual kernel: 512)

// Work-item id of a 1-D block WITHOUT keeping v0 (the packed ids the hardware delivers) alive: lane id from v_mbcnt, wave index read once
// at kernel entry into an SGPR (trx_wave_index).  With five bodies inlined behind a dispatch the allocator spilled v0 to scratch at entry and
// every body paid a memory round trip to get it back.
__device__ __forceinline__ int trx_lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int trx_wave_index() { return __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6); }

template <int NV, int NW>
__device__ __forceinline__ void block_reduce_store_nw(const float (&vals)[NV], float *__restrict__ out, float *smem, int wave)
{
    // smem: >= NW*16*65 + NW*16 floats of scratch (aliases the tile box); wave: this wave's index in the block (uniform)
    constexpr int CH = 16;
    float(*red)[CH][65] = reinterpret_cast<float(*)[CH][65]>(smem);
    float(*wsum)[CH] = reinterpret_cast<float(*)[CH]>(smem + NW * CH * 65);
    const int lane = trx_lane_id(), tid = wave * 64 + lane;
#pragma unroll
    for (int c0 = 0; c0 < NV; c0 += CH) {
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (c0 + j < NV) red[wave][j][lane] = vals[c0 + j];
        __syncthreads();
        if (lane < CH && c0 + lane < NV) {
            float s = 0.f;
#pragma unroll 16
            for (int i = 0; i < 64; i++) s += red[wave][lane][i];
            wsum[wave][lane] = s;
        }
        __syncthreads();
        if (tid < CH && c0 + tid < NV) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) s += wsum[w][tid];
            out[c0 + tid] = s;
        }
        __syncthreads();
    }
}

// value is identical in every lane: pin it to an SGPR so it does not occupy a VGPR
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// wave-uniform pointer pinned to an SGPR pair (a no-op when the compiler already knows it is uniform); the asm blocks of the
// tile kernel take their base addresses as "s" operands
template <class T>
__device__ __forceinline__ T *uni_ptr(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float lane_bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// Requires vol.xn / vol.yn / vol.zn != NULL (the launcher materialises them when the caller passes NULL).
//
// Work decomposition: a 512-thread block owns one (x-tile, z-tile) COLUMN of the volume and walks it
// along y, 16 rows per tile.  Thread (lx, lz, half) keeps its voxel column (x, z) for the whole block, so
// only sum(q grad) and sum(q grad yn) live in registers (23 accumulators); the xn / zn columns of the 41
// sums are one multiply at the very end.
//
// Sample coordinates are formed as  (ATen's identity coordinate of this voxel) + (deviation of theta
// from identity): i_x = id_x(x) + (W/2)((t00-1) xn + t01 yn + t02 zn + t03), etc.  id_c uses ATen's exact
// un-normalisation roundings (unnorm<3>), so at theta = identity the coordinates are BITWISE those of
// the reference (every sample sits on a voxel there and the one-sided derivative depends on the last
// bit); for any other theta this is the same affine map to within fp32 rounding, at 4 VALU ops / voxel.
// s[100:101] (the DMA base walker of the fast loop) are outside the compiler's allocatable SGPR range on gfx950 - it warns
// that it will not preserve them, which is exactly why they are safe to use; LDS addresses are 32-bit integers by construction.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
// LTH (round 6, the carry form of a step): `theta` points at the 12 floats of THIS pair in LDS (the block's prologue has just computed them) instead of at the
// batch's theta array in global memory.
template <int MODE, class G, bool LTH = false>
__device__ __forceinline__ void tile_body(const trx_volumes &vol, const float *__restrict__ theta, const TileGeom &tg, int channels,
                                          float *__restrict__ partials, float *box, const int bx, const int by, const int rows_stride, const int wave_in)
{
    // geometry of this instantiation (bx, by: the block's index in the (blocks_per_pair, pairs x channels) grid)
    constexpr int kTX = G::TX, kTY = G::TY, kTZ = G::TZ, kBW = G::BW, kBH = G::BH, kBD = G::BD, kPP = G::PP, kBufs = G::Bufs;
    constexpr int kNH = G::NH, kRows = G::Rows, kTileWaves = G::Waves, kBW4 = G::BW4, kPlaneSlots = G::PlaneSlots, kPieces = G::Pieces;
    constexpr int kPieceFloats = G::PieceFloats, kBoxFloats = G::BoxFloats;
    (void)kNH; (void)kBD; (void)kPlaneSlots;
    // MODE 0: F1 sums, MODE 1: moments only, MODE 2: generic warp backward (`vol.target` = grad_out [B][channels][D][H][W],
    // 12 sums per (pair, channel)), MODE 3: forward warp (writes the warped volume to `partials` = out[B][channels][D][H][W])
    // MODE 4: the step kernel for losses without an NCC term (MSE and / or SSD - what the reference's rigid / affine drivers always run,
    // SURVEY Q2): 13 sums per block (sum d^2, 12 x sum d J with d = warped - target) instead of 41, 8 accumulation instructions per
    // voxel instead of 15
    constexpr int NQ = (MODE == 0) ? 3 : ((MODE == 2 || MODE == 4) ? 1 : 0);
    constexpr int NP = (MODE == 0) ? np_full(3) : (MODE == 2 ? 12 : (MODE == 4 ? kNpMse : 5));
    constexpr bool kGrad = (MODE == 0) || (MODE == 2) || (MODE == 4);
    constexpr bool kPerChannel = (MODE == 2) || (MODE == 3);   // blockIdx.y enumerates (pair, channel); channels share theta
    const int b = kPerChannel ? by / channels : by;
    const int ch = kPerChannel ? by - b * channels : 0;
    const int D = vol.D, H = vol.H, W = vol.W;
    const float *__restrict__ th = LTH ? theta : uni_ptr(theta + (size_t)b * TRX_PSTRIDE);
    const float *__restrict__ mov = uni_ptr(vol.moving + (size_t)b * vol.moving_stride + (size_t)ch * D * H * W);
    // MODE 3 has no target: `tgt` is the OUTPUT volume of this (pair, channel)
    float *__restrict__ wout = uni_ptr(partials + (size_t)by * D * H * W);
    const float *__restrict__ tgt = (MODE == 3) ? wout : uni_ptr(vol.target + (size_t)b * vol.target_stride + (MODE == 2 ? (size_t)ch * D * H * W : 0));
    const float *__restrict__ xtab = uni_ptr(vol.xn), *__restrict__ ytab = uni_ptr(vol.yn), *__restrict__ ztab = uni_ptr(vol.zn);
    const int lane = trx_lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);          // provably wave-uniform (SGPR)
    const int tid = wave * 64 + lane;
    const int lx = tid & (kTX - 1), lz = (tid / kTX) & (kTZ - 1), lh = wave / (kTileWaves / kNH);
    const float fW = (float)W, fH = (float)H, fD = (float)D;
    const float hW = 0.5f * fW, hH = 0.5f * fH, hD = 0.5f * fD;
    auto thv = [&](int k) { return LTH ? uni(th[k]) : th[k]; };   // (LDS reads are vector loads: pin the uniform values to scalar registers as the scalar loads do)
    const float t00 = thv(0), t01 = thv(1), t02 = thv(2), t03 = thv(3);
    const float t10 = thv(4), t11 = thv(5), t12 = thv(6), t13 = thv(7);
    const float t20 = thv(8), t21 = thv(9), t22 = thv(10), t23 = thv(11);
    const float sx = uni(hW * t01), sy = uni(hH * (t11 - 1.0f)), sz = uni(hD * t21);

    // XCD-aware column order: blocks b, b+8, b+16, ... share an XCD (and its L2); give each XCD a
    // contiguous slab of columns so that the halo re-reads of neighbouring columns hit the same L2.
    const int ncol = tg.ntx * tg.ntz;
    const int yseg = bx / ncol, cb = bx - yseg * ncol;
    int col = cb;
    if ((ncol & 7) == 0) col = (cb & 7) * (ncol >> 3) + (cb >> 3);
    const int X0 = (col % tg.ntx) * kTX, Z0 = (col / tg.ntx) * kTZ;
    const int nx = min(kTX, W - X0), nz = min(kTZ, D - Z0);
    // per-thread voxel column (x, z); idle lanes are clamped so every load stays in bounds
    const bool act = (lx < nx) && (lz < nz);
    const int x = X0 + (act ? lx : 0), z = Z0 + (act ? lz : 0);
    const float xn = xtab[x], zn = ztab[z];
    const float base_x = unnorm<3>(xn, fW) + hW * fmaf(t00 - 1.0f, xn, fmaf(t02, zn, t03));
    const float base_y = hH * fmaf(t10, xn, fmaf(t12, zn, t13));
    const float base_z = unnorm<3>(zn, fD) + hD * fmaf(t20, xn, fmaf(t22 - 1.0f, zn, t23));

    // Pre-image bounding box of a tile = image of its (X0, Y0, Z0) corner + a tile-independent extent:
    // the map is affine, so extremes sit at corners; d(i_c)/d(voxel step along axis a) = t_ca * S_c / S_a.
    float ext_lo[3], ext_hi[3];
    {
        const float ex[3] = {(float)(kTX - 1), (float)(kTY - 1), (float)(kTZ - 1)};
        const float slope[3][3] = {{t00, t01 * fW / fH, t02 * fW / fD}, {t10 * fH / fW, t11, t12 * fH / fD}, {t20 * fD / fW, t21 * fD / fH, t22}};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            ext_lo[c] = ext_hi[c] = 0.f;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float e = slope[c][a] * ex[a];
                ext_lo[c] += fminf(e, 0.f); ext_hi[c] += fmaxf(e, 0.f);
            }
            ext_lo[c] = uni(ext_lo[c]); ext_hi[c] = uni(ext_hi[c]);
        }
    }
    const float cxn = xtab[X0], czn = ztab[Z0];
    const float corner_x = uni(unnorm<3>(cxn, fW) + hW * fmaf(t00 - 1.0f, cxn, fmaf(t02, czn, t03)));
    const float corner_y = uni(hH * fmaf(t10, cxn, fmaf(t12, czn, t13)));
    const float corner_z = uni(unnorm<3>(czn, fD) + hD * fmaf(t20, cxn, fmaf(t22 - 1.0f, czn, t23)));

    // LDS-DMA slot of a thread in piece 0 (tile independent): box float4 (pz, dy, dx4), pz = 0 .. kPP-1.
    // rb0 = its byte offset from the box origin voxel inside the volume, d0 = packed (dz << 16 | dy << 8 | dx4).
    // Piece k: d = d0 + (kPP k << 16), byte offset rb0 + k * (kPP H W 4).  Spare lanes get dz = 100 (never needed).
    auto slot_geom = [&](int ln, unsigned &rb0, int &d0) {
        const int q = wave * 64 + ln;
        const int pz = q / kPlaneSlots, r = q - pz * kPlaneSlots;
        const int dy = r / kBW4, dx4 = r - dy * kBW4;
        const bool valid = q < kPP * kPlaneSlots;
        rb0 = valid ? (unsigned)((pz * H + dy) * W + dx4 * 4) * 4u : 0u;
        d0 = valid ? ((pz << 16) | (dy << 8) | dx4) : (100 << 16);
    };
    const unsigned piece_stride = (unsigned)(kPP * H * W) * 4u;   // bytes between the pieces of one thread inside the volume
    constexpr int kDShift = (kPP == 2) ? 17 : 18;                  // d of piece k = d0 + (k << kDShift)
    // A needed float4 slot that the DMA does not fetch - outside the volume, or straddling its +x face when W % 4 != 0
    // (global_load_lds_dwordx4 itself takes any 4-byte aligned address) - is filled element by element with zero padding.
    auto fill_slot = [&](float *dst, int gz, int gy, int gx) {
        const bool rowin = ((unsigned)gz < (unsigned)D) && ((unsigned)gy < (unsigned)H);
        const float *row = mov + ((size_t)(rowin ? gz : 0) * H + (rowin ? gy : 0)) * W;
        float4 v;
        v.x = (rowin && (unsigned)(gx + 0) < (unsigned)W) ? row[gx + 0] : 0.f;
        v.y = (rowin && (unsigned)(gx + 1) < (unsigned)W) ? row[gx + 1] : 0.f;
        v.z = (rowin && (unsigned)(gx + 2) < (unsigned)W) ? row[gx + 2] : 0.f;
        v.w = (rowin && (unsigned)(gx + 3) < (unsigned)W) ? row[gx + 3] : 0.f;
        *reinterpret_cast<float4 *>(dst) = v;
    };
    static_assert(kPP == 2 || kPP == 4, "piece planes");

    F1Acc acc;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
    acc.M01 = acc.M23 = (f2)(0.f);
    acc.M4 = 0.f;
    const int j0 = lh * kRows;                              // first row of this thread's half
    const int toff = (z * H + j0) * W + x;                  // this thread's target offset inside a tile's row block

    // Geometry of up to 64 tiles at a time, ONE TILE PER LANE (the per-tile cost is then a few
    // v_readlane instead of ~40 VALU instructions): box origin, needed extent, fits / interior flags.
    const int ty_begin = yseg * tg.tiles_per_seg, ty_end = min((yseg + 1) * tg.tiles_per_seg, tg.nty);
    int g_ox = 0, g_oy = 0, g_oz = 0, g_pk = 0;
    auto lane_geometry = [&](int ty_first) {
        const int ty = min(ty_first + lane, tg.nty - 1);
        const float yn0 = ytab[ty * kTY], yid0 = unnorm<3>(yn0, fH);
        const float cx = fmaf(sx, yn0, corner_x), cy = yid0 + fmaf(sy, yn0, corner_y), cz = fmaf(sz, yn0, corner_z);
        const float slack = 0.05f;   // fp32 rounding + table non-uniformity of interior points vs the corner + extent bound
        bool fits = (fabsf(cx) < 1.0e6f) && (fabsf(cy) < 1.0e6f) && (fabsf(cz) < 1.0e6f);   // also rejects NaN
        int ox = 0, oy = 0, oz = 0, ex4 = 0, ey = 0, ez = 0;
        bool interior = false;
        if (fits) {
            const int lx0 = (int)floorf(cx + ext_lo[0] - slack), hx1 = (int)floorf(cx + ext_hi[0] + slack) + 1;
            oy = (int)floorf(cy + ext_lo[1] - slack); const int hy1 = (int)floorf(cy + ext_hi[1] + slack) + 1;
            oz = (int)floorf(cz + ext_lo[2] - slack); const int hz1 = (int)floorf(cz + ext_hi[2] + slack) + 1;
            ox = lx0 & ~3;
            ex4 = ((hx1 - ox) >> 2) + 1; ey = hy1 - oy + 1; ez = hz1 - oz + 1;
            fits = (ex4 <= kBW4) && (ey <= kBH) && (ez <= kBD);
            // the whole box capacity lies inside the volume: no zero padding needed for this tile
            interior = (ox >= 0) && (oy >= 0) && (oz >= 0) && (ox + kBW <= W) && (oy + kBH <= H) && (oz + kBD <= D);
        }
        g_ox = ox; g_oy = oy; g_oz = oz;
        g_pk = fits ? ((ex4 - 1) | ((ey - 1) << 8) | ((ez - 1) << 16) | (1 << 24) | ((interior ? 1 : 0) << 25)) : 0;
    };
    int ty = ty_begin;
    // ================= fast loop: full 16-row tiles whose box fits =================
    // No per-lane branch around the accumulation: lanes outside a partial x / z tile work on the clamped
    // column (x, z) = (X0, Z0) - every address stays valid - and their sums are discarded after the loop, so
    // the 23 packed accumulators live in one set of registers with no copies at control-flow joins.
    // The VALU is the busiest unit of this kernel (rocprof: ~70 % issue utilisation), so the loop is written to
    // the instruction: addresses that are (uniform base + per-thread 32-bit offset) use the SGPR-base form of
    // global_load (no VALU address arithmetic), per-row constants stay in SGPRs, the LDS address is a
    // shift-add + two mad24 with SGPR strides.
    {
        const unsigned toffb = (unsigned)toff * 4u;
        const unsigned lane7b = (unsigned)(lane & (kRows - 1)) * 4u;
        float sxv, syv, szv;   // VGPR copies of the uniform slopes: the per-row yn can then be the (single) SGPR operand
        asm("v_mov_b32 %0, %1" : "=v"(sxv) : "s"(sx));
        asm("v_mov_b32 %0, %1" : "=v"(syv) : "s"(sy));
        asm("v_mov_b32 %0, %1" : "=v"(szv) : "s"(sz));
        int ys_s, zs_s;        // LDS strides (bytes) pinned in SGPRs: gfx9 VOP3 takes no literal operand
        asm("s_mov_b32 %0, %1" : "=s"(ys_s) : "i"(kBW * 4));
        asm("s_mov_b32 %0, %1" : "=s"(zs_s) : "i"(kBW * kBH * 4));
        const unsigned box_lds = (unsigned)(uintptr_t)box;   // LDS byte address of the box
        unsigned rb0;                                        // byte offset of this thread's piece-0 DMA slot
        {
            int d0;
            slot_geom(lane, rb0, d0);
        }
        // More than 8 pieces (GeomR's deep box): one exec mask per piece would not fit the SGPR file (the reloads cost 18 %), but the
        // masks are structured - lane (pz, dy, dx4) of piece k fetches iff its (dy, dx4) is wanted and plane 2k + pz is: two masks
        // (pz = 0 / 1 lanes with a wanted (dy, dx4)) and one bit per box plane rebuild each piece's mask with scalar instructions.
        constexpr bool kZMask = kPieces > 8;
        static_assert(!kZMask || (kPP == 2 && kBD <= 32), "plane-bit masks: pieces of two planes");
        constexpr int kNM = kZMask ? 1 : kPieces;
        unsigned long long m_ld[kNM];                        // cached exec masks of the DMA pieces (wave-uniform)
#pragma unroll
        for (int k = 0; k < kNM; k++) m_ld[k] = 0;
        unsigned long long m_a0 = 0, m_a1 = 0;               // kZMask: lanes of plane 0 / 1 of a piece whose (dy, dx4) is fetched
        unsigned m_zb = 0;                                   // kZMask: bit z = box plane z is fetched
        unsigned m_oob = 0;                                  // bit k: slot k of this thread is needed but outside the volume
        unsigned m_part = 0;                                 // bit k: slot k straddles x = W (W % 4 != 0): zero its tail after landing
        // Fetched extent = the LARGEST pre-image extent any tile of this theta can have (capped at the box): the exact
        // extent of a tile flips between two values with the fractional position of its corner, and every change
        // would invalidate the cached masks; one extra row / plane / float4 of DMA is cheaper than that.
        int lim_blk;
        {
            const float slack2 = 0.1f;
            const int ex4m = (((int)floorf(ext_hi[0] - ext_lo[0] + slack2) + 5) >> 2) + 1;   // hx1 - lx0 <= floor(span) + 2, + 3 of alignment
            const int eym = (int)floorf(ext_hi[1] - ext_lo[1] + slack2) + 3, ezm = (int)floorf(ext_hi[2] - ext_lo[2] + slack2) + 3;
            lim_blk = __builtin_amdgcn_readfirstlane(((min(ezm, kBD) - 1) << 16) | ((min(eym, kBH) - 1) << 8) | (min(ex4m, kBW4) - 1));
        }
        int m_lim = -1, m_lo = -1, m_hi = -1, m_px = -2;
        TRX_TM_INIT();
        typedef const __attribute__((address_space(3))) f2u *lds_f2;
        // Tiles come in chunks of 64 (geometry: one tile per lane); the leading run of fast tiles of a chunk is a plain
        // counted loop - no exit in the middle, so the accumulators stay in one register set.
        while (ty < ty_end) {
          lane_geometry(ty);
          const int chunk = min(64, ty_end - ty);
          // W % 4 != 0: a tile whose needed extent reaches the float4 that straddles x = W (only columns at the +x face)
          // (the straddling float4 of a row is fetched whole, i.e. up to 12 bytes into the next row - except on the last row
          // of the volume, where that would leave the allocation: that one tile per pair goes to the generic loop)
          const bool xpart = ((W & 3) != 0) && (g_ox + 4 * ((lim_blk & 0xff) + 1) > W - (W & 3)) &&
                             (g_oy + ((lim_blk >> 8) & 0xff) >= H - 1) && (g_oz + (lim_blk >> 16) >= D - 1);
          const bool ok = (lane < chunk) && ((g_pk >> 24) & 1) && ((ty + lane + 1) * kTY <= H) && !xpart;
          const unsigned long long bad = ~__builtin_amdgcn_ballot_w64(ok);
          const int nf = bad ? __builtin_ctzll(bad) : 64;
          // ---- box of fast tile `g` of this chunk -> LDS buffer `buf`: refresh the cached exec masks if the tile's slot
          // range changed, issue the DMA pieces (no wait), zero-fill needed cells that lie outside the volume.
          const char *dma_base = nullptr;   // of the box prepared last by issue_box (for piece-wise issue)
          unsigned dma_lds = 0;
          auto issue_box = [&](int g, int buf, bool spread) {
              const int ox = __builtin_amdgcn_readlane(g_ox, g), oy = __builtin_amdgcn_readlane(g_oy, g), oz = __builtin_amdgcn_readlane(g_oz, g);
              const int lim = lim_blk;         // (ez-1) << 16 | (ey-1) << 8 | (ex4-1)
              const int loz = max(0, -oz), loy = max(0, -oy), lox = max(0, -(ox >> 2));
              const int hiz = min(lim >> 16, D - 1 - oz), hiy = min((lim >> 8) & 0xff, H - 1 - oy), hix = min(lim & 0xff, ((W - ox + 3) >> 2) - 1);   // includes the float4 straddling x = W (W % 4 != 0)
              const bool none = (hiz < loz) || (hiy < loy) || (hix < lox);   // the whole pre-image lies outside the volume
              const int lo = none ? 0x7f7f7f : ((loz << 16) | (loy << 8) | lox);
              const int hi = none ? 0 : ((hiz << 16) | (hiy << 8) | hix);
              // float4 index of the slot straddling x = W (W % 4 != 0) if this tile fetches it.  It moves with ox, so it is part of
              // the cache key: two tiles of a column can share (lim, lo, hi) while the straddler sits in different slots.
              const int part_x = ((W & 3) && !none && ((W - ox) >> 2) <= hix) ? ((W - ox) >> 2) : -1;
              if (lim != m_lim || lo != m_lo || hi != m_hi || part_x != m_px) {
                  m_lim = lim; m_lo = lo; m_hi = hi; m_px = part_x;
                  m_oob = 0;
                  m_part = 0;
                  int ln = lane;   // opaque copy: keeps the slot decode inside this (rarely taken) branch
                  asm volatile("" : "+v"(ln));
                  unsigned rbx;
                  int d0;
                  slot_geom(ln, rbx, d0);
#pragma unroll
                  for (int k = 0; k < kPieces; k++) {   // per-field compares on the packed (dz, dy, dx4): no field may borrow
                      const int d = d0 + (k << kDShift);
                      const bool need = (((lim - d) & 0x80808080) == 0);
                      const bool ld = need && (((hi - d) & 0x80808080) == 0) && (((d - lo) & 0x80808080) == 0);
                      if constexpr (!kZMask) m_ld[k] = __builtin_amdgcn_ballot_w64(ld);
                      if (need && !ld) m_oob |= 1u << k;
                      if (ld && (d & 0xff) == part_x) m_part |= 1u << k;   // fetched whole; its tail past x = W is zeroed after landing
                  }
                  if constexpr (kZMask) {
                      const int dyx = d0 & 0xffff;
                      const bool xy = (((((lim & 0xffff) - dyx) | ((hi & 0xffff) - dyx) | (dyx - (lo & 0xffff))) & 0x8080) == 0);
                      m_a0 = __builtin_amdgcn_ballot_w64(xy && (d0 >> 16) == 0);
                      m_a1 = __builtin_amdgcn_ballot_w64(xy && (d0 >> 16) == 1);
                      m_zb = (unsigned)__builtin_amdgcn_readfirstlane((int)(none ? 0u : (((2u << hiz) - 1u) & ~((1u << loz) - 1u))));   // hiz <= the needed depth by construction
                  }
              }
              dma_base = reinterpret_cast<const char *>(mov + (ptrdiff_t)((oz * H + oy) * W + ox));   // uniform; may point below `mov` (those lanes are masked)
              dma_lds = box_lds + (unsigned)buf * (kBoxFloats * 4u) + (unsigned)wave * 1024u;
              if (TRX_DBG_SKIP != 2 && !spread) {
                  const float *mbase = reinterpret_cast<const float *>(dma_base);
                  const unsigned lds0 = dma_lds;
                  unsigned long long sv;
                  unsigned m0s;
                  static_assert((kPieces >= 6 && kPieces <= 8) || (kPieces >= 9 && kPieces <= 13), "the DMA block below is written for 6, 7, 8 or 9..13 pieces");
                  // one exec mask + one SGPR-base load per piece; s[100:101] walks the volume by kPP planes per piece
#if TRX_DMA_EXECZ_SKIP
#define TRX_DMA_SKIP "s_cbranch_execz 1f\n\t"
#else
#define TRX_DMA_SKIP
#endif
#define TRX_DMA_NEXT(K)                                  \
    "s_add_u32 s100, s100, %[vstr]\n\t"                  \
    "s_addc_u32 s101, s101, 0\n\t"                       \
    "s_add_u32 m0, m0, %[pstr]\n\t"                      \
    "s_mov_b64 exec, %[k" #K "]\n\t"                     \
    TRX_DMA_SKIP                                         \
    "global_load_lds_dwordx4 %[off], s[100:101]" TRX_BOX_POLICY "\n\t" \
    "1:\n\t"
#define TRX_DMA_HEAD                                     \
    "s_mov_b64 %[sv], exec\n\t"                          \
    "s_mov_b32 %[m0s], m0\n\t"                           \
    "s_mov_b64 s[100:101], %[base]\n\t"                  \
    "s_mov_b32 m0, %[lds]\n\t"                           \
    "s_mov_b64 exec, %[k0]\n\t"                          \
    TRX_DMA_SKIP                                         \
    "global_load_lds_dwordx4 %[off], s[100:101]" TRX_BOX_POLICY "\n\t"     \
    "1:\n\t"                                             \
    TRX_DMA_NEXT(1) TRX_DMA_NEXT(2) TRX_DMA_NEXT(3) TRX_DMA_NEXT(4) TRX_DMA_NEXT(5)
#define TRX_DMA_TAIL "s_mov_b64 exec, %[sv]\n\t" "s_mov_b32 m0, %[m0s]"
                  if constexpr (kZMask) {
                      static_assert(kPieces <= 13, "plane-bit DMA block: up to 13 pieces (the plane bits of pieces a shallower box lacks are never set)");
                      unsigned long long t0, t1;
#define TRX_DMA_ZSEL(B0, B1)                             \
    "s_bitcmp1_b32 %[zb], " #B0 "\n\t"                   \
    "s_cselect_b64 %[t0], %[a0], 0\n\t"                 \
    "s_bitcmp1_b32 %[zb], " #B1 "\n\t"                   \
    "s_cselect_b64 %[t1], %[a1], 0\n\t"                 \
    "s_or_b64 exec, %[t0], %[t1]\n\t"                   \
    "s_cbranch_execz 1f\n\t"                            \
    "global_load_lds_dwordx4 %[off], s[100:101]" TRX_BOX_POLICY "\n\t" \
    "1:\n\t"
#define TRX_DMA_ZNEXT(B0, B1)                            \
    "s_add_u32 s100, s100, %[vstr]\n\t"                 \
    "s_addc_u32 s101, s101, 0\n\t"                      \
    "s_add_u32 m0, m0, %[pstr]\n\t" TRX_DMA_ZSEL(B0, B1)
                      asm volatile("s_mov_b64 %[sv], exec\n\t"
                                   "s_mov_b32 %[m0s], m0\n\t"
                                   "s_mov_b64 s[100:101], %[base]\n\t"
                                   "s_mov_b32 m0, %[lds]\n\t" TRX_DMA_ZSEL(0, 1)
                                   TRX_DMA_ZNEXT(2, 3) TRX_DMA_ZNEXT(4, 5) TRX_DMA_ZNEXT(6, 7) TRX_DMA_ZNEXT(8, 9) TRX_DMA_ZNEXT(10, 11)
                                   TRX_DMA_ZNEXT(12, 13) TRX_DMA_ZNEXT(14, 15) TRX_DMA_ZNEXT(16, 17) TRX_DMA_ZNEXT(18, 19)
                                   TRX_DMA_ZNEXT(20, 21) TRX_DMA_ZNEXT(22, 23) TRX_DMA_ZNEXT(24, 25) TRX_DMA_TAIL
                                   : [sv] "=&s"(sv), [m0s] "=&s"(m0s), [t0] "=&s"(t0), [t1] "=&s"(t1)
                                   : [lds] "s"(lds0), [base] "s"(mbase), [pstr] "i"(kPieceFloats * 4), [vstr] "s"(piece_stride), [off] "v"(rb0),
                                     [a0] "s"(m_a0), [a1] "s"(m_a1), [zb] "s"(__builtin_amdgcn_readfirstlane((int)m_zb))
                                   : "memory", "scc", "s100", "s101");
#undef TRX_DMA_ZSEL
#undef TRX_DMA_ZNEXT
                  } else if constexpr (kPieces == 8) {
                      asm volatile(TRX_DMA_HEAD TRX_DMA_NEXT(6) TRX_DMA_NEXT(7) TRX_DMA_TAIL
                                   : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                                   : [lds] "s"(lds0), [base] "s"(mbase), [pstr] "i"(kPieceFloats * 4), [vstr] "s"(piece_stride), [off] "v"(rb0),
                                     [k0] "s"(m_ld[0]), [k1] "s"(m_ld[1]), [k2] "s"(m_ld[2]), [k3] "s"(m_ld[3]), [k4] "s"(m_ld[4]),
                                     [k5] "s"(m_ld[5]), [k6] "s"(m_ld[kPieces > 6 ? 6 : 0]), [k7] "s"(m_ld[kPieces - 1])
                                   : "memory", "scc", "s100", "s101");
                  } else if constexpr (kPieces == 7) {
                      asm volatile(TRX_DMA_HEAD TRX_DMA_NEXT(6) TRX_DMA_TAIL
                                   : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                                   : [lds] "s"(lds0), [base] "s"(mbase), [pstr] "i"(kPieceFloats * 4), [vstr] "s"(piece_stride), [off] "v"(rb0),
                                     [k0] "s"(m_ld[0]), [k1] "s"(m_ld[1]), [k2] "s"(m_ld[2]), [k3] "s"(m_ld[3]), [k4] "s"(m_ld[4]),
                                     [k5] "s"(m_ld[5]), [k6] "s"(m_ld[kPieces - 1])
                                   : "memory", "scc", "s100", "s101");
                  } else {
                      asm volatile(TRX_DMA_HEAD TRX_DMA_TAIL
                                   : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                                   : [lds] "s"(lds0), [base] "s"(mbase), [pstr] "i"(kPieceFloats * 4), [vstr] "s"(piece_stride), [off] "v"(rb0),
                                     [k0] "s"(m_ld[0]), [k1] "s"(m_ld[1]), [k2] "s"(m_ld[2]), [k3] "s"(m_ld[3]), [k4] "s"(m_ld[4]),
                                     [k5] "s"(m_ld[5])
                                   : "memory", "scc", "s100", "s101");
                  }
#undef TRX_DMA_NEXT
#undef TRX_DMA_SKIP
#undef TRX_DMA_HEAD
#undef TRX_DMA_TAIL
              }
              if (m_oob) {   // zero padding: needed cells outside the volume (tiles at a volume face only)
                  float zero;
                  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));   // materialised here, not kept live across the loop
#pragma unroll
                  for (int k = 0; k < kPieces; k++)
                      if (m_oob & (1u << k))
                          *reinterpret_cast<float4 *>(box + buf * kBoxFloats + k * kPieceFloats + (wave * 64 + lane) * 4) = make_float4(zero, zero, zero, zero);
              }
          };
          // W % 4 != 0: the float4 that straddles the +x face was fetched whole (its tail belongs to the next row): zero the tail
          auto zero_tails = [&](int buf) {
#pragma unroll
              for (int k = 0; k < kPieces; k++)
                  if (m_part & (1u << k)) {
                      float *sl = box + buf * kBoxFloats + k * kPieceFloats + (wave * 64 + lane) * 4;
                      for (int e = W & 3; e < 4; e++) sl[e] = 0.f;
                  }
          };
          // one DMA piece of the box prepared by issue_box(.., spread = true): issued between the rows of the gather so that
          // the TA drains the pieces (64 B/clk per CU) while the VALU works, instead of every wave queueing all of them first
          auto issue_piece = [&](int k) {
              if (TRX_DBG_SKIP == 2) return;
              unsigned long long sv;
              unsigned m0s;
              asm volatile("s_mov_b64 %[sv], exec\n\t"
                           "s_mov_b32 %[m0s], m0\n\t"
                           "s_mov_b32 m0, %[lds]\n\t"
                           "s_mov_b64 exec, %[mk]\n\t"
                           "global_load_lds_dwordx4 %[off], %[base]\n\t"
                           "s_mov_b64 exec, %[sv]\n\t"
                           "s_mov_b32 m0, %[m0s]"
                           : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                           : [lds] "s"(dma_lds + (unsigned)k * (kPieceFloats * 4u)), [base] "s"(dma_base + (size_t)k * piece_stride), [off] "v"(rb0),
                             [mk] "s"(kZMask ? ((((m_zb >> (2 * k)) & 1u) ? m_a0 : 0ull) | (((m_zb >> (2 * k + 1)) & 1u) ? m_a1 : 0ull)) : m_ld[kZMask ? 0 : k])
                           : "memory");
          };
          // ---- the 8 rows of this thread in fast tile `g`, gathered from LDS buffer `buf`.  tnext != nullptr: after row j is
          // consumed, row j of the tile at `tnext` is fetched into the same register (target prefetch without extra VGPRs).
          auto gather_tile = [&](int g, int buf, float yn_l, float yid_l, float (&tv)[kRows], const float *tnext, bool dma_next) {
              const int Y0 = (ty + g) * kTY;
              const int ox = __builtin_amdgcn_readlane(g_ox, g), oy = __builtin_amdgcn_readlane(g_oy, g), oz = __builtin_amdgcn_readlane(g_oz, g);
              float yn_r[kRows], yid_r[kRows];
#pragma unroll
              for (int j = 0; j < kRows; j++) { yn_r[j] = lane_bcast(yn_l, j); yid_r[j] = lane_bcast(yid_l, j); }
              if (TRX_DBG_SKIP == 1) return;
              const int bpb = (int)box_lds + buf * (kBoxFloats * 4) - ((oz * kBH + oy) * kBW + ox) * 4;   // LDS byte address of voxel (0,0,0) of the volume
              // software pipeline: the 4 LDS reads of row j+1 are issued before the arithmetic of row j
              struct Fetch { f2 r00, r01, r10, r11; float fx, fy, fz; };
              auto fetch = [&](int j) -> Fetch {
                  const float yn = yn_r[j];
                  const float ix = fmaf(sxv, yn, base_x);
                  const float iy = yid_r[j] + fmaf(syv, yn, base_y);
                  const float iz = fmaf(szv, yn, base_z);
                  int a0, a1, a2, a3;
                  asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a0) : "v"(floor_to_int(ix)), "s"(bpb));
                  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a1) : "v"(floor_to_int(iy)), "s"(ys_s), "v"(a0));
                  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a2) : "v"(floor_to_int(iz)), "s"(zs_s), "v"(a1));
                  asm("v_add_u32 %0, %1, %2" : "=v"(a3) : "s"(zs_s), "v"(a2));
                  Fetch f;
                  f.r00 = *(lds_f2)(unsigned)a2; f.r01 = *(lds_f2)(unsigned)(a2 + kBW * 4);
                  f.r10 = *(lds_f2)(unsigned)a3; f.r11 = *(lds_f2)(unsigned)(a3 + kBW * 4);
                  f.fx = __builtin_amdgcn_fractf(ix); f.fy = __builtin_amdgcn_fractf(iy); f.fz = __builtin_amdgcn_fractf(iz);
                  return f;
              };
              Fetch cur = fetch(0);
#pragma unroll
              for (int j = 0; j < kRows; j++) {
                  Fetch nxt;
                  if (TRX_SWP && j + 1 < kRows) nxt = fetch(j + 1);
                  if (kBufs == 2 && j < kPieces && dma_next) issue_piece(j);
                  const Samp3 sm = lerp3_pairs<kGrad>(cur.r00, cur.r01, cur.r10, cur.r11, cur.fx, cur.fy, cur.fz);
                  if constexpr (MODE == 3) { if (act) wout[(size_t)Y0 * W + (unsigned)(toff + j * W)] = sm.v; }
                  else f1_accumulate_pk<MODE>(sm, tv[j], yn_r[j], acc);
                  if (kBufs == 2 && TRX_DBG_SKIP != 3 && MODE != 3)
                      asm volatile("global_load_dword %0, %1, %2" : "=v"(tv[j]) : "v"(toffb), "s"(tnext + (size_t)j * W));
                  if (j + 1 < kRows) cur = TRX_SWP ? nxt : fetch(j + 1);
              }
          };
          auto load_targets = [&](int g, float (&tv)[kRows]) {
              const float *trow = tgt + (size_t)(ty + g) * kTY * W;   // uniform (toffb holds the row offset of this half)
#pragma unroll
              for (int j = 0; j < kRows; j++) {
                  if (TRX_DBG_SKIP == 3 || MODE == 3) tv[j] = 1.f;
                  else asm volatile("global_load_dword %0, %1, %2" TRX_TGT_POLICY : "=v"(tv[j]) : "v"(toffb), "s"(trow + (size_t)j * W) : "memory");
              }
          };
          float tv[kRows];
          if constexpr (kBufs == 1) {
            // one box, two blocks per CU: burst - wait - barrier - gather - barrier
            for (int gl = 0; gl < nf; gl++) {
              TRX_TM_STAMP(tm0);
              // row tables of this wave's 8 rows in lanes 0..7 (broadcast in gather_tile with constant-lane v_readlane)
              float yn_l;
              asm volatile("global_load_dword %0, %1, %2" : "=v"(yn_l) : "v"(lane7b), "s"(ytab + (ty + gl) * kTY + j0) : "memory");
              __builtin_amdgcn_s_setprio(TRX_STAGE_PRIO);
              load_targets(gl, tv);
              issue_box(gl, 0, false);
              __builtin_amdgcn_s_setprio(0);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // box pieces and the target column have landed
              if (m_part) zero_tails(0);
              {
#pragma unroll
                  for (int j = 0; j < kRows; j++) asm volatile("" : "+v"(tv[j]));
                  asm volatile("" : "+v"(yn_l));
              }
              const float yid_l = unnorm<3>(yn_l, fH);
              TRX_TM_STAMP(tm1);
              __syncthreads();
              TRX_TM_STAMP(tm2);
              gather_tile(gl, 0, yn_l, yid_l, tv, nullptr, false);
              TRX_TM_STAMP(tm3);
              __syncthreads();   // the box is overwritten by the next tile
              TRX_TM_TILE_DONE();
            }
          } else {
            // two boxes, one block per CU: the DMA of tile g+1 (and, row by row, its target column) is in flight while
            // tile g is gathered; one barrier per tile (it also tells that every wave is done with the other box)
            if (nf > 0) {
                load_targets(0, tv);
                issue_box(0, 0, false);
            }
            for (int gl = 0; gl < nf; gl++) {
              TRX_TM_STAMP(tm0);
              float yn_l;
              asm volatile("global_load_dword %0, %1, %2" : "=v"(yn_l) : "v"(lane7b), "s"(ytab + (ty + gl) * kTY + j0) : "memory");
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's box pieces (issued one tile ago) and target column
              if (m_part) zero_tails(gl & 1);
              {
#pragma unroll
                  for (int j = 0; j < kRows; j++) asm volatile("" : "+v"(tv[j]));
                  asm volatile("" : "+v"(yn_l));
              }
              const float yid_l = unnorm<3>(yn_l, fH);
              TRX_TM_STAMP(tm1);
              __syncthreads();
              TRX_TM_STAMP(tm2);
              const int gn = (gl + 1 < nf) ? gl + 1 : gl;
              if (gl + 1 < nf) issue_box(gl + 1, (gl + 1) & 1, TRX_DMA_SPREAD);
              TRX_TM_STAMP(tm3);
              gather_tile(gl, gl & 1, yn_l, yid_l, tv, tgt + (size_t)(ty + gn) * kTY * W, TRX_DMA_SPREAD && gl + 1 < nf);
              TRX_TM_TILE_DONE();
            }
            // the prefetch issued during the last tile is still in flight: its destination registers must not be reused,
            // and the generic loop below must not overwrite box 0 while a slower wave still gathers from it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < kRows; j++) asm volatile("" : "+v"(tv[j]));
            __syncthreads();
          }
          ty += nf;
          if (nf < chunk) break;   // the generic loop takes over at tile ty (same 64-tile chunking, geometry already in g_*)
        }
        TRX_TM_STORE();
        if (!act) {
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
            acc.M01 = acc.M23 = (f2)(0.f);
            acc.M4 = 0.f;
        }
    }

    // slot geometry for the generic loop, recomputed here from an opaque copy of the lane id so that it is not
    // live across the fast loop
    unsigned rb0;
    int d0;
    {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        slot_geom(ln, rb0, d0);
    }
    int prev_lim = -1;
    unsigned needmask = 0;    // bit k: DMA slot k of this thread lies inside the needed extent of the current tile

    // ---- box of one tile straight into LDS (LDS-DMA, no staging VGPRs); returns after the data has landed (this
    // wave's part: the caller still needs the block barrier).  Only the float4 slots inside the tile's actual pre-image
    // extent (ex4 x ey x ez) are fetched: lanes outside it are masked off, so the bytes a CU ingests track the need,
    // not the box capacity.
    auto stage_box = [&](int pk, int ox, int oy, int oz) {
        const bool interior = (pk >> 25) & 1;
        const int lim = pk & 0xffffff;   // (ez-1) << 16 | (ey-1) << 8 | (ex4-1)
        if (lim != prev_lim) {           // extents rarely change along a column: refresh the slot mask only then
            prev_lim = lim;
            needmask = 0;
#pragma unroll
            for (int k = 0; k < kPieces; k++)   // per-field compare (dz,dy,dx4) <= lim: no field of lim-d may borrow
                if (((lim - (d0 + (k << kDShift))) & 0x80808080) == 0) needmask |= 1u << k;
        }
        const int obase = (oz * H + oy) * W + ox;
        unsigned oob = 0;   // bit k: slot k of this thread is needed but lies outside the volume
        if (interior) {
            const char *__restrict__ mbase = reinterpret_cast<const char *>(mov + obase);    // uniform; obase >= 0 here
#pragma unroll
            for (int k = 0; k < kPieces; k++)
                if ((needmask & (1u << k)) && TRX_DBG_SKIP != 2)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(mbase + (size_t)k * piece_stride + rb0),
                                                     box + k * kPieceFloats + wave * 256, 16, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < kPieces; k++)
                if ((needmask & (1u << k)) && TRX_DBG_SKIP != 2) {
                    const int d = d0 + (k << kDShift);
                    const int gz = oz + (d >> 16), gy = oy + ((d >> 8) & 0xff), gx = ox + (d & 0xff) * 4;
                    const bool inb = ((unsigned)gz < (unsigned)D) && ((unsigned)gy < (unsigned)H) && (gx >= 0) && (gx + 4 <= W);   // whole slot inside
                    const unsigned idx = inb ? (unsigned)(obase + (int)((rb0 + k * piece_stride) >> 2)) : 0u;
                    if (!inb) oob |= 1u << k;
                    __builtin_amdgcn_global_load_lds(mov + idx, box + k * kPieceFloats + wave * 256, 16, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (oob) {   // slots the DMA skipped (boundary tiles only): zero padding / partial rows
#pragma unroll
            for (int k = 0; k < kPieces; k++)
                if (oob & (1u << k)) {
                    const int d = d0 + (k << kDShift);
                    fill_slot(box + k * kPieceFloats + (wave * 64 + lane) * 4, oz + (d >> 16), oy + ((d >> 8) & 0xff), ox + (d & 0xff) * 4);
                }
        }
    };
    // ---- one voxel gathered from the LDS box: coordinates, 4 paired reads, trilinear value (+ gradient)
    auto gather = [&](const float *bp, float yn, float yid) -> Samp3 {
        const float ix = fmaf(sx, yn, base_x);
        const float iy = yid + fmaf(sy, yn, base_y);
        const float iz = fmaf(sz, yn, base_z);
        const int a = __mul24(floor_to_int(iz), kBH * kBW) + __mul24(floor_to_int(iy), kBW) + floor_to_int(ix);
        const float *p = bp + a;
        const f2 r00 = *reinterpret_cast<const f2u *>(p), r01 = *reinterpret_cast<const f2u *>(p + kBW);
        const f2 r10 = *reinterpret_cast<const f2u *>(p + kBW * kBH), r11 = *reinterpret_cast<const f2u *>(p + kBW * kBH + kBW);
        return lerp3_pairs<kGrad>(r00, r01, r10, r11, __builtin_amdgcn_fractf(ix), __builtin_amdgcn_fractf(iy), __builtin_amdgcn_fractf(iz));
    };

    // ================= generic loop: partial last tile, tiles whose box does not fit, W % 4 != 0 =================
    for (; ty < ty_end; ty++) {
        const int gl = (ty - ty_begin) & 63;
        if (gl == 0) lane_geometry(ty);
        const int Y0 = ty * kTY;
        const int ny = min(kTY, H - Y0);
        // row tables of this tile, one row per lane (lanes 0..15), broadcast later with v_readlane
        const float yn_l = ytab[Y0 + min(lane & (kTY - 1), ny - 1)];
        const float yid_l = unnorm<3>(yn_l, fH);
        const int pk = __builtin_amdgcn_readlane(g_pk, gl);
        const int ox = __builtin_amdgcn_readlane(g_ox, gl), oy = __builtin_amdgcn_readlane(g_oy, gl), oz = __builtin_amdgcn_readlane(g_oz, gl);
        const bool fits = (pk >> 24) & 1;
        const float *__restrict__ trow = tgt + (size_t)Y0 * W;   // uniform base of this tile's target rows

        if (fits) {
            // staging waves outrank the co-resident block's gather waves: their loads should enter the memory
            // system as early as possible, the VALU work they displace is short
            __builtin_amdgcn_s_setprio(TRX_STAGE_PRIO);
            float tv[kRows];
#pragma unroll
            for (int j = 0; j < kRows; j++)
                tv[j] = (TRX_DBG_SKIP == 3 || MODE == 3) ? 1.f : trow[(unsigned)(toff + (min(j0 + j, ny - 1) - j0) * W)];
            stage_box(pk, ox, oy, oz);
            __syncthreads();
            // wave-uniform row constants of this wave's half (rows j0 .. j0+7) -> SGPRs.  The v_readlane MUST
            // run here, in uniform control flow: inside `if (act)` lanes 0..15 may be inactive (partial x tile)
            // and the compiler is free to sink the computation of yn_l / yid_l into that branch.
            float yn_r[kRows], yid_r[kRows];
#pragma unroll
            for (int j = 0; j < kRows; j++) { yn_r[j] = lane_bcast(yn_l, j0 + j); yid_r[j] = lane_bcast(yid_l, j0 + j); }
            if (act && TRX_DBG_SKIP != 1) {
                const float *bp = box - ((oz * kBH + oy) * kBW + ox);
#pragma unroll
                for (int j = 0; j < kRows; j++)
                    if (j0 + j < ny) {
                        const Samp3 sm = gather(bp, yn_r[j], yid_r[j]);
                        if constexpr (MODE == 3) wout[(size_t)Y0 * W + (unsigned)(toff + j * W)] = sm.v;
                        else f1_accumulate_pk<MODE>(sm, tv[j], yn_r[j], acc);
                    }
            }
            __syncthreads();   // the box is overwritten by the next tile
        } else if (act) {
            // large deformation (or W % 4 != 0): gather straight from global memory (L2), two rows in flight:
            // the 4 pair loads + the target of row j+1 are issued before the arithmetic of row j
            struct GFetch { f2 r00, r01, r10, r11; float fx, fy, fz, yn, yv; };
            auto gfetch = [&](int j) -> GFetch {
                GFetch g;
                g.yn = ytab[Y0 + j];
                const float ix = fmaf(sx, g.yn, base_x);
                const float iy = unnorm<3>(g.yn, fH) + fmaf(sy, g.yn, base_y);
                const float iz = fmaf(sz, g.yn, base_z);
                const float flx = floorf(ix), fly = floorf(iy), flz = floorf(iz);
                g.fx = ix - flx; g.fy = iy - fly; g.fz = iz - flz;
                const int x0 = (int)flx, y0 = (int)fly, z0 = (int)flz;
                const bool interior = ((unsigned)x0 < (unsigned)(W - 1)) & ((unsigned)y0 < (unsigned)(H - 1)) & ((unsigned)z0 < (unsigned)(D - 1));
                if (__all(interior)) {
                    const float *p = mov + ((size_t)z0 * H + y0) * W + x0, *q = p + (size_t)H * W;
                    g.r00 = *reinterpret_cast<const f2u *>(p); g.r01 = *reinterpret_cast<const f2u *>(p + W);
                    g.r10 = *reinterpret_cast<const f2u *>(q); g.r11 = *reinterpret_cast<const f2u *>(q + W);
                } else {
                    const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
                    const bool bx0 = (unsigned)x0 < (unsigned)W, bx1 = (unsigned)x1 < (unsigned)W;
                    const bool by0 = (unsigned)y0 < (unsigned)H, by1 = (unsigned)y1 < (unsigned)H;
                    const bool bz0 = (unsigned)z0 < (unsigned)D, bz1 = (unsigned)z1 < (unsigned)D;
                    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
                    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
                    const int cz0 = min(max(z0, 0), D - 1), cz1 = min(max(z1, 0), D - 1);
                    const float *a00 = mov + ((size_t)cz0 * H + cy0) * W, *a01 = mov + ((size_t)cz0 * H + cy1) * W;
                    const float *a10 = mov + ((size_t)cz1 * H + cy0) * W, *a11 = mov + ((size_t)cz1 * H + cy1) * W;
                    g.r00 = f2{(bz0 & by0 & bx0) ? a00[cx0] : 0.f, (bz0 & by0 & bx1) ? a00[cx1] : 0.f};
                    g.r01 = f2{(bz0 & by1 & bx0) ? a01[cx0] : 0.f, (bz0 & by1 & bx1) ? a01[cx1] : 0.f};
                    g.r10 = f2{(bz1 & by0 & bx0) ? a10[cx0] : 0.f, (bz1 & by0 & bx1) ? a10[cx1] : 0.f};
                    g.r11 = f2{(bz1 & by1 & bx0) ? a11[cx0] : 0.f, (bz1 & by1 & bx1) ? a11[cx1] : 0.f};
                }
                g.yv = (MODE == 3) ? 0.f : trow[(unsigned)(toff + (j - j0) * W)];
                return g;
            };
            const int jend = min(j0 + kRows, ny);
            if (j0 < jend) {
                GFetch cur = gfetch(j0);
#pragma unroll 1
                for (int j = j0; j < jend; j++) {
                    GFetch nxt = cur;
                    if (j + 1 < jend) nxt = gfetch(j + 1);
                    const Samp3 sm = lerp3_pairs<kGrad>(cur.r00, cur.r01, cur.r10, cur.r11, cur.fx, cur.fy, cur.fz);
                    if constexpr (MODE == 3) wout[(size_t)Y0 * W + (unsigned)(toff + (j - j0) * W)] = sm.v;
                    else f1_accumulate_pk<MODE>(sm, cur.yv, cur.yn, acc);
                    cur = nxt;
                }
            }
        }
    }

    if constexpr (MODE == 3) return;
    float vals[NP];
    int o = 0;
    if constexpr (MODE == 4) {
        vals[0] = acc.M4;
        o = 1;
    } else if constexpr (MODE != 2) {
        vals[0] = acc.M01.x; vals[1] = acc.M01.y; vals[2] = acc.M23.x; vals[3] = acc.M23.y; vals[4] = acc.M4;
        o = 5;
    }
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float a = acc.AB[q][c].x;
            vals[o++] = xn * a; vals[o++] = acc.AB[q][c].y; vals[o++] = zn * a; vals[o++] = a;
        }
    block_reduce_store_nw<NP, kTileWaves>(vals, partials + ((size_t)by * rows_stride + bx) * NP, box, wave);
}

// The primary kernel: one geometry (GeomP) for every block.
template <int MODE>
__global__ __launch_bounds__(GeomP::Threads, TRX_TILE_MIN_WAVES) void affine_tile_kernel(trx_volumes vol, const float *__restrict__ theta,
                                                                                          TileGeom tg, int channels, float *__restrict__ partials)
{
    __shared__ __attribute__((aligned(16))) float box[GeomP::BoxAlloc + TRX_DEV_LDS_PAD];
    tile_body<MODE, GeomP>(vol, theta, tg, channels, partials, box, blockIdx.x, blockIdx.y, gridDim.x, trx_wave_index());
}

#pragma clang diagnostic pop
#include "affine_zstream.h"   // z-streaming F1 body for transforms near the identity: the fifth per-pair choice of the step kernels
#include "affine_eft.h"       // exact-footprint F1 body for rotated transforms: the sixth per-pair choice of the step kernels
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"

// Does geometry G's box hold the pre-image of one of its tiles for this theta?  (theta-only: the tile-independent maximum extent, the same
// bound the fast loop fetches.)  NaN / huge theta compare false: GeomR, whose own per-tile test then sends everything to the fallback.
template <class G>
__device__ __forceinline__ bool dual_fits(const float *__restrict__ th, float fD, float fH, float fW)
{
    const float slope[3][3] = {{th[0], th[1] * fW / fH, th[2] * fW / fD}, {th[4] * fH / fW, th[5], th[6] * fH / fD}, {th[8] * fD / fW, th[9] * fD / fH, th[10]}};
    const float ex[3] = {(float)(G::TX - 1), (float)(G::TY - 1), (float)(G::TZ - 1)};
    float span[3];
#pragma unroll
    for (int c = 0; c < 3; c++) span[c] = fabsf(slope[c][0]) * ex[0] + fabsf(slope[c][1]) * ex[1] + fabsf(slope[c][2]) * ex[2];
    return (span[0] < 1.0e6f) && (span[1] < 1.0e6f) && (span[2] < 1.0e6f) &&
           ((((int)floorf(span[0] + 0.1f) + 5) >> 2) + 1 <= G::BW4) && ((int)floorf(span[1] + 0.1f) + 3 <= G::BH) &&
           ((int)floorf(span[2] + 0.1f) + 3 <= G::BD);
}
// The geometry a pair's blocks run (0 = GeomD, the deep tile, step kernels only; 1 = GeomA; 2 = GeomR; 3 = GeomRD, step kernels only): evaluated identically by every
// block of the pair and by the step's finalise kernel.
__device__ __forceinline__ int dual_choice(const float *__restrict__ th, float fD, float fH, float fW, bool with_deep, bool with_rd = false, int zs_planes = 0)
{
    if (zs_planes > 0 && zs_nsub<ZS64>(th, fD, fH, fW, zs_planes) > 0) return 4;   // 4 = the z-streaming body (step kernels only, transforms next to the identity)
    if (with_deep && dual_fits<GeomD>(th, fD, fH, fW)) return 0;
    if (dual_fits<GeomA>(th, fD, fH, fW)) return 1;
    return (with_rd && dual_fits<GeomRD>(th, fD, fH, fW)) ? 3 : 2;   // 3 = GeomRD: GeomR's box under a 16 x 16 x 16 tile, where the pre-image still fits it
}

// The dual kernel: per pair, GeomA where its box holds the pre-image of a GeomA tile for this theta (decided from the
// tile-independent maximum extent, the same bound the fast loop fetches), GeomR otherwise.  The grid is sized for the geometry
// with more blocks; the surplus blocks of the other one write a zero partial row and leave.
// WHICH = 0: every body in one kernel (3: only GeomA and GeomR, whatever the mode).  WHICH = 1 / 2: only the GeomA / GeomR body - the pair of launches (1 then 2) does the
// same job with each body compiled on its own (measured alternative: bench.py 0.328 ms per step against 0.323 ms for the two-body
// kernel and 0.311-0.320 ms for the single-geometry kernel): blocks of a pair that the other geometry owns leave at once.
#ifndef TRX_EF_RULE
#define TRX_EF_RULE 1   // 0: round 4's first rule (GeomRD keeps every pair whose rotation is mostly about z)
#endif
__device__ __forceinline__ bool eft_wants(int choice, const float *__restrict__ th, float fD, float fH, float fW)
{
    if (choice == 2) return true;
    if (choice != 3) return false;
    const float tilt = fabsf(th[8] * fD / fW) + fabsf(th[9] * fD / fH);                 // how far the tile's pre-image leans out of its z planes
    const float zspan = (tilt + fabsf(th[10])) * (float)(ECfg::TZ - 1);
#ifdef TRX_EF_ZSPAN
    return zspan > TRX_EF_ZSPAN;   // development
#endif
#if TRX_EF_RULE == 0
    return zspan > 17.0f;
#else
    if (zspan > 17.0f || tilt > 0.07f) return true;
    const float s = fmaxf(fabsf(th[1] * fW / fH), fabsf(th[4] * fH / fW));              // in-plane rotation (its sine, in voxels)
    return s > 0.15f && s < 0.37f;   // GeomRD's rows are at their longest just past GeomD's window (NaN compares false: GeomRD)
#endif
}

// The kernel arguments of affine_tile_dual_kernel as one struct: the kernel reads its arguments THROUGH THE KERNARG SEGMENT at the point
// of use (a few scalar loads in front of the body that needs them) instead of through its parameters, which the compiler loads at entry
// and keeps in SGPRs across the dispatch to five inlined bodies: that cost ~60 SGPRs, pushed the bodies' own scalars into VGPR lanes
// and the work-item id into scratch, whose reload (a memory round trip in front of everything) made every launch 5-6 us longer -
// 1 x 64^3: 15.5 -> 10 us per launch (tools/kbench.hip).  Field order and types = the parameter list (same layout rules).
// CARRY (round 6): the finalise of iteration t - reduction of the partial rows, loss, dL/dtheta, optimiser, theta of the next forward - folded into the
// prologue of iteration t + 1's step kernel, so that an iteration of a launch-bound registration (one pair up to ~128^3: a 12 us kernel and a 4 us
// finalise launch behind it, profiles/r05e_configs.txt) is ONE launch.  No rendezvous between blocks: EVERY block of a pair reduces that pair's rows of
// the previous launch (other parity of the two partial buffers) in the same fixed order and computes the same theta; the pair's first block alone
// writes the state (into the carry buffer of this parity: blocks of this launch that start later still read the other one), the loss curve and the
// best-theta record.  trx_affine_run enqueues iters such launches and one finalise kernel behind the last (the flush).
struct CarryKArgs {
    const float *prev_partials;   // partial rows of the previous iteration (nullptr: nothing pending - the first launch of a run)
    const int *prev_rows_used;    // ... and its per-pair notes
    int prev_nblk;                // row stride of prev_partials
    int mse_rows;
    const float *state_prev;      // carry buffer the state is read from (nullptr: the caller's arrays - the first launch of a run)
    float *state_next;            // carry buffer the pair's first block writes
    double nvox;
    trx_loss_cfg lc;
    trx_opt_cfg oc;
    trx_affine_state st;
};
typedef const __attribute__((address_space(4))) CarryKArgs *CarryKPtr;
template <int MODE>
__device__ __forceinline__ void carry_prologue(CarryKPtr c, int b, bool designated, int D, int H, int W, float *scratch, float *s_th, int wave, int lane);

struct DualKArgs {
    trx_volumes vol;
    const float *theta;
    TileGeom tgA, tgR;
    int channels;
    float *partials;
    int zero_surplus;
    TileGeom tgD, tgRD;
    ZGeom zg;
    int *rows_used;
    int rows_stride;
    int eft;
    CarryKArgs carry;
};

template <int MODE, int WHICH = 0>
__global__ __launch_bounds__(512, TRX_DUAL_MIN_WAVES) void affine_tile_dual_kernel(trx_volumes vol_, const float *__restrict__ theta_, TileGeom tgA_,
                                                                                   TileGeom tgR_, int channels_, float *__restrict__ partials_,
                                                                                   int zero_surplus_ = 1, TileGeom tgD_ = TileGeom{}, TileGeom tgRD_ = TileGeom{},
                                                                                   ZGeom zg_ = ZGeom{}, int *__restrict__ rows_used_ = nullptr, int rows_stride_ = 0,
                                                                                   int eft_ = 0, CarryKArgs carry_ = CarryKArgs{})
{
    static_assert(GeomA::Threads == 512 && GeomR::Threads == 512 && GeomD::Threads == 512 && GeomRD::Threads == 512, "every geometry runs 512-thread blocks");
    static_assert(GeomRD::BoxAlloc <= GeomR::BoxAlloc, "GeomRD lives in GeomR's box");
    // the step kernels (MODE 0 / 4) of the one-launch form also carry the deep tile for transforms next to the identity
    // WHICH = 4: the same without the z-streaming body - behind affine_zs_step_kernel, which has taken the pairs next to the identity
    // WHICH = 5: WHICH = 4 plus the exact-footprint body - ONE kernel behind the z-streaming kernel instead of two (a launch boundary and an
    // empty 512-block dispatch less per step: 4-5 us; inside the five-body kernel of round 4 the same merge cost the z-streaming loop 7-12 %,
    // which no longer lives here)
    // WHICH = 6: WHICH = 3 (GeomA / GeomR, classic grid) in the CARRY form - the block's prologue finalises the previous iteration of its pair and theta comes from LDS
    constexpr bool kCarry = WHICH == 6;
    static_assert(!kCarry || MODE == 0 || MODE == 4, "the carry form exists for the step kernels");
    constexpr bool kDeep = (WHICH == 0 || WHICH == 4 || WHICH == 5) && (MODE == 0 || MODE == 4) && (TRX_TILE_CFG == 0) && (TRX_DEEP_TILE != 0);
    constexpr bool kZs = kDeep && WHICH == 0;
    constexpr bool kEft = kDeep && WHICH == 5;
    constexpr int kAllocAR = GeomA::BoxAlloc > GeomR::BoxAlloc ? GeomA::BoxAlloc : GeomR::BoxAlloc;
    constexpr int kAllocD = WHICH == 1 ? GeomA::BoxAlloc : (WHICH == 2 ? GeomR::BoxAlloc : ((kDeep && GeomD::BoxAlloc > kAllocAR) ? GeomD::BoxAlloc : kAllocAR));
    constexpr int kAllocZ = (kZs && ZS64::Alloc > kAllocD) ? ZS64::Alloc : kAllocD;
    constexpr int kAlloc = (kEft && ECfg::Alloc > kAllocZ) ? ECfg::Alloc : kAllocZ;
    __shared__ __attribute__((aligned(16))) float box[kAlloc];
    constexpr bool kPerChannel = (MODE == 2) || (MODE == 3);
    constexpr int NP = (MODE == 0) ? np_full(3) : (MODE == 2 ? 12 : (MODE == 4 ? kNpMse : 5));
    // a fresh view of the arguments: the empty asm hides the pointer's identity, so loads through it stay where they are written
    typedef const __attribute__((address_space(4))) DualKArgs *KArgs;
    auto args = [&]() {
        KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return p;
    };
    typedef const __attribute__((address_space(4))) TileGeom *KTile;
    auto tile_of = [](KTile t) { return TileGeom{t->ntx, t->nty, t->ntz, t->ntiles, t->blocks_per_pair, t->ysplit, t->tiles_per_seg}; };   // (field by field: scalar loads)
    float fD, fH, fW;
    bool with_d, with_rd;
    int zs_planes, rows_stride, nA, nR, nD, nRD, nZ;
    bool with_ef;
    const float *theta;
    {
        KArgs a = args();
        fD = (float)a->vol.D; fH = (float)a->vol.H; fW = (float)a->vol.W;
        nA = a->tgA.blocks_per_pair; nR = a->tgR.blocks_per_pair; nD = a->tgD.blocks_per_pair; nRD = a->tgRD.blocks_per_pair; nZ = a->zg.blocks_per_pair;
        with_d = kDeep && nD > 0; with_rd = kDeep && nRD > 0;
        zs_planes = kZs && nZ > 0 ? a->zg.planes_per_seg : 0;   // (WHICH = 4: the z-streaming test is the kernel's in front - never repeated here, where it could round differently)
        rows_stride = a->rows_stride;
        theta = a->theta;
        with_ef = kDeep && a->rows_used != nullptr && (((TRX_EFT_BODY != 0) && a->eft != 0) || WHICH == 4 || WHICH == 5);   // pairs with rows_used < 0 belong to a kernel that ran in front
        if constexpr (WHICH == 4 || WHICH == 5) {   // the kernels in front left the number of pairs they did NOT take: none - the usual case next to the identity - and this launch is over
            const int *plan = a->rows_used + a->vol.B;
            if (__builtin_amdgcn_readfirstlane(*plan) == 0) return;
        }
    }
    // 6 = taken by the exact-footprint kernel (affine_eft_step_kernel, launched IN FRONT of this one: it left rows_used[b] = -(its row
    // count) for the pairs it took - rotated pairs that GeomR would run and whose plan fits its buffers): no block here
    // 7 = (WHICH = 5) the exact-footprint body inside this kernel: its 16^3 tiling is GeomRD's
    auto blocks_of = [&](int choice) { return choice == 6 ? 0 : (choice == 7 ? nRD : (choice == 4 ? nZ : (choice == 0 ? nD : (choice == 3 ? nRD : (choice == 1 ? nA : nR))))); };
    // Work items of this block: (pair / slab `by`, block index v of that pair's geometry).  Classic grid: exactly one, from blockIdx.
    // FLAT grid (rows_stride > 0; the launcher's choice for big batches of the step kernels): gridDim.x persistent blocks share one list
    // of work items - pair 0's blocks, then pair 1's, ... each pair with the block count of the body ITS theta selects - and block p runs
    // items p, p + gridDim.x, ...  Why: a (blocks of the largest geometry) x (pairs) grid makes every pair of a smaller geometry dispatch
    // hundreds of blocks that exit at once, in front of the next pair's working blocks (3584 of them cost the 8 x 256^3 launch 33 us); a
    // per-pair loop over x removes them but runs all pairs side by side on every XCD, and the halo re-reads of the rotated geometries
    // then miss the L2 (GeomR: +28 %).  Pair-major items keep the dispatcher's order: all blocks on one pair, each XCD on its slab of
    // columns (block counts are multiples of 8, so item % 8 is the XCD of v as before).
    const bool flat = kDeep && rows_stride > 0;
    const int wave_idx = trx_wave_index();   // the only use of threadIdx in this kernel
    const int lane = trx_lane_id();
    const int tid_ = wave_idx * 64 + lane;
    // flat path: per pair (lane) the body its theta selects and the inclusive prefix sum of the block counts - kept in LDS, not in
    // registers, across the item loop (two more live VGPRs are two spilled ones in the bodies that sit at the register limit)
    __shared__ int s_choice[64], s_pre[64];
    __shared__ float s_th[16];   // (carry) theta of this block's pair
    if constexpr (kCarry) {
        carry_prologue<MODE>(&args()->carry, (int)blockIdx.y, blockIdx.x == 0, args()->vol.D, args()->vol.H, args()->vol.W, box, s_th, wave_idx, lane);
        theta = s_th - (size_t)blockIdx.y * TRX_PSTRIDE;   // (the choice below indexes by pair; the bodies take s_th itself)
    }
    int my_choice = 2, total = 1;
    if (flat) {
        KArgs a = args();
        const int B = a->vol.B;
        int *rows_used = a->rows_used;
        int cnt = 0, ch = 2;
        if (lane < B) ch = dual_choice(theta + (size_t)lane * TRX_PSTRIDE, fD, fH, fW, with_d, with_rd, zs_planes);
        if (with_ef && lane < B && rows_used[lane] < 0) ch = 6;   // taken by the exact-footprint kernel, which ran in front of this one
        if constexpr (kEft) {
            // the exact-footprint body's pairs (affine_eft_step_kernel's decision, here): those GeomR or (eft_wants) GeomRD would run and whose plan
            // fits its buffers - counted exactly, one candidate pair per wave and round
            const float *thl = theta + (size_t)(lane < B ? lane : 0) * TRX_PSTRIDE;
            const bool cand_l = lane < B && (ch == 2 || ch == 3) && eft_wants(ch, thl, fD, fH, fW) && ef_candidate(thl, fD, fH, fW);
            const unsigned long long cand = __builtin_amdgcn_ballot_w64(cand_l);
            if (cand) {   // (block-uniform)
                if (wave_idx == 0) s_pre[lane] = 0;
                __syncthreads();
                unsigned long long rest = cand;
                for (int k = 0; rest; k++) {
                    const int pb = __builtin_ctzll(rest);
                    rest &= rest - 1;
                    if ((k & 7) != wave_idx) continue;
                    const EfMap m = ef_map(theta + (size_t)pb * TRX_PSTRIDE, fD, fH, fW);
                    const EfDims d = ef_dims(m);
                    const int g = ef_plan_granules_wave(m, d, lane);
                    if (lane == 0) s_pre[pb] = (d.ok && g > 0 && g <= ECfg::GCap) ? 1 : 0;
                }
                __syncthreads();
                if (cand_l && s_pre[lane] != 0) ch = 7;
                __syncthreads();   // (s_pre is rewritten below)
            }
        }
        if (lane < B) {
            cnt = blocks_of(ch);
            if (rows_used && blockIdx.x == 0 && wave_idx == 0 && ch != 6) rows_used[lane] = ch == 7 ? -rows_note(cnt, 7) : rows_note(cnt, 1 + ch);   // for the step's finalise kernel
        }
        int pre = cnt;   // inclusive prefix sum over the lanes (pairs)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(pre, d, 64);
            if (lane >= d) pre += t;
        }
        total = __builtin_amdgcn_readlane(pre, 63);
        if (wave_idx == 0) { s_choice[lane] = ch; s_pre[lane] = pre; }
        __syncthreads();
    } else {
        KArgs a = args();
        const int channels = a->channels;
        const int b = kPerChannel ? blockIdx.y / channels : blockIdx.y;
        my_choice = __builtin_amdgcn_readfirstlane(dual_choice(theta + (size_t)b * TRX_PSTRIDE, fD, fH, fW, with_d, with_rd, zs_planes));
        if (with_ef && a->rows_used[b] < 0) my_choice = 6;   // taken by the exact-footprint kernel, which ran in front of this one
        const bool useA = my_choice == 1;
        if ((WHICH == 1 && !useA) || (WHICH == 2 && useA)) return;   // the other launch owns this pair (and its surplus rows)
        const int mine = blocks_of(my_choice);
        // zero_surplus = 0: surplus blocks write nothing; the reader (the step's finalise kernel) learns the pair's row count from
        // rows_used[], written here by the pair's first block - it does not repeat the choice (two inlined copies of a float test could
        // disagree by an ulp)
        int *rows_used = a->rows_used;
        if (rows_used && blockIdx.x == 0 && tid_ == 0 && my_choice != 6) rows_used[blockIdx.y] = rows_note(mine, 1 + my_choice);
        if ((int)blockIdx.x >= mine) {
            if (MODE != 3 && a->zero_surplus && tid_ < NP) a->partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NP + tid_] = 0.f;
            return;
        }
    }
    total = __builtin_amdgcn_readfirstlane(total);
    const int it0 = flat ? (int)blockIdx.x : 0, it_step = flat ? (int)gridDim.x : 1;
    for (int item = __builtin_amdgcn_readfirstlane(it0); item < total; item += it_step) {
        int choice, v, by, stride;
        if (flat) {
            const int pre = *(volatile int *)&s_pre[lane];
            // (round 5) odd iterations of a run (TRX_FLAG_WALK_DOWN) take the list backwards, in groups of 8 (item % 8 stays the XCD): they start on the pairs the Infinity Cache still holds
            const int it = (TRX_DUAL_PINGPONG && WHICH == 4 && (args()->vol.flags & TRX_FLAG_WALK_DOWN) && (total & 7) == 0) ? total - 8 - (item & ~7) + (item & 7) : item;
            const int pair = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__builtin_amdgcn_ballot_w64(pre <= it)));   // lanes >= B hold the total: never counted
            const int off = pair > 0 ? __builtin_amdgcn_readlane(pre, pair - 1) : 0;
            choice = *(volatile int *)&s_choice[pair];
            v = __builtin_amdgcn_readfirstlane(it - off); by = pair; stride = rows_stride;
            if (item != it0) __syncthreads();   // the previous body's reduction scratch aliases the box
        } else {
            choice = __builtin_amdgcn_readfirstlane(my_choice); v = blockIdx.x; by = blockIdx.y; stride = gridDim.x;   // (my_choice is per-lane on the flat path: keep this one provably uniform)
        }
        choice = __builtin_amdgcn_readfirstlane(choice); v = __builtin_amdgcn_readfirstlane(v);   // all four ARE wave-uniform; say so to the
        by = __builtin_amdgcn_readfirstlane(by); stride = __builtin_amdgcn_readfirstlane(stride);  // compiler (the bodies pin them to SGPRs)
        KArgs a = args();   // this item's view of the arguments: the volumes and ONE geometry
        const trx_volumes vol = {a->vol.moving, a->vol.target, a->vol.moving_stride, a->vol.target_stride, a->vol.ndim, a->vol.B, a->vol.D, a->vol.H, a->vol.W,
                                 a->vol.xn, a->vol.yn, a->vol.zn, a->vol.flags};
        float *partials = a->partials;
        const int channels = a->channels;
        if constexpr (kZs) {
            if (choice == 4) {
                const ZGeom zg = {a->zg.ntx, a->zg.nty, a->zg.nzseg, a->zg.planes_per_seg, a->zg.blocks_per_pair};
                zstream_body<MODE, ZS64>(vol, theta, zg, partials, box, v, by, stride, wave_idx,
                                         (int)((blockIdx.y * gridDim.x + blockIdx.x) * 2 >= gridDim.x * gridDim.y));   // (second half of the grid = the later block of its CU)
                continue;
            }
        }
        if constexpr (kEft) {
            if (choice == 7) {   // (columns differ widely in cost: the column index is rotated per pair, by a multiple of 8 - see affine_eft_step_kernel)
                const TileGeom t = tile_of(&a->tgRD);
                const int ve = __builtin_amdgcn_readfirstlane((v + 104 * by) % t.blocks_per_pair);
                EfPlanRegs pr;
                eft_body<MODE>(vol, theta, t, partials, box, ve, by, stride, wave_idx, pr, true);
                continue;
            }
        }
        if constexpr (kDeep) {
            if (choice == 0) {
                const TileGeom t = tile_of(&a->tgD);
                tile_body<MODE, GeomD>(vol, theta, t, channels, partials, box, v, by, stride, wave_idx);
                continue;
            }
            if (choice == 3) {
                const TileGeom t = tile_of(&a->tgRD);
                tile_body<MODE, GeomRD>(vol, theta, t, channels, partials, box, v, by, stride, wave_idx);
                continue;
            }
        }
        if constexpr (WHICH != 1) {
            if (choice != 1) {
                const TileGeom t = tile_of(&a->tgR);
                if constexpr (kCarry) tile_body<MODE, GeomR, true>(vol, s_th, t, channels, partials, box, v, by, stride, wave_idx);
                else tile_body<MODE, GeomR>(vol, theta, t, channels, partials, box, v, by, stride, wave_idx);
                continue;
            }
        }
        if constexpr (WHICH != 2) {
            const TileGeom t = tile_of(&a->tgA);
            if constexpr (kCarry) tile_body<MODE, GeomA, true>(vol, s_th, t, channels, partials, box, v, by, stride, wave_idx);
            else tile_body<MODE, GeomA>(vol, theta, t, channels, partials, box, v, by, stride, wave_idx);
        }
    }
}

// The exact-footprint kernel of a step, launched IN FRONT of affine_tile_dual_kernel.  It decides which pairs are its own - those that
// kernel would give to GeomR or (eft_wants) GeomRD (dual_choice, same arguments) and whose plan fits its buffers (counted exactly, one pair per wave) - leaves
// rows_used[b] = -(partial rows it writes) for them and 0 for the others (the step kernel behind it skips the negative ones; so does the
// finalise kernel's reading of the count: |rows_used|), and runs them.  Why a kernel of its own: inlined into affine_tile_dual_kernel as a
// sixth body - or called from it, or with only this decision in its prologue - it changed the code the compiler makes of the z-streaming
// loop (a reload and a vmcnt(0) per step): the headline lost 7-12 % (profiles/r04c_eft_placement_ab.txt).
// stride > 0: FLAT grid of persistent blocks over the pair-major list of those pairs' blocks (partial rows of stride `stride` per pair);
// stride < 0: block (x, pair) of a (blocks_per_pair, pairs) grid, rows of stride -stride.  No pair of its own: every block leaves at once.
// Which of the fused kernel's choices this kernel takes over: GeomR's pairs (2) always; GeomRD's (3) when the tile's pre-image leans out of its z
// planes (zspan > 17 source planes per 16^3 tile, or a tilt of more than 0.07 voxels per voxel: GeomRD's rows get long), or when the in-plane rotation
// lies between the end of GeomD's window and a sine of 0.37 - pure rotations about z beyond that stay with GeomRD.  Measured, 8 x 256^3, us per
// pair-iteration, GeomRD -> this kernel: R_x(0.6) 63.7 -> 47.6, R(.3,.3,.3) 63.8 -> 53.3, R_z(0.25) 54.3 -> 48.6, R(.1,0,.3) 55.5 -> 48.1, but
// R_z(0.6) 50.3 -> 54.5, R_z(1.0) 50.7 -> 56.2 (profiles/r04g_rotation_sweep.txt, profiles/r04h_eft_offer_rule.txt).
// The z-streaming body as a kernel of its OWN for launches that fill the chip (flat grid), IN FRONT of the exact-footprint kernel and the
// tile kernel (round 5).  Inside affine_tile_dual_kernel its loop shared the register allocation of four other bodies: the same source
// measured 253-262 us stand-alone and 261-318 us fused, by which way the allocator fell (profiles/r05a_zstream_fused_vs_alone.txt).  This
// kernel takes every pair whose theta passes zs_nsub - the ONLY place that test is evaluated for such a launch -, leaves
// rows_used[b] = -(its rows) for them and 0 for the rest, the number of pairs it did NOT take in rows_used[B] and the bit mask of those it
// took in rows_used[B + 1, B + 2]: the two kernels behind it return at once when the count is zero (a launch boundary each, ~1.5 us), and
// otherwise skip the pairs of the mask.
#ifndef TRX_ZS_FLAT
#define TRX_ZS_FLAT 1   // the z-streaming kernel also carries the FLAT tile (ZSF: 64 x 16 voxels per plane, ring of 8 planes of 80 x 30) for pairs the 64 x 32 tile
                        // does not take: poses of the convergence basin (rotations up to ~0.15 rad about z, zooms to 1.15, 3.7 planes of tilt) run 10-11 %
                        // faster on it than on the deep tile kernel (profiles/r05a_zs_flat_tile.txt); 0: never
#endif
// FB (round 6): the ONE-KERNEL form of a step (TRX_FLAG_ONE_KERNEL) - nothing is launched behind this kernel, so the pairs its two streaming tiles do
// not take run HERE, on GeomR's body (the tile whose box holds the pre-image under any rotation; 78.6 KB, inside the ring's allocation): a pair that
// leaves the window costs ~2 x its usual time for as long as it stays outside, instead of two launches that find nothing to do on every step of every
// run: 3.2-3.5 us per step, measured (profiles/r06a_one_kernel_ab.txt; their traced durations, 5.3 + 4.8 us, overstate it).  FB = 0 is the kernel of
// round 5: the FB code is compiled out of it (same instructions: tools/isa_loops.py).
struct ZsOneKArgs {   // the kernel arguments of affine_zs_one_kernel as one struct (field order and types = its parameter list)
    trx_volumes vol;
    const float *theta;
    ZGeom zg, zgf;
    float *partials;
    int *rows_used;
    int stride;
    TileGeom tgR;
};
template <int MODE, int FB>
__device__ __forceinline__ void zs_step_main(const trx_volumes &vol, const float *__restrict__ theta, const ZGeom &zg, const ZGeom &zgf, float *__restrict__ partials,
                                             int *__restrict__ rows_used, int stride, const int fb_blocks, float *ring)
{
    const int wave = trx_wave_index(), lane = trx_lane_id();
    const float fD = (float)vol.D, fH = (float)vol.H, fW = (float)vol.W;
    // per pair (lane): 1 = the 64 x 32 tile takes it, 2 = the flat tile does, 0 = left to the kernels behind (FB: 3 = GeomR's body, here)
    int which = 0;
    if (lane < vol.B) {
        if (zs_nsub<ZS64>(theta + (size_t)lane * TRX_PSTRIDE, fD, fH, fW, zg.planes_per_seg) > 0) which = 1;
        else if (TRX_ZS_FLAT && zgf.blocks_per_pair > 0 && zs_nsub<ZSF>(theta + (size_t)lane * TRX_PSTRIDE, fD, fH, fW, zgf.planes_per_seg) > 0) which = 2;
        else if (FB) which = 3;
    }
    const unsigned long long take = __builtin_amdgcn_ballot_w64(which != 0), take1 = __builtin_amdgcn_ballot_w64(which == 1);
    const int mine = which == 1 ? zg.blocks_per_pair : (which == 2 ? zgf.blocks_per_pair : ((FB && which == 3) ? fb_blocks : 0));
    if (blockIdx.x == 0 && wave == 0) {
        if (lane < vol.B) rows_used[lane] = -rows_note(mine, which == 3 ? 3 : (which == 2 ? 8 : 6));
        if (lane == 0) {
            rows_used[vol.B] = vol.B - __builtin_popcountll(take);
            rows_used[vol.B + 1] = (int)(unsigned)take; rows_used[vol.B + 2] = (int)(unsigned)(take >> 32);   // which pairs: nobody rewrites this
        }
        if (lane < 8) rows_used[vol.B + 3 + lane] = 0;   // work tickets of the exact-footprint kernel behind this one (one queue per XCD)
    }
    if (take == 0) return;
    int pre = mine;   // inclusive prefix sum over the lanes (pairs)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(pre, d, 64);
        if (lane >= d) pre += t;
    }
    const int total = __builtin_amdgcn_readlane(pre, 63);
    const int second = (int)(blockIdx.x * 2 >= gridDim.x);   // the later of the two blocks of a CU (TRX_ZS_PRIO)
    if constexpr (FB != 0) {
        // the pair's body rides in the low two bits of its prefix sum: the masks of the FB = 0 form (two scalar-register pairs, live across the streaming
        // loops, which have none to spare) would be three here
        const int prew = (pre << 2) | which;
        for (int item = blockIdx.x; item < total; item += gridDim.x) {
            const int pair = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__builtin_amdgcn_ballot_w64((prew >> 2) <= item)));
            const int off = pair > 0 ? (__builtin_amdgcn_readlane(prew, pair - 1) >> 2) : 0;
            const int w = __builtin_amdgcn_readlane(prew, pair) & 3;
            const int v = __builtin_amdgcn_readfirstlane(item - off);
            if (item != (int)blockIdx.x) __syncthreads();   // the previous item's reduction scratch aliases the ring
            if (__builtin_expect(w == 3, 0)) {
                // (the tile geometry is read through the kernarg segment HERE, as affine_tile_dual_kernel reads its arguments: loaded at the kernel's entry
                // its seven scalars stay live across the streaming loops.  Inlined: as a function of its own - __noinline__, arguments through the kernarg
                // segment - the call's register conventions cost the streaming loops 25 more scalar reloads per two planes than the inlined body's 2)
                typedef const __attribute__((address_space(4))) ZsOneKArgs *KArgs;
                KArgs a = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(a));
                const TileGeom t = {a->tgR.ntx, a->tgR.nty, a->tgR.ntz, a->tgR.ntiles, a->tgR.blocks_per_pair, a->tgR.ysplit, a->tgR.tiles_per_seg};
                tile_body<MODE, GeomR>(vol, theta, t, 1, partials, ring, v, pair, stride, wave);
            } else if (!TRX_ZS_FLAT || w == 1) zstream_body<MODE, ZS64>(vol, theta, zg, partials, ring, v, pair, stride, wave, second);
            else zstream_body<MODE, ZSF>(vol, theta, zgf, partials, ring, v, pair, stride, wave, second);
        }
        return;
    }
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int pair = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__builtin_amdgcn_ballot_w64(pre <= item)));
        const int off = pair > 0 ? __builtin_amdgcn_readlane(pre, pair - 1) : 0;
        const int v = __builtin_amdgcn_readfirstlane(item - off);
        if (item != (int)blockIdx.x) __syncthreads();   // the previous item's reduction scratch aliases the ring
        if (!TRX_ZS_FLAT || ((take1 >> pair) & 1ull)) zstream_body<MODE, ZS64>(vol, theta, zg, partials, ring, v, pair, stride, wave, second);
        else zstream_body<MODE, ZSF>(vol, theta, zgf, partials, ring, v, pair, stride, wave, second);
    }
}

template <int MODE>
__global__ __launch_bounds__(ZS64::Threads, TRX_ZS_MIN_WAVES) void affine_zs_step_kernel(trx_volumes vol, const float *__restrict__ theta, ZGeom zg, ZGeom zgf, float *__restrict__ partials,
                                                                                         int *__restrict__ rows_used, int stride)
{
    constexpr int kAlloc = (TRX_ZS_FLAT && ZSF::Alloc > ZS64::Alloc) ? ZSF::Alloc : ZS64::Alloc;
    __shared__ __attribute__((aligned(16))) float ring[kAlloc];
    zs_step_main<MODE, 0>(vol, theta, zg, zgf, partials, rows_used, stride, 0, ring);
}

template <int MODE>
__global__ __launch_bounds__(ZS64::Threads, TRX_ZS_MIN_WAVES) void affine_zs_one_kernel(trx_volumes vol, const float *__restrict__ theta, ZGeom zg, ZGeom zgf, float *__restrict__ partials,
                                                                                        int *__restrict__ rows_used, int stride, TileGeom tgR)
{
    constexpr int kAllocZ = (TRX_ZS_FLAT && ZSF::Alloc > ZS64::Alloc) ? ZSF::Alloc : ZS64::Alloc;
    constexpr int kAlloc = GeomR::BoxAlloc > kAllocZ ? GeomR::BoxAlloc : kAllocZ;
    __shared__ __attribute__((aligned(16))) float ring[kAlloc];
    zs_step_main<MODE, 1>(vol, theta, zg, zgf, partials, rows_used, stride, tgR.blocks_per_pair, ring);
}

template <int MODE>
__global__ __launch_bounds__(ECfg::Threads, 4) void affine_eft_step_kernel(trx_volumes vol, const float *__restrict__ theta, TileGeom tg, float *__restrict__ partials,
                                                                           int *__restrict__ rows_used, int stride, int with_d, int with_rd, int zs_planes_)
{
    const bool zs_first = zs_planes_ < 0;   // (flat launches behind affine_zs_step_kernel: the z-streaming test is not repeated here)
    const int zs_planes = zs_first ? 0 : zs_planes_;
    __shared__ __attribute__((aligned(16))) float lds[ECfg::Alloc];
    __shared__ int s_ef[64];
    const int wave = trx_wave_index(), lane = trx_lane_id();
    const float fD = (float)vol.D, fH = (float)vol.H, fW = (float)vol.W;
    if (stride < 0) {   // (small launches: the body is offered by TRX_FLAG_EFT only)
        if ((int)blockIdx.x >= tg.blocks_per_pair) return;
        const int b = blockIdx.y;
        const float *th = theta + (size_t)b * TRX_PSTRIDE;
        bool take = __builtin_amdgcn_readfirstlane((int)eft_wants(dual_choice(th, fD, fH, fW, with_d != 0, with_rd != 0, zs_planes), th, fD, fH, fW)) &&
                    __builtin_amdgcn_readfirstlane((int)ef_candidate(th, fD, fH, fW));
        if (take) {   // (block-uniform) the plan's granules, counted by the whole block: two rows per thread
            const EfMap m = ef_map(th, fD, fH, fW);
            const EfDims d = ef_dims(m);
            int cnt = 0;
#pragma unroll
            for (int h = 0; h < (ECfg::NY * ECfg::NZ) / ECfg::Threads; h++) {
                const int r = wave * 64 + lane + h * ECfg::Threads, iy = r & (ECfg::NY - 1), iz = r >> 5;
                int wlo;
                if (iy < d.ny && iz < d.nz) cnt += ef_row_window(m, d.dy0 + iy, d.dz0 + iz, wlo);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if (lane == 0) s_ef[wave] = cnt;
            __syncthreads();
            int g = 0;
#pragma unroll
            for (int w = 0; w < ECfg::Waves; w++) g += s_ef[w];
            g = __builtin_amdgcn_readfirstlane(g);
            take = d.ok && g > 0 && g <= ECfg::GCap;
            __syncthreads();   // s_ef is not touched again, but the body's prologue reuses LDS right away: keep the phases apart
        }
        if (blockIdx.x == 0 && wave == 0 && lane == 0) rows_used[b] = take ? -rows_note(tg.blocks_per_pair, 7) : 0;
        if (!take) return;
    }
    // ONE call site of the body for both grids (as in affine_tile_dual_kernel): with a second inlined copy behind the classic grid's branch the
    // compiler produced a kernel whose classic-grid launches faulted as soon as round 5 touched the flat branch (tools/repro_eft2.py)
    const bool flat = stride >= 0;
#if TRX_EF_STAMP
    if (flat && wave == 0 && lane == 0) { for (int k = 1; k < 8; k++) trx_ef_blocks[blockIdx.x * 8 + k] = 0; trx_ef_blocks[blockIdx.x * 8] = __builtin_amdgcn_s_memrealtime(); }
#endif
    int pre = 0, total = 1;
    if (flat) {
    // flat: per pair (lane) the decision, one candidate pair per wave and round
    // zs_first: affine_zs_step_kernel ran in front - no pair left (rows_used[B] = 0): done; else its pairs are those of the mask in rows_used[B + 1, B + 2]
    if (zs_first && __builtin_amdgcn_readfirstlane(rows_used[vol.B]) == 0) return;
    unsigned long long zs_mask = 0;   // pairs of the kernel in front (its mask is not rewritten by anybody: block 0 of THIS kernel rewrites rows_used[b] of the free pairs)
    if (zs_first) zs_mask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rows_used[vol.B + 1]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rows_used[vol.B + 2]) << 32);
    const bool free_l = lane < vol.B && !((zs_mask >> lane) & 1ull);
    const bool cand_l = free_l && eft_wants(dual_choice(theta + (size_t)lane * TRX_PSTRIDE, fD, fH, fW, with_d != 0, with_rd != 0, zs_planes),
                                                  theta + (size_t)lane * TRX_PSTRIDE, fD, fH, fW) &&
                        ef_candidate(theta + (size_t)lane * TRX_PSTRIDE, fD, fH, fW);
    const unsigned long long cand = __builtin_amdgcn_ballot_w64(cand_l);
    unsigned long long fit = 0;
    if (cand) {
        if (wave == 0) s_ef[lane] = 0;
        __syncthreads();
        unsigned long long rest = cand;
        for (int k = 0; rest; k++) {
            const int pb = __builtin_ctzll(rest);
            rest &= rest - 1;
            if ((k & (ECfg::Waves - 1)) != wave) continue;
            const EfMap m = ef_map(theta + (size_t)pb * TRX_PSTRIDE, fD, fH, fW);
            const EfDims d = ef_dims(m);
            const int g = ef_plan_granules_wave(m, d, lane);
            if (lane == 0) s_ef[pb] = (d.ok && g > 0 && g <= ECfg::GCap) ? 1 : 0;
        }
        __syncthreads();
        fit = __builtin_amdgcn_ballot_w64(s_ef[lane] != 0) & cand;
    }
    const int mine = ((fit >> lane) & 1ull) ? tg.blocks_per_pair : 0;
    if (blockIdx.x == 0 && wave == 0 && free_l) rows_used[lane] = -rows_note(mine, 7);
    if (fit == 0) return;
    pre = mine;   // inclusive prefix sum over the lanes (pairs)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(pre, d, 64);
        if (lane >= d) pre += t;
    }
    total = __builtin_amdgcn_readlane(pre, 63);
    }
    total = __builtin_amdgcn_readfirstlane(total);
    const int rows_stride = __builtin_amdgcn_readfirstlane(flat ? stride : -stride);
    // A block's items (pair-major index blockIdx + k gridDim) would be the SAME column in pair after pair - and columns differ widely in
    // cost (those that leave the source volume early are cheap): the column index is rotated per pair (by a multiple of 8: blocks b, b + 8,
    // ... still share an XCD's L2 with the neighbouring columns), which evens the blocks' loads without any shared counter.
    // TRX_EF_CHUNK (round 5, measured alternative, off - see affine_eft.h): a block takes `ipb` CONSECUTIVE items of the pair-major list instead of every
    // gridDim-th one, i.e. (when the pairs' item counts are multiples of ipb) items of ONE pair, whose plan it then makes once (eft_body's `replan`).
    // Block jb of a pair (XCD jb & 7) takes the columns (jb & 7) + 8 (s nblk / 8 + jb / 8), s = 0 .. ipb - 1, of its XCD's slab.
    // TRX_EF_TICKETS (round 5): behind affine_zs_step_kernel (which zeroes them) the flat grid's blocks DRAW their items - items cost 43 k ... 345 k ticks, and
    // with every gridDim-th item a launch waited 460 us for blocks whose median finished at 349 (profiles/r05c_eft_item_timeline.txt).  One queue per XCD
    // (block b draws from queue b & 7: the columns v = b & 7 (mod 8) of every pair, pair-major - the slab of its XCD's L2, as before); a block's first item is the
    // one it had (no draw), the ticket of the next item is drawn when the current one is done (TRX_EF_TICKETS 2: at its START - no round trip to wait for, but every block then
    // holds an item that nobody else can take: measured worse).
    const bool tickets = TRX_EF_TICKETS && !TRX_EF_CHUNK && flat && zs_first && (tg.blocks_per_pair & 7) == 0 && (gridDim.x & 7) == 0 && (total % tg.blocks_per_pair) == 0;
    const int ipb = __builtin_amdgcn_readfirstlane(flat ? (TRX_EF_CHUNK ? (total + (int)gridDim.x - 1) / (int)gridDim.x : 1) : 1);
    const int it0 = flat ? (TRX_EF_CHUNK ? (int)blockIdx.x * ipb : (int)blockIdx.x) : 0, it_step = flat ? (TRX_EF_CHUNK ? 1 : (int)gridDim.x) : 1;
    const int it_end = flat ? (TRX_EF_CHUNK ? min(total, it0 + ipb) : total) : 1;
    EfPlanRegs pr;
    int planned = -1;   // the pair the registers and the row table in LDS belong to
    int q = blockIdx.x & 7;                                                                   // (tickets) the queue the current item came from: this block's XCD's, at the end any
    const int qlen = total >> 3, per_pair_q = tg.blocks_per_pair >> 3;                        // (tickets) a queue's length, a pair's items in it
    int qk = (int)blockIdx.x >> 3;                                                            // (tickets) position in the queue of the current item
    int drawn = 0;                                                                            // (tickets 2; lane 0 of wave 0) the ticket drawn for the next item
    for (int item = __builtin_amdgcn_readfirstlane(it0); tickets ? (qk < qlen) : (item < it_end); item += it_step) {
        int v = blockIdx.x, pair = blockIdx.y;   // classic grid: the one item of this block
        if (flat) {
            if (tickets) {
                // (order of a queue: pair-major.  The pairs' inner columns first and their columns on a face of the volume - the cheap items - last was measured
                // worse, 429 against 408 us: profiles/r05c_eft_item_timeline.txt)
                int ord = qk / per_pair_q;                                    // the ord-th pair of those that fit ...
                const int j = qk - ord * per_pair_q;                          // ... position in the XCD's slab
                if (TRX_EF_PINGPONG && (vol.flags & TRX_FLAG_WALK_DOWN)) ord = total / tg.blocks_per_pair - 1 - ord;   // odd iterations of a run take the pairs in the opposite order: they start on what the Infinity Cache still holds
                pair = __builtin_popcountll(__builtin_amdgcn_ballot_w64(pre <= ord * tg.blocks_per_pair));
                v = q + 8 * j;
                if (TRX_EF_TICKETS == 2 && wave == 0 && lane == 0) drawn = __hip_atomic_fetch_add(rows_used + vol.B + 3 + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
            pair = __builtin_popcountll(__builtin_amdgcn_ballot_w64(pre <= item));
            const int off = pair > 0 ? __builtin_amdgcn_readlane(pre, pair - 1) : 0;
            if (TRX_EF_CHUNK) {
                const int idx = item - off, nblk = tg.blocks_per_pair / ipb;
                if (nblk * ipb == tg.blocks_per_pair && (nblk & 7) == 0 && (off % ipb) == 0) {
                    const int jb = idx / ipb, sidx = idx - jb * ipb;
                    v = (jb & 7) + 8 * (sidx * (nblk >> 3) + (jb >> 3));
                } else v = idx;
            } else v = (item - off + 104 * pair) % tg.blocks_per_pair;
            }
        }
        v = __builtin_amdgcn_readfirstlane(v); pair = __builtin_amdgcn_readfirstlane(pair);
        if (item != it0) __syncthreads();   // the previous item's reduction scratch aliases the buffers
        eft_body<MODE>(vol, theta, tg, partials, lds, v, pair, rows_stride, wave, pr, pair != planned);
        planned = pair;
#if TRX_EF_STAMP
        if (flat && wave == 0 && lane == 0) { trx_ef_blocks[blockIdx.x * 8 + 1 + min((item - it0) / it_step, 5)] = __builtin_amdgcn_s_memrealtime(); }
#endif
        if (tickets) {   // the next item: the ticket drawn above, counted behind the (gridDim / 8) items the queue's blocks started with
            if (wave == 0 && lane == 0) {
                const int first = (int)gridDim.x >> 3;
                int nq = blockIdx.x & 7, k = drawn + first;
                if (TRX_EF_TICKETS != 2) {   // own queue first; when it is empty the others in turn (the XCDs' slabs differ in cost: edge slabs leave the source volume early)
                    k = qlen;
                    for (int t = 0; t < 8 && k >= qlen; t++) {
                        nq = ((int)blockIdx.x + t) & 7;
                        k = __hip_atomic_fetch_add(rows_used + vol.B + 3 + nq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + first;
                    }
                }
                s_ef[63] = k < qlen ? ((nq << 24) | k) : -1;
            }
            __syncthreads();
            const int nx = __builtin_amdgcn_readfirstlane(s_ef[63]);
            if (nx < 0) break;
            q = nx >> 24; qk = nx & 0xffffff;
        }
    }
}

// GeomA / GeomR per pair: one two-body launch or the pair of single-body launches (TRX_AFFINE_DUAL = 1 / 2, default 1: the pair costs one more launch and gains nothing).
template <int MODE>
static void launch_dual(dim3 grid, hipStream_t s, const trx_volumes &v, const float *theta, const TileGeom &ta, const TileGeom &tr, int channels, float *out, int how,
                        int zero_surplus = 1, TileGeom td = TileGeom{}, TileGeom trd = TileGeom{}, ZGeom zg = ZGeom{}, int *rows_used = nullptr, int rows_stride = 0,
                        int eft = 0, const CarryKArgs *carry = nullptr, bool *carry_used = nullptr)
{
    if (how == 1) {
        // WHICH = 3: the two-body (GeomA / GeomR) instance for launches that offer nothing else - the five-body kernel's entry costs 2-3 us
        // more (its arguments are fetched in front of the body that needs them), which is what a step of a small volume takes in all
        // (and no exact-footprint kernel in front: the two-body instance does not read its marks and would run its pairs a second time - ADVICE r4)
        if (td.blocks_per_pair == 0 && trd.blocks_per_pair == 0 && zg.blocks_per_pair == 0 && rows_stride == 0 && eft == 0) {
            if constexpr (MODE == 0 || MODE == 4) {
                if (carry != nullptr && rows_used != nullptr && zero_surplus == 0) {   // the carry form of the same launch (trx_affine_run): theta from the block's own prologue
                    hipLaunchKernelGGL((affine_tile_dual_kernel<MODE, 6>), grid, dim3(512), 0, s, v, theta, ta, tr, channels, out, zero_surplus, td, trd, zg, rows_used, rows_stride, 0, *carry);
                    *carry_used = true;
                    return;
                }
            }
            hipLaunchKernelGGL((affine_tile_dual_kernel<MODE, 3>), grid, dim3(512), 0, s, v, theta, ta, tr, channels, out, zero_surplus, td, trd, zg, rows_used, rows_stride);
        }
        else
            hipLaunchKernelGGL((affine_tile_dual_kernel<MODE, 0>), grid, dim3(512), 0, s, v, theta, ta, tr, channels, out, zero_surplus, td, trd, zg, rows_used, rows_stride, eft);
    } else {
        hipLaunchKernelGGL((affine_tile_dual_kernel<MODE, 1>), grid, dim3(512), 0, s, v, theta, ta, tr, channels, out, 1);
        hipLaunchKernelGGL((affine_tile_dual_kernel<MODE, 2>), grid, dim3(512), 0, s, v, theta, ta, tr, channels, out, 1);
    }
}

#pragma clang diagnostic pop


// closed-form base coordinates for callers that pass no tables: (2i+1)/S - 1
__global__ void fill_tables_kernel(float *__restrict__ tab, int W, int H, int D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W) tab[i] = (float)(2 * i + 1) / (float)W - 1.0f;
    if (i < H) tab[W + i] = (float)(2 * i + 1) / (float)H - 1.0f;
    if (i < D) tab[W + H + i] = (float)(2 * i + 1) / (float)D - 1.0f;
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
        double cps, sps;
        sincos((double)p[0], &sps, &cps);
        double cth, sth;
        sincos((double)p[1], &sth, &cth);
        double cph, sph;
        sincos((double)p[2], &sph, &cph);
        th[0] = cps * cth; th[1] = sph * sps * cth - cph * sth; th[2] = cph * sps * cth + sph * sth; th[3] = 0.25 * tanh((double)p[3]);
        th[4] = cps * sth; th[5] = sph * sps * sth + cph * cth; th[6] = cph * sps * sth - sph * cth; th[7] = 0.25 * tanh((double)p[4]);
        th[8] = -sps;      th[9] = sph * cps;                   th[10] = cph * cps;                  th[11] = 0.25 * tanh((double)p[5]);
    } else {
        double c, s;
        sincos((double)p[0], &s, &c);
        th[0] = c; th[1] = -s; th[2] = p[1];
        th[3] = s; th[4] = c;  th[5] = p[2];
    }
}

template <int ND>
__device__ void pose_vjp(const float *p, const double *g, double *dx)
{
    if constexpr (ND == 3) {
        double cps, sps;
        sincos((double)p[0], &sps, &cps);
        double cth, sth;
        sincos((double)p[1], &sth, &cth);
        double cph, sph;
        sincos((double)p[2], &sph, &cph);
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
        double c, s;
        sincos((double)p[0], &s, &c);
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
        // batches of 16 independent loads per thread (all in flight together: one memory round trip for up to 256 rows),
        // fixed summation order
        for (int blk0 = grp; blk0 < nblk; blk0 += 16 * NG) {
            float a[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int blk = blk0 + i * NG;
                const float v = part[(size_t)min(blk, nblk - 1) * NP + k];   // clamped, unconditional: a predicated load compiles to
                a[i] = (blk < nblk) ? v : 0.f;                              // branch + s_waitcnt per load (16 serial round trips)
            }
            s += ((((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3])) + (((double)a[4] + (double)a[5]) + ((double)a[6] + (double)a[7]))) +
                 ((((double)a[8] + (double)a[9]) + ((double)a[10] + (double)a[11])) + (((double)a[12] + (double)a[13]) + ((double)a[14] + (double)a[15])));
        }
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

// The per-pair epilogue of an iteration, shared by the finalise kernel and by the CARRY prologue of the step kernels (round 6: the finalise of
// iteration t folded into the first kernel of iteration t + 1): lanes 0 .. 63 of ONE wave, lane i owns parameter i.  fin_load issues every load of
// the pair's state at once (callers put the partial rows' reduction between the two, so that state and rows share one wait); fin_apply turns the
// reduced sums S[] into loss, dL/dtheta, (rigid) the pose chain rule, the optimiser update and theta of the next forward.
template <int ND>
struct FinRegs {
    int t;
    float best_prev, theta_old, p_old, m_old, v_old;
    float pose_old[(ND == 3) ? 6 : 3];
};
template <int ND>
__device__ __forceinline__ FinRegs<ND> fin_load(const float *param, const float *theta, const float *am, const float *av, const int *step, const float *best_loss,
                                                bool adam, bool rigid, int i)
{
    constexpr int NT = ND * (ND + 1), NPOSE = (ND == 3) ? 6 : 3;
    FinRegs<ND> r;
    const int ic = min(i, NT - 1);
    r.t = *step;
    r.best_prev = best_loss ? *best_loss : 0.f;
    r.theta_old = theta[ic]; r.p_old = param[ic];
    r.m_old = r.v_old = 0.f;
    if (adam) { r.m_old = am[ic]; r.v_old = av[ic]; }
#pragma unroll
    for (int k = 0; k < NPOSE; k++) r.pose_old[k] = rigid ? param[k] : 0.f;
    return r;
}
// Outputs: the pair's state (param / theta / Adam moments / step: nullable - the carry prologue's non-designated blocks write none), the caller-visible
// per-iteration records (losses[t], best theta / loss / index, grad: `user`), and theta of the next forward into LDS (`theta_lds`, nullable).
template <int ND>
__device__ __forceinline__ void fin_apply(const double *S, const FinRegs<ND> &r, int b, int i, double nvox, int D, int H, int W, const trx_loss_cfg &lc, const trx_opt_cfg &oc,
                                          const trx_affine_state &st, int mse_rows, float *param_out, float *theta_out, float *m_out, float *v_out, int *step_out, bool user,
                                          float *theta_lds, double *sh_dth, float *sh_pose)
{
    constexpr int NT = ND * (ND + 1);
    constexpr int NPOSE = (ND == 3) ? 6 : 3;
    const bool rigid = st.mode == TRX_PARAM_RIGID;
    const int np = rigid ? NPOSE : NT;
    const int ic = min(i, NT - 1);
    const int t = r.t;
    double bc1 = 1.0, rsbc2 = 1.0;
    if (oc.kind == TRX_OPT_ADAM) {   // beta^(t+1) by repeated squaring: a dozen fp64 multiplies instead of two pow() calls
        bc1 = 1.0 - ipow((double)oc.beta1, t + 1);
        rsbc2 = 1.0 / sqrt(1.0 - ipow((double)oc.beta2, t + 1));
    }
    const double scale[3] = {0.5 * W, 0.5 * H, 0.5 * D};
    double dth_i, total;
    if (ND == 3 && mse_rows) {   // S[0] = sum (w - y)^2, S[1 + i] = sum (w - y) J_i:  L = (w_mse / n + w_ssd alpha) S[0],  dL/dw_p = q (w_p - y_p)
        const double q = (double)lc.w_mse * 2.0 / nvox + (double)lc.w_ssd * (double)lc.ssd_alpha * 2.0;
        total = 0.5 * q * S[0];
        dth_i = scale[ic / (ND + 1)] * q * S[1 + ic];
    } else {
        const LossCoef L = loss_from_moments(S, nvox, lc);
        total = L.total;
        dth_i = scale[ic / (ND + 1)] * (L.c0 * S[5 + ic] + L.cy * S[5 + NT + ic] + L.cw * S[5 + 2 * NT + ic]);
    }
    const float lossf = (float)total;
    // best = first strict minimum, theta of THIS forward (ref:warpings.py:85-93)
    const bool is_best = (t == 0) || (lossf < r.best_prev);
    if (user) {
        if (is_best && i < NT) st.best_theta[(size_t)b * TRX_PSTRIDE + i] = r.theta_old;
        if (i == 0) {
            if (st.losses && t < st.losses_capacity) st.losses[(size_t)b * st.losses_capacity + t] = lossf;
            if (is_best) { st.best_loss[b] = lossf; st.best_idx[b] = t; }
        }
    }
    if (i == 0 && step_out) *step_out = t + 1;

    double g_i = dth_i;
    if (rigid) {
        if (i < NT) sh_dth[i] = dth_i;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        double dth[NT], g[NT];
#pragma unroll
        for (int k = 0; k < NT; k++) dth[k] = sh_dth[k];
        pose_vjp<ND>(r.pose_old, dth, g);
        g_i = 0.0;
#pragma unroll
        for (int k = 0; k < NPOSE; k++) g_i = (k == i) ? g[k] : g_i;
    }
    float p_new = r.p_old;
    if (i < np) {
        const float gf = (float)g_i;
        if (user && st.grad) st.grad[(size_t)b * TRX_PSTRIDE + i] = gf;
        if (oc.kind == TRX_OPT_ADAM) {
            const float mi = r.m_old + (gf - r.m_old) * (1.0f - oc.beta1);
            const float vi = oc.beta2 * r.v_old + (1.0f - oc.beta2) * gf * gf;
            if (m_out) { m_out[i] = mi; v_out[i] = vi; }
            const float denom = (float)(sqrt((double)vi) * rsbc2) + oc.eps;
            p_new = r.p_old - (float)((double)oc.lr / bc1) * (mi / denom);
        } else {
            p_new = r.p_old - oc.lr * gf;
        }
        if (param_out) param_out[i] = p_new;
    }
    float th_new = p_new;
    if (rigid) {
        if (i < NPOSE) sh_pose[i] = p_new;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        float pose_new[NPOSE];
#pragma unroll
        for (int k = 0; k < NPOSE; k++) pose_new[k] = sh_pose[k];
        double thd[NT];
        theta_from_pose<ND>(pose_new, thd);
        double th_i = 0.0;
#pragma unroll
        for (int k = 0; k < NT; k++) th_i = (k == i) ? thd[k] : th_i;
        th_new = (float)th_i;
    }
    if (i < NT) {
        if (theta_out) theta_out[i] = th_new;
        if (theta_lds) theta_lds[i] = th_new;
    }
}

// The carry buffers of trx_affine_run's one-launch iterations (two of them, by iteration parity): per pair 64 floats -
// theta[12] | param[12] | Adam m[12] | Adam v[12] | step (int) - the state the iteration's FIRST kernel computes in its prologue and every block of it reads.
constexpr int kCarryStride = 64;
constexpr int kCarryTheta = 0, kCarryParam = 12, kCarryM = 24, kCarryV = 36, kCarryStep = 48;

// state_in != nullptr: the pair's state is read from a carry buffer (the flush behind the last one-launch iteration of a run) instead of the caller's arrays
template <int ND>
__global__ __launch_bounds__(TRX_FIN_THREADS) void affine_finalize_kernel(const float *__restrict__ partials, int nblk,
                                                                          double nvox, int D, int H, int W,
                                                                          trx_loss_cfg lc, trx_opt_cfg oc,
                                                                          trx_affine_state st, const int *__restrict__ rows_used = nullptr, int mse_rows = 0,
                                                                          const float *__restrict__ state_in = nullptr)
{
    constexpr int NP = np_full(ND);
    constexpr int NT = ND * (ND + 1);
    constexpr int NPOSE = (ND == 3) ? 6 : 3;
    __shared__ double S[64];
    const int b = blockIdx.x;
    // Lane-parallel epilogue: lane i owns parameter i.  Every load of the per-pair state is issued up front
    // by all lanes at once (ONE memory round trip; a single thread looping over the 12 parameters paid a
    // dependent load->store chain per parameter, ~20 us) and BEFORE the partial rows, so that the state and the rows share
    // one wait; the scalar fp64 math is computed redundantly.
    __shared__ double sh_dth[NT];
    __shared__ float sh_pose[NPOSE];
    const int i = threadIdx.x;
    const bool rigid = st.mode == TRX_PARAM_RIGID;
    float *param = st.param + (size_t)b * TRX_PSTRIDE;
    float *theta = st.theta + (size_t)b * TRX_PSTRIDE;
    FinRegs<ND> r;
    if (i < 64) {
        if (state_in) {
            const float *sb = state_in + (size_t)b * kCarryStride;
            r = fin_load<ND>(sb + kCarryParam, sb + kCarryTheta, sb + kCarryM, sb + kCarryV, reinterpret_cast<const int *>(sb + kCarryStep), st.best_loss + b,
                             oc.kind == TRX_OPT_ADAM, rigid, i);
        } else {
            r = fin_load<ND>(param, theta, st.adam_m ? st.adam_m + (size_t)b * TRX_PSTRIDE : nullptr, st.adam_v ? st.adam_v + (size_t)b * TRX_PSTRIDE : nullptr, st.step + b,
                             st.best_loss + b, oc.kind == TRX_OPT_ADAM, rigid, i);
        }
    }
    {
        // rows the F1 pass wrote for this pair: the dual kernel lays a pair's rows out with stride nblk and fills the first
        // blocks_per_pair of the geometry it chose for the pair - it left that count in rows_used[b]
        int rows = nblk;
        if (rows_used) rows = min(abs(rows_used[b]) & kRowsMask, nblk);   // (negative: the pair was left to the exact-footprint kernel, which wrote |rows_used| rows)
        if (ND == 3 && mse_rows) reduce_partials<kNpMse>(partials + (size_t)b * nblk * kNpMse, rows, S);   // rows of the MSE / SSD-only step kernel
        else reduce_partials<NP>(partials + (size_t)b * nblk * NP, rows, S);
    }
    if (i >= 64) return;
    fin_apply<ND>(S, r, b, i, nvox, D, H, W, lc, oc, st, mse_rows, param, theta, st.adam_m ? st.adam_m + (size_t)b * TRX_PSTRIDE : nullptr,
                  st.adam_v ? st.adam_v + (size_t)b * TRX_PSTRIDE : nullptr, st.step + b, true, nullptr, sh_dth, sh_pose);
}

// The carry prologue of a 512-thread step block (3-D): see CarryKArgs.  `scratch`: the block's tile box (free until the body starts); on return s_th[0 .. 11]
// holds theta of this launch's forward and every wave has passed a barrier behind it.
template <int MODE>
__device__ __forceinline__ void carry_prologue(CarryKPtr c, int b, bool designated, int D, int H, int W, float *scratch, float *s_th, int wave, int lane)
{
    constexpr int NP = (MODE == 4) ? kNpMse : np_full(3);
    constexpr int NT = 12;
    const float *prev = c->prev_partials;
    const float *sp = c->state_prev;
    float *sn = c->state_next;
    const trx_affine_state st = {c->st.mode, c->st.param, c->st.theta, c->st.adam_m, c->st.adam_v, c->st.best_theta, c->st.best_loss, c->st.best_idx, c->st.losses,
                                 c->st.losses_capacity, c->st.step, c->st.grad};
    const trx_opt_cfg oc = {c->oc.kind, c->oc.lr, c->oc.beta1, c->oc.beta2, c->oc.eps};
    const bool adam = oc.kind == TRX_OPT_ADAM, rigid = st.mode == TRX_PARAM_RIGID;
    // where the pair's state is read from: the carry buffer of the other parity, or (first launch of a run) the caller's arrays
    const float *src_theta = sp ? sp + (size_t)b * kCarryStride + kCarryTheta : st.theta + (size_t)b * TRX_PSTRIDE;
    const float *src_param = sp ? sp + (size_t)b * kCarryStride + kCarryParam : st.param + (size_t)b * TRX_PSTRIDE;
    const float *src_m = sp ? sp + (size_t)b * kCarryStride + kCarryM : (st.adam_m ? st.adam_m + (size_t)b * TRX_PSTRIDE : nullptr);
    const float *src_v = sp ? sp + (size_t)b * kCarryStride + kCarryV : (st.adam_v ? st.adam_v + (size_t)b * TRX_PSTRIDE : nullptr);
    const int *src_step = sp ? reinterpret_cast<const int *>(sp + (size_t)b * kCarryStride + kCarryStep) : st.step + b;
    float *nxt = (designated && sn) ? sn + (size_t)b * kCarryStride : nullptr;
    if (prev == nullptr) {   // nothing pending: theta of this forward is the state's; the pair's first block seeds the carry buffer
        if (wave == 0) {
            const int ic = min(lane, NT - 1);
            const float th = src_theta[ic], pa = src_param[ic];
            const float m0 = (adam && src_m) ? src_m[ic] : 0.f, v0 = (adam && src_v) ? src_v[ic] : 0.f;
            const int t = *src_step;
            if (lane < NT) {
                s_th[lane] = th;
                if (nxt) { nxt[kCarryTheta + lane] = th; nxt[kCarryParam + lane] = pa; nxt[kCarryM + lane] = m0; nxt[kCarryV + lane] = v0; }
            }
            if (lane == 0 && nxt) *reinterpret_cast<int *>(nxt + kCarryStep) = t;
        }
        __syncthreads();
        return;
    }
    double *acc = reinterpret_cast<double *>(scratch);   // [8][64] | S[64] | dtheta[12] | pose (floats)
    double *S = acc + 8 * 64, *sh_dth = S + 64;
    float *sh_pose = reinterpret_cast<float *>(sh_dth + NT);
    FinRegs<3> r;
    if (wave == 0) r = fin_load<3>(src_param, src_theta, src_m, src_v, src_step, designated ? st.best_loss + b : nullptr, adam, rigid, lane);
    {
        // the pair's rows of the previous launch: wave w sums rows w, w + 8, ... of column `lane` in batches of 16 independent loads, a fixed order, the same
        // in every block of the pair.  (Measured alternatives, profiles/r06b_carry.txt: copying the rows into LDS with float4 loads first, +0.4 ... +1.2 us;
        // a grid without the surplus blocks of the geometry the pair does not run, +-0.)
        const int nblk = c->prev_nblk;
        const int rows = min(abs(c->prev_rows_used[b]) & kRowsMask, nblk);
        const float *part = prev + (size_t)b * nblk * NP;
        double s = 0.0;
        if (lane < NP) {
            constexpr int NB = TRX_CARRY_BATCH;   // loads in flight per lane
            for (int r0 = wave; r0 < rows; r0 += NB * 8) {
                float a[NB];
#pragma unroll
                for (int i = 0; i < NB; i++) {
                    const int row = r0 + i * 8;
                    const float v = part[(size_t)min(row, rows - 1) * NP + lane];   // clamped, unconditional (reduce_partials: a predicated load is a branch + a wait each)
                    a[i] = (row < rows) ? v : 0.f;
                }
                double d[NB];
#pragma unroll
                for (int i = 0; i < NB; i++) d[i] = (double)a[i];
#pragma unroll
                for (int w = 1; w < NB; w <<= 1)   // pairwise tree, fixed order
#pragma unroll
                    for (int i = 0; i + w < NB; i += 2 * w) d[i] += d[i + w];
                s += d[0];
            }
        }
        acc[wave * 64 + lane] = s;
    }
    __syncthreads();
    if (wave == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 8; w++) t += acc[w * 64 + lane];
        S[lane] = t;
    }
    __syncthreads();
    if (wave == 0) {
        const trx_loss_cfg lc = {c->lc.w_mse, c->lc.w_ncc, c->lc.ncc_alpha, c->lc.w_ssd, c->lc.ssd_alpha};
        fin_apply<3>(S, r, b, lane, c->nvox, D, H, W, lc, oc, st, c->mse_rows, nxt ? nxt + kCarryParam : nullptr, nxt ? nxt + kCarryTheta : nullptr,
                     nxt ? nxt + kCarryM : nullptr, nxt ? nxt + kCarryV : nullptr, nxt ? reinterpret_cast<int *>(nxt + kCarryStep) : nullptr, designated, s_th, sh_dth, sh_pose);
    }
    __syncthreads();
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
    const float fW = (float)W, fH = (float)H, fD = (float)D;
    for (VoxelWalk vw(blockIdx.x * TRX_BLOCK + threadIdx.x, gridDim.x * TRX_BLOCK, H, W); vw.i < nvox; vw.next(H, W)) {
        const size_t i = vw.i;
        const int x = vw.x, y = vw.y, z = vw.z;
        const float xn = base_coord(vol.xn, x, W), yn = base_coord(vol.yn, y, H);
        if constexpr (ND == 3) {
            const float zn = base_coord(vol.zn, z, D);
            const float ix = unnorm<3>(fmaf(th[1], yn, fmaf(th[0], xn, fmaf(th[2], zn, th[3]))), fW);
            const float iy = unnorm<3>(fmaf(th[5], yn, fmaf(th[4], xn, fmaf(th[6], zn, th[7]))), fH);
            const float iz = unnorm<3>(fmaf(th[9], yn, fmaf(th[8], xn, fmaf(th[10], zn, th[11]))), fD);
            for (int ch = 0; ch < channels; ch++) o[ch * nvox + i] = sample3(mov + ch * chan_stride, D, H, W, ix, iy, iz).v;
        } else {
            const float ix = unnorm<2>(fmaf(th[1], yn, fmaf(th[0], xn, th[2])), fW);
            const float iy = unnorm<2>(fmaf(th[4], yn, fmaf(th[3], xn, th[5])), fH);
            for (int ch = 0; ch < channels; ch++) o[ch * nvox + i] = sample2(mov + ch * chan_stride, H, W, ix, iy).v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Forward warp and its theta-backward on a SUB-LATTICE of the output grid: the NMI loss (ref:utils.py:236-252) only ever looks at
// F.interpolate(warped, size, mode="nearest"), i.e. at the output voxels (iz[kz], iy[ky], ix[kx]) - 10^6 of the 1.7e7 voxels of a
// 256^3 volume.  Evaluating the warp there directly replaces a full-volume warp, the nearest down-sampling, its autograd backward
// (a scatter into a full-volume gradient) and a full-volume warp backward.  Same coordinate arithmetic as affine_warp_kernel.
// ------------------------------------------------------------------------------------------
struct LatticeIdx {
    const int *iz, *iy, *ix;
    int nz, ny, nx;
};

template <int ND>
__device__ __forceinline__ void lattice_coords(const trx_volumes &vol, const float *__restrict__ th, int x, int y, int z, float &xn, float &yn,
                                               float &zn, float &ix, float &iy, float &iz)
{
    xn = base_coord(vol.xn, x, vol.W); yn = base_coord(vol.yn, y, vol.H); zn = 0.f; iz = 0.f;
    if constexpr (ND == 3) {
        zn = base_coord(vol.zn, z, vol.D);
        ix = unnorm<3>(fmaf(th[1], yn, fmaf(th[0], xn, fmaf(th[2], zn, th[3]))), (float)vol.W);
        iy = unnorm<3>(fmaf(th[5], yn, fmaf(th[4], xn, fmaf(th[6], zn, th[7]))), (float)vol.H);
        iz = unnorm<3>(fmaf(th[9], yn, fmaf(th[8], xn, fmaf(th[10], zn, th[11]))), (float)vol.D);
    } else {
        ix = unnorm<2>(fmaf(th[1], yn, fmaf(th[0], xn, th[2])), (float)vol.W);
        iy = unnorm<2>(fmaf(th[4], yn, fmaf(th[3], xn, th[5])), (float)vol.H);
    }
}

// One pass over the lattice, four points per thread and trip: the three dependent memory round trips of a point (index tables ->
// coordinate tables -> the eight corners) are each issued for all four points before the first is used (branch-free sampler), so a
// thread pays ~3 latencies per four points instead of twelve.  BWD = false: out[b][k] = warped value; BWD = true: acc += go[k] * J_k.
template <int ND, bool BWD>
__device__ __forceinline__ void lattice_pass(const trx_volumes &vol, const float *__restrict__ th, const float *__restrict__ mov, const LatticeIdx &L,
                                             const float *__restrict__ go, float *__restrict__ out, float (&acc)[ND * (ND + 1)], float &vmin, float &vmax)
{
    constexpr int U = 4;
    const unsigned n = (unsigned)L.nz * L.ny * L.nx;   // < 2^31 (checked by the caller): 32-bit index arithmetic
    const unsigned stride = gridDim.x * TRX_BLOCK;
    for (unsigned i0 = blockIdx.x * TRX_BLOCK + threadIdx.x; i0 < n; i0 += U * stride) {
        unsigned idx[U];
        bool ok[U];
        int x[U], y[U], z[U];
        float g[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned i = i0 + u * stride;
            ok[u] = i < n;
            idx[u] = ok[u] ? i : i0;                   // a thread past the end repeats its first point (discarded below)
            const unsigned r = idx[u] / (unsigned)L.nx, kx = idx[u] - r * L.nx, kz = r / (unsigned)L.ny, ky = r - kz * L.ny;
            x[u] = L.ix[kx]; y[u] = L.iy[ky]; z[u] = (ND == 3) ? L.iz[kz] : 0;
            g[u] = BWD ? go[idx[u]] : 0.f;
        }
        float xn[U], yn[U], zn[U], ix[U], iy[U], iz[U];
#pragma unroll
        for (int u = 0; u < U; u++) lattice_coords<ND>(vol, th, x[u], y[u], z[u], xn[u], yn[u], zn[u], ix[u], iy[u], iz[u]);
        float v[U], gq[U][ND];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (ND == 3) {
                const Samp3 sm = sample3_padded(mov, vol.D, vol.H, vol.W, ix[u], iy[u], iz[u]);
                v[u] = sm.v; gq[u][0] = sm.dx; gq[u][1] = sm.dy; gq[u][2] = sm.dz;
            } else {
                const Samp2 sm = sample2(mov, vol.H, vol.W, ix[u], iy[u]);
                v[u] = sm.v; gq[u][0] = sm.dx; gq[u][1] = sm.dy;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (!BWD) {
                if (ok[u]) { out[idx[u]] = v[u]; vmin = fminf(vmin, v[u]); vmax = fmaxf(vmax, v[u]); }
            } else {
                const float w = ok[u] ? g[u] : 0.f;
#pragma unroll
                for (int c = 0; c < ND; c++) {   // the partial-row layout of affine_bwd_finalize_kernel: per component (xn, yn[, zn], 1)
                    const float q = w * gq[u][c];
                    acc[c * (ND + 1) + 0] = fmaf(q, xn[u], acc[c * (ND + 1) + 0]);
                    acc[c * (ND + 1) + 1] = fmaf(q, yn[u], acc[c * (ND + 1) + 1]);
                    if constexpr (ND == 3) acc[c * (ND + 1) + 2] = fmaf(q, zn[u], acc[c * (ND + 1) + 2]);
                    acc[c * (ND + 1) + ND] += q;
                }
            }
        }
    }
}

template <int ND>
__global__ __launch_bounds__(TRX_BLOCK) void affine_warp_lattice_kernel(trx_volumes vol, const float *__restrict__ theta, LatticeIdx L,
                                                                        float *__restrict__ out, float *__restrict__ block_minmax)
{
    const int b = blockIdx.y;
    const size_t n = (size_t)L.nz * L.ny * L.nx;
    float acc[ND * (ND + 1)];
    float vmin = INFINITY, vmax = -INFINITY;
    lattice_pass<ND, false>(vol, theta + (size_t)b * TRX_PSTRIDE, vol.moving + (size_t)b * vol.moving_stride, L, nullptr, out + (size_t)b * n, acc, vmin, vmax);
    if (block_minmax) {   // extrema of this block's values (the NMI sample lines run between the extrema of the warped samples): one pair per block
        __shared__ float red[2][TRX_WAVES];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, m, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, m, 64)); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = vmin; red[1][threadIdx.x >> 6] = vmax; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < TRX_WAVES; w++) { vmin = fminf(vmin, red[0][w]); vmax = fmaxf(vmax, red[1][w]); }
            block_minmax[((size_t)b * gridDim.x + blockIdx.x) * 2 + 0] = vmin;
            block_minmax[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = vmax;
        }
    }
}

// The two sample lines of the NMI loss for pair b's P patches (ref:utils.py:40-48 get_pdf: linspace(max, min, bins) of the samples the
// PDF is taken of): line A between the extrema of the warped samples (reduced here from the lattice kernel's per-block pairs), line B
// between the extrema of warped and target samples pooled; xis[b * P + p][0 .. bins) = A, [bins .. 2 bins) = B.  The points follow
// torch.lerp's two-sided formula on the ramp k / (bins - 1), like the torch composition this replaces (aminmax, 2 x lerp, maximum,
// minimum, cat: six launches).  mm_out[b] = (min, max) of the warped samples.
__global__ __launch_bounds__(1024) void nmi_lines_kernel(const float *__restrict__ block_minmax, int nblk, const float *__restrict__ mm_target, int P, int bins,
                                                         float *__restrict__ xis, float *__restrict__ mm_out)
{
    __shared__ float red[2][16];
    const int b = blockIdx.x, tid = threadIdx.x;
    float vmin = INFINITY, vmax = -INFINITY;
    for (int i = tid; i < nblk; i += 1024) {
        vmin = fminf(vmin, block_minmax[((size_t)b * nblk + i) * 2 + 0]);
        vmax = fmaxf(vmax, block_minmax[((size_t)b * nblk + i) * 2 + 1]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { vmin = fminf(vmin, __shfl_xor(vmin, m, 64)); vmax = fmaxf(vmax, __shfl_xor(vmax, m, 64)); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = vmin; red[1][tid >> 6] = vmax; }
    __syncthreads();
    vmin = red[0][0]; vmax = red[1][0];
#pragma unroll
    for (int w = 1; w < 16; w++) { vmin = fminf(vmin, red[0][w]); vmax = fmaxf(vmax, red[1][w]); }
    const float tlo = mm_target[b * 2 + 0], thi = mm_target[b * 2 + 1];
    const float lo2 = fminf(vmin, tlo), hi2 = fmaxf(vmax, thi);
    if (tid == 0 && mm_out) { mm_out[b * 2 + 0] = vmin; mm_out[b * 2 + 1] = vmax; }
    auto lerp = [](float start, float end, float w) { const float d = end - start; return (fabsf(w) < 0.5f) ? start + w * d : end - d * (1.0f - w); };
    for (int i = tid; i < P * 2 * bins; i += 1024) {
        const int p = i / (2 * bins), k = i - p * 2 * bins;
        const bool second = k >= bins;
        const float w = (float)(second ? k - bins : k) / (float)(bins - 1);
        xis[((size_t)b * P + p) * 2 * bins + k] = second ? lerp(hi2, lo2, w) : lerp(vmax, vmin, w);
    }
}

// The tail of one iteration of the default-criterion loop (ref:warpings.py:80-93 / :146-159 after error.backward()): the loss of this
// iteration = sum of the NMI terms + the fused terms' loss, the theta of this forward into the history, g = g_a + g_b, SGD on theta -
// or, rigid, on the pose through Theta's vector-Jacobian product - and theta of the next forward into `param_copy` (the fused solver's
// parameter).  One wave; replaces ~8 element-wise launches per iteration.
template <int ND>
__global__ __launch_bounds__(64) void nmi_loop_update_kernel(float *__restrict__ theta, float *__restrict__ pose, const float *__restrict__ g_a,
                                                             const float *__restrict__ g_b, float lr, const float *__restrict__ loss_terms, int n_terms,
                                                             const float *__restrict__ loss_b, float *__restrict__ hist_loss_t,
                                                             float *__restrict__ hist_theta_t, float *__restrict__ param_copy)
{
    constexpr int NT = ND * (ND + 1), NPOSE = (ND == 3) ? 6 : 3;
    const int i = threadIdx.x;
    if (i == 0) {
        float tot = 0.f;
        for (int k = 0; k < n_terms; k++) tot += loss_terms[k];   // torch's terms.sum() of <= 8 values, then + the fused loss (fp32 like the composition)
        if (loss_b) tot += *loss_b;
        *hist_loss_t = tot;
    }
    const float th_i = (i < TRX_PSTRIDE) ? theta[i] : 0.f;
    if (i < TRX_PSTRIDE) hist_theta_t[i] = th_i;
    float gi = 0.f;
    if (i < TRX_PSTRIDE) gi = g_a[i] + (g_b ? g_b[i] : 0.f);
    float th_new = th_i;
    if (pose) {
        float p[NPOSE];
        double g[NT], dx[NPOSE];
#pragma unroll
        for (int k = 0; k < NPOSE; k++) p[k] = pose[k];
#pragma unroll
        for (int k = 0; k < NT; k++) g[k] = (double)(g_a[k] + (g_b ? g_b[k] : 0.f));
        pose_vjp<ND>(p, g, dx);
#pragma unroll
        for (int k = 0; k < NPOSE; k++) p[k] = p[k] - lr * (float)dx[k];   // pose.sub_(dpose, alpha = lr) on the fp32 dpose
        double th[NT];
        theta_from_pose<ND>(p, th);
        double v = 0.0, pv = 0.0;
#pragma unroll
        for (int k = 0; k < NT; k++) v = (k == i) ? th[k] : v;
#pragma unroll
        for (int k = 0; k < NPOSE; k++) pv = (k == i) ? (double)p[k] : pv;
        if (i < NT) th_new = (float)v;
        __syncthreads();   // every lane has read the old pose
        if (i < NPOSE) pose[i] = (float)pv;
    } else if (i < NT) {
        th_new = th_i - lr * gi;                                   // theta.sub_(g, alpha = lr)
    }
    if (i < TRX_PSTRIDE) { theta[i] = th_new; if (param_copy) param_copy[i] = th_new; }
}

template <int ND>
__global__ __launch_bounds__(TRX_BLOCK) void affine_lattice_bwd_kernel(trx_volumes vol, const float *__restrict__ theta, LatticeIdx L,
                                                                       const float *__restrict__ grad_out, float *__restrict__ partials)
{
    constexpr int NT = ND * (ND + 1);
    const int b = blockIdx.y;
    const size_t n = (size_t)L.nz * L.ny * L.nx;
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = 0.f;
    float vmin = 0.f, vmax = 0.f;
    lattice_pass<ND, true>(vol, theta + (size_t)b * TRX_PSTRIDE, vol.moving + (size_t)b * vol.moving_stride, L, grad_out + (size_t)b * n, nullptr, acc, vmin, vmax);
    block_reduce_store<NT>(acc, partials + ((size_t)b * gridDim.x + blockIdx.x) * NT);
}

// Theta (ref:utils.py:287-310) and its vector-Jacobian product for callers that assemble dL/dtheta themselves (the default-criterion
// loop: NMI's gradient arrives outside the fused step): one wave per pair, fp64 like the finalise kernel.
template <int ND>
__global__ __launch_bounds__(64) void theta_chain_kernel(const float *__restrict__ pose, const float *__restrict__ dtheta, float *__restrict__ theta_out,
                                                         float *__restrict__ dpose_out)
{
    constexpr int NT = ND * (ND + 1), NPOSE = (ND == 3) ? 6 : 3;
    const int b = blockIdx.x, i = threadIdx.x;
    float p[NPOSE];
#pragma unroll
    for (int k = 0; k < NPOSE; k++) p[k] = pose[(size_t)b * TRX_PSTRIDE + k];
    if (theta_out) {
        double th[NT];
        theta_from_pose<ND>(p, th);
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < NT; k++) v = (k == i) ? th[k] : v;
        if (i < NT) theta_out[(size_t)b * TRX_PSTRIDE + i] = (float)v;
    }
    if (dtheta && dpose_out) {
        double g[NT], dx[NPOSE];
#pragma unroll
        for (int k = 0; k < NT; k++) g[k] = (double)dtheta[(size_t)b * TRX_PSTRIDE + k];
        pose_vjp<ND>(p, g, dx);
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < NPOSE; k++) v = (k == i) ? dx[k] : v;
        if (i < NPOSE) dpose_out[(size_t)b * TRX_PSTRIDE + i] = (float)v;
    }
}

static int check_vol(const trx_volumes *v, bool need_target)
{
    if (!v || !v->moving || (need_target && !v->target)) return TRX_ERR_ARG;
    if (v->ndim != 2 && v->ndim != 3) return TRX_ERR_NDIM;
    if (v->B < 1 || v->D < 1 || v->H < 1 || v->W < 1) return TRX_ERR_ARG;
    if (v->ndim == 2 && v->D != 1) return TRX_ERR_NDIM;
    if (v->B > 65535) return TRX_ERR_ARG;
    if ((size_t)v->D * v->H * v->W >= ((size_t)1 << 31)) return TRX_ERR_ARG;   // 32-bit voxel indices inside one volume
    return TRX_OK;
}

constexpr int kTargetBlocks = 2048;

}  // namespace trx

using namespace trx;

// 0 = primary geometry only (TRX_FLAG_SINGLE_GEOM), 1 = GeomA / GeomR chosen per pair inside one two-body kernel, 2 = the same choice
// as a pair of single-body launches (compile-time alternative, TRX_DUAL_DEFAULT).
static int use_dual(const trx_volumes *vol)
{
    if (vol->flags & TRX_FLAG_SINGLE_GEOM) return 0;
    return TRX_TILE_CFG == 0 ? TRX_DUAL_DEFAULT : 0;
}

// Does a launch of the step kernels offer the z-streaming body to its pairs?  Sizes only (the theta part is zs_fits on the device):
// the shape must tile, and the launch must fill the chip with blocks of useful length (small problems stay with the tile kernels).
// Block slots of the device for the 512-thread step kernels: two per CU (LDS and registers allow exactly two) - 512 on MI355X (256 CUs).
// The size of the flat grid and the yardstick of the offer rules below; queried per launch (the runtime caches device attributes).
static int persistent_blocks()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return TRX_PERSISTENT_BLOCKS;
    return 2 * cus;
}

static ZGeom zs_launch_geom(const trx_volumes &v)
{
    ZGeom none = ZGeom{};
    if ((v.flags & TRX_FLAG_NO_ZSTREAM) || !zs_shape_ok<ZS64>(v)) return none;
    const ZGeom g = zs_geom<ZS64>(v);
    if (v.flags & TRX_FLAG_ZSTREAM) return g;
    const long zb = (long)g.blocks_per_pair * v.B;
    if (zb < TRX_ZS_MIN_BLOCKS || g.planes_per_seg < TRX_ZS_MIN_PLANES) return none;
    // between one block per CU and a full round the streaming blocks share CUs unevenly; where the tile kernels fit ONE round of block
    // slots they win that case (3 x 192^3: 62 us against 69), everywhere else measured the streaming body is ahead
    const int slots = persistent_blocks();
    if (zb > slots / 2 && zb < slots && (long)tile_geom<GeomA>(v).blocks_per_pair * v.B <= slots) return none;
    return g;
}

// partial rows per pair that a tile-path launch may write (the dual grid is sized for the geometry with more blocks)
static size_t tile_rows_per_pair(const trx_volumes &v)
{
    size_t n = (size_t)tile_geom<GeomP>(v).blocks_per_pair;
    const size_t a = (size_t)tile_geom<GeomA>(v).blocks_per_pair, r = (size_t)tile_geom<GeomR>(v).blocks_per_pair;
    const size_t d = (size_t)tile_geom<GeomD>(v).blocks_per_pair, rd = (size_t)tile_geom<GeomRD>(v).blocks_per_pair;
    size_t z = zs_shape_ok<ZS64>(v) ? (size_t)zs_geom<ZS64>(v).blocks_per_pair : 0;
    if (TRX_ZS_FLAT && zs_shape_ok<ZSF>(v) && (size_t)zs_geom<ZSF>(v).blocks_per_pair > z) z = (size_t)zs_geom<ZSF>(v).blocks_per_pair;
    if (a > n) n = a;
    if (r > n) n = r;
    if (d > n) n = d;
    if (rd > n) n = rd;
    if (z > n) n = z;
    return n;
}

// Workspace of the affine entry points: [B][rows][41] partial sums | coordinate tables (callers that pass none) | rows_used[B + 11]
// ... | (3-D) second partial buffer | second rows_used[B + 11] | two carry buffers [B][64] floats  (the parity buffers of trx_affine_run's one-launch iterations)
struct AffineWs {
    size_t rows, off_tab, off_rows_used, off_part2, off_rows_used2, off_carry, bytes;
};
static AffineWs affine_ws(const trx_volumes &v)
{
    AffineWs w;
    AffineGeom g = affine_geom(v, kTargetBlocks);
    w.rows = (size_t)g.nblk;
    if (v.ndim == 3) {
        const size_t t = tile_rows_per_pair(v);
        if (t > w.rows) w.rows = t;
    }
    w.off_tab = ((size_t)v.B * w.rows * np_full(3) * sizeof(float) + 255) & ~(size_t)255;
    w.off_rows_used = w.off_tab + (((size_t)(v.W + v.H + v.D) * sizeof(float) + 255) & ~(size_t)255);
    const size_t notes = ((size_t)(v.B + 11) * sizeof(int) + 255) & ~(size_t)255;   // rows_used[B] | pairs left by the z-streaming kernel | its pair mask (2) | the exact-footprint kernel's work tickets, one per XCD (8)
    w.off_part2 = w.off_rows_used + notes;
    w.off_rows_used2 = w.off_part2 + (v.ndim == 3 ? w.off_tab : 0);
    w.off_carry = w.off_rows_used2 + (v.ndim == 3 ? notes : 0);
    w.bytes = w.off_carry + (v.ndim == 3 ? (((size_t)2 * v.B * kCarryStride * sizeof(float) + 255) & ~(size_t)255) : 0);
    return w;
}

extern "C" size_t trx_affine_workspace_bytes(const trx_volumes *vol)
{
    if (check_vol(vol, false) != TRX_OK) return 0;
    return affine_ws(*vol).bytes;
}

extern "C" size_t trx_affine_workspace_rows_offset(const trx_volumes *vol)
{
    if (check_vol(vol, false) != TRX_OK) return 0;
    return affine_ws(*vol).off_rows_used;
}

// [host] Would a step of this batch run entirely on the z-streaming kernel's two tiles at these (HOST) thetas?  The same test the kernel
// evaluates per pair on the device (zs_nsub), for callers that decide about TRX_FLAG_ONE_KERNEL from values they hold on the host.
extern "C" int trx_affine_near_identity(const trx_volumes *vol, const float *theta_host)
{
    if (check_vol(vol, false) != TRX_OK || !theta_host || vol->ndim != 3) return 0;
    const ZGeom zg = zs_launch_geom(*vol);
    if (zg.blocks_per_pair == 0) return 0;
    const bool flat_ok = TRX_ZS_FLAT && zs_shape_ok<ZSF>(*vol) && !(vol->flags & TRX_FLAG_NO_ZS_FLAT);
    const ZGeom zgf = flat_ok ? zs_geom<ZSF>(*vol) : ZGeom{};
    const float fD = (float)vol->D, fH = (float)vol->H, fW = (float)vol->W;
    for (int b = 0; b < vol->B; b++) {
        const float *th = theta_host + (size_t)b * TRX_PSTRIDE;
        if (zs_nsub<ZS64>(th, fD, fH, fW, zg.planes_per_seg) > 0) continue;
        if (flat_ok && zs_nsub<ZSF>(th, fD, fH, fW, zgf.planes_per_seg) > 0) continue;
        return 0;
    }
    return 1;
}

static bool use_tile_path(const trx_volumes *vol)
{
    if (vol->ndim != 3) return false;
    return !(vol->flags & TRX_FLAG_GATHER_PATH);   // the un-tiled kernel stays reachable so that the tests can compare the two
}

// MODE 0 / 1 dispatch: LDS-tiled kernel for 3-D, row-walking gather kernel otherwise.
// Returns the number of partial rows per pair through *nblk.
struct CarryLaunch {   // trx_affine_run's request for the carry form of a step launch: which parity buffers it writes, the kernel's carry arguments; `used` = it was launched
    int parity;
    CarryKArgs args;
    bool used;
    float *partials_out;   // where the launch wrote its rows (the parity buffer), whether or not the carry form was used
};
template <int MODE>
static int launch_f1(const trx_volumes *vol, const float *theta, float *partials, int *nblk, hipStream_t s, bool dual, const int **rows_used = nullptr, CarryLaunch *cl = nullptr);

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

// rows_used != nullptr: the caller's reduction reads the row count of every pair from the device array returned through it (the
// step's finalise kernel): surplus blocks of the dual grid then write nothing, and the step kernels may offer their extra bodies.
template <int MODE>
static int launch_f1(const trx_volumes *vol, const float *theta, float *partials, int *nblk, hipStream_t s, bool dual, const int **rows_used, CarryLaunch *cl)
{
    if (rows_used) *rows_used = nullptr;
    if (cl) cl->used = false;
    if (use_tile_path(vol)) {
        TileGeom t = tile_geom(*vol);
        trx_volumes v = *vol;
        const AffineWs ws = affine_ws(*vol);
        float *const ws_base = partials;   // (tables and notes sit at fixed offsets from the workspace's start; the rows go to the parity buffer)
        const bool odd = cl != nullptr && cl->parity != 0;
        if (odd) partials = (float *)((char *)ws_base + ws.off_part2);
        if (cl) cl->partials_out = partials;
        if (!v.xn || !v.yn || !v.zn) {
            // tables live behind the partials (trx_affine_workspace_bytes reserves the room)
            float *tab = (float *)((char *)ws_base + ws.off_tab);
            const int n = max(v.W, max(v.H, v.D));
            hipLaunchKernelGGL(fill_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, s, tab, v.W, v.H, v.D);
            TRX_CHECK_LAUNCH();
            v.xn = tab; v.yn = tab + v.W; v.zn = tab + v.W + v.H;
        }
        if (dual && use_dual(vol)) {
            const TileGeom ta = tile_geom<GeomA>(*vol), tr = tile_geom<GeomR>(*vol);
            const int gx = ta.blocks_per_pair > tr.blocks_per_pair ? ta.blocks_per_pair : tr.blocks_per_pair;
            const bool aware = rows_used != nullptr && use_dual(vol) == 1;
            const bool step_kernel = aware && (MODE == 0 || MODE == 4);
            // the deep tile joins the choice only where the reader of the partial rows knows which rows were written (the step's finalise
            // kernel) and where its 128-row slabs still fill the chip (8192 voxels per tile: big batches)
            TileGeom td = TileGeom{};
            if (step_kernel) {
                const TileGeom cand = tile_geom<GeomD>(*vol);
                // big batches as before; smaller ones where the deep tiling still fills every block slot with blocks of at least four tiles
                // (1 or 2 pairs of 256^3: -3 %; never where it would leave slots empty - 2 x 128^3 measured +29 % with it)
                const long blocks_d = (long)cand.blocks_per_pair * vol->B;
                const long slots_d = persistent_blocks();   // block slots of these 512-thread kernels (MI355X: 512) - the same number the flat grid and the z-streaming / exact-footprint rules use
                const long rem_d = blocks_d % slots_d;  // a last round that is less than ~60 % full costs more than the deeper tile saves (6 x 256^3: +6 %)
                const bool small_ok = TRX_DEEP_SMALL && blocks_d >= slots_d && cand.tiles_per_seg >= 4 && (rem_d == 0 || rem_d >= slots_d * 5 / 8);
                if (TRX_DEEP_TILE && (blocks_d >= 2 * slots_d || small_ok || (vol->flags & TRX_FLAG_DEEP_TILE))) td = cand;
            }
            // GeomRD (GeomR's box under a 16 x 16 x 16 tile) joins under the same condition: rotations whose pre-image still fits that box
            TileGeom trd = TileGeom{};
            if (step_kernel && TRX_ROT_DEEP_TILE && TRX_DEEP_TILE && !(vol->flags & TRX_FLAG_NO_ROT_DEEP_TILE)) {
                const TileGeom cand = tile_geom<GeomRD>(*vol);
                // tiny volumes (<= 64^3: at most 64 of these tiles) stay with GeomR's smaller tiles: measured +5 ... +14 % otherwise; and a launch
                // of fewer than 1024 of these tiles (one pair up to 128^3) is a step of ~15 us, of which the five-body kernel's entry is 2-3
                if ((cand.ntiles >= 128 && (long)cand.ntiles * vol->B >= 2l * persistent_blocks()) || (vol->flags & TRX_FLAG_DEEP_TILE)) trd = cand;
            }
            // the z-streaming body: pairs next to the identity (zs_fits) in launches that fill the chip
            ZGeom zg = ZGeom{};
            if (step_kernel && TRX_DEEP_TILE) zg = zs_launch_geom(*vol);
            int gxx = gx;
            if (td.blocks_per_pair > gxx) gxx = td.blocks_per_pair;
            if (trd.blocks_per_pair > gxx) gxx = trd.blocks_per_pair;
            if (zg.blocks_per_pair > gxx) gxx = zg.blocks_per_pair;
            if (TRX_ZS_FLAT && zg.blocks_per_pair > 0 && zs_shape_ok<ZSF>(*vol) && zs_geom<ZSF>(*vol).blocks_per_pair > gxx) gxx = zs_geom<ZSF>(*vol).blocks_per_pair;
            int *ru = aware ? (int *)((char *)ws_base + (odd ? ws.off_rows_used2 : ws.off_rows_used)) : nullptr;
            // big batches of the step kernels: a flat grid of persistent blocks over a pair-major work list (no surplus blocks; see the kernel)
            const int slots = persistent_blocks();
            // (B <= 64: the flat grid keeps one pair's state per lane)
            const bool flat = step_kernel && TRX_DEEP_TILE && TRX_FLAT_GRID && vol->B <= 64 && (long)gxx * vol->B >= 2 * slots;
            // the exact-footprint kernel (rotated pairs that GeomR would take) behind launches that fill the chip: columns of at most 64 of its tiles
            const TileGeom tef = tile_geom<GeomRD>(*vol);   // (its 16^3 tiling is GeomRD's)
            // (sizes: its row / plane pitches are 24-bit multiplier operands, its byte offsets 32-bit)
            const bool eft_sizes = (long)vol->H * vol->W * 4 < (1l << 24) && (size_t)vol->D * vol->H * vol->W < ((size_t)1 << 29);
            const int eft = (TRX_EFT_BODY && step_kernel && TRX_DEEP_TILE && ru && tef.tiles_per_seg <= 64 && eft_sizes && !(vol->flags & TRX_FLAG_NO_EFT) &&
                             (flat || (vol->flags & TRX_FLAG_EFT))) ? tef.blocks_per_pair : 0;   // = the partial rows per pair of that kernel
            if (eft && tef.blocks_per_pair > gxx) gxx = tef.blocks_per_pair;
            // flat launches that offer the z-streaming body run it as a kernel of its own, FIRST (affine_zs_step_kernel): it alone evaluates the
            // test, the kernels behind it skip its pairs and return at once when it has taken them all
            const bool zs_first = flat && zg.blocks_per_pair > 0 && ru != nullptr && !(vol->flags & TRX_FLAG_ZS_FUSED);
            // ... and the exact-footprint body then rides in the tile kernel (WHICH = 5) when its tiling is the one GeomRD is offered with
            const bool eft_merged = TRX_EFT_MERGED && zs_first && eft != 0 && trd.blocks_per_pair == tef.blocks_per_pair && (MODE == 0 || MODE == 4);
            if constexpr (MODE == 0 || MODE == 4) {
                if (zs_first) {
                    const ZGeom zgf = (TRX_ZS_FLAT && zs_shape_ok<ZSF>(*vol) && !(vol->flags & TRX_FLAG_NO_ZS_FLAT)) ? zs_geom<ZSF>(*vol) : ZGeom{};
                    if (vol->flags & TRX_FLAG_ONE_KERNEL) {
                        // the one-kernel form of a step: the streaming kernel also runs the pairs outside its window (GeomR's body), nothing behind it
                        hipLaunchKernelGGL((affine_zs_one_kernel<MODE>), dim3(slots, 1), dim3(ZS64::Threads), 0, s, v, theta, zg, zgf, partials, ru, gxx, tr);
                        TRX_CHECK_LAUNCH();
                        *nblk = gxx;
                        *rows_used = ru;
                        return TRX_OK;
                    }
                    hipLaunchKernelGGL((affine_zs_step_kernel<MODE>), dim3(slots, 1), dim3(ZS64::Threads), 0, s, v, theta, zg, zgf, partials, ru, gxx);
                    TRX_CHECK_LAUNCH();
                }
                if (eft && !eft_merged) {   // in front of the tile kernel: takes its pairs and marks them rows_used < 0 (none - the usual case next to the identity - costs ~3 us, ~1.5 behind the z-streaming kernel)
                    const int wd = td.blocks_per_pair > 0, wrd = trd.blocks_per_pair > 0, zp = zs_first ? -1 : (zg.blocks_per_pair > 0 ? zg.planes_per_seg : 0);   // (-1: the z-streaming kernel ran in front)
                    if (flat) hipLaunchKernelGGL((affine_eft_step_kernel<MODE>), dim3(slots, 1), dim3(ECfg::Threads), 0, s, v, theta, tef, partials, ru, gxx, wd, wrd, zp);
                    else hipLaunchKernelGGL((affine_eft_step_kernel<MODE>), dim3(tef.blocks_per_pair, vol->B), dim3(ECfg::Threads), 0, s, v, theta, tef, partials, ru, -gxx, wd, wrd, zp);
                    TRX_CHECK_LAUNCH();
                }
            }
            if (eft_merged) hipLaunchKernelGGL((affine_tile_dual_kernel<MODE, 5>), dim3(slots, 1), dim3(512), 0, s, v, theta, ta, tr, 1, partials, 0, td, trd, zg, ru, gxx, eft);
            else if (zs_first) hipLaunchKernelGGL((affine_tile_dual_kernel<MODE, 4>), dim3(slots, 1), dim3(512), 0, s, v, theta, ta, tr, 1, partials, 0, td, trd, zg, ru, gxx, eft);
            else if (flat) launch_dual<MODE>(dim3(slots, 1), s, v, theta, ta, tr, 1, partials, 1, 0, td, trd, zg, ru, gxx, eft);
            else launch_dual<MODE>(dim3(gxx, vol->B), s, v, theta, ta, tr, 1, partials, use_dual(vol), aware ? 0 : 1, td, trd, zg, ru, 0, eft, cl ? &cl->args : nullptr, cl ? &cl->used : nullptr);
            TRX_CHECK_LAUNCH();
            *nblk = gxx;
            if (aware) *rows_used = ru;
            return TRX_OK;
        }
        hipLaunchKernelGGL((affine_tile_kernel<MODE>), dim3(t.blocks_per_pair, vol->B), dim3(kTileThreads), 0, s, v, theta, t, 1, partials);
        TRX_CHECK_LAUNCH();
        *nblk = t.blocks_per_pair;
        return TRX_OK;
    }
    if constexpr (MODE == 4) {
        return TRX_ERR_ARG;   // (the MSE-only rows exist in the tile kernels only; trx_affine_step asks for them on the tile path)
    } else {
        AffineGeom g = affine_geom(*vol, kTargetBlocks);
        *nblk = g.nblk;
        return launch_accum<MODE>(vol, theta, g, 1, 0, partials, s);
    }
}

static int launch_finalize(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt, const trx_affine_state *st, const float *partials, int nblk,
                           const int *rows_used, bool mse_only, const float *state_in, hipStream_t s);

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
    float *partials = (float *)workspace;
    int nblk = 0;
    // Every step picks GeomA / GeomR per pair in the kernel (TRX_FLAG_SINGLE_GEOM: the primary geometry only).  Rigid runs start from a
    // random pose (reference: torch.rand, up to 1 rad) and live at large rotations; affine runs start at the identity, where the
    // GeomA body is all that runs, but may rotate away from it: the single-geometry kernel then gathers from L2 at 3.2x the cost.
    const int *rows_used = nullptr;
    // without an NCC term only d = warped - target matters: the step kernel then keeps 13 sums instead of 41 (3-D tile path)
    const bool mse_only = (loss->w_ncc == 0.f) && use_tile_path(vol);
    rc = mse_only ? launch_f1<4>(vol, st->theta, partials, &nblk, s, true, &rows_used) : launch_f1<0>(vol, st->theta, partials, &nblk, s, true, &rows_used);
    if (rc) return rc;
    return launch_finalize(vol, loss, opt, st, partials, nblk, rows_used, mse_only, nullptr, s);
}

extern "C" int trx_affine_accumulate(const trx_volumes *vol, const float *theta, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_vol(vol, true);
    if (rc) return rc;
    if (!theta || !workspace) return TRX_ERR_ARG;
    if (workspace_bytes < trx_affine_workspace_bytes(vol)) return TRX_ERR_WORKSPACE;
    int nblk = 0;
    const int *rows_used = nullptr;
    return launch_f1<0>(vol, theta, (float *)workspace, &nblk, (hipStream_t)stream, true, &rows_used);   // exactly the launch of a step (profiling aid)
}

// The finalise kernel behind a step launch (trx_affine_step, and the flush behind the last carry launch of trx_affine_run: state_in = its carry buffer)
static int launch_finalize(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt, const trx_affine_state *st, const float *partials, int nblk,
                           const int *rows_used, bool mse_only, const float *state_in, hipStream_t s)
{
    const double nvox = (double)vol->D * vol->H * vol->W;
    if (vol->ndim == 3)
        hipLaunchKernelGGL((affine_finalize_kernel<3>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, nblk, nvox,
                           vol->D, vol->H, vol->W, *loss, *opt, *st, rows_used, mse_only ? 1 : 0, state_in);
    else
        hipLaunchKernelGGL((affine_finalize_kernel<2>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, nblk, nvox,
                           vol->D, vol->H, vol->W, *loss, *opt, *st);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_affine_run(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                              const trx_affine_state *st, int iters, void *workspace, size_t workspace_bytes, void *stream)
{
    if (iters < 0 || !st) return TRX_ERR_ARG;
    if (st->losses && iters > st->losses_capacity) return TRX_ERR_CAPACITY;
    if (!vol) return TRX_ERR_ARG;
    trx_volumes v = *vol;
    int first = 0;
    // CARRY (CarryKArgs): launch-bound 3-D steps - one (GeomA / GeomR) kernel and a finalise kernel behind it - run as ONE launch per iteration: the finalise
    // of iteration k rides in the prologue of iteration k + 1's kernel, and one finalise kernel behind the last launch flushes the run.  The launches alternate
    // between the workspace's two partial / note / carry buffers; the LAST one writes the primary set (what trx_affine_workspace_rows_offset describes).
    if (iters >= 2 && TRX_CARRY && !(vol->flags & TRX_FLAG_NO_CARRY) && check_vol(vol, true) == TRX_OK && vol->ndim == 3 && use_tile_path(vol) && loss && opt && workspace &&
        st->param && st->theta && st->best_theta && st->best_loss && st->best_idx && st->step && (opt->kind == TRX_OPT_SGD || (opt->kind == TRX_OPT_ADAM && st->adam_m && st->adam_v)) &&
        (st->mode == TRX_PARAM_AFFINE || st->mode == TRX_PARAM_RIGID) && workspace_bytes >= trx_affine_workspace_bytes(vol)) {
        hipStream_t s = (hipStream_t)stream;
        const AffineWs ws = affine_ws(*vol);
        const bool mse_only = loss->w_ncc == 0.f;
        float *const base = (float *)workspace;
        float *const carry_buf = (float *)((char *)workspace + ws.off_carry);
        const float *prev_partials = nullptr;
        const int *prev_notes = nullptr;
        int prev_nblk = 0;
        for (int k = 0; k < iters; k++) {
            CarryLaunch cl;
            cl.parity = (iters - 1 - k) & 1;
            cl.used = false;
            cl.partials_out = nullptr;
            cl.args = CarryKArgs{prev_partials, prev_notes, prev_nblk, mse_only ? 1 : 0, k > 0 ? carry_buf + (size_t)(cl.parity ^ 1) * vol->B * kCarryStride : nullptr,
                                 carry_buf + (size_t)cl.parity * vol->B * kCarryStride, (double)vol->D * vol->H * vol->W, *loss, *opt, *st};
            v.flags = vol->flags;
            if ((k & 1) && !(vol->flags & TRX_FLAG_NO_PINGPONG)) v.flags ^= TRX_FLAG_WALK_DOWN;
            int nblk = 0;
            const int *notes = nullptr;
            int rc = mse_only ? launch_f1<4>(&v, st->theta, base, &nblk, s, true, &notes, &cl) : launch_f1<0>(&v, st->theta, base, &nblk, s, true, &notes, &cl);
            if (rc) return rc;
            if (!cl.used) {
                // not a launch the carry form exists for (only possible at k = 0: the choice depends on sizes and flags alone): the launch above was an
                // ordinary step launch - finish that step and run the rest the ordinary way
                rc = launch_finalize(vol, loss, opt, st, cl.partials_out, nblk, notes, mse_only, nullptr, s);
                if (rc) return rc;
                first = 1;
                prev_partials = nullptr;
                break;
            }
            prev_partials = cl.partials_out; prev_notes = notes; prev_nblk = nblk;
        }
        if (prev_partials != nullptr)   // every launch was a carry launch: flush the last iteration (its buffers are the primary set: parity 0)
            return launch_finalize(vol, loss, opt, st, prev_partials, prev_nblk, prev_notes, mse_only, carry_buf, s);
    }
    for (int i = first; i < iters; i++) {
        // odd iterations walk the z-streaming columns downward: the tail of one pass is the head of the next (TRX_FLAG_WALK_DOWN)
        v.flags = vol->flags;
        if ((i & 1) && !(vol->flags & TRX_FLAG_NO_PINGPONG)) v.flags ^= TRX_FLAG_WALK_DOWN;
        int rc = trx_affine_step(&v, loss, opt, st, workspace, workspace_bytes, stream);
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
    float *partials = (float *)workspace;
    int nblk = 0;
    rc = launch_f1<1>(vol, theta, partials, &nblk, s, true);
    if (rc) return rc;
    const double nvox = (double)vol->D * vol->H * vol->W;
    hipLaunchKernelGGL(affine_loss_finalize_kernel, dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, nblk, nvox, *loss, terms);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_affine_warp(const trx_volumes *vol, const float *theta, int channels, float *out, void *stream)
{
    int rc = check_vol(vol, false);
    if (rc) return rc;
    if (!theta || !out || channels < 1) return TRX_ERR_ARG;
    if ((long)vol->B * channels > 65535) return TRX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t nvox = (size_t)vol->D * vol->H * vol->W;
    if (use_tile_path(vol) && vol->xn && vol->yn && vol->zn) {
        // LDS-tiled forward warp (same staging as the F1 pass); needs the caller's coordinate tables
        trx_volumes v = *vol;
        v.B = vol->B * channels;                       // geometry: every (pair, channel) is one slab of blocks
        TileGeom t = tile_geom(v);
        if (use_dual(vol)) {
            const TileGeom ta = tile_geom<GeomA>(v), tr = tile_geom<GeomR>(v);
            v.B = vol->B;
            const int gx = ta.blocks_per_pair > tr.blocks_per_pair ? ta.blocks_per_pair : tr.blocks_per_pair;
            launch_dual<3>(dim3(gx, vol->B * channels), s, v, theta, ta, tr, channels, out, use_dual(vol));
            TRX_CHECK_LAUNCH();
            return TRX_OK;
        }
        v.B = vol->B;
        hipLaunchKernelGGL((affine_tile_kernel<3>), dim3(t.blocks_per_pair, vol->B * channels), dim3(kTileThreads), 0, s, v, theta, t, channels, out);
        TRX_CHECK_LAUNCH();
        return TRX_OK;
    }
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
    if (use_tile_path(vol) && vol->xn && vol->yn && vol->zn && channels == 1) {
        // LDS-tiled kernel (same staging as the F1 pass); one partial row per block, 12 sums each.  (Several channels or
        // caller-less tables: the row-walking gather kernel below; the workspace is sized for single-channel tiles.)
        TileGeom t = tile_geom(*vol);
        int rows = t.blocks_per_pair;
        if (use_dual(vol)) {
            const TileGeom ta = tile_geom<GeomA>(*vol), tr = tile_geom<GeomR>(*vol);
            rows = ta.blocks_per_pair > tr.blocks_per_pair ? ta.blocks_per_pair : tr.blocks_per_pair;
            launch_dual<2>(dim3(rows, vol->B), s, v, theta, ta, tr, 1, partials, use_dual(vol));
        } else {
            hipLaunchKernelGGL((affine_tile_kernel<2>), dim3(t.blocks_per_pair, vol->B), dim3(kTileThreads), 0, s, v, theta, t, 1, partials);
        }
        TRX_CHECK_LAUNCH();
        hipLaunchKernelGGL((affine_bwd_finalize_kernel<3>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, rows, vol->D, vol->H,
                           vol->W, dtheta);
        TRX_CHECK_LAUNCH();
        return TRX_OK;
    }
    rc = launch_accum<2>(&v, theta, g, channels, nvox, partials, s);
    if (rc) return rc;
    if (vol->ndim == 3)
        hipLaunchKernelGGL((affine_bwd_finalize_kernel<3>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, g.nblk, vol->D, vol->H, vol->W, dtheta);
    else
        hipLaunchKernelGGL((affine_bwd_finalize_kernel<2>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, g.nblk, vol->D, vol->H, vol->W, dtheta);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

static int check_lattice(const trx_volumes *vol, const int *iz, int nz, const int *iy, int ny, const int *ix, int nx)
{
    if (!iy || !ix || ny < 1 || nx < 1) return TRX_ERR_ARG;
    if (vol->ndim == 3 ? (!iz || nz < 1) : (nz != 1)) return TRX_ERR_ARG;
    if ((double)nz * ny * nx >= 2147483648.0) return TRX_ERR_ARG;
    return TRX_OK;
}

static int lattice_blocks(size_t n)   // 4 lattice points per thread and trip, at most 1024 blocks per pair (one partial row each: the finalise reads them all)
{
    size_t nb = (n + (size_t)TRX_BLOCK * 4 - 1) / ((size_t)TRX_BLOCK * 4);
    return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}

static unsigned lattice_fwd_blocks(size_t n)
{
    size_t nb = (n + (size_t)TRX_BLOCK * 4 - 1) / ((size_t)TRX_BLOCK * 4);
    return (unsigned)(nb > 8192 ? 8192 : nb);
}

static int launch_lattice_fwd(const trx_volumes *vol, const float *theta, const LatticeIdx &L, float *out, float *block_minmax, hipStream_t s)
{
    dim3 grid(lattice_fwd_blocks((size_t)L.nz * L.ny * L.nx), vol->B), block(TRX_BLOCK);
    if (vol->ndim == 3) hipLaunchKernelGGL((affine_warp_lattice_kernel<3>), grid, block, 0, s, *vol, theta, L, out, block_minmax);
    else hipLaunchKernelGGL((affine_warp_lattice_kernel<2>), grid, block, 0, s, *vol, theta, L, out, block_minmax);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_affine_warp_lattice(const trx_volumes *vol, const float *theta, const int *iz, int nz, const int *iy, int ny, const int *ix,
                                       int nx, float *out, void *stream)
{
    int rc = check_vol(vol, false);
    if (rc) return rc;
    if (!theta || !out) return TRX_ERR_ARG;
    if ((rc = check_lattice(vol, iz, nz, iy, ny, ix, nx)) != TRX_OK) return rc;
    const LatticeIdx L = {iz, iy, ix, nz, ny, nx};
    return launch_lattice_fwd(vol, theta, L, out, nullptr, (hipStream_t)stream);
}

extern "C" int trx_nmi_lattice_lines(const trx_volumes *vol, const float *theta, const int *iz, int nz, const int *iy, int ny, const int *ix, int nx,
                                     float *out, const float *minmax_target, int patches, int bins, float *xis, float *minmax_warped, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    int rc = check_vol(vol, false);
    if (rc) return rc;
    if (!theta || !out || !minmax_target || !xis || !workspace || patches < 1 || bins < 2 || bins > 1024) return TRX_ERR_ARG;
    if ((rc = check_lattice(vol, iz, nz, iy, ny, ix, nx)) != TRX_OK) return rc;
    const LatticeIdx L = {iz, iy, ix, nz, ny, nx};
    const unsigned nblk = lattice_fwd_blocks((size_t)nz * ny * nx);
    if (workspace_bytes < (size_t)vol->B * nblk * 2 * sizeof(float)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = launch_lattice_fwd(vol, theta, L, out, (float *)workspace, s)) != TRX_OK) return rc;
    hipLaunchKernelGGL(nmi_lines_kernel, dim3(vol->B), dim3(1024), 0, s, (const float *)workspace, (int)nblk, minmax_target, patches, bins, xis, minmax_warped);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_nmi_loop_update(int ndim, float *theta, float *pose, const float *grad_a, const float *grad_b, float lr, const float *loss_terms,
                                   int n_terms, const float *loss_b, float *hist_loss_t, float *hist_theta_t, float *param_copy, void *stream)
{
    if ((ndim != 2 && ndim != 3)) return TRX_ERR_NDIM;
    if (!theta || !grad_a || !loss_terms || n_terms < 1 || !hist_loss_t || !hist_theta_t) return TRX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (ndim == 3) hipLaunchKernelGGL((nmi_loop_update_kernel<3>), dim3(1), dim3(64), 0, s, theta, pose, grad_a, grad_b, lr, loss_terms, n_terms, loss_b,
                                      hist_loss_t, hist_theta_t, param_copy);
    else hipLaunchKernelGGL((nmi_loop_update_kernel<2>), dim3(1), dim3(64), 0, s, theta, pose, grad_a, grad_b, lr, loss_terms, n_terms, loss_b, hist_loss_t,
                            hist_theta_t, param_copy);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_affine_warp_lattice_backward(const trx_volumes *vol, const float *theta, const int *iz, int nz, const int *iy, int ny,
                                                const int *ix, int nx, const float *grad_out, float *dtheta, void *workspace,
                                                size_t workspace_bytes, void *stream)
{
    int rc = check_vol(vol, false);
    if (rc) return rc;
    if (!theta || !grad_out || !dtheta || !workspace) return TRX_ERR_ARG;
    if ((rc = check_lattice(vol, iz, nz, iy, ny, ix, nx)) != TRX_OK) return rc;
    const int nblk = lattice_blocks((size_t)nz * ny * nx);
    if (workspace_bytes < trx_affine_workspace_bytes(vol) || workspace_bytes < (size_t)vol->B * nblk * 12 * sizeof(float)) return TRX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float *partials = (float *)workspace;
    const LatticeIdx L = {iz, iy, ix, nz, ny, nx};
    dim3 grid((unsigned)nblk, vol->B), block(TRX_BLOCK);
    if (vol->ndim == 3) {
        hipLaunchKernelGGL((affine_lattice_bwd_kernel<3>), grid, block, 0, s, *vol, theta, L, grad_out, partials);
        TRX_CHECK_LAUNCH();
        hipLaunchKernelGGL((affine_bwd_finalize_kernel<3>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, nblk, vol->D, vol->H, vol->W, dtheta);
    } else {
        hipLaunchKernelGGL((affine_lattice_bwd_kernel<2>), grid, block, 0, s, *vol, theta, L, grad_out, partials);
        TRX_CHECK_LAUNCH();
        hipLaunchKernelGGL((affine_bwd_finalize_kernel<2>), dim3(vol->B), dim3(TRX_FIN_THREADS), 0, s, partials, nblk, vol->D, vol->H, vol->W, dtheta);
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_theta_chain(const float *pose, const float *dtheta, int ndim, int B, float *theta_out, float *dpose_out, void *stream)
{
    if (!pose || B < 1 || (!theta_out && !(dtheta && dpose_out))) return TRX_ERR_ARG;
    if (ndim != 2 && ndim != 3) return TRX_ERR_NDIM;
    hipStream_t s = (hipStream_t)stream;
    if (ndim == 3) hipLaunchKernelGGL((theta_chain_kernel<3>), dim3(B), dim3(64), 0, s, pose, dtheta, theta_out, dpose_out);
    else hipLaunchKernelGGL((theta_chain_kernel<2>), dim3(B), dim3(64), 0, s, pose, dtheta, theta_out, dpose_out);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}
