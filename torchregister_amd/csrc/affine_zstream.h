// Z-STREAMING variant of the fused F1 pass (3-D) for transforms near the identity.  Included by affine.hip inside namespace trx.
//
// Why a second kernel family: the tile kernels (tile_body) stage the whole pre-image box of a tile, wait for it, gather, and start
// over; requests are in flight only while a block is in its burst phase, and every tile re-fetches its z / y halo.  Here a 512-thread
// block owns an in-plane tile (TX x TY voxels) and walks it along z, one output plane per step:
//   * the source planes live in an LDS RING of NZ slots (slot = source plane mod NZ), filled by LDS-DMA ahead of the gather: a source
//     plane is fetched once per block (no z halo) and requests are in flight all the time;
//   * one barrier and one s_waitcnt per step; the DMA of a plane is 16 pieces, two per wave, with exec masks and per-lane offsets
//     computed once per anchor (the window origin is fixed for `sublen` planes), so a step's loader work is a few scalar instructions;
//   * NZ need not be a power of two: the slot of floor(iz) comes from an 8-entry byte table in two SGPRs through v_perm_b32 - the
//     instruction count of a mask - which buys 40 rows of window where a power-of-two ring leaves 35;
//   * thread (x, rows r0 .. r0+R-1) is fixed for the whole block: sum(q grad) and sum(q grad yn) accumulate as in the tile kernel,
//     the zn column comes from U = sum over planes of the running sum (sum_z zn_z P_z = zn_0 S + d (n S - U), zn uniform in z).
// A pair whose theta does not keep the pre-image inside the window (zs_nsub, a function of theta and the shape only) is left to the
// tile kernels.  Per-voxel arithmetic is that of tile_body's fast loop: coordinates are ATen's identity coordinate + deviation.
// Measured (8 x 256^3, MI355X): 253-265 us per launch against 300-313 us for the tile kernels on the same box; L2 requests 12.3 M
// (18.0 M), HBM-side traffic 1.03 x the algorithmic bytes (1.38 x); barriers and waits are not binding (deleting them changes nothing):
// the pass is bound by its vector instructions at the clock the chip holds under this load.

#ifndef TRX_ZS_DBG
#define TRX_ZS_DBG 0   // development ablation (tools/zbench.hip): bits: 1 = no ring DMA, 2 = no target loads, 4 = no gather, 8 = no barrier (racy), 16 = no counted wait (racy), 32 = no LDS reads (fake corners), 64 = no accumulation of the pose sums, 128 = the lower plane's two pairs are copies of the upper plane's (4 LDS reads per voxel instead of 8: the upper bound of carrying the upper plane's pairs into the next step; timing / power only)
#endif
#ifndef TRX_ZS_MIN_WAVES
#define TRX_ZS_MIN_WAVES 4
#endif
#ifndef TRX_ZS_V2
#define TRX_ZS_V2 1   // step kernels: row coordinates and yn products on SGPR PAIRS (one packed instruction per two rows / two components), accumulators
                      // as (x, y) pairs - an instruction with a scalar-register operand costs a SIMD 4.3 cycles where an all-VGPR one costs 2.5
                      // (profiles/r05a_mfma_coissue_and_op_costs.txt), so the scalar operand should serve two results; 0 = the round-3 form
#endif
#ifndef TRX_ZS_PRIO
#define TRX_ZS_PRIO 2   // Fair sharing of a CU between its two resident blocks.  The SIMD arbiter serves the OLDER wave first: of the two blocks of a
                        // CU the one dispatched first ran its 128 steps in 172 us, the other one needed 262 us and spent the last 90 us alone on the
                        // CU at 2 waves per SIMD (profiles/r05a_zstream_block_timeline.txt).  2 = the block of the launch's second half of the grid
                        // raises its priority in alternate time slices of 2^TRX_ZS_PRIO_BIT shader cycles (s_memtime), the other one in the
                        // slices between: end skew 112 -> 44 us, launch -1.5 ... -2 %; 1 = by step parity (measured alternative); 0 = off
#endif
#ifndef TRX_ZS_PRIO_BIT
#define TRX_ZS_PRIO_BIT 14
#endif
#ifndef TRX_ZS_STAMP
#define TRX_ZS_STAMP 0   // development (tools/zbench.hip): per-wave s_memtime sums of the four phases of a step -> trx_zs_stamps
#endif
#if TRX_ZS_STAMP
__device__ unsigned long long trx_zs_stamps[512 * 8 * 8];   // [block][wave][wait, barrier, issue, gather, total ticks, steps, realtime start, realtime end]
#endif
#ifndef TRX_ZS_LEAD
#define TRX_ZS_LEAD 1   // steps between the issue of a ring plane and the first step that may touch it (1: one more resident plane, Span = NZ - 1)
#endif

// Running sums of the z-streaming step body (TRX_ZS_V2): for the weightings q = 1, y, w: Axy = (sum q gx, sum q gy), Bxy = the same times yn,
// ABz = (sum q gz, sum q gz yn); M01 = (Sy, Sw), M23 = (Syy, Sww), M4 = Syw.
struct ZAcc {
    f2 Axy[3], Bxy[3], ABz[3], M01, M23;
    float M4;
};
__device__ __forceinline__ unsigned long long sgpr_pair(float lo, float hi)   // two wave-uniform floats as one 64-bit scalar operand
{
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(lo)), h = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(hi));
    return ((unsigned long long)h << 32) | l;
}
// Trilinear sample + gradient from the four x-pairs, results where the accumulation wants them: yw = (target value, warped value),
// gxy = (d/dx, d/dy), gz.  The three x-stage results are written by NON-destructive v_fma_f32 straight into the halves of their pairs (the
// compiler's v_fmac_f32 accumulates in place and then needs a v_mov per pair).
__device__ __forceinline__ void zs_lerp_accumulate(f2 r00, f2 r01, f2 r10, f2 r11, float tx, float ty, float tz, float yv, float yn, unsigned long long yn2, ZAcc &a)
{
    const f2 dz0 = r10 - r00, dz1 = r11 - r01;
    const f2 z0 = dz0 * tz + r00, z1 = dz1 * tz + r01;
    const f2 dy = z1 - z0;
    const f2 vv = dy * ty + z0;
    const f2 dzy = (dz1 - dz0) * ty + dz0;
    f2 yw, gxy, gz2;
    yw.x = yv;
    gxy.x = vv.y - vv.x;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(yw.y) : "v"(tx), "v"(gxy.x), "v"(vv.x));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(gxy.y) : "v"(tx), "v"(dy.y - dy.x), "v"(dy.x));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(gz2.x) : "v"(tx), "v"(dzy.y - dzy.x), "v"(dzy.x));
    gz2.y = yn * gz2.x;
    f2 ygxy;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(ygxy) : "v"(gxy), "s"(yn2));
    a.M01 += yw;
    a.M23 = yw * yw + a.M23;
    a.M4 = fmaf(yw.x, yw.y, a.M4);
    a.Axy[0] += gxy; a.Bxy[0] += ygxy; a.ABz[0] += gz2;
    a.Axy[1] = gxy * yw.x + a.Axy[1]; a.Bxy[1] = ygxy * yw.x + a.Bxy[1]; a.ABz[1] = gz2 * yw.x + a.ABz[1];
    a.Axy[2] = gxy * yw.y + a.Axy[2]; a.Bxy[2] = ygxy * yw.y + a.Bxy[2]; a.ABz[2] = gz2 * yw.y + a.ABz[2];
}

template <int TX_, int TY_, int NZ_, int BW_, int BH_>
struct ZCfg {
    static constexpr int TX = TX_, TY = TY_, NZ = NZ_, BW = BW_, BH = BH_;
    static constexpr int Threads = 512, Waves = 8;
    static constexpr int XW = TX / 64, RG = Waves / XW, R = TY / RG;   // x wave-groups, row groups, rows per thread and plane
    static constexpr int BW4 = BW / 4, PlaneSlots = BH * BW4;          // float4 slots of one ring plane
    static constexpr int LPP = (PlaneSlots + 15) / 16;                 // lanes per DMA piece: 16 pieces per plane, two per wave
    static constexpr int PlaneFloats = BW * BH, PlaneBytes = PlaneFloats * 4;
    static constexpr int RingFloats = NZ * PlaneFloats;
    static constexpr int Span = NZ - TRX_ZS_LEAD;                                // source planes a step may touch: [p - Span + 1, p], p = pbase + step
    static constexpr int ReduceScratch = Waves * 16 * 65 + Waves * 16;
    static constexpr int Alloc = (RingFloats + 4 > ReduceScratch) ? RingFloats + 4 : ReduceScratch;   // + one float4 the dummy DMAs write
    static_assert(TX % 64 == 0 && Waves % XW == 0 && TY % RG == 0 && BW % 4 == 0 && LPP <= 64 && NZ >= 4 && NZ <= 8 && Span <= 7, "geometry");   // (the slot tables hold eight planes: floor(z) - zlo <= Span - 1 and its upper neighbour <= 7)
};
// 64 x 32 voxels per plane, ring of 7 planes of 72 x 40 floats = 80.6 KB: two blocks per CU.  The ring size is not a power of two: the
// slot of a source plane comes from an 8-entry byte table (v_perm_b32), at the instruction count of a mask.
#ifndef TRX_ZS_GEOM
#define TRX_ZS_GEOM 0   // 1: ring of 6 planes of 80 x 42 floats - the same 80.6 KB with 9.7 / 5.4 instead of 1.7 / 3.4 voxels of slack in x / y, one plane less in z:
                        // measured alternative (profiles/r05a_zs_window_variants.txt: R_z(0.1) joins the window, the identity loses 4 %, a converging run 12 %)
#endif
// development (tools/zbench.hip): a FLAT tile - 64 x 16 voxels per plane under a ring of 8 planes of 80 x 30 floats (76.8 KB): 9.7 / 10.7 voxels of slack in
// x / y and 3.7 planes of tilt, for poses of the convergence basin that the 64 x 32 tile leaves to the deep tile kernel
using ZSF = ZCfg<64, 16, 8, 80, 30>;
#if TRX_ZS_GEOM == 1
using ZS64 = ZCfg<64, 32, 6, 80, 42>;
#else
using ZS64 = ZCfg<64, 32, 7, 72, 40>;
#endif

struct ZGeom {
    int ntx, nty, nzseg, planes_per_seg, blocks_per_pair;
};

template <class C>
static ZGeom zs_geom(const trx_volumes &v)
{
    ZGeom g;
    g.ntx = v.W / C::TX; g.nty = v.H / C::TY;
    const long cols = (long)v.B * g.ntx * g.nty;
    int nseg = cols >= 512 ? 1 : (int)((512 + cols - 1) / cols);
    const int cap = v.D / 32 > 1 ? v.D / 32 : 1;   // a segment pays ~6 planes of pipeline fill
    if (nseg > cap) nseg = cap;
    g.planes_per_seg = (v.D + nseg - 1) / nseg;
    g.nzseg = (v.D + g.planes_per_seg - 1) / g.planes_per_seg;
    g.blocks_per_pair = g.ntx * g.nty * g.nzseg;
    return g;
}

template <class C>
static bool zs_shape_ok(const trx_volumes &v)
{
    if (v.ndim != 3 || v.W % C::TX || v.H % C::TY || v.D < 8) return false;
    return (size_t)v.D * v.H * v.W < ((size_t)1 << 29);   // 32-bit byte offsets inside one volume
}

// Does the pre-image of EVERY block of a pair stay inside its window when a block re-anchors the window every `len` planes?  theta and
// sizes only (the map is affine: extents do not depend on the position), so every block of the pair, the launcher's surplus test and the
// body agree; zstream_body's own per-anchor test (exact, with the actual corners) is implied by this one - the bounds below are its
// worst case over the alignment of the origin.
template <class C>
__host__ __device__ __forceinline__ bool zs_fits_len(const float *__restrict__ th, float fD, float fH, float fW, int len)
{
    const float ex = (float)(C::TX - 1), ey = (float)(C::TY - 1), ez = (float)(len - 1);
    const float sx = fabsf(th[0]) * ex + fabsf(th[1] * fW / fH) * ey + fabsf(th[2] * fW / fD) * ez;
    const float sy = fabsf(th[4] * fH / fW) * ex + fabsf(th[5]) * ey + fabsf(th[6] * fH / fD) * ez;
    const float sz = fabsf(th[8] * fD / fW) * ex + fabsf(th[9] * fD / fH) * ey + fabsf(th[10] - 1.0f) * ez;
    // x: hx - ox <= span + 2 slack + 5 (floor, +1 neighbour, 3 of alignment);  y: + 2;  z: span + drift <= Span - 3 - 2 slack
    return (sx <= (float)C::BW - 6.3f) && (sy <= (float)C::BH - 3.3f) && (sz <= (float)C::Span - 3.3f);   // NaN compares false
}
// Number of sub-segments (1, 2, 4 or 8) a block of `planes_per_seg` planes re-anchors its window in; 0: the pair does not fit.
template <class C>
__host__ __device__ __forceinline__ int zs_nsub(const float *__restrict__ th, float fD, float fH, float fW, int planes_per_seg)
{
    if (zs_fits_len<C>(th, fD, fH, fW, planes_per_seg)) return 1;
    if (planes_per_seg >= 64 && zs_fits_len<C>(th, fD, fH, fW, (planes_per_seg + 1) / 2)) return 2;
    if (planes_per_seg >= 128 && zs_fits_len<C>(th, fD, fH, fW, (planes_per_seg + 3) / 4)) return 4;
    if (planes_per_seg >= 128 && zs_fits_len<C>(th, fD, fH, fW, (planes_per_seg + 7) / 8)) return 8;
    return 0;
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <int MODE, class C>
__device__ __forceinline__ void zstream_body(const trx_volumes &vol, const float *__restrict__ theta, const ZGeom &zg, float *__restrict__ partials,
                                             float *ring, const int bx, const int by, int rows_per_pair, const int wave_in, const int second_slot = 0)
{
    const int walk_down = __builtin_amdgcn_readfirstlane((int)(vol.flags & TRX_FLAG_WALK_DOWN));
    // walk_down (TRX_FLAG_WALK_DOWN; wave-uniform): the column is walked from its last plane to its first.  Nothing else changes - the sums are
    // order-independent up to rounding - but a registration that alternates the direction from iteration to iteration finds the planes it
    // read LAST in the 256 MiB Infinity Cache when it starts the next pass with them (profiles/r05a_power_cap.txt: a quarter of a 1 GiB pass,
    // at 40 % of the energy of an HBM read).
    const int dir = walk_down ? -1 : 1;
    const bool down = walk_down != 0;
    static_assert(MODE == 0 || MODE == 1 || MODE == 4, "step kernels and the moments pass");
    constexpr int NQ = (MODE == 0) ? 3 : (MODE == 4 ? 1 : 0);
    constexpr int NP = (MODE == 0) ? np_full(3) : (MODE == 4 ? kNpMse : 5);
    constexpr bool kGrad = MODE != 1;
    constexpr int R = C::R;
    const int b = by;
    const int D = vol.D, H = vol.H, W = vol.W;
    const float *__restrict__ th = uni_ptr(theta + (size_t)b * TRX_PSTRIDE);
    const float *__restrict__ mov = uni_ptr(vol.moving + (size_t)b * vol.moving_stride);
    const float *__restrict__ tgt = uni_ptr(vol.target + (size_t)b * vol.target_stride);
    const float *__restrict__ xtab = uni_ptr(vol.xn), *__restrict__ ytab = uni_ptr(vol.yn), *__restrict__ ztab = uni_ptr(vol.zn);
    const int lane = trx_lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);
    const int tid = wave * 64 + lane;
    const float fW = (float)W, fH = (float)H, fD = (float)D;
    const float hW = 0.5f * fW, hH = 0.5f * fH, hD = 0.5f * fD;
    const float t00 = th[0], t01 = th[1], t02 = th[2], t03 = th[3];
    const float t10 = th[4], t11 = th[5], t12 = th[6], t13 = th[7];
    const float t20 = th[8], t21 = th[9], t22 = th[10], t23 = th[11];

    // block -> (column, z segment).  Blocks b, b + 8, ... share an XCD: give each XCD a contiguous run of logical ids, i.e. a patch
    // of neighbouring columns of one z segment, so that the x / y halo of a column is in its XCD's L2.
    const int ncol = zg.ntx * zg.nty, nblk = zg.blocks_per_pair;
    int lb = bx;
    if ((nblk & 7) == 0) lb = (bx & 7) * (nblk >> 3) + (bx >> 3);
    const int zseg = lb / ncol, col = lb - zseg * ncol;
    const int X0 = (col % zg.ntx) * C::TX, Y0 = (col / zg.ntx) * C::TY;
    const int zb0 = zseg * zg.planes_per_seg, ze0 = min(zb0 + zg.planes_per_seg, D);
    // the window is re-anchored every `sublen` planes (1, 2, 4 or 8 sub-segments, from theta: the same rule for every block of the pair)
    const int nsub = max(1, __builtin_amdgcn_readfirstlane(zs_nsub<C>(th, fD, fH, fW, zg.planes_per_seg)));
    const int sublen = (zg.planes_per_seg + nsub - 1) / nsub;

    const int xw = wave % C::XW, rg = wave / C::XW;
    const int x = X0 + xw * 64 + lane, yrow0 = Y0 + rg * R;
    const float xn = xtab[x];
    // sample coordinates of voxel (x, row j, plane z):  i_c = (c_c + k_cz zn_z [+ zid_z]) + row constant_j
    const float cx = unnorm<3>(xn, fW) + hW * fmaf(t00 - 1.0f, xn, t03);
    const float cy = hH * fmaf(t10, xn, t13);
    const float cz = hD * fmaf(t20, xn, t23);
    const float kxz = uni(hW * t02), kyz = uni(hH * t12), kzz = uni(hD * (t22 - 1.0f));
    float yn_r[R], sx_r[R], ey_r[R], sz_r[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        const float yn = ytab[yrow0 + j];
        yn_r[j] = uni(yn);
        sx_r[j] = uni((hW * t01) * yn);
        ey_r[j] = uni(unnorm<3>(yn, fH) + (hH * (t11 - 1.0f)) * yn);
        sz_r[j] = uni((hD * t21) * yn);
    }
    constexpr bool kV2 = (TRX_ZS_V2 != 0) && MODE == 0 && (R % 2 == 0);
    unsigned long long sx2[(R + 1) / 2], ey2[(R + 1) / 2], sz2[(R + 1) / 2], yn2[R];   // scalar-register pairs: rows (2k, 2k + 1) of sx / ey / sz, (yn, yn) per row
    if constexpr (kV2) {
#pragma unroll
        for (int k = 0; k < R / 2; k++) {
            sx2[k] = sgpr_pair(sx_r[2 * k], sx_r[2 * k + 1]);
            ey2[k] = sgpr_pair(ey_r[2 * k], ey_r[2 * k + 1]);
            sz2[k] = sgpr_pair(sz_r[2 * k], sz_r[2 * k + 1]);
        }
#pragma unroll
        for (int j = 0; j < R; j++) yn2[j] = sgpr_pair(yn_r[j], yn_r[j]);
    }
    auto coord = [&](int xi, int yi, int zi, float &ix, float &iy, float &iz) {
        const float a = xtab[xi], bb = ytab[yi], c = ztab[zi];
        ix = unnorm<3>(a, fW) + hW * fmaf(t00 - 1.0f, a, fmaf(t01, bb, fmaf(t02, c, t03)));
        iy = unnorm<3>(bb, fH) + hH * fmaf(t10, a, fmaf(t11 - 1.0f, bb, fmaf(t12, c, t13)));
        iz = unnorm<3>(c, fD) + hD * fmaf(t20, a, fmaf(t21, bb, fmaf(t22 - 1.0f, c, t23)));
    };

    const unsigned ring_lds = (unsigned)(uintptr_t)ring;
    const unsigned dummy_lds = ring_lds + C::RingFloats * 4u;
    const unsigned pl0 = ring_lds + (unsigned)(wave * C::LPP * 16), pl1 = ring_lds + (unsigned)((wave + 8) * C::LPP * 16);
    const size_t plane_bytes = (size_t)H * W * 4;
    const unsigned toffb = (unsigned)((yrow0 * W) + x) * 4u;
    int ys_s, ps_s;   // LDS strides (bytes) pinned in SGPRs
    asm("s_mov_b32 %0, %1" : "=s"(ys_s) : "i"(C::BW * 4));
    asm("s_mov_b32 %0, %1" : "=s"(ps_s) : "i"(C::PlaneBytes));
    typedef const __attribute__((address_space(3))) f2u *lds_f2;

    F1Acc acc;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
    acc.M01 = acc.M23 = (f2)(0.f);
    acc.M4 = 0.f;
    ZAcc acc2;
    f2 Uxy[3];
    float Uz[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        acc2.Axy[q] = acc2.Bxy[q] = acc2.ABz[q] = Uxy[q] = (f2)(0.f);
        Uz[q] = 0.f;
    }
    acc2.M01 = acc2.M23 = (f2)(0.f);
    acc2.M4 = 0.f;
    float U[3][3];   // sum over planes of the running sum(q grad): the zn column
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) U[q][c] = 0.f;
    bool ok = true;
#if TRX_ZS_STAMP
    unsigned long long stamp_sum[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    for (int si = 0; si < nsub; si++) {
        const int sub = down ? nsub - 1 - si : si;
        const int zb = zb0 + sub * sublen, ze = min(zb + sublen, ze0);
        if (zb >= ze) continue;
        const int nsteps = ze - zb, last = nsteps - 1;
        const int zf = down ? ze - 1 : zb, zl = down ? zb : ze - 1;   // the planes of the first and of the last step
        // ---- window of this anchor: x / y origin fixed, source plane p(s) = pbase + s is the highest one step s may touch
        float mnx = 1e30f, mxx = -1e30f, mny = 1e30f, mxy = -1e30f, mnz0 = 1e30f, mxz0 = -1e30f, mnz1 = 1e30f, mxz1 = -1e30f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float ix, iy, iz;
            coord(X0 + ((k & 1) ? C::TX - 1 : 0), Y0 + ((k & 2) ? C::TY - 1 : 0), (k & 4) ? zl : zf, ix, iy, iz);
            mnx = fminf(mnx, ix); mxx = fmaxf(mxx, ix); mny = fminf(mny, iy); mxy = fmaxf(mxy, iy);
            if (k & 4) { mnz1 = fminf(mnz1, iz); mxz1 = fmaxf(mxz1, iz); } else { mnz0 = fminf(mnz0, iz); mxz0 = fmaxf(mxz0, iz); }
        }
        const float slack = 0.05f;
        bool fits = (fabsf(mnx) < 1.0e6f) && (fabsf(mxx) < 1.0e6f) && (fabsf(mny) < 1.0e6f) && (fabsf(mxy) < 1.0e6f) && (fabsf(mnz0) < 1.0e6f) &&
                    (fabsf(mxz0) < 1.0e6f) && (fabsf(mnz1) < 1.0e6f) && (fabsf(mxz1) < 1.0e6f);   // also rejects NaN
        int ox = 0, oy = 0, pbase = 0, need_rows = 0, need_c4 = 0;   // need_*: the part of the window this anchor can touch (the rest is not fetched)
        if (fits) {
            ox = (int)floorf(mnx - slack) & ~3;
            oy = (int)floorf(mny - slack);
            const int hx = (int)floorf(mxx + slack) + 1, hy = (int)floorf(mxy + slack) + 1;
            need_rows = hy - oy; need_c4 = (hx - ox) >> 2;
            const int hi0 = (int)floorf(mxz0 + slack) + 1, hi1 = (int)floorf(mxz1 + slack) + 1;
            const int lo0 = (int)floorf(mnz0 - slack), lo1 = (int)floorf(mnz1 - slack);
            // pbase = the LEADING plane of the window at step 0: walking up the highest plane a step may touch (window [pbase + s - (Span - 1),
            // pbase + s]), walking down the lowest one (window [pbase - s, pbase - s + (Span - 1)])
            if (!down) {
                pbase = max(hi0, hi1 - last);
                fits = (hx <= ox + C::BW - 1) && (hy <= oy + C::BH - 1) && (lo0 >= pbase - (C::Span - 1)) && (lo1 >= pbase + last - (C::Span - 1));
            } else {
                pbase = min(lo0, lo1 + last);
                fits = (hx <= ox + C::BW - 1) && (hy <= oy + C::BH - 1) && (hi0 <= pbase + (C::Span - 1)) && (hi1 <= pbase - last + (C::Span - 1));
            }
        }
        ox = __builtin_amdgcn_readfirstlane(ox); oy = __builtin_amdgcn_readfirstlane(oy); pbase = __builtin_amdgcn_readfirstlane(pbase);
        need_rows = __builtin_amdgcn_readfirstlane(need_rows); need_c4 = __builtin_amdgcn_readfirstlane(need_c4);
        if (!__builtin_amdgcn_readfirstlane((int)fits)) { ok = false; break; }   // (cannot happen while zs_nsub holds; reported as NaN rows, never silent)

        // ---- ring: zero everything (cells outside the volume in x / y are never written by a DMA: they ARE the zero padding)
        for (int i = tid; i < C::RingFloats / 4; i += C::Threads) reinterpret_cast<float4 *>(ring)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        // DMA pieces of this wave: q = wave and wave + 8; lane -> float4 slot (row, c4) of the plane
        unsigned long long mk[2];
        unsigned voff[2];
        bool pvalid[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int q = wave + 8 * k;
            const int slot = q * C::LPP + lane;
            const bool valid = (lane < C::LPP) && (slot < C::PlaneSlots);
            const int row = slot / C::BW4, c4 = slot - row * C::BW4;
            const int gy = oy + row, gx = ox + 4 * c4;
            const bool inb = valid && (row <= need_rows) && (c4 <= need_c4) && (gy >= 0) && (gy < H) && (gx >= 0) && (gx + 4 <= W);
            mk[k] = __builtin_amdgcn_ballot_w64(inb);
            voff[k] = inb ? (unsigned)((row * W + 4 * c4) * 4) : 0u;
            pvalid[k] = valid;
        }
        // Per piece: a piece with nothing to fetch (its lanes all lie outside the volume in x / y) still issues one lane into a dummy
        // float4, so that every wave has exactly two vector-memory operations per plane and one s_waitcnt immediate serves all.
        const bool real0 = mk[0] != 0, real1 = mk[1] != 0;
        const unsigned long long em0 = real0 ? mk[0] : 1ull, em1 = real1 ? mk[1] : 1ull;
        auto dma2 = [&](const char *b0, const char *b1, unsigned l0, unsigned l1, unsigned long long k0, unsigned long long k1) {
            unsigned long long sv;
            unsigned m0s;
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "s_mov_b32 %[m0s], m0\n\t"
                         "s_mov_b32 m0, %[l0]\n\t"
                         "s_mov_b64 exec, %[k0]\n\t"
                         "global_load_lds_dwordx4 %[o0], %[b0]" TRX_ZS_RING_POLICY "\n\t"
                         "s_mov_b32 m0, %[l1]\n\t"
                         "s_mov_b64 exec, %[k1]\n\t"
                         "global_load_lds_dwordx4 %[o1], %[b1]" TRX_ZS_RING_POLICY "\n\t"
                         "s_mov_b64 exec, %[sv]\n\t"
                         "s_mov_b32 m0, %[m0s]"
                         : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                         : [l0] "s"(l0), [l1] "s"(l1), [b0] "s"(b0), [b1] "s"(b1), [o0] "v"(voff[0]), [o1] "v"(voff[1]), [k0] "s"(k0), [k1] "s"(k1)
                         : "memory");
        };
        // plane p -> ring slot `slot` = p mod NZ.  `gb` = address of (plane p, row oy, column ox) - uniform, may point outside the volume (masked lanes).
        auto issue_plane = [&](int p, int slot, const char *gb) {
            if (TRX_ZS_DBG & 1) return;
            const unsigned so = (unsigned)slot * (unsigned)C::PlaneBytes;
            const char *mv = reinterpret_cast<const char *>(mov);
            if ((unsigned)p < (unsigned)D) {
                dma2(real0 ? gb : mv, real1 ? gb : mv, real0 ? pl0 + so : dummy_lds, real1 ? pl1 + so : dummy_lds, em0, em1);
            } else {   // a source plane outside the volume: zero padding (and two dummy operations: vmcnt stays uniform)
                dma2(mv, mv, dummy_lds, dummy_lds, 1ull, 1ull);
#pragma unroll
                for (int k = 0; k < 2; k++)
                    if (pvalid[k]) *reinterpret_cast<float4 *>(ring + slot * C::PlaneFloats + ((wave + 8 * k) * C::LPP + lane) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto plane_ptr = [&](int p) { return reinterpret_cast<const char *>(mov + ((ptrdiff_t)p * H + oy) * W + ox); };
        auto pmod = [&](int p) { const int m = p % C::NZ; return m < 0 ? m + C::NZ : m; };   // (uniform, a few times per anchor)
        // this thread's target values of one plane (R rows at its x)
        auto issue_targets = [&](const float *trow /* uniform: plane z of the target */, float (&tv)[R]) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (TRX_ZS_DBG & 2) { tv[j] = 1.f; continue; }
                asm volatile("global_load_dword %0, %1, %2" TRX_ZS_TGT_POLICY : "=v"(tv[j]) : "v"(toffb), "s"(trow + (size_t)j * W) : "memory");
            }
        };
        const int bpb = (int)ring_lds - (oy * C::BW + ox) * 4;   // LDS byte address of (x = 0, y = 0) of ring slot 0

        // zn / unnorm(zn) of 64 steps at a time, one step per lane (a per-step load of ztab[z] would sit in front of every gather and,
        // being a vector-memory operation the compiler counts, drain the DMA pipeline with its s_waitcnt vmcnt(0))
        float zn_l = 0.f, zid_l = 0.f;
        auto load_ztab = [&](int s0) {
            zn_l = ztab[max(0, min(zf + dir * (s0 + lane), D - 1))];
            zid_l = unnorm<3>(zn_l, fD);
        };
        // Ring slots of the source planes step s may touch, zlo = pbase + s - (Span - 1) being the lowest: byte r of `tabA` = slot of
        // plane zlo + r, of `tabB` = slot of plane zlo + r + 1 (r = 0 .. 7); `selc` + floor(iz) = the byte selector of v_perm_b32.
        unsigned long long tabA = 0, tabB = 0;
        unsigned selc = 0;
        int mlo = 0;   // zlo mod NZ
        auto set_tables = [&](int zlo) {
            mlo = pmod(zlo);
            tabA = tabB = 0;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                tabA |= (unsigned long long)((mlo + r) % C::NZ) << (8 * r);
                tabB |= (unsigned long long)((mlo + r + 1) % C::NZ) << (8 * r);
            }
            selc = 0x0c0c0c00u - (unsigned)zlo;
        };
        auto advance_tables = [&]() {   // zlo -> zlo + 1 (scalar unit)
            mlo = (mlo + 1 == C::NZ) ? 0 : mlo + 1;
            const int m9 = (mlo + 8) % C::NZ;
            tabA = tabB;
            tabB = (tabB >> 8) | ((unsigned long long)m9 << 56);
            selc -= 1u;
        };

        auto gather_plane = [&](int s, float (&tv)[R]) {
            if (TRX_ZS_DBG & 4) return;
            const float zn = lane_bcast(zn_l, s & 63), zid = lane_bcast(zid_l, s & 63);
            const float bxz = fmaf(kxz, zn, cx), byz = fmaf(kyz, zn, cy), bzz = zid + fmaf(kzz, zn, cz);
            f2 ixp[(R + 1) / 2], iyp[(R + 1) / 2], izp[(R + 1) / 2];
            if constexpr (kV2) {
                const f2 bx2 = {bxz, bxz}, by2 = {byz, byz}, bz2 = {bzz, bzz};
#pragma unroll
                for (int k = 0; k < R / 2; k++) {
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(ixp[k]) : "v"(bx2), "s"(sx2[k]));
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(iyp[k]) : "v"(by2), "s"(ey2[k]));
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(izp[k]) : "v"(bz2), "s"(sz2[k]));
                }
            }
            unsigned selv = selc;   // (v2: in a vector register - one move per plane instead of a scalar operand in every voxel's add)
            if constexpr (kV2) asm("v_mov_b32 %0, %1" : "=v"(selv) : "s"(selc));
            struct Fetch { f2 r00, r01, r10, r11; float fx, fy, fz; };
            auto fetch = [&](int j) -> Fetch {
                float ix, iy, iz;
                if constexpr (kV2) {
                    ix = (j & 1) ? ixp[j >> 1].y : ixp[j >> 1].x; iy = (j & 1) ? iyp[j >> 1].y : iyp[j >> 1].x; iz = (j & 1) ? izp[j >> 1].y : izp[j >> 1].x;
                } else {
                    ix = bxz + sx_r[j]; iy = byz + ey_r[j]; iz = bzz + sz_r[j];
                }
                const unsigned sel = (unsigned)floor_to_int(iz) + selv;
                int a0, a1, aA, aB;
                asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a0) : "v"(floor_to_int(ix)), "s"(bpb));
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a1) : "v"(floor_to_int(iy)), "s"(ys_s), "v"(a0));
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(aA) : "v"(__builtin_amdgcn_perm((unsigned)(tabA >> 32), (unsigned)tabA, sel)), "s"(ps_s), "v"(a1));
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(aB) : "v"(__builtin_amdgcn_perm((unsigned)(tabB >> 32), (unsigned)tabB, sel)), "s"(ps_s), "v"(a1));
                Fetch f;
                if (TRX_ZS_DBG & 32) {
                    f.r00 = f2{__int_as_float(aA), ix}; f.r01 = f2{iy, __int_as_float(aB)}; f.r10 = f2{iz, ix}; f.r11 = f2{iy, iz};
                } else {
                if (!(TRX_ZS_DBG & 128)) { f.r00 = *(lds_f2)(unsigned)aA; f.r01 = *(lds_f2)(unsigned)(aA + C::BW * 4); }
                f.r10 = *(lds_f2)(unsigned)aB; f.r11 = *(lds_f2)(unsigned)(aB + C::BW * 4);
                if (TRX_ZS_DBG & 128) {   // (opaque copies: the arithmetic on them stays)
                    f.r00 = f.r10; f.r01 = f.r11;
                    asm volatile("" : "+v"(f.r00.x), "+v"(f.r00.y), "+v"(f.r01.x), "+v"(f.r01.y), "+v"(aA));
                }
                }
                f.fx = __builtin_amdgcn_fractf(ix); f.fy = __builtin_amdgcn_fractf(iy); f.fz = __builtin_amdgcn_fractf(iz);
                return f;
            };
            Fetch cur = fetch(0);
#pragma unroll
            for (int j = 0; j < R; j++) {
                Fetch nxt;
                if (j + 1 < R) nxt = fetch(j + 1);
                if (TRX_ZS_DBG & 64) {
                    const Samp3 sm = lerp3_pairs<kGrad>(cur.r00, cur.r01, cur.r10, cur.r11, cur.fx, cur.fy, cur.fz);
                    acc2.M4 += sm.v + sm.dx + sm.dy + sm.dz + tv[j];
                } else if constexpr (kV2) {
                    zs_lerp_accumulate(cur.r00, cur.r01, cur.r10, cur.r11, cur.fx, cur.fy, cur.fz, tv[j], yn_r[j], yn2[j], acc2);
                } else {
                    const Samp3 sm = lerp3_pairs<kGrad>(cur.r00, cur.r01, cur.r10, cur.r11, cur.fx, cur.fy, cur.fz);
                    f1_accumulate_pk<MODE>(sm, tv[j], yn_r[j], acc);
                }
                if (j + 1 < R) cur = nxt;
            }
            if constexpr (kV2) {
#pragma unroll
                for (int q = 0; q < 3; q++) { Uxy[q] += acc2.Axy[q]; Uz[q] += acc2.ABz[q].x; }
            } else if constexpr (kGrad) {
#pragma unroll
                for (int q = 0; q < (NQ > 0 ? NQ : 1); q++)
#pragma unroll
                    for (int c = 0; c < 3; c++) U[q][c] += acc.AB[q][c].x;
            }
        };

        auto retreat_tables = [&]() {   // zlo -> zlo - 1 (walking down)
            mlo = (mlo == 0) ? C::NZ - 1 : mlo - 1;
            tabB = tabA;
            tabA = (tabA << 8) | (unsigned long long)mlo;
            selc += 1u;
        };

        // ---- one step.  Program order of a wave:  T(s+1) | D(s+2) || wait: all but D(s+2) | barrier | T(s+2) | D(s+3) | gather(s+1) ...
        //   T(s+1): this thread's targets of the next plane into the other register set
        //   D(s+2): ring plane pbase + s + 2 into the slot of plane pbase + s + 2 - NZ, which no step >= s reads
        const ptrdiff_t tstep = (ptrdiff_t)dir * H * W, dstep = (ptrdiff_t)dir * (ptrdiff_t)plane_bytes;
        const float *tnext = tgt + (ptrdiff_t)(zf + dir) * H * W;    // target plane of the next step
        const char *dnext = plane_ptr(pbase + dir * TRX_ZS_LEAD);     // ring plane TRX_ZS_LEAD steps ahead ...
        int dslot = pmod(pbase + dir * TRX_ZS_LEAD);                  // ... and its slot
        auto step = [&](int s, float (&use)[R], float (&load)[R]) {
#if TRX_ZS_STAMP
            const unsigned long long st0 = __builtin_amdgcn_s_memtime();
            unsigned long long st1 = st0, st2 = st0;
#endif
#if TRX_ZS_PRIO == 1
            if ((s + second_slot) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#elif TRX_ZS_PRIO == 2
            if (((__builtin_amdgcn_s_memtime() >> TRX_ZS_PRIO_BIT) + second_slot) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
            if (s > 0) {
                if ((s & 63) == 0) {   // next chunk of the z tables (compiler-counted loads: the pipeline drains here, once per 64 steps)
                    load_ztab(s);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (TRX_ZS_DBG & 16) {
                } else if (TRX_ZS_LEAD == 2 && s + 1 <= last) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if TRX_ZS_STAMP
                st1 = __builtin_amdgcn_s_memtime();
#endif
                if (!(TRX_ZS_DBG & 8)) __syncthreads();
#if TRX_ZS_STAMP
                st2 = __builtin_amdgcn_s_memtime();
#endif
                if (down) retreat_tables(); else advance_tables();
            }
#pragma unroll
            for (int j = 0; j < R; j++) asm volatile("" : "+v"(use[j]));
            if (s + 1 <= last) { issue_targets(tnext, load); tnext += tstep; }
            if (s + TRX_ZS_LEAD <= last) {
                issue_plane(pbase + dir * (s + TRX_ZS_LEAD), dslot, dnext);
                dnext += dstep;
                dslot += dir;
                dslot = (dslot == C::NZ) ? 0 : (dslot < 0 ? C::NZ - 1 : dslot);
            }
#if TRX_ZS_STAMP
            const unsigned long long st3 = __builtin_amdgcn_s_memtime();
#endif
            gather_plane(s, use);
#if TRX_ZS_STAMP
            const unsigned long long st4 = __builtin_amdgcn_s_memtime();
            stamp_sum[0] += st1 - st0; stamp_sum[1] += st2 - st1; stamp_sum[2] += st3 - st2; stamp_sum[3] += st4 - st3; stamp_sum[5] += 1;
#endif
        };

        float tvA[R], tvB[R];
#pragma unroll
        for (int j = 0; j < R; j++) tvA[j] = tvB[j] = 0.f;
        load_ztab(0);
        set_tables(down ? pbase : pbase - (C::Span - 1));
        __syncthreads();   // the ring is zeroed
        // fill the pipeline: the planes steps 0 and 1 touch, the targets of step 0
        {
            const int ahead = min(last, TRX_ZS_LEAD - 1);
            const int pa = down ? pbase - ahead : pbase - (C::Span - 1), pb = down ? pbase + (C::Span - 1) : pbase + ahead;
            for (int p = pa; p <= pb; p++) issue_plane(p, pmod(p), plane_ptr(p));
        }
        issue_targets(tgt + (size_t)zf * H * W, tvA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int s = 0;
        while (s < nsteps) {
            step(s, tvA, tvB);
            s++;
            if (s >= nsteps) break;
            step(s, tvB, tvA);
            s++;
        }
        __syncthreads();   // every wave is done with the ring (it is re-zeroed by the next anchor / becomes the reduction scratch)
    }

#if TRX_ZS_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (!ok) {
        if (tid < NP) partials[((size_t)by * rows_per_pair + bx) * NP + tid] = __builtin_nanf("");
        return;
    }
    float vals[NP];
    int o = 0;
    if constexpr (kV2) {
        vals[0] = acc2.M01.x; vals[1] = acc2.M01.y; vals[2] = acc2.M23.x; vals[3] = acc2.M23.y; vals[4] = acc2.M4;
        o = 5;
    } else if constexpr (MODE == 4) {
        vals[0] = acc.M4;
        o = 1;
    } else {
        vals[0] = acc.M01.x; vals[1] = acc.M01.y; vals[2] = acc.M23.x; vals[3] = acc.M23.y; vals[4] = acc.M4;
        o = 5;
    }
    if constexpr (kGrad) {
        // sum_z zn_z P_z = zn_0 S + d (n S - U): zn_z = zn_0 + d (z - zb0), S = the final running sum, U = sum over planes of the running sums
        // (in step order: walking down, the first step's plane is the segment's last one and zn falls by the same amount per step)
        const int nall = ze0 - zb0;
        const float zn0 = ztab[down ? ze0 - 1 : zb0];
        const float dzn = nall > 1 ? (ztab[down ? zb0 : ze0 - 1] - zn0) / (float)(nall - 1) : 0.f;
        const float fn = (float)nall;
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float a, bsum, u;
                if constexpr (kV2) {
                    a = c == 0 ? acc2.Axy[q].x : (c == 1 ? acc2.Axy[q].y : acc2.ABz[q].x);
                    bsum = c == 0 ? acc2.Bxy[q].x : (c == 1 ? acc2.Bxy[q].y : acc2.ABz[q].y);
                    u = c == 0 ? Uxy[q].x : (c == 1 ? Uxy[q].y : Uz[q]);
                } else {
                    a = acc.AB[q][c].x; bsum = acc.AB[q][c].y; u = U[q][c];
                }
                vals[o++] = xn * a; vals[o++] = bsum; vals[o++] = fmaf(dzn, fmaf(fn, a, -u), zn0 * a); vals[o++] = a;
            }
    }
    block_reduce_store_nw<NP, C::Waves>(vals, partials + ((size_t)by * rows_per_pair + bx) * NP, ring, wave);
#if TRX_ZS_STAMP
    if (lane == 0) {
        unsigned long long *o = trx_zs_stamps + ((size_t)((by * rows_per_pair + bx) & 511) * 8 + wave) * 8;
        stamp_sum[4] = __builtin_amdgcn_s_memtime() - stamp_t0;
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        for (int k = 0; k < 5; k++) o[k] = stamp_sum[k];
        o[5] = stamp_sum[5] | ((unsigned long long)(hwid & 0xffff) << 16) | ((unsigned long long)(xcc & 0xf) << 32);
        o[6] = stamp_r0; o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int MODE, class C>
__global__ __launch_bounds__(C::Threads, TRX_ZS_MIN_WAVES) void affine_zstream_kernel(trx_volumes vol, const float *__restrict__ theta, ZGeom zg,
                                                                                      float *__restrict__ partials, int rows_per_pair)
{
    __shared__ __attribute__((aligned(16))) float ring[C::Alloc];
    if ((int)blockIdx.x >= zg.blocks_per_pair) return;
    if (zs_nsub<C>(theta + (size_t)blockIdx.y * TRX_PSTRIDE, (float)vol.D, (float)vol.H, (float)vol.W, zg.planes_per_seg) == 0) {
        if (threadIdx.x < 41) partials[((size_t)blockIdx.y * rows_per_pair + blockIdx.x) * 41 + threadIdx.x] = __builtin_nanf("");
        return;
    }
    zstream_body<MODE, C>(vol, theta, zg, partials, ring, blockIdx.x, blockIdx.y, rows_per_pair, trx_wave_index(),
                          (int)((blockIdx.y * gridDim.x + blockIdx.x) * 2 >= gridDim.x * gridDim.y));
}
#pragma clang diagnostic pop
