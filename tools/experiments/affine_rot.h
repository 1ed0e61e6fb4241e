// PACKED-BOX variant of the tile body for rotated transforms (3-D, step kernels) - an experiment of round 3, measured and SHELVED (numbers
// at the end of this comment, DESIGN.md 8 item 0c, profiles/r03d_packed_box_experiment.txt).  Included by tools/rbench.hip inside namespace trx.
//
// What bounds GeomR / GeomRD on a general rotation is staging (613 of 614 us), and what staging costs is the NUMBER of
// LDS-DMA instructions, not their bytes: a global_load_lds_dwordx4 occupies the CU's address path for ~48 cycles whether 64 or 32 of its
// lanes are live (half the lanes: -4 %; half the lines: -16 %).  tile_body lays its box out with compile-time strides (28 x 27 x 26),
// so a piece of 512 lanes covers two whole box planes of 7 x 27 float4 slots of which the needed extent fills 45-75 %, and 13 pieces
// per wave go out for every tile of 2048 voxels.  Here the box has the layout of what is NEEDED: its dimensions (NX4 float4 x NY x NZ)
// are computed once per block from theta (the pre-image of a tile is position independent up to rounding: an affine map), slots are
// numbered densely in that layout, and a thread's global offset per piece is computed once and kept in a register - per tile only the
// origin moves.  With a 16 x 16 x 16 tile (eight rows per thread, the fixed per-tile work amortised over 4096 voxels) a general rotation
// stages 4-5 floats per voxel in 8-10 dense pieces per wave: a third of the DMA instructions per voxel.
// The gather pays three vector adds per voxel for strides that are no longer immediates.  Tiles whose box crosses a face of the volume
// decode their slots again (float reciprocals) and zero-fill what lies outside: grid_sample's zero padding.
// A pair whose box does not fit the LDS budget stays with tile_body (rot_dims returns false: general rotations beyond ~0.4 rad per axis).
// Measured (8 x 256^3, MI355X, tools/rbench.hip; sums equal to the tile kernels' to the rounding of the summation order): R(.5,.4,.3)
// 559 us against 612 for GeomR (-9 %), but R_z(0.6) 480 against 433, R(.2,.2,.2) 503 against 463, identity 441 against 379 - although it
// issues a third of the DMA instructions and 46 vector instructions per voxel (trimming them from 58 changed the time by 2.5 %).  A tile
// round - stage, wait, barrier, gather, with two blocks per CU taking turns - lasts ~7 us per 4096-voxel tile whatever is staged: the
// vector work of the CU's 16 waves accounts for half of it, the rest is the latency of the burst.  What the z-streaming body has and no
// tile body can have within 160 KB is a second box: staging of tile t + 1 under the gather of tile t.  Not offered by the launcher.

struct RotCfg {
    static constexpr int TX = 16, TY = 16, TZ = 16, Threads = 512, Waves = 8, Rows = 8;
    static constexpr int BoxFloats = 19968;                 // 78 KB: what the dual kernel's LDS allocation holds anyway
    static constexpr int MaxPieces = (BoxFloats / 4 + Threads - 1) / Threads;   // 10
    static constexpr int ReduceScratch = Waves * 16 * 65 + Waves * 16;
    static_assert(ReduceScratch <= BoxFloats, "the box doubles as the reduction scratch");
};

// Dimensions of the packed box for theta: cells per axis the pre-image of a 16^3 tile can touch, x in float4 units with room for the
// alignment of the origin.  Returns false when it does not fit (or theta is not finite).
__device__ __forceinline__ bool rot_dims(const float *__restrict__ th, float fD, float fH, float fW, int &NX4, int &NY, int &NZ, float (&ext_lo)[3])
{
    const float ex[3] = {(float)(RotCfg::TX - 1), (float)(RotCfg::TY - 1), (float)(RotCfg::TZ - 1)};
    const float slope[3][3] = {{th[0], th[1] * fW / fH, th[2] * fW / fD}, {th[4] * fH / fW, th[5], th[6] * fH / fD}, {th[8] * fD / fW, th[9] * fD / fH, th[10]}};
    int n[3];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float e = slope[c][a] * ex[a];
            lo += fminf(e, 0.f); hi += fmaxf(e, 0.f);
        }
        ext_lo[c] = lo;
        const float span = hi - lo;
        ok = ok && (span < 200.0f);          // (NaN compares false)
        n[c] = (int)(span + 0.004f) + 3;     // floor(pmin - m) .. floor(pmax + m) + 1
    }
    NX4 = (n[0] + 3 + 3) >> 2; NY = n[1]; NZ = n[2];
    return ok && (NX4 * NY * NZ * 4 <= RotCfg::BoxFloats);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <int MODE>
__device__ __forceinline__ void rot_body(const trx_volumes &vol, const float *__restrict__ theta, const TileGeom &tg, float *__restrict__ partials, float *box,
                                         const int bx, const int by, const int rows_stride, const int wave_in)
{
    using C = RotCfg;
    constexpr int NQ = (MODE == 0) ? 3 : (MODE == 4 ? 1 : 0);
    constexpr int NP = (MODE == 0) ? np_full(3) : (MODE == 4 ? kNpMse : 5);
    constexpr bool kGrad = (MODE == 0) || (MODE == 4);
    const int D = vol.D, H = vol.H, W = vol.W;
    const float *__restrict__ th = uni_ptr(theta + (size_t)by * TRX_PSTRIDE);
    const float *__restrict__ mov = uni_ptr(vol.moving + (size_t)by * vol.moving_stride);
    const float *__restrict__ tgt = vol.target + (size_t)by * vol.target_stride;
    const float *__restrict__ xtab = vol.xn, *__restrict__ ytab = vol.yn, *__restrict__ ztab = vol.zn;
    const int lane = trx_lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);
    const int tid = wave * 64 + lane;
    const int lx = tid & 15, lz = (tid >> 4) & 15, lh = wave >> 2;   // rows lh * 8 .. lh * 8 + 7 of a tile: wave-uniform
    const float fW = (float)W, fH = (float)H, fD = (float)D;
    const float t00 = th[0], t01 = th[1], t02 = th[2], t03 = th[3];
    const float t10 = th[4], t11 = th[5], t12 = th[6], t13 = th[7];
    const float t20 = th[8], t21 = th[9], t22 = th[10], t23 = th[11];

    // column of this block (XCD-aware order as in tile_body) and its y segment
    const int ncol = tg.ntx * tg.ntz;
    const int yseg = bx / ncol, cb = bx - yseg * ncol;
    int col = cb;
    if ((ncol & 7) == 0) col = (cb & 7) * (ncol >> 3) + (cb >> 3);
    const int X0 = (col % tg.ntx) * C::TX, Z0 = (col / tg.ntx) * C::TZ;
    const bool act = (X0 + lx < W) && (Z0 + lz < D);
    const int x = X0 + (act ? lx : 0), z = Z0 + (act ? lz : 0);
    const float xn = xtab[x], zn = ztab[z];
    // sample point of voxel (x, y, z) in source index units: p_c = P_c + k_c yn(y), P_c = unnorm(theta_c0 xn + theta_c2 zn + theta_c3) per thread,
    // k_c = theta_c1 S_c / 2 (one fma per axis and voxel; a single rounding where the normalised-coordinate form has three)
    const float hW = 0.5f * fW, hH = 0.5f * fH, hD = 0.5f * fD;
    const float Px = unnorm<3>(fmaf(t00, xn, fmaf(t02, zn, t03)), fW), Py = unnorm<3>(fmaf(t10, xn, fmaf(t12, zn, t13)), fH), Pz = unnorm<3>(fmaf(t20, xn, fmaf(t22, zn, t23)), fD);
    const float kx = hW * t01, ky = hH * t11, kz = hD * t21;
    const float cxn = xtab[X0], czn = ztab[Z0];
    const float cPx = uni(unnorm<3>(fmaf(t00, cxn, fmaf(t02, czn, t03)), fW)), cPy = uni(unnorm<3>(fmaf(t10, cxn, fmaf(t12, czn, t13)), fH)),
                cPz = uni(unnorm<3>(fmaf(t20, cxn, fmaf(t22, czn, t23)), fD));

    // the packed box of this theta
    int NX4, NY, NZ;
    float ext_lo[3];
    const bool fits = rot_dims(th, fD, fH, fW, NX4, NY, NZ, ext_lo);
    NX4 = __builtin_amdgcn_readfirstlane(NX4); NY = __builtin_amdgcn_readfirstlane(NY); NZ = __builtin_amdgcn_readfirstlane(NZ);
    const float elx = uni(ext_lo[0]), ely = uni(ext_lo[1]), elz = uni(ext_lo[2]);
    const int plane_slots = NX4 * NY, slots = plane_slots * NZ;
    const int K = (slots + C::Threads - 1) / C::Threads;
    const float inv_plane = 1.0f / (float)plane_slots, inv_row = 1.0f / (float)NX4;
    // slot of this thread in piece k: s = k * 512 + tid -> (dz, dy, dx4); its byte offset from the box origin inside the volume
    unsigned voff[C::MaxPieces];
    unsigned long long last_mask = 0ull;
#pragma unroll
    for (int k = 0; k < C::MaxPieces; k++) {
        const int s = k * C::Threads + tid;
        const int dz = (int)(((float)s + 0.5f) * inv_plane), r = s - dz * plane_slots;
        const int dy = (int)(((float)r + 0.5f) * inv_row), dx4 = r - dy * NX4;
        const bool valid = s < slots;
        voff[k] = valid ? (unsigned)(((dz * H + dy) * W + 4 * dx4) * 4) : 0u;
        if (k == K - 1) last_mask = __builtin_amdgcn_ballot_w64(valid);
    }
    const unsigned box_lds = (unsigned)(uintptr_t)box;
    const int NXf = NX4 * 4;
    const unsigned row_bytes = (unsigned)(NX4 * 16), plane_bytes = (unsigned)(plane_slots * 16);
    typedef const __attribute__((address_space(3))) f2u *lds_f2;

    F1Acc acc;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
    acc.M01 = acc.M23 = (f2)(0.f);
    acc.M4 = 0.f;

    const int ty0 = yseg * tg.tiles_per_seg, ty1 = min(ty0 + tg.tiles_per_seg, tg.nty);
    const unsigned col_off = (unsigned)((z * H) * W + x);   // this thread's voxel column inside the target
    for (int ty = ty0; ty < ty1 && fits; ty++) {
        const int Y0 = ty * C::TY;
        // origin of the box: the image of the tile's (X0, Y0, Z0) corner + the tile-independent lower extent
        const float cyn = ytab[Y0];
        const float p0x = fmaf(kx, cyn, cPx), p0y = fmaf(ky, cyn, cPy), p0z = fmaf(kz, cyn, cPz);
        const int ox = (__builtin_amdgcn_readfirstlane(floor_to_int(p0x + elx - 0.002f)) >> 2) << 2;
        const int oy = __builtin_amdgcn_readfirstlane(floor_to_int(p0y + ely - 0.002f)), oz = __builtin_amdgcn_readfirstlane(floor_to_int(p0z + elz - 0.002f));
        const unsigned cbase = box_lds - (unsigned)(((oz * NY + oy) * NXf + ox) * 4);   // LDS address of source index (0, 0, 0) in this tile's box
        const bool interior = (ox >= 0) && (ox + 4 * NX4 <= W) && (oy >= 0) && (oy + NY <= H) && (oz >= 0) && (oz + NZ <= D);
        // this thread's target values of the tile: requested before the box, consumed after it
        const int ybase = Y0 + lh * C::Rows;
        float tv[C::Rows];
#pragma unroll
        for (int j = 0; j < C::Rows; j++) tv[j] = (act && ybase + j < H) ? tgt[col_off + (unsigned)((ybase + j) * W)] : 0.f;

        __syncthreads();   // every wave is done with the previous tile's box
        const long long origin_off = ((long long)oz * H + oy) * W + ox;   // (may be negative for a box that crosses a face: only in-volume lanes use it)
        const char *gbase = reinterpret_cast<const char *>(mov) + origin_off * 4;
        if (interior) {
#pragma unroll
            for (int k = 0; k < C::MaxPieces; k++) {
                if (k < K) {
                    const unsigned long long mk = (k == K - 1) ? last_mask : ~0ull;
                    if (mk != 0ull) {
                        unsigned long long sv;
                        unsigned m0s;
                        const unsigned dst = box_lds + (unsigned)(k * C::Threads + wave * 64) * 16u;
                        asm volatile("s_mov_b64 %[sv], exec\n\t"
                                     "s_mov_b32 %[m0s], m0\n\t"
                                     "s_mov_b32 m0, %[l0]\n\t"
                                     "s_mov_b64 exec, %[k0]\n\t"
                                     "global_load_lds_dwordx4 %[o0], %[b0]\n\t"
                                     "s_mov_b64 exec, %[sv]\n\t"
                                     "s_mov_b32 m0, %[m0s]"
                                     : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                                     : [l0] "s"(dst), [b0] "s"(gbase), [o0] "v"(voff[k]), [k0] "s"(mk)
                                     : "memory");
                    }
                }
            }
        } else {
            for (int k = 0; k < K; k++) {
                const int s = k * C::Threads + tid;
                const int dz = (int)(((float)s + 0.5f) * inv_plane), r = s - dz * plane_slots;
                const int dy = (int)(((float)r + 0.5f) * inv_row), dx4 = r - dy * NX4;
                const int gz = oz + dz, gy = oy + dy, gx = ox + 4 * dx4;
                const bool valid = s < slots;
                const bool inb = valid && ((unsigned)gz < (unsigned)D) && ((unsigned)gy < (unsigned)H) && (gx >= 0) && (gx + 4 <= W);
                const unsigned off = inb ? (unsigned)(((dz * H + dy) * W + 4 * dx4) * 4) : 0u;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(inb);
                if (mk != 0ull) {
                    unsigned long long sv;
                    unsigned m0s;
                    const unsigned dst = box_lds + (unsigned)(k * C::Threads + wave * 64) * 16u;
                    asm volatile("s_mov_b64 %[sv], exec\n\t"
                                 "s_mov_b32 %[m0s], m0\n\t"
                                 "s_mov_b32 m0, %[l0]\n\t"
                                 "s_mov_b64 exec, %[k0]\n\t"
                                 "global_load_lds_dwordx4 %[o0], %[b0]\n\t"
                                 "s_mov_b64 exec, %[sv]\n\t"
                                 "s_mov_b32 m0, %[m0s]"
                                 : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                                 : [l0] "s"(__builtin_amdgcn_readfirstlane(dst)), [b0] "s"(gbase), [o0] "v"(off), [k0] "s"(mk)
                                 : "memory");
                }
                if (valid && !inb) *reinterpret_cast<float4 *>(box + s * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        if (act) {
#pragma unroll
            for (int j = 0; j < C::Rows; j++) {
                const int y = ybase + j;
                if (y < H) {   // (wave-uniform)
                    const float yn = ytab[y];
                    const float ix = fmaf(kx, yn, Px), iy = fmaf(ky, yn, Py), iz = fmaf(kz, yn, Pz);
                    int t1, t2;
                    unsigned a0;
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(floor_to_int(iz)), "s"(NY), "v"(floor_to_int(iy)));
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t2) : "v"(t1), "s"(NXf), "v"(floor_to_int(ix)));
                    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a0) : "v"(t2), "s"(cbase));
                    const unsigned a1 = a0 + row_bytes, a2 = a0 + plane_bytes, a3 = a2 + row_bytes;
                    const f2 r00 = *(lds_f2)a0, r01 = *(lds_f2)a1, r10 = *(lds_f2)a2, r11 = *(lds_f2)a3;
                    const Samp3 sm = lerp3_pairs<kGrad>(r00, r01, r10, r11, __builtin_amdgcn_fractf(ix), __builtin_amdgcn_fractf(iy), __builtin_amdgcn_fractf(iz));
                    f1_accumulate_pk<MODE>(sm, tv[j], yn, acc);
                }
            }
        }
    }
    __syncthreads();   // the box becomes the reduction scratch

    if (!fits) {
        if (tid < NP) partials[((size_t)by * rows_stride + bx) * NP + tid] = __builtin_nanf("");
        return;
    }
    float vals[NP];
    int o = 0;
    if constexpr (MODE == 4) {
        vals[0] = acc.M4;
        o = 1;
    } else {
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
    block_reduce_store_nw<NP, C::Waves>(vals, partials + ((size_t)by * rows_stride + bx) * NP, box, wave);
}

#pragma clang diagnostic pop

template <int MODE>
__global__ __launch_bounds__(RotCfg::Threads, 4) void affine_rot_kernel(trx_volumes vol, const float *__restrict__ theta, TileGeom tg, float *__restrict__ partials,
                                                                        int rows_stride)
{
    __shared__ __attribute__((aligned(16))) float box[RotCfg::BoxFloats];
    if ((int)blockIdx.x >= tg.blocks_per_pair) return;
    rot_body<MODE>(vol, theta, tg, partials, box, blockIdx.x, blockIdx.y, rows_stride, trx_wave_index());
}
