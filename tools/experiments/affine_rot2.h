// DOUBLE-BUFFERED packed-box tile body for rotated transforms (3-D, step kernels): tools/experiments/affine_rot.h with the one thing a
// tile body was missing - staging of tile t + 1 under the gather of tile t inside ONE block.  1024 threads own a 16 x 16 x 16 tile (four
// rows per thread) and both 78 KB halves of a CU's LDS: while the 16 waves gather tile t from one box, their LDS-DMA pieces of tile t + 1
// land in the other; every wave issues exactly kPieces DMA instructions and Rows target loads per tile (dummy one-lane pieces where its
// share of the box is empty), so one s_waitcnt immediate separates "tile t has landed" from "tile t + 1 is in flight".
// Included inside namespace trx after affine_rot.h (RotCfg / rot_dims).

struct Rot2Cfg {
    static constexpr int TX = 16, TY = 16, TZ = 16, Threads = 1024, Waves = 16, Rows = 4;
    static constexpr int BoxFloats = RotCfg::BoxFloats;
    static constexpr int Pieces = (BoxFloats / 4 + Threads - 1) / Threads;   // 5 per wave and tile
    static constexpr int Alloc = 2 * BoxFloats + 4;                           // two boxes + the float4 the dummy pieces write
    static_assert(Alloc * 4 <= 160 * 1024 && Waves * 16 * 65 + Waves * 16 <= BoxFloats, "LDS");
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
#pragma clang diagnostic ignored "-Winline-asm"
template <int MODE>
__device__ __forceinline__ void rot2_body(const trx_volumes &vol, const float *__restrict__ theta, const TileGeom &tg, float *__restrict__ partials, float *lds,
                                          const int bx, const int by, const int rows_stride, const int wave_in)
{
    using C = Rot2Cfg;
    constexpr int NQ = (MODE == 0) ? 3 : (MODE == 4 ? 1 : 0);
    constexpr int NP = (MODE == 0) ? np_full(3) : (MODE == 4 ? kNpMse : 5);
    constexpr bool kGrad = (MODE == 0) || (MODE == 4);
    const int D = vol.D, H = vol.H, W = vol.W;
    const float *__restrict__ th = uni_ptr(theta + (size_t)by * TRX_PSTRIDE);
    const float *__restrict__ mov = uni_ptr(vol.moving + (size_t)by * vol.moving_stride);
    const float *__restrict__ tgt = uni_ptr(vol.target + (size_t)by * vol.target_stride);
    const float *__restrict__ xtab = vol.xn, *__restrict__ ytab = vol.yn, *__restrict__ ztab = vol.zn;
    const int lane = trx_lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);
    const int tid = wave * 64 + lane;
    const int lx = tid & 15, lz = (tid >> 4) & 15, lq = wave >> 2;   // rows lq * 4 .. lq * 4 + 3 of a tile: wave-uniform
    const float fW = (float)W, fH = (float)H, fD = (float)D;
    const float t00 = th[0], t01 = th[1], t02 = th[2], t03 = th[3];
    const float t10 = th[4], t11 = th[5], t12 = th[6], t13 = th[7];
    const float t20 = th[8], t21 = th[9], t22 = th[10], t23 = th[11];

    const int ncol = tg.ntx * tg.ntz;
    const int yseg = bx / ncol, cb = bx - yseg * ncol;
    int col = cb;
    if ((ncol & 7) == 0) col = (cb & 7) * (ncol >> 3) + (cb >> 3);
    const int X0 = (col % tg.ntx) * C::TX, Z0 = (col / tg.ntx) * C::TZ;
    const bool act = (X0 + lx < W) && (Z0 + lz < D);
    const int x = X0 + (act ? lx : 0), z = Z0 + (act ? lz : 0);
    const float xn = xtab[x], zn = ztab[z];
    const float hW = 0.5f * fW, hH = 0.5f * fH, hD = 0.5f * fD;
    const float Px = unnorm<3>(fmaf(t00, xn, fmaf(t02, zn, t03)), fW), Py = unnorm<3>(fmaf(t10, xn, fmaf(t12, zn, t13)), fH), Pz = unnorm<3>(fmaf(t20, xn, fmaf(t22, zn, t23)), fD);
    const float kx = hW * t01, ky = hH * t11, kz = hD * t21;
    const float cxn = xtab[X0], czn = ztab[Z0];
    const float cPx = uni(unnorm<3>(fmaf(t00, cxn, fmaf(t02, czn, t03)), fW)), cPy = uni(unnorm<3>(fmaf(t10, cxn, fmaf(t12, czn, t13)), fH)),
                cPz = uni(unnorm<3>(fmaf(t20, cxn, fmaf(t22, czn, t23)), fD));

    int NX4, NY, NZ;
    float ext_lo[3];
    const bool fits = rot_dims(th, fD, fH, fW, NX4, NY, NZ, ext_lo);
    NX4 = __builtin_amdgcn_readfirstlane(NX4); NY = __builtin_amdgcn_readfirstlane(NY); NZ = __builtin_amdgcn_readfirstlane(NZ);
    const float elx = uni(ext_lo[0]), ely = uni(ext_lo[1]), elz = uni(ext_lo[2]);
    const int plane_slots = NX4 * NY, slots = plane_slots * NZ;
    const float inv_plane = 1.0f / (float)plane_slots, inv_row = 1.0f / (float)NX4;
    unsigned voff[C::Pieces];
    unsigned long long vmask[C::Pieces];
#pragma unroll
    for (int k = 0; k < C::Pieces; k++) {
        const int s = k * C::Threads + tid;
        const int dz = (int)(((float)s + 0.5f) * inv_plane), r = s - dz * plane_slots;
        const int dy = (int)(((float)r + 0.5f) * inv_row), dx4 = r - dy * NX4;
        const bool valid = s < slots;
        voff[k] = valid ? (unsigned)(((dz * H + dy) * W + 4 * dx4) * 4) : 0u;
        vmask[k] = __builtin_amdgcn_ballot_w64(valid);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const unsigned dummy_lds = lds0 + 2u * C::BoxFloats * 4u;
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
    const unsigned col_bytes = (unsigned)((z * H) * W + x) * 4u;   // this thread's voxel column inside the target

    // one LDS-DMA instruction of this wave: `mk` lanes fetch their float4 at gb + off into dst + lane * 16 (mk == 0: one lane into the dummy slot)
    auto dma = [&](const char *gb, unsigned off, unsigned dst, unsigned long long mk) {
        unsigned long long sv;
        unsigned m0s;
        const bool real = mk != 0ull;
        const char *b = real ? gb : reinterpret_cast<const char *>(mov);
        const unsigned d = real ? dst : dummy_lds;
        const unsigned long long m = real ? mk : 1ull;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b32 %[m0s], m0\n\t"
                     "s_mov_b32 m0, %[l0]\n\t"
                     "s_mov_b64 exec, %[k0]\n\t"
                     "global_load_lds_dwordx4 %[o0], %[b0]\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_mov_b32 m0, %[m0s]"
                     : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                     : [l0] "s"(__builtin_amdgcn_readfirstlane(d)), [b0] "s"(b), [o0] "v"(real ? off : 0u), [k0] "s"(m)
                     : "memory");
    };
    // stage tile `ty` into box `bi` and request this thread's target values of it: exactly Pieces + Rows vector-memory operations per wave
    unsigned cbase_n = 0;
    float tv_n[C::Rows];
    auto issue = [&](int ty, int bi) {
        const int Y0 = ty * C::TY;
        const float cyn = ytab[Y0];
        const float p0x = fmaf(kx, cyn, cPx), p0y = fmaf(ky, cyn, cPy), p0z = fmaf(kz, cyn, cPz);
        const int ox = (__builtin_amdgcn_readfirstlane(floor_to_int(p0x + elx - 0.002f)) >> 2) << 2;
        const int oy = __builtin_amdgcn_readfirstlane(floor_to_int(p0y + ely - 0.002f)), oz = __builtin_amdgcn_readfirstlane(floor_to_int(p0z + elz - 0.002f));
        const unsigned box_lds = lds0 + (unsigned)bi * (C::BoxFloats * 4u);
        cbase_n = box_lds - (unsigned)(((oz * NY + oy) * NXf + ox) * 4);
        const bool interior = (ox >= 0) && (ox + 4 * NX4 <= W) && (oy >= 0) && (oy + NY <= H) && (oz >= 0) && (oz + NZ <= D);
        const long long origin_off = ((long long)oz * H + oy) * W + ox;
        const char *gbase = reinterpret_cast<const char *>(mov) + origin_off * 4;
        if (interior) {
#pragma unroll
            for (int k = 0; k < C::Pieces; k++) dma(gbase, voff[k], box_lds + (unsigned)(k * C::Threads + wave * 64) * 16u, vmask[k]);
        } else {
#pragma unroll
            for (int k = 0; k < C::Pieces; k++) {
                const int s = k * C::Threads + tid;
                const int dz = (int)(((float)s + 0.5f) * inv_plane), r = s - dz * plane_slots;
                const int dy = (int)(((float)r + 0.5f) * inv_row), dx4 = r - dy * NX4;
                const int gz = oz + dz, gy = oy + dy, gx = ox + 4 * dx4;
                const bool valid = s < slots;
                const bool inb = valid && ((unsigned)gz < (unsigned)D) && ((unsigned)gy < (unsigned)H) && (gx >= 0) && (gx + 4 <= W);
                dma(gbase, inb ? voff[k] : 0u, box_lds + (unsigned)(k * C::Threads + wave * 64) * 16u, __builtin_amdgcn_ballot_w64(inb));
                if (valid && !inb) *reinterpret_cast<float4 *>(lds + bi * C::BoxFloats + s * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const int ybase = Y0 + lq * C::Rows;
#pragma unroll
        for (int j = 0; j < C::Rows; j++) {
            const int yy = min(ybase + j, H - 1);
            const unsigned off = col_bytes + (unsigned)(yy * W) * 4u;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(tv_n[j]) : "v"(off), "s"(tgt) : "memory");
        }
    };

    if (fits && ty0 < ty1) {
        issue(ty0, 0);
        for (int ty = ty0; ty < ty1; ty++) {
            const int bi = (ty - ty0) & 1;
            const unsigned cbase = cbase_n;
            float tv[C::Rows];
#pragma unroll
            for (int j = 0; j < C::Rows; j++) tv[j] = tv_n[j];
            if (ty + 1 < ty1) {
                __syncthreads();   // every wave is done gathering from the other box (tile ty - 1)
                issue(ty + 1, bi ^ 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::Pieces + C::Rows) : "memory");   // tile ty has landed; tile ty + 1 stays in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < C::Rows; j++) asm volatile("" : "+v"(tv[j]));
            __syncthreads();       // ... in every wave
            const int ybase = ty * C::TY + lq * C::Rows;
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
    }
    __syncthreads();   // box 0 becomes the reduction scratch

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
    block_reduce_store_nw<NP, C::Waves>(vals, partials + ((size_t)by * rows_stride + bx) * NP, lds, wave);
}
#pragma clang diagnostic pop

template <int MODE>
__global__ __launch_bounds__(Rot2Cfg::Threads) void affine_rot2_kernel(trx_volumes vol, const float *__restrict__ theta, TileGeom tg, float *__restrict__ partials,
                                                                      int rows_stride)
{
    __shared__ __attribute__((aligned(16))) float lds[Rot2Cfg::Alloc];
    if ((int)blockIdx.x >= tg.blocks_per_pair) return;
    rot2_body<MODE>(vol, theta, tg, partials, lds, blockIdx.x, blockIdx.y, rows_stride, trx_wave_index());
}
