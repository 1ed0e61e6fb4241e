// Device helpers shared by the affine and flow kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/trx.h"

#define TRX_BLOCK 256
#define TRX_WAVES (TRX_BLOCK / 64)

#define TRX_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return TRX_ERR_HIP; \
    } while (0)

namespace trx {

// ------------------------------------------------------------------------------------------
// Trilinear / bilinear sample with zero padding (grid_sample 'bilinear', padding_mode='zeros').
// Returns the value and the derivative wrt the (un-normalised) voxel coordinates.
// Fast path: when every active lane of the wave has all 8 (4) corners inside the volume the
// loads are issued unmasked; otherwise corner addresses are clamped and values masked.
// ------------------------------------------------------------------------------------------
struct Samp3 {
    float v, dx, dy, dz;
};
// Packed fp32: one v_pk_* instruction works on a 64-bit VGPR pair.  The VALU issues one wave64
// instruction per 4 cycles per SIMD whether it is packed or not (measured: SQ_ACTIVE_INST_VALU ==
// SQ_INSTS_VALU quad-cycles), so explicit pairs halve the cost of the interpolation / accumulation math.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));   // 4-byte aligned pair -> ds_read2_b32

// floor(x) as int32 in ONE instruction (v_cvt_flr_i32_f32) instead of v_floor_f32 + v_cvt_i32_f32
// (z, y, x) along a grid-stride walk over a volume of < 2^31 voxels: one 32-bit division at the start, carries afterwards
// (`i % W`, `i / W % H` on a size_t index compile to ~45 VALU instructions per voxel, a third of a streaming kernel's loop).
struct VoxelWalk {
    unsigned i, stride;
    int x, y, z, sx, sy, sz;
    __device__ __forceinline__ VoxelWalk(unsigned i0, unsigned stride_, int H, int W) : i(i0), stride(stride_)
    {
        const unsigned r = i0 / (unsigned)W, rs = stride_ / (unsigned)W;
        x = (int)(i0 - r * (unsigned)W); z = (int)(r / (unsigned)H); y = (int)(r - (unsigned)z * (unsigned)H);
        sx = (int)(stride_ - rs * (unsigned)W); sz = (int)(rs / (unsigned)H); sy = (int)(rs - (unsigned)sz * (unsigned)H);
    }
    __device__ __forceinline__ void next(int H, int W)
    {
        i += stride;
        x += sx;
        if (x >= W) { x -= W; y++; }
        y += sy;
        if (y >= H) { y -= H; z++; }
        z += sz;
    }
};

// b^n by repeated squaring: the Adam bias terms 1 - beta^t in the finalise / coefficient kernels (two fp64 pow() calls cost ~1 us on one lane)
__device__ __forceinline__ double ipow(double b, int n)
{
    double r = 1.0;
    for (; n > 0; n >>= 1, b *= b)
        if (n & 1) r *= b;
    return r;
}

__device__ __forceinline__ int floor_to_int(float x)
{
    int i;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(x));
    return i;
}

// Trilinear sample + gradient from the four x-pairs (x0, x0+1) at (z0,y0), (z0,y1), (z1,y0), (z1,y1):
// z and y are interpolated on pairs (8 packed instructions), x last (6 scalar ones).
template <bool GRAD>
__device__ __forceinline__ Samp3 lerp3_pairs(f2 r00, f2 r01, f2 r10, f2 r11, float tx, float ty, float tz)
{
    const f2 dz0 = r10 - r00, dz1 = r11 - r01;
    const f2 z0 = dz0 * tz + r00, z1 = dz1 * tz + r01;
    const f2 dy = z1 - z0;
    const f2 yv = dy * ty + z0;
    Samp3 s;
    s.dx = yv.y - yv.x;
    s.v = fmaf(tx, s.dx, yv.x);
    if constexpr (GRAD) {
        const f2 dzy = (dz1 - dz0) * ty + dz0;
        s.dy = fmaf(tx, dy.y - dy.x, dy.x);
        s.dz = fmaf(tx, dzy.y - dzy.x, dzy.x);
    } else {
        s.dy = s.dz = 0.f;
    }
    return s;
}
struct Samp2 {
    float v, dx, dy;
};

__device__ __forceinline__ Samp3 lerp3(float v000, float v001, float v010, float v011, float v100, float v101,
                                       float v110, float v111, float tx, float ty, float tz)
{
    // vzyx naming: v{z}{y}{x}
    float dx00 = v001 - v000, dx01 = v011 - v010, dx10 = v101 - v100, dx11 = v111 - v110;
    float c00 = fmaf(tx, dx00, v000), c01 = fmaf(tx, dx01, v010);
    float c10 = fmaf(tx, dx10, v100), c11 = fmaf(tx, dx11, v110);
    float dy0 = c01 - c00, dy1 = c11 - c10;
    float e0 = fmaf(ty, dy0, c00), e1 = fmaf(ty, dy1, c10);
    Samp3 s;
    s.dz = e1 - e0;
    s.v = fmaf(tz, s.dz, e0);
    s.dy = fmaf(tz, dy1 - dy0, dy0);
    float gx0 = fmaf(ty, dx01 - dx00, dx00), gx1 = fmaf(ty, dx11 - dx10, dx10);
    s.dx = fmaf(tz, gx1 - gx0, gx0);
    return s;
}

// zero-padded trilinear sample, branch-free form (every corner clamped + predicated): the general path of sample3, and the one to call
// where several independent samples should have their loads in flight together (no wave-level branch between them)
__device__ __forceinline__ Samp3 sample3_padded(const float *__restrict__ mov, int D, int H, int W, float ix, float iy, float iz)
{
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    float tx = ix - fx, ty = iy - fy, tz = iz - fz;
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    bool bx0 = (unsigned)x0 < (unsigned)W, bx1 = (unsigned)x1 < (unsigned)W;
    bool by0 = (unsigned)y0 < (unsigned)H, by1 = (unsigned)y1 < (unsigned)H;
    bool bz0 = (unsigned)z0 < (unsigned)D, bz1 = (unsigned)z1 < (unsigned)D;
    int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    int cz0 = min(max(z0, 0), D - 1), cz1 = min(max(z1, 0), D - 1);
    const float *r00 = mov + ((size_t)cz0 * H + cy0) * W, *r01 = mov + ((size_t)cz0 * H + cy1) * W;
    const float *r10 = mov + ((size_t)cz1 * H + cy0) * W, *r11 = mov + ((size_t)cz1 * H + cy1) * W;
    float v000 = (bz0 & by0 & bx0) ? r00[cx0] : 0.f, v001 = (bz0 & by0 & bx1) ? r00[cx1] : 0.f;
    float v010 = (bz0 & by1 & bx0) ? r01[cx0] : 0.f, v011 = (bz0 & by1 & bx1) ? r01[cx1] : 0.f;
    float v100 = (bz1 & by0 & bx0) ? r10[cx0] : 0.f, v101 = (bz1 & by0 & bx1) ? r10[cx1] : 0.f;
    float v110 = (bz1 & by1 & bx0) ? r11[cx0] : 0.f, v111 = (bz1 & by1 & bx1) ? r11[cx1] : 0.f;
    return lerp3(v000, v001, v010, v011, v100, v101, v110, v111, tx, ty, tz);
}

__device__ __forceinline__ Samp3 sample3(const float *__restrict__ mov, int D, int H, int W, float ix, float iy,
                                         float iz)
{
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    float tx = ix - fx, ty = iy - fy, tz = iz - fz;
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    bool interior = ((unsigned)x0 < (unsigned)(W - 1)) & ((unsigned)y0 < (unsigned)(H - 1)) &
                    ((unsigned)z0 < (unsigned)(D - 1));
    const size_t HW = (size_t)H * W;
    if (__all(interior)) {
        const float *p = mov + ((size_t)z0 * H + y0) * W + x0;
        float v000 = p[0], v001 = p[1], v010 = p[W], v011 = p[W + 1];
        const float *q = p + HW;
        float v100 = q[0], v101 = q[1], v110 = q[W], v111 = q[W + 1];
        return lerp3(v000, v001, v010, v011, v100, v101, v110, v111, tx, ty, tz);
    }
    return sample3_padded(mov, D, H, W, ix, iy, iz);
}

__device__ __forceinline__ Samp2 sample2(const float *__restrict__ mov, int H, int W, float ix, float iy)
{
    float fx = floorf(ix), fy = floorf(iy);
    float tx = ix - fx, ty = iy - fy;
    int x0 = (int)fx, y0 = (int)fy;
    int x1 = x0 + 1, y1 = y0 + 1;
    bool bx0 = (unsigned)x0 < (unsigned)W, bx1 = (unsigned)x1 < (unsigned)W;
    bool by0 = (unsigned)y0 < (unsigned)H, by1 = (unsigned)y1 < (unsigned)H;
    int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    const float *r0 = mov + (size_t)cy0 * W, *r1 = mov + (size_t)cy1 * W;
    float v00 = (by0 & bx0) ? r0[cx0] : 0.f, v01 = (by0 & bx1) ? r0[cx1] : 0.f;
    float v10 = (by1 & bx0) ? r1[cx0] : 0.f, v11 = (by1 & bx1) ? r1[cx1] : 0.f;
    float dx0 = v01 - v00, dx1 = v11 - v10;
    float c0 = fmaf(tx, dx0, v00), c1 = fmaf(tx, dx1, v10);
    Samp2 s;
    s.dy = c1 - c0;
    s.v = fmaf(ty, s.dy, c0);
    s.dx = fmaf(ty, dx1 - dx0, dx0);
    return s;
}

// grid_sample's un-normalisation (align_corners=False), written with the SAME roundings as the
// ATen CPU kernels the reference runs on: 3-D scalar path ((c+1)*S-1)/2 (the final fma(a,.5,-.5)
// is bitwise (a-1)/2), 2-D vectorised path fma(c+1, S/2, -0.5).  Mathematically identical to a
// single fma, but at theta = identity (the start of every affine run) each sample sits exactly on
// a voxel, where the trilinear derivative is one-sided and the side is decided by this last bit.
template <int ND>
__device__ __forceinline__ float unnorm(float c, float S)
{
    if constexpr (ND == 3) return fmaf((c + 1.0f) * S, 0.5f, -0.5f);
    else return fmaf(c + 1.0f, 0.5f * S, -0.5f);
}

// base coordinate of affine_grid(align_corners=False)
__device__ __forceinline__ float base_coord(const float *__restrict__ tab, int i, int S)
{
    return tab ? tab[i] : (float)(2 * i + 1) / (float)S - 1.0f;
}

// ------------------------------------------------------------------------------------------
// Block reduction of NV per-thread floats -> out[NV] (written by the first NV threads).
// Deterministic (fixed order).  Staged through LDS in chunks of CH values so the footprint
// stays at TRX_WAVES*CH*65 floats.
// ------------------------------------------------------------------------------------------
template <int NV, int CH = 16>
__device__ __forceinline__ void block_reduce_store(const float (&vals)[NV], float *__restrict__ out)
{
    __shared__ float red[TRX_WAVES][CH][65];
    __shared__ float wsum[TRX_WAVES][CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
            for (int w = 0; w < TRX_WAVES; w++) s += wsum[w][tid];
            out[c0 + tid] = s;
        }
        __syncthreads();
    }
}

}  // namespace trx
