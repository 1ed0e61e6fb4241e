// Micro-benchmark (development): does an fp32 MFMA (v_mfma_f32_4x4x1_16b_f32: 16 blocks of 4x4 outer products, K = 1)
// co-issue with a saturated VALU stream on gfx950, or does it take VALU issue slots?  And what do the vector instructions of the F1
// loop cost per SIMD when 2 / 4 waves share it (wall clock, since the chip's clock moves with the load)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_coissue.hip -o build/mfma_coissue && build/mfma_coissue
// Part 1: per loop iteration NV independent VALU instructions (packed or plain fp32) with NM MFMAs spread between them.
// Part 2: one instruction kind at a time, 32 independent instances per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// KIND 0: v_pk_fma_f32, 1: v_fma_f32 (VOP3), 2: v_add_f32 (VOP2), 3: the F1 loop's mix (20 packed + 32 plain per 52)
template <int KIND, int NV, int NM>
__global__ __launch_bounds__(1024) void coissue(unsigned long long *out, int iters, float seed)
{
    float a[12];
    f2 p[12];
    f4 acc[9];
#pragma unroll
    for (int i = 0; i < 12; i++) { a[i] = seed + i + threadIdx.x; p[i] = f2{seed + i, seed - i}; }
#pragma unroll
    for (int i = 0; i < 9; i++) acc[i] = f4{seed, seed, seed, seed};
    const float b = seed * 1.0001f, c = seed * 0.5f;
    const f2 pb = {b, c};
    const float wa = seed + (threadIdx.x & 3), wb = seed * (threadIdx.x & 63);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        int m = 0;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int r = i % 12;
            if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[r]) : "v"(pb));
            if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            if constexpr (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            if constexpr (KIND == 3) {
                if (i % 13 < 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[r]) : "v"(pb));
                else if (i % 13 < 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            }
            // spread the NM MFMAs evenly between the VALU instructions
            if (NM > 0 && NV > 0 && ((i + 1) * NM) / NV > m) { MFMA(acc[m % 9], wa, wb); m++; }
        }
        if constexpr (NV == 0) {
#pragma unroll
            for (int k = 0; k < NM; k++) MFMA(acc[k % 9], wa, wb);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) s += a[i] + p[i].x + p[i].y;
#pragma unroll
    for (int i = 0; i < 9; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 1.2345f) out[1] = 1;
    if (threadIdx.x == 0) { out[2 + 2 * blockIdx.x] = t1 - t0; out[3 + 2 * blockIdx.x] = r1 - r0; }
}

enum Op { ADD, SUB, MUL, FMAC, FMA, PKFMA, PKADD, PKMUL, FLOOR, FRACT, CVTFLR, CVTI, CVTU, PERM, MADU24, MADI24, MULU24, ADDU, AND, LSHL, LSHLADD, ADDLSHL, ADD3, MOV, CNDMASK, BFE,
          READLANE, DPPADD, MFMA4, ADDS, FMAS, FMAK, FMACS, PKFMAB, PKMULS, FMA3, DSR32, DSR2, DSR2ST64, DSR64, DSR128, NOPS };
static const char *opname[] = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_floor_f32", "v_fract_f32",
                               "v_cvt_flr_i32_f32", "v_cvt_i32_f32", "v_cvt_u32_f32", "v_perm_b32 (s, v, v)", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_u32_u24", "v_add_u32", "v_and_b32",
                               "v_lshlrev_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_add3_u32", "v_mov_b32", "v_cndmask_b32", "v_bfe_u32", "v_readlane_b32", "v_add_f32 dpp quad_perm",
                               "v_mfma_f32_4x4x1_16b_f32", "v_add_f32 v,s,v", "v_fma_f32 v,v,s,v", "v_fma_f32 v,v,2.0,v", "v_fmac_f32 v,s,v", "v_pk_fma_f32 a,a,b,a", "v_pk_mul_f32 v,v,s", "v_fma_f32 d,a,b,c (4 regs)", "ds_read_b32", "ds_read2_b32 off 0,1", "ds_read2st64_b32 off 0,45", "ds_read_b64", "ds_read_b128"};

template <int OP>
__global__ __launch_bounds__(1024) void opcost(unsigned long long *out, int iters, float seed)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    float a[8];
    f2 p[8];
    f4 q[8];
    int n[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = f2{seed + i, seed - i}; n[i] = (int)threadIdx.x + i; q[i] = f4{seed, seed, seed, seed}; }
    const float b = seed * 1.0001f, c = seed * 0.5f;
    const f2 pb = {b, c};
    int s1;
    asm volatile("s_mov_b32 %0, 0x07060504" : "=s"(s1));
    unsigned long long s64;
    asm volatile("s_mov_b64 %0, 0x3f800000" : "=s"(s64));
    const unsigned la = (unsigned)(uintptr_t)lds + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 288;   // consecutive floats per lane: the gather's pattern at theta = I
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
                if constexpr (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if constexpr (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if constexpr (OP == FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
                if constexpr (OP == FRACT) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
                if constexpr (OP == CVTFLR) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
                if constexpr (OP == CVTI) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
                if constexpr (OP == CVTU) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
                if constexpr (OP == PERM) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(n[i]) : "s"(s1), "v"(n[(i + 1) & 7]));
                if constexpr (OP == MADU24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));
                if constexpr (OP == MADI24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));
                if constexpr (OP == MULU24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if constexpr (OP == ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if constexpr (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if constexpr (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(n[i]));
                if constexpr (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if constexpr (OP == ADDLSHL) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if constexpr (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));
                if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(n[(i + 1) & 7]));
                if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 7]) : "vcc");
                if constexpr (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(n[i]));
                if constexpr (OP == READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(n[i])); asm volatile("" :: "s"(s)); }
                if constexpr (OP == DPPADD) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (OP == MFMA4) MFMA(q[i], b, c);
                if constexpr (OP == ADDS) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s1));
                if constexpr (OP == FMAS) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s1), "v"(c));
                if constexpr (OP == FMAK) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a[i]) : "v"(c));
                if constexpr (OP == FMACS) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s1), "v"(c));
                if constexpr (OP == PKFMAB) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(pb));
                if constexpr (OP == PKMULS) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "s"(s64));
                if constexpr (OP == FMA3) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(b), "v"(c));
                if constexpr (OP == DSR32) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(la));
                if constexpr (OP == DSR2) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(p[i]) : "v"(la));
                if constexpr (OP == DSR2ST64) asm volatile("ds_read2st64_b32 %0, %1 offset1:45" : "=v"(p[i]) : "v"(la));
                if constexpr (OP == DSR64) asm volatile("ds_read_b64 %0, %1" : "=v"(p[i]) : "v"(la * 2));
                if constexpr (OP == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(la * 4));
            }
            if constexpr (OP >= DSR32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + n[i] + q[i].x + q[i].y + q[i].z + q[i].w;
    if (s == 1.2345f) out[1] = 1;
    if (threadIdx.x == 0) { out[2 + 2 * blockIdx.x] = t1 - t0; out[3 + 2 * blockIdx.x] = r1 - r0; }
}

struct Res { double wall_us, ticks, real_us; };
template <typename K>
static Res launch(K kern, unsigned long long *out, int wps, int iters)
{
    const int blocks = 256, threads = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 + 2 * blocks);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0;
    for (int i = 0; i < blocks; i++) { ticks += (double)h[2 + 2 * i]; real += (double)h[3 + 2 * i]; }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return {ms * 1e3, ticks / blocks, real / blocks / 100.0};   // s_memrealtime: 100 MHz
}

template <int KIND, int NV, int NM>
static void co(unsigned long long *out, const char *name)
{
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {
        const Res r = launch(coissue<KIND, NV, NM>, out, wps, iters);
        const double ns_iter_simd = r.real_us * 1e3 / iters / wps;   // time one SIMD needs for one wave's iteration
        printf("%-28s NV %2d NM %2d  %d waves/SIMD: in-kernel %8.1f us  wall %8.1f us  %7.2f ns per iteration per SIMD  (s_memtime %.0f ticks = %.2f per ns)\n", name, NV, NM, wps, r.real_us,
               r.wall_us, ns_iter_simd, r.ticks, r.ticks / (r.real_us * 1e3));
    }
}

template <int OP>
static void oc(unsigned long long *out)
{
    const int iters = 2000;
    printf("%-26s", opname[OP]);
    for (int wps : {1, 2, 4}) {
        const Res r = launch(opcost<OP>, out, wps, iters);
        const double ns = (r.wall_us - 6.0) * 1e3 / (iters * 32.0 * wps);   // wall clock (- ~6 us of launch): thread 0's own stamps only time the OLDEST wave, which the arbiter favours
        printf("  %dw: %6.3f ns = %5.2f cyc", wps, ns, ns * r.ticks / (r.real_us * 1e3));
    }
    printf("\n");
}

template <int... OPS>
static void oc_all(unsigned long long *out) { (oc<OPS>(out), ...); }

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long *out;
    hipMalloc(&out, (2 + 2 * 4096) * 8);
    hipMemset(out, 0, (2 + 2 * 4096) * 8);
    printf("# part 1: VALU stream with MFMA 4x4x1 fp32 spread through it (per loop iteration: NV VALU + NM MFMA)\n");
    co<0, 36, 0>(out, "v_pk_fma_f32");
    co<0, 36, 9>(out, "v_pk_fma_f32 + mfma");
    co<0, 45, 0>(out, "v_pk_fma_f32");
    co<1, 36, 0>(out, "v_fma_f32");
    co<1, 36, 9>(out, "v_fma_f32 + mfma");
    co<1, 45, 0>(out, "v_fma_f32");
    co<2, 36, 0>(out, "v_add_f32");
    co<2, 36, 9>(out, "v_add_f32 + mfma");
    co<2, 36, 18>(out, "v_add_f32 + mfma");
    co<3, 39, 0>(out, "F1 mix (15 pk, 12 add, 12 fma)");
    co<3, 39, 9>(out, "F1 mix + mfma");
    co<3, 52, 0>(out, "F1 mix (20 pk, 16 add, 16 fma)");
    co<0, 0, 9>(out, "mfma only");
    co<0, 0, 18>(out, "mfma only");
    printf("# part 2: one instruction kind, 32 independent instances per iteration\n");
    oc_all<ADD, SUB, MUL, FMAC, FMA, PKFMA, PKADD, PKMUL, FLOOR, FRACT, CVTFLR, CVTI, CVTU, PERM, MADU24, MADI24, MULU24, ADDU, AND, LSHL, LSHLADD, ADDLSHL, ADD3, MOV, CNDMASK, BFE, READLANE, DPPADD, MFMA4,
           ADDS, FMAS, FMAK, FMACS, PKFMAB, PKMULS, FMA3, DSR32, DSR2, DSR2ST64, DSR64, DSR128>(out);
    return 0;
}
