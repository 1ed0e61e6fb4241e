// Micro-benchmark (development): issue rate of the VALU / LDS instructions the F1 tile kernel is made of, per SIMD, at 1, 2 and 4
// resident waves per SIMD.  Answers: does a wave64 v_fma_f32 cost 2 or 4 cycles of a SIMD, is v_pk_fma_f32 one slot or two, and
// what do v_cvt_flr / v_fract / v_mad_i32_i24 / v_readlane / ds_read2_b32 cost.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { FMA, PKFMA, PKADD, PKMUL, ADD, CVTFLR, FRACT, MAD24, LSHLADD, READLANE, DSREAD2, DSREAD64, MIX };
static const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f32", "v_cvt_flr_i32_f32", "v_fract_f32",
                              "v_mad_i32_i24", "v_lshl_add_u32", "v_readlane_b32", "ds_read2_b32", "ds_read_b64", "fma+pk_fma 1:1"};

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters, float seed)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    float a[8];
    f2 p[8];
    int n[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = f2{seed + i, seed - i}; n[i] = (int)threadIdx.x + i; }
    const float b = seed * 1.0001f, c = seed * 0.5f;
    const f2 pb = {b, c};
    const unsigned la = (unsigned)(uintptr_t)lds + (threadIdx.x & 63) * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
                if constexpr (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if constexpr (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if constexpr (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if constexpr (OP == CVTFLR) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
                if constexpr (OP == FRACT) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
                if constexpr (OP == MAD24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));
                if constexpr (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if constexpr (OP == READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(n[i])); asm volatile("" :: "s"(s)); }
                if constexpr (OP == DSREAD2) asm volatile("ds_read2_b32 %0, %1 offset1:44" : "=v"(p[i]) : "v"(la));
                if constexpr (OP == DSREAD64) asm volatile("ds_read_b64 %0, %1" : "=v"(p[i]) : "v"(la));
                if constexpr (OP == MIX) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
                }
            }
            if constexpr (OP == DSREAD2 || OP == DSREAD64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + n[i];
    if (s == 1.2345f) out[1] = 1;
    if (threadIdx.x == 0) out[2 + blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(unsigned long long *out, int blocks_per_cu)   // = waves per SIMD: ONE block of 256 * n threads per CU (a grid of 256 blocks lands one per CU)
{
    const int iters = 2000, blocks = 256, threads = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 + blocks);
    hipMemcpy(h.data(), out, (2 + blocks) * 8, hipMemcpyDeviceToHost);
    double ticks = 0;
    for (int i = 0; i < blocks; i++) ticks += (double)h[2 + i];
    ticks /= blocks;
    const double per_wave = (double)iters * 32 * (OP == MIX ? 2 : 1);      // instructions per wave
    const double per_simd = per_wave * blocks_per_cu;                      // one wave of each block per SIMD
    printf("%-20s %d waves/SIMD: %7.2f cycles per instruction per SIMD (in-kernel ticks %.0f, wall %.1f us -> %.2f GHz)\n", names[OP], blocks_per_cu,
           ticks / per_simd, ticks, ms * 1e3, ticks / (ms * 1e3) / 1e3);
}

template <int OP>
static void all(unsigned long long *out)
{
    run<OP>(out, 1); run<OP>(out, 2); run<OP>(out, 4);
}

int main()
{
    unsigned long long *out;
    hipMalloc(&out, (2 + 4096) * 8);
    hipMemset(out, 0, (2 + 4096) * 8);
    all<FMA>(out); all<PKFMA>(out); all<PKADD>(out); all<PKMUL>(out); all<ADD>(out); all<MIX>(out); all<CVTFLR>(out); all<FRACT>(out);
    all<MAD24>(out); all<LSHLADD>(out); all<READLANE>(out); all<DSREAD2>(out); all<DSREAD64>(out);
    return 0;
}
