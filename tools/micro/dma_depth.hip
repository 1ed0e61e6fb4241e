// Micro-benchmark (development): how many LDS-DMA loads does ONE wave keep in flight?
// A single wave issues N global_load_lds_dwordx4 (or N plain global_load_dwordx4) back to back on L2-resident data and
// waits for all of them; ticks per batch vs N shows whether the loads pipeline (flat) or serialise (linear in N).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_depth.hip -o build/dma_depth && build/dma_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N, bool DMA>
__global__ __launch_bounds__(64) void k(const float *src, unsigned long long *out, int reps, size_t stride_floats, size_t ws_floats)
{
    __shared__ __attribute__((aligned(16))) float lds[16 * 256];
    const float *p0 = src + (size_t)blockIdx.x * ws_floats + threadIdx.x * 4;
    unsigned long long t = 0;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r = 0; r < reps + 1; r++) {
        // walk a per-block working set larger than the 32 KB L1 so that every batch misses L1 (and hits L2 when it fits)
        const float *p = p0 + ((size_t)r * N * 256) % (ws_floats - N * 256 + 1) / 256 * 256;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < N; i++) __builtin_amdgcn_global_load_lds(p + (size_t)i * stride_floats, lds + i * 256, 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            float4 v[N];
#pragma unroll
            for (int i = 0; i < N; i++) v[i] = *reinterpret_cast<const float4 *>(p + (size_t)i * stride_floats);
#pragma unroll
            for (int i = 0; i < N; i++) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (r > 0) t += t1 - t0;   // first pass warms L2
    }
    if (threadIdx.x == 0) out[blockIdx.x] = t / reps;
    if (acc.x == 12345.f) out[0] = (unsigned long long)lds[threadIdx.x];
}

template <int N, bool DMA>
void run(const float *src, unsigned long long *out, int blocks, size_t stride, size_t ws)
{
    hipLaunchKernelGGL((k<N, DMA>), dim3(blocks), dim3(64), 0, 0, src, out, 200, stride, ws);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%s ws=%4zuKB N=%2d blocks=%4d : %8.0f ticks per batch  (%6.1f per load)\n", DMA ? "lds-dma " : "vgpr    ", ws * 4 / 1024, N, blocks, s / blocks, s / blocks / N);
}

int main()
{
    float *src;
    unsigned long long *out;
    hipMalloc(&src, 1024);
    hipMalloc(&out, 4096 * 8);
    const size_t n2 = (size_t)2048 * 256 * 1024;
    hipFree(src);
    hipMalloc(&src, n2 * 4);
    hipMemset(src, 0, n2 * 4);
    // L2-resident but L1-thrashing: 12 KB per block x 8 blocks per CU = 96 KB per CU (> 32 KB L1), 3 MB per XCD (< 4 MB L2)
    for (int blocks : {2048, 4096}) {
        run<4, true>(src, out, blocks, 256, 12 * 256); run<8, true>(src, out, blocks, 256, 12 * 256);
        run<4, false>(src, out, blocks, 256, 12 * 256); run<8, false>(src, out, blocks, 256, 12 * 256);
    }
    for (size_t wsk : {16, 64, 1024}) {          // KB per block: L1-resident, L2-resident (16 MB at 256 blocks), HBM (256 MB)
        const size_t ws = wsk * 256;             // floats
        for (int blocks : {256, 2048}) {
            if ((size_t)blocks * ws > n2) continue;
            run<1, true>(src, out, blocks, 256, ws); run<4, true>(src, out, blocks, 256, ws); run<8, true>(src, out, blocks, 256, ws);
            run<16, true>(src, out, blocks, 256, ws);
            run<4, false>(src, out, blocks, 256, ws); run<16, false>(src, out, blocks, 256, ws);
        }
    }
    return 0;
}
