// Micro-test (development): does global_load_lds_dwordx4 accept global addresses that are only 4-byte aligned?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_align.hip -o build/dma_align && build/dma_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k(const float *src, float *out, int off)
{
    __shared__ __attribute__((aligned(16))) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -1.f;
    __syncthreads();
    __builtin_amdgcn_global_load_lds(src + off + threadIdx.x * 4, lds + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 256, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    volatile float *vp = lds;   // volatile: the compiler does not see the DMA writes
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = vp[i];
}

int main()
{
    float *src, *out;
    std::vector<float> h(1024), o(256);
    for (int i = 0; i < 1024; i++) h[i] = (float)i;
    hipMalloc(&src, 4096); hipMalloc(&out, 1024);
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    for (int off : {0, 1, 2, 3, 5}) {
        hipMemset(out, 0, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, off);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; i++) bad += (o[i] != (float)(i + off));
        printf("offset %d floats: %s, %d mismatches (first values %g %g %g %g)\n", off, hipGetErrorString(e), bad, o[0], o[1], o[2], o[3]);
    }
    return 0;
}
