// MFMA issue-rate probe (gfx950): cycles per v_mfma_f32_16x16x32_bf16 for NACC independent accumulators, A operand reused for RA consecutive MFMAs,
// with and without an s_waitcnt after every G MFMAs.  One wave per SIMD (256 threads, 1 workgroup per CU).  s_memtime stamps around 64 rounds.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int NACC, int RA, int G>
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NA = NACC / RA;                 // A fragments per round; B fragments: RA
    bf16x8 a[NA], b[RA];
    for (int i = 0; i < NA; i++) a[i] = __builtin_bit_cast(bf16x8, src[threadIdx.x + 256 * i]);
    for (int j = 0; j < RA; j++) b[j] = __builtin_bit_cast(bf16x8, src[threadIdx.x + 256 * (NA + j)]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // operands in registers before the clock starts
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j < RA; j++) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i * RA + j]) : "v"(a[i]), "v"(b[j]));      // accumulators pinned in AGPRs: no copies
                if (G > 0 && ((i * RA + j) % G) == G - 1) asm volatile("s_waitcnt lgkmcnt(0)");
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int RA, int G>
static void run(const uint4* src, float* sink, unsigned long long* cyc, int grid) {
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<NACC, RA, G>), dim3(grid), dim3(256), 0, 0, src, sink, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("grid %3d  %2d accumulators, A reused %d x, s_waitcnt every %2d MFMAs: median %.2f cycles per MFMA (p10 %.2f, p90 %.2f)\n", grid, NACC, RA, G,
           h[h.size() / 2] / (64.0 * NACC), h[h.size() / 10] / (64.0 * NACC), h[h.size() * 9 / 10] / (64.0 * NACC));
}

int main() {
    uint4* src; float* sink; unsigned long long* cyc;
    hipMalloc(&src, 256 * 64 * 16); hipMemset(src, 0x3c, 256 * 64 * 16);
    hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    for (int grid : {256, 64}) {
        run<32, 4, 0>(src, sink, cyc, grid); run<32, 4, 16>(src, sink, cyc, grid);
        run<16, 4, 0>(src, sink, cyc, grid); run<16, 2, 0>(src, sink, cyc, grid); run<16, 2, 8>(src, sink, cyc, grid); run<16, 4, 16>(src, sink, cyc, grid);
        run<8, 2, 0>(src, sink, cyc, grid); run<4, 2, 0>(src, sink, cyc, grid);
    }
    return 0;
}
