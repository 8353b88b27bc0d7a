// Development microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 with N filler VALU ops (and optional LDS reads)
// between MFMAs, one wave per SIMD, two alternating accumulators.  Build: hipcc --offload-arch=gfx950 -O3 mfma_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NR, int AG = 0, int NW = 0>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63;
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, (short)lane};
    f32x16 c0 = {0}, c1 = {0};
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = lane * 0.5f + i;
    bf16x8 r = b;
    if (AG) asm volatile("" : "+a"(a));                                      // keep the A operand in AGPRs
    bf16x8 ring[4] = {b, b, b, b};
    const unsigned char* p = lds + lane * 16;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (NR == 3) { ring[(j + 3) & 3] = *reinterpret_cast<const bf16x8*>(p + (((j + 3) * 1024) & 65535)); r = ring[j & 3]; }
            if (NR == 4) { ring[(2 * j + 3) & 3] = *reinterpret_cast<const bf16x8*>(p + (((2 * j + 3) * 1024) & 65535)); r = ring[(2 * j) & 3]; }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, r, c0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) f[v & 7] = fmaxf(f[v & 7] * 1.0001f, 0.5f);
            if (NR == 1 || NR == 2) r = *reinterpret_cast<const bf16x8*>(p + ((j * 1024) & 65535));
            __builtin_amdgcn_sched_barrier(0);
            if (NR == 4) { ring[(2 * j + 4) & 3] = *reinterpret_cast<const bf16x8*>(p + (((2 * j + 4) * 1024) & 65535)); r = ring[(2 * j + 1) & 3]; }
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, r, c1, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) f[(v + 4) & 7] = fmaxf(f[(v + 4) & 7] * 1.0001f, 0.5f);
            if (NR == 2) r = *reinterpret_cast<const bf16x8*>(p + ((j * 1024 + 512) & 65535));
            if (NW && (j & 1)) *reinterpret_cast<float2*>(lds + 32768 + lane * 8 + ((j * 512) & 16383)) = make_float2(f[0], f[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int NR, int AG = 0, int NW = 0>
void run(float* out, long long* cyc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, NR, AG, NW><<<256, 256>>>(out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, NR, AG, NW><<<256, 256>>>(out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 64.0 * 32;
    printf("VALU/gap=%d NR=%d AG=%d NW=%d: %.1f counter ticks per MFMA, %.2f ns per MFMA (kernel %.1f us) -> counter %.2f GHz\n", NV * 2, NR, AG, NW, c / n,
           ms * 1e6 / n, ms * 1e3, c / (ms * 1e6));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0, 4>(out, cyc); run<0, 3>(out, cyc);
    run<0, 0>(out, cyc); run<3, 0>(out, cyc); run<4, 0>(out, cyc);
    run<0, 0, 1>(out, cyc); run<2, 0, 1>(out, cyc);
    run<0, 3>(out, cyc); run<1, 3>(out, cyc); run<2, 3>(out, cyc);
    run<0, 3, 0, 1>(out, cyc); run<1, 3, 0, 1>(out, cyc); run<2, 3, 0, 1>(out, cyc); run<2, 3, 1, 1>(out, cyc);
    return 0;
}
