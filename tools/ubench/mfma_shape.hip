// Development microbenchmark: the two fp16 MFMA shapes at the parity kernel's operating point -- fragments re-read from LDS
// (ds_read_b128), 8 waves per workgroup (two per SIMD), random operands, long launches -- FLOP/s and cycles of
// v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 (the guide: MI355X_MICROARCH.md, DVFS give-back item 7).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const _Float16* __restrict__ src, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 131072 / 16; i += 512)
        reinterpret_cast<f16x8*>(lds)[i] = reinterpret_cast<const f16x8*>(src)[(blockIdx.x * 37 + i) & 65535];
    __syncthreads();
    f16x8 a[4];
    for (int i = 0; i < 4; ++i) a[i] = reinterpret_cast<const f16x8*>(src)[(wave * 64 + lane + 977 * i) & 65535];
    const unsigned char* p = lds + lane * 16 + wave * 1024;
    long long t0 = __builtin_readcyclecounter();
    float s = 0;
    if (SHAPE == 32) {
        f32x16 c[4] = {{0}, {0}, {0}, {0}};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f16x8 b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const f16x8*>(p + ((j * 4 + t) * 8192 & 131071));
#pragma unroll
                for (int t = 0; t < 4; ++t)                       // 4 tiles x 3 terms, as the kernel's k-step
#pragma unroll
                    for (int q = 0; q < 3; ++q) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q], b[t], c[t], 0, 0, 0);
            }
        }
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += c[t][i];
    } else {
        f32x4 c[16];
        for (int t = 0; t < 16; ++t) c[t] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                        // one 32-k step = two of the other shape's
                f16x8 b[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) b[t] = *reinterpret_cast<const f16x8*>(p + ((j * 8 + t) * 8192 & 131071));
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int q = 0; q < 3; ++q) c[2 * t + f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(q + f) & 3], b[t], c[2 * t + f], 0, 0, 0);
            }
        }
        for (int t = 0; t < 16; ++t) for (int i = 0; i < 4; ++i) s += c[t][i];
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE>
void run(const _Float16* src, float* out, long long* cyc, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<SHAPE><<<256, 512, 131072>>>(src, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) k<SHAPE><<<256, 512, 131072>>>(src, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // per wave and iteration: 8 x 12 MFMAs of 32x32x16 (32768 flop) | 4 x 48 MFMAs of 16x16x32 (16384 flop): equal flops
    const double flop = 256.0 * 8 * iters * 8 * 12 * 32768.0;
    printf("shape %2d: %8.3f ms  %7.1f TFLOP/s  %lld cycles (wave 0)  -> %.3f GHz effective, %.2f cycles per 32x32x16-equivalent\n", SHAPE, ms,
           flop / ms / 1e9, c, c / (ms * 1e6), (double)c / (iters * 96.0) / 2.0);
}

int main() {
    const int n = 65536 * 8;
    _Float16* h = (_Float16*)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    _Float16* src; float* out; long long* cyc;
    hipMalloc(&src, n * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    for (int r = 0; r < 3; ++r) {
        run<32>(src, out, cyc, 4000);
        run<16>(src, out, cyc, 4000);
    }
    return 0;
}
