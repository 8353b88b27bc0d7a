// Development micro-benchmark: HBM write rate of a [M][256] bf16 image stored (a) the way a stack layer of the forward-with-save
// kernel stores it -- per instruction, lane (r31, h) writes 16 bytes of row r31, chunk 2k+h: 32 rows x 32 contiguous bytes --
// and (b) fully coalesced (a wave instruction writes 1 KB = two whole rows).  One wave per SIMD (256 threads, 1 block per CU),
// like the fused kernel.   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(uint16_t* out, long long M) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long ntiles = M / 128;
    const i32x4 v = {tid, 1, 2, 3};
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        unsigned char* base = reinterpret_cast<unsigned char*>(out) + tile * 128 * 512;
        if (MODE == 0) {                       // fragment pattern: wave w stores row tile w, 16 k-steps
            const int r31 = lane & 31, h = lane >> 5;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                *reinterpret_cast<i32x4*>(base + (32 * wave + r31) * 512 + (2 * kk + h) * 16) = v;
        } else {                               // coalesced: wave w stores rows [32w, 32w+32): 2 rows per instruction
#pragma unroll
            for (int i = 0; i < 16; ++i)
                *reinterpret_cast<i32x4*>(base + (32 * wave + 2 * i + (lane >> 5)) * 512 + (lane & 31) * 16) = v;
        }
    }
}

int main() {
    const long long M = 196608LL * 8;          // 805 MB
    uint16_t* d;
    hipMalloc(&d, M * 512);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(s);
            for (int i = 0; i < 5; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, M);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, M);
            }
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            printf("mode %d (%s): %.1f us per 805 MB = %.2f TB/s\n", mode, mode ? "coalesced rows" : "fragment pattern", ms / 5 * 1e3, M * 512.0 / (ms / 5 * 1e-3) / 1e12);
        }
    }
    return 0;
}
