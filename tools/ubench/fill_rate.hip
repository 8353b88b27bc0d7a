// Development microbenchmark: how fast a CU fills LDS from an L2-resident buffer, (a) with LDS-DMA (global_load_lds_dwordx4: no
// registers, the form the GEMM kernels stage their operands in) and (b) through registers (global_load_dwordx4 + ds_write_b128),
// for several numbers of 16-byte requests in flight per lane.  Every workgroup reads its own 64 KB window of a 16 MB buffer over
// and over (L2 / MALL hits after the first pass), 256 threads, W workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 fill_rate.hip -o fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ void copy16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// MODE 0: LDS-DMA, DEPTH copies per lane in flight; MODE 1: register-staged, DEPTH loads per lane in flight;
// MODE 2 / 3: LDS-DMA with the GEMM kernels' row-strided requests: a wave instruction fetches 8 rows x 128 B (MODE 2: the 64-wide
// k-tiles) or 16 rows x 64 B (MODE 3: the 32-wide k-tiles) of a matrix with 2 016-byte rows
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned char* base = src + (size_t)(blockIdx.x % 256) * 65536;
    float s = 0.0f;
    if (MODE >= 2) {
        constexpr int LPR = MODE == 2 ? 8 : 4;                   // lanes per row
        const unsigned char* mat = src + (size_t)(blockIdx.x % 64) * 64 * 2016;   // 64-row blocks of a 4 096 x 1 008 bf16 matrix
        const int row = lane / LPR, c = lane % LPR;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int r0 = ((d * 4 + wave) * (64 / LPR)) & 63, k0 = ((it * (LPR * 16)) % 1920);
                copy16(mat + (size_t)(r0 + row) * 2016 + k0 + c * 16, lds + ((d * 4 + wave) * 1024));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s = reinterpret_cast<float*>(lds)[tid];
    } else if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)                      // wave w, copy d: 1 KB at window offset ((it * DEPTH + d) * 4 + w) KB
                copy16(base + ((((it * DEPTH + d) * 4 + wave) * 1024) & 65535) + lane * 16, lds + ((d * 4 + wave) * 1024));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s = reinterpret_cast<float*>(lds)[tid];
    } else {
        uint4 r[DEPTH];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                r[d] = *reinterpret_cast<const uint4*>(base + ((((it * DEPTH + d) * 4 + wave) * 1024) & 65535) + lane * 16);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<uint4*>(lds + ((d * 4 + wave) * 1024) + lane * 16) = r[d];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        s = reinterpret_cast<float*>(lds)[tid];
    }
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int DEPTH>
void run(const unsigned char* src, float* out, int wg_per_cu) {
    const int iters = 4096 / DEPTH;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k<MODE, DEPTH><<<grid, 256, DEPTH * 4096>>>(src, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) k<MODE, DEPTH><<<grid, 256, DEPTH * 4096>>>(src, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double bytes = (double)grid * iters * DEPTH * 4096.0;
    printf("%s depth %2d, %d workgroups per CU: %7.1f GB/s per CU, %6.2f TB/s over the card\n",
           MODE == 0 ? "LDS-DMA  " : (MODE == 1 ? "registers" : (MODE == 2 ? "DMA 8x128" : "DMA 16x64")), DEPTH,
           wg_per_cu, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
}

int main() {
    unsigned char* src; float* out;
    hipMalloc(&src, 16 << 20); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemset(src, 1, 16 << 20);
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 4>(src, out, w); run<0, 8>(src, out, w); run<0, 12>(src, out, w);
        run<1, 4>(src, out, w); run<1, 8>(src, out, w); run<1, 12>(src, out, w);
        run<2, 4>(src, out, w); run<2, 12>(src, out, w); run<3, 4>(src, out, w); run<3, 12>(src, out, w);
    }
    return 0;
}
