// Development microbenchmark: what a 16-byte-per-lane weight-fragment load costs beside MFMAs (one wave per SIMD).
// MODE 0: no loads; 1: global_load_dwordx4 -> VGPR; 2: global_load_dwordx4 -> AGPR; 3: buffer_load_dwordx4 (32-bit offsets);
// EVERY: one load per EVERY MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 vmem_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int EVERY, int LDS = 0>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* __restrict__ w, float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* lp = lds + lane * 16;
    bf16x8 xr[4];
    bf16x8 b = {1, 1, 1, 1, 1, 1, 1, (short)lane};
    f32x16 c0 = {0}, c1 = {0};
    bf16x8 ring[32];
    for (int i = 0; i < 32; ++i) ring[i] = b;
    for (int i = 0; i < 4; ++i) xr[i] = b;
    const bf16x8* p = w + wave * 2048 + lane;                    // 32 KB per wave, 1 KB per fragment
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 20, 0x27000);
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            bf16x8 a = ring[j];
            if (MODE == 2) asm volatile("" : "+a"(a));
            bf16x8 x = b;
            if (LDS) {
                if (!(j & 1)) xr[((j >> 1) + 3) & 3] = *reinterpret_cast<const bf16x8*>(lp + ((((j >> 1) + 3) * 1024) & 65535));
                x = xr[(j >> 1) & 3];
            }
            if (j & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE != 0 && (j % EVERY) == EVERY - 1) {
                const int f = ((j / EVERY) + it) & 31;
                if (MODE == 3) {
                    i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (wave * 2048 + lane) * 16 + f * 1024, 0, 0);
                    ring[j] = __builtin_bit_cast(bf16x8, v);
                } else {
                    ring[j] = p[f * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int EVERY, int LDS = 0>
void run(const bf16x8* w, float* out, long long* cyc) {
    k<MODE, EVERY, LDS><<<256, 256>>>(w, out, cyc);
    hipDeviceSynchronize();
    k<MODE, EVERY, LDS><<<256, 256>>>(w, out, cyc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d, one load per %d MFMAs, lds reads %d: %.1f cycles per MFMA\n", MODE, EVERY, LDS, c / (64.0 * 32));
}

int main() {
    bf16x8* w; float* out; long long* cyc;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0, 4>(w, out, cyc);
    run<0, 4, 1>(w, out, cyc); run<1, 4, 1>(w, out, cyc); run<1, 2, 1>(w, out, cyc); run<3, 4, 1>(w, out, cyc);
    run<1, 4>(w, out, cyc); run<1, 2>(w, out, cyc); run<1, 1>(w, out, cyc);
    run<2, 4>(w, out, cyc); run<2, 2>(w, out, cyc);
    run<3, 4>(w, out, cyc); run<3, 2>(w, out, cyc); run<3, 1>(w, out, cyc);
    return 0;
}
