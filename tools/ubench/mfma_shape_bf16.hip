// Development microbenchmark: the bf16 inference kernel's operating point (dhaug_mlp.hip: stack_layer) in the two MFMA shapes.
// ONE wave per SIMD (256 threads, all of the CU's LDS), weights resident in registers, per 64 clocks of matrix issue one
// ds_read_b128 of activations, per 128 clocks one ds_write_b64 + 2 cvt_pk + 2 pk_max + one 16-byte buffer load of the next
// layer's weights -- pinned between the matrix instructions as the kernel pins them.  Prints FLOP/s and cycles per layer-tile
// (128 rows x 256 x 256 per workgroup = 4 096 clocks of matrix issue per wave).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape_bf16.hip -o mfma_shape_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32v2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_relu(float a, float b) {
    f32v2 f = {a, b};
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(f, bf16v2)), z));
}

// FILL bits: 1 LDS reads, 2 LDS writes, 4 weight loads, 8 pack + ReLU
template <int SHAPE, int FILL>
__global__ __launch_bounds__(256, 1) void k(const uint16_t* __restrict__ src, float* out, long long* cyc, int layers) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 163840 / 16; i += 256)
        reinterpret_cast<bf16x8*>(lds)[i] = reinterpret_cast<const bf16x8*>(src)[(blockIdx.x * 37 + i) & 65535];
    __syncthreads();
    bf16x8 w[32];                                                        // 128 registers of resident weights
    for (int i = 0; i < 32; ++i) w[i] = reinterpret_cast<const bf16x8*>(src)[(wave * 64 + lane + 977 * i) & 65535];
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(src), 0, 1 << 20, 0x27000);
    const int r31 = lane & 31, h = lane >> 5, x = lane & 15, q4 = lane >> 4;
    const unsigned char* rd32 = lds + (r31 * 512 | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4));
    const int rd16 = x * 512 | ((q4 ^ (x & 3)) << 4) | ((x & 12) << 4);                   // ^ (k-step << 6)
    unsigned char* wr32 = lds + 65536 + (r31 * 512 | (((4 * wave) ^ x) << 4) | (h << 3));
    const int wr16 = x * 512 | (((8 * wave + (q4 >> 1)) ^ x) << 4) | ((q4 & 1) << 3);      // ^ (feature tile << 5)
    float dm[16];                                                        // FILL & 32: the packs read these instead of the accumulators
    for (int i = 0; i < 16; ++i) { dm[i] = (float)src[lane + i * 64]; asm volatile("" : "+v"(dm[i])); }
    long long t0 = __builtin_readcyclecounter();
    float s = 0;
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.0f;
        bf16x8 fx[4];
#pragma unroll 1
        for (int l = 0; l < layers; ++l) {
            for (int i = 0; i < 3; ++i) fx[i] = *reinterpret_cast<const bf16x8*>(rd32 + (i << 5));
            uint32_t st0 = 0, st1 = 0;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const int sidx = mt * 16 + kk;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[mt & 1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t * 16 + kk], fx[sidx & 3], acc[mt & 1][t], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if ((FILL & 1) && t == 0)
                            fx[(sidx + 3) & 3] = *reinterpret_cast<const bf16x8*>(rd32 + ((((kk + 3) & 7) << 5) ^ (((kk + 3) & 8) << 5)) + ((mt * 16384) & 65535));
                        if ((FILL & 14) && t == 1) {
                            if (kk & 1) {
                                if (FILL & 4) w[16 + (kk >> 1)] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (l & 7) * 8192 + kk * 1024, 0));
                                if (FILL & 8) {
                                    if (FILL & 32) {
                                        st0 = pack_relu(dm[kk & 14], dm[(kk & 14) + 1]);
                                        st1 = pack_relu(dm[(kk + 2) & 14], dm[((kk + 2) & 14) + 1]);
                                    } else {
                                        st0 = pack_relu(acc[(mt + 1) & 1][0][kk & 14], acc[(mt + 1) & 1][0][(kk & 14) + 1]);
                                        st1 = pack_relu(acc[(mt + 1) & 1][1][kk & 14], acc[(mt + 1) & 1][1][(kk & 14) + 1]);
                                    }
                                }
                            } else if (FILL & 2) {
                                const u32x2 oo = {st0, st1};
                                *reinterpret_cast<u32x2*>(wr32 + (kk >> 1) * 16 * 0 + ((kk >> 1) << 8 & 0x300) + mt * 16384) = oo;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();
        }
        for (int a = 0; a < 2; ++a) for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) s += acc[a][t][e];
    } else {
        f32x4 acc[2][4];
        for (int a = 0; a < 2; ++a) for (int t = 0; t < 4; ++t) acc[a][t] = f32x4{0, 0, 0, 0};
        bf16x8 fx[4];
#pragma unroll 1
        for (int l = 0; l < layers; ++l) {
            for (int i = 0; i < 3; ++i) fx[i] = *reinterpret_cast<const bf16x8*>(lds + (rd16 ^ (i << 6)));
            uint32_t st0 = 0, st1 = 0;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int sidx = mt * 8 + kk;
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        acc[mt & 1][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[f * 8 + kk], fx[sidx & 3], acc[mt & 1][f], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if ((FILL & 1) && f == 0)
                            fx[(sidx + 3) & 3] = *reinterpret_cast<const bf16x8*>(lds + (rd16 ^ (((kk + 3) & 7) << 6)) + ((mt * 8192) & 65535));
                        if (FILL & 16) {                                  // the same work, no slot with more than one long instruction
                            if (f == 2 && (kk & 1)) w[16 + (kk >> 1) + 4 * (mt & 1)] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (l & 7) * 8192 + kk * 1024, 0));
                            if (f == 1 && (kk & 1)) st0 = pack_relu(acc[(mt + 1) & 1][kk >> 1][0], acc[(mt + 1) & 1][kk >> 1][1]);
                            if (f == 3 && (kk & 1)) st1 = pack_relu(acc[(mt + 1) & 1][kk >> 1][2], acc[(mt + 1) & 1][kk >> 1][3]);
                            if (f == 2 && !(kk & 1)) {
                                const u32x2 oo = {st0, st1};
                                *reinterpret_cast<u32x2*>(lds + 65536 + (wr16 ^ ((kk >> 1) << 5)) + mt * 8192) = oo;
                            }
                        } else if ((FILL & 14) && f == 2) {
                            if (kk & 1) {
                                if (FILL & 4) w[16 + (kk >> 1) + 4 * (mt & 1)] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (l & 7) * 8192 + kk * 1024, 0));
                                if (FILL & 8) {
                                    st0 = pack_relu(acc[(mt + 1) & 1][kk >> 1][0], acc[(mt + 1) & 1][kk >> 1][1]);
                                    st1 = pack_relu(acc[(mt + 1) & 1][kk >> 1][2], acc[(mt + 1) & 1][kk >> 1][3]);
                                }
                            } else if (FILL & 2) {
                                const u32x2 oo = {st0, st1};
                                *reinterpret_cast<u32x2*>(lds + 65536 + (wr16 ^ ((kk >> 1) << 5)) + mt * 8192) = oo;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();
        }
        for (int a = 0; a < 2; ++a) for (int t = 0; t < 4; ++t) for (int e = 0; e < 4; ++e) s += acc[a][t][e];
    }
    long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 32; ++i) s += (float)w[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int FILL>
void run(const uint16_t* src, float* out, long long* cyc, int layers) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<SHAPE, FILL><<<256, 256, 163840>>>(src, out, cyc, layers);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) k<SHAPE, FILL><<<256, 256, 163840>>>(src, out, cyc, layers);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = 256.0 * layers * 2.0 * 128 * 256 * 256;
    printf("shape %2d fill %d: %8.3f ms  %7.1f TFLOP/s  %7.0f cycles per layer-tile (4096 of matrix issue)  %.3f GHz\n", SHAPE, FILL, ms,
           flop / ms / 1e9, (double)c / layers, c / (ms * 1e6));
}

int main() {
    const int n = 65536 * 8;
    uint16_t* h = (uint16_t*)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
        uint32_t b; memcpy(&b, &f, 4);
        h[i] = (uint16_t)(b >> 16);
    }
    uint16_t* src; float* out; long long* cyc;
    hipMalloc(&src, (1 << 20) + n * 2); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
    for (int r = 0; r < 1; ++r) {
        run<32, 0>(src, out, cyc, 2000);
        run<16, 0>(src, out, cyc, 2000);
        run<32, 1>(src, out, cyc, 2000);
        run<16, 1>(src, out, cyc, 2000);
        run<32, 3>(src, out, cyc, 2000);
        run<16, 3>(src, out, cyc, 2000);
        run<32, 5>(src, out, cyc, 2000);
        run<16, 5>(src, out, cyc, 2000);
        run<32, 9>(src, out, cyc, 2000);
        run<16, 9>(src, out, cyc, 2000);
        run<32, 11>(src, out, cyc, 2000);
        run<32, 43>(src, out, cyc, 2000);
        run<32, 15>(src, out, cyc, 2000);
        run<32, 47>(src, out, cyc, 2000);
        run<16, 15>(src, out, cyc, 2000);
        run<16, 31>(src, out, cyc, 2000);
    }
    return 0;
}
