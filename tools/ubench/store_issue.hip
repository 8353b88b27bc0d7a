// Development micro-benchmark: what a 16-byte-per-lane global store costs a wave that is feeding the matrix pipe, by address pattern.
// One wave per SIMD (256 threads, one block per CU), like the bf16 fused kernel: per "k-step" two dependent-chain-free MFMAs
// (32 x 32 x 16 bf16, 2 x 32 clocks of matrix issue), optionally two 1 KB weight-fragment loads (the fused kernel's 128 KB per layer
// and tile through the vector-memory path) and one store per k-step:
//   mode 0  no store
//   mode 1  fragment pattern: lane (r31, h) writes 16 bytes of row r31 at chunk 2k + h -- 32 rows x 32 contiguous bytes
//   mode 2  row pattern: the instruction writes two whole 512-byte rows (consecutive lanes = consecutive 16-byte chunks)
//   mode 3  row pattern fed by a ds_read_b128 of a swizzled LDS image (what an in-layer copy would issue)
// The store target is either streamed (every tile new rows: HBM-bound if anything) or a 64 KB region per block written again
// and again (stays in L2: isolates the CU's own vector-memory path).
//   hipcc --offload-arch=gfx950 -O3 store_issue.hip -o store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, bool WLOAD>
__global__ __launch_bounds__(256, 1) void k(uint16_t* out, const uint16_t* w, long long ntiles, int stream, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<i32x4*>(smem)[i] = i32x4{i, 1, 2, 3};
    __syncthreads();
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    bf16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
    i32x4 v = {tid, 1, 2, 3};
    const int r31 = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(w), 0, 0x7fffffff, 0x27000);
    i32x4 wf[4];
    for (int i = 0; i < 4; ++i) wf[i] = i32x4{0, 0, 0, 0};
    for (long long tile = 0; tile < ntiles; ++tile) {
        unsigned char* base = reinterpret_cast<unsigned char*>(out) + (stream ? (tile * gridDim.x + blockIdx.x) * 65536LL : blockIdx.x * 65536LL);
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(base, 0, 65536, 0x27000);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (WLOAD && (kk & 1)) wf[(kk >> 1) & 3] = __builtin_amdgcn_raw_buffer_load_b128(wr, lane << 4, (wave * 32 + mt * 8 + (kk >> 1)) * 1024, 0);   // 32 KB per wave and layer-tile
                if (mt == wave || MODE == 3) {
                    if (MODE == 1 && mt == wave) __builtin_amdgcn_raw_buffer_store_b128(v, sr, (32 * wave + r31) * 512 + (2 * kk + h) * 16, 0, 0);
                    if (MODE == 2 && mt == wave) __builtin_amdgcn_raw_buffer_store_b128(v, sr, (32 * wave + 2 * kk + h) * 512 + r31 * 16, 0, 0);
                }
                if (MODE == 3 && (kk & 3) == 0) {           // one row pair per four k-steps in every tile: read now, store two k-steps on
                    const int row = 32 * wave + 2 * (mt * 4 + (kk >> 2)) + h;
                    v = *reinterpret_cast<const i32x4*>(smem + row * 512 + ((r31 ^ (row & 15)) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 3 && (kk & 3) == 2) {
                    const int row = 32 * wave + 2 * (mt * 4 + (kk >> 2)) + h;
                    __builtin_amdgcn_raw_buffer_store_b128(v, sr, row * 512 + r31 * 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    for (int i = 0; i < 4; ++i) s += (float)wf[i][0];
    if (s == 12345.678f) sink[tid] = s;
}

template <int MODE, bool WLOAD>
static void run(uint16_t* d, uint16_t* w, float* sink, int stream, const char* name) {
    const long long ntiles = 96;               // 96 tiles x 64 MFMA pairs: 96 x 4 096 clocks of matrix issue per wave
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL((k<MODE, WLOAD>), dim3(256), dim3(256), 65536, 0, d, w, ntiles, stream, sink);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (rep && ms < best) best = ms;
    }
    const double per_tile_us = best * 1e3 / ntiles;
    const double bytes = MODE ? 256.0 * ntiles * 65536 : 0.0;
    printf("%-44s %s %s: %7.3f us per layer-tile (matrix issue alone 2.05 us at 2.0 GHz), stores %.2f TB/s\n", name, WLOAD ? "wload" : "     ",
           stream ? "stream" : "L2    ", per_tile_us, bytes / (best * 1e-3) / 1e12);
}

int main() {
    uint16_t *d, *w;
    float* sink;
    hipMalloc(&d, 256LL * 96 * 65536);         // 1.6 GB
    hipMalloc(&w, 4 * 128 * 1024 + 4096);
    hipMalloc(&sink, 4096);
    hipMemset(w, 0, 4 * 128 * 1024 + 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int stream = 0; stream < 2; ++stream) {
        run<0, false>(d, w, sink, stream, "no store");
        run<0, true>(d, w, sink, stream, "no store");
        run<1, false>(d, w, sink, stream, "fragment pattern (32 rows x 32 B)");
        run<1, true>(d, w, sink, stream, "fragment pattern (32 rows x 32 B)");
        run<2, false>(d, w, sink, stream, "row pattern (2 rows x 512 B)");
        run<2, true>(d, w, sink, stream, "row pattern (2 rows x 512 B)");
        run<3, false>(d, w, sink, stream, "row pattern from LDS, spread over the tiles");
        run<3, true>(d, w, sink, stream, "row pattern from LDS, spread over the tiles");
    }
    return 0;
}
