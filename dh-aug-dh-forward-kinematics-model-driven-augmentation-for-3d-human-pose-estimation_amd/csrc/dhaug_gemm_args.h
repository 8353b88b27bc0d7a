// Argument blocks of the NT layer GEMMs, shared by the translation units that hold their kernels (dhaug_gemm.hip: the specialised and
// generic kernels and the C-ABI entry points; dhaug_gemm_p8.hip: the 256 x 256-tile ping-pong kernel of the DenseDim-1000 layers).
#pragma once
#include "dhaug_common.h"

namespace dhaug_gemm {

struct GemmArgs {
    const uint16_t* A; long long lda;
    const uint16_t* B; long long ldb;
    const float* bias;
    const uint16_t* res; long long ld_res;
    const float* resf; long long ld_resf;
    uint16_t* cb; long long ldcb; long long npad;
    float* cf; long long ldcf;
    long long M, N, K, W;          // W = output width covered by tiles (N, or the zero-padded width)
    int act; float slope;
    const uint16_t* dmask; long long ld_dmask; float dneg;   // optional: result *= (dmask > 0 ? 1 : dneg)
    const uint32_t* dbits;                                   // the same mask as a sign-bit array (dhaug_mlp_unit.bits layout)
    const uint32_t* dbits2;                                  // ... of output columns 256..511 (gemm_nt_ws_kernel: a layer whose output is two 256-wide blocks)
    int abl;                                                 // development (big-tile kernels): 1 no epilogue, 2 no reads / MFMAs, 4 no copies
    const float* dmaskf; long long ld_dmaskf;                // the same mask from an fp32 activation (split-operand arithmetic): kernels
                                                             // whose epilogue is nt_store_tile / the ping-pong kernel's
    // The ping-pong kernel only (dhaug_gemm_bf16x6_planes): A holds the THREE distinct bf16 pieces of a split fp32 operand as planes
    // [hi | mid | lo] of xp_kp columns each (lda >= 3 xp_kp) and K = 6 xp_kp: K-segment s of the contraction reads plane
    // (xp_map >> 2 s) & 3.  xp_lg = 1 + log2(xp_kp / 64); 0: A is an ordinary K-wide operand.  IEEE-half operands (dhaug_gemm_f16x3_planes):
    // two pieces [hi | lo], lda >= 2 xp_kp, K = 3 xp_kp, the result's planes likewise two.
    int xp_lg; unsigned xp_map; long long xp_kp;
    // ... and its result once more as such planes (N == cp_kp columns per piece, ldcp >= 3 N): what dhaug_split_bf16(c_f32, mode 2) would
    // make of it, written by the epilogue that has the values in registers -- the next layer's operand without a split launch
    uint16_t* cp; long long ldcp; long long cp_kp;          // (cp_kp >= N columns per piece, % 8 == 0; columns [N, cp_kp) are written as zeros)
};

// Up to eight independent GEMMs of ONE shape as one launch (dhaug_gemm_bf16_group)
constexpr int NT_GROUP_MAX = 8;
struct GemmGroupArgs { GemmArgs g[NT_GROUP_MAX]; };

// A GemmArgs copied word by word out of the kernarg segment (the grouped kernels) holds pointers hipcc knows nothing about: every
// access through them was a FLAT instruction -- which counts in the LDS counter as well, so each wait for an LDS read in the epilogue also
// waited for the row stores before it.  Passing the pointers through the global address space restores global_load / global_store
// (measured: the grouped launches take what they took, 28.9 / 17.3 us for four 1 536- / 512-row members -- the epilogue is not what waits).
template <class T> __device__ __forceinline__ T* as_global(T* q) { return (T*)(__attribute__((address_space(1))) T*)reinterpret_cast<uintptr_t>(q); }
__device__ __forceinline__ void globalize(GemmArgs& p) {
    p.A = as_global(p.A); p.B = as_global(p.B); p.bias = as_global(p.bias); p.res = as_global(p.res); p.resf = as_global(p.resf);
    p.cb = as_global(p.cb); p.cf = as_global(p.cf); p.dmask = as_global(p.dmask); p.dbits = as_global(p.dbits); p.dbits2 = as_global(p.dbits2);
    p.dmaskf = as_global(p.dmaskf); p.cp = as_global(p.cp);
}

// the member's arguments of a grouped launch, read from the kernarg segment with scalar loads (a by-value array indexed dynamically
// would be copied to scratch)
__device__ __forceinline__ void load_group_member(GemmArgs& p, int member) {
    static_assert(sizeof(GemmArgs) % 8 == 0, "copied as 8-byte words");
    const unsigned long long __attribute__((address_space(4)))* src = (const unsigned long long __attribute__((address_space(4)))*)(
        (const unsigned char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + member * sizeof(GemmArgs));
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&p);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(GemmArgs) / 8); ++i) dst[i] = src[i];
    globalize(p);
}

// branch-free activation: v > 0 ? v : v * neg, neg = 0 (ReLU) / slope (LeakyReLU) / 1 (identity)
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    const float neg = act == DHAUG_ACT_RELU ? 0.0f : (act == DHAUG_ACT_LRELU ? slope : 1.0f);
    return v > 0.0f ? v : v * neg;
}

}  // namespace dhaug_gemm

// dhaug_gemm_p8.hip: the 256 x 256 x 64 ping-pong kernel.  `supported` says whether the kernel takes the problem (shape, strides,
// alignment); the launchers enqueue on `s` and return a DHAUG_* / hipError_t code.
bool dhaug_p8_supported(const dhaug_gemm::GemmArgs& p);
int dhaug_p8_launch(hipStream_t s, const dhaug_gemm::GemmArgs& p);
int dhaug_p8_launch_f16(hipStream_t s, const dhaug_gemm::GemmArgs& p);     // the same tiles on IEEE-half operands (dhaug_gemm_f16x3)
int dhaug_p8_launch_group(hipStream_t s, const dhaug_gemm::GemmGroupArgs& g, int n);
