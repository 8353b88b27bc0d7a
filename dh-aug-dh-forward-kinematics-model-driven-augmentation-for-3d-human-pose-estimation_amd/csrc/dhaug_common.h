// Shared helpers for the libdhaug HIP sources (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dhaug.h"

#define DHAUG_WAVE 64

// Development switches (ablations: "timing only, results wrong"; phase stamps; structure variants of the fused kernels) exist in
// these sources for tools/build_*_abl.sh.  A PRODUCT build defines none of them: any of the names below on the command line is a
// compile error unless DHAUG_ABLATION_BUILD is defined as well (tools/ define it; __graft_entry__.build_lib never does unless the
// environment asks for an ablation build by name) -- a stray -DX3_ABL_NOSPLIT cannot yield a library that loads, exports every
// symbol and computes garbage (tests/test_cpu_boundary.py::test_ablation_switches_need_an_ablation_build).
#if !defined(DHAUG_ABLATION_BUILD)
#if defined(T4_ABL_NOSTORE) || defined(T4_ABL_NOCOMPUTE) || defined(ABL_NOWRITE) || defined(ABL_NOREAD) || defined(ABL_NOWLOAD) || \
    defined(SAVE_ABL_NOSTORE) || defined(SAVE_ABL_BRANCH) || defined(SAVE_ABL_NOBITS) || defined(SAVE_ABL_NONANSAFE) || defined(SAVE_ABL_NULLSTORES) || defined(MOVE_BATCH_OVERRIDE) || defined(LOAD_ABL_NOROWS32) || \
    defined(DHAUG_MLP_TIMING) || defined(DHAUG_MLP_TIMING_UNITS) || defined(DHAUG_STAMP_TID) || defined(DHAUG_PIPE_TIMING) || defined(DHAUG_TOP_TIMING) || \
    defined(W_NO_XCD_MAP) || defined(X3_NWAVES) || defined(X3_SPREAD) || defined(X3_RING) || defined(X3_TIMING) || \
    defined(X3_STAMP_TID) || defined(X3_REG_STASH) || defined(X3_WS_NT) || defined(X3_EPI_FENCE) || defined(X3_WRITE128) || \
    defined(X3_ABL_NOSPLIT) || defined(X3_ABL_NOWRITE) || defined(X3_ABL_NOWS) || defined(X3_ABL_NOREAD) || \
    defined(X3_ABL_NOWLOAD) || defined(X3_ABL_NOEPI) || defined(X3_AB_SPLIT) || defined(X3_PRIO_SEL) || \
    defined(P8_ABL_NOMMA) || defined(P8_ABL_NOREAD) || defined(P8_ABL_NOCOPY) || defined(P8_ABL_NOEPI) || defined(P8_ABL_NOSTAGGER) || \
    defined(P8_ABL_NOPRIO) || defined(P8_TIMING)
#error "a development / ablation switch is defined without -DDHAUG_ABLATION_BUILD: this would build a library with wrong results"
#endif
#endif
// run-time ablation selectors (DHAUG_TN256_ABL, DHAUG_BIG_ABL: phases of a kernel switched off, results wrong) are read from the
// environment in ablation builds only
#if defined(DHAUG_ABLATION_BUILD)
#define DHAUG_ABL_ENV(name) (getenv(name) ? atoi(getenv(name)) : 0)
#else
#define DHAUG_ABL_ENV(name) 0
#endif

#define DHAUG_CHECK_PTR(p)        do { if ((p) == nullptr) return DHAUG_EINVAL; } while (0)
#define DHAUG_CHECK(cond, code)   do { if (!(cond)) return (code); } while (0)

static inline int dhaug_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DHAUG_OK : (int)e;
}

static inline bool dhaug_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Workgroups a one-per-CU (persistent) launch may take: 256, or what dhaug_set_workgroup_cap() left it -- two chains of such
// launches on two streams then run SIDE BY SIDE on disjoint sets of CUs instead of queueing for the whole card.
extern "C" int dhaug_workgroup_cap_;
// dhaug_set_nan_propagation(): the fused inference programs apply ReLU as max(v, v * 0) (NaN-propagating) instead of an integer
// max on the bit pattern (which turns the matrix pipe's -NaN into 0)
extern "C" int dhaug_nan_propagation_;
static inline unsigned dhaug_persistent_grid(long long tiles) {
    const long long cap = dhaug_workgroup_cap_ > 0 && dhaug_workgroup_cap_ < 256 ? dhaug_workgroup_cap_ : 256;
    return (unsigned)(tiles < cap ? tiles : cap);
}

// Grid for one-tile-per-wave streaming kernels: enough workgroups to fill 256 CUs several times over,
// capped so the tail is a grid-stride loop (cdna_hip_programming.md Guideline 11).
static inline int dhaug_stream_grid(int64_t tiles, int tiles_per_block, int max_blocks = 256 * 8) {
    int64_t b = (tiles + tiles_per_block - 1) / tiles_per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

__device__ __forceinline__ uint16_t dhaug_f32_to_bf16(float f) {
    // round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
// clamp to [-1, 1] as torch.clamp does it (R/common/camera.py:62-94 project_to_2d): a NaN stays a NaN -- fminf / fmaxf return the
// other operand.  The test reads the bit pattern, so it also holds in the translation units built with -ffinite-math-only.
__host__ __device__ __forceinline__ float dhaug_clamp_pm1(float x) {
    const float c = fminf(fmaxf(x, -1.0f), 1.0f);
    return (__builtin_bit_cast(uint32_t, x) & 0x7fffffffu) > 0x7f800000u ? x : c;
}
__device__ __forceinline__ float dhaug_bf16_to_f32(uint16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}
