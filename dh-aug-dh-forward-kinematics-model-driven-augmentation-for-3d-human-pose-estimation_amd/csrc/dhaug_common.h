// Shared helpers for the libdhaug HIP sources (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dhaug.h"

#define DHAUG_WAVE 64

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
__device__ __forceinline__ float dhaug_bf16_to_f32(uint16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}
