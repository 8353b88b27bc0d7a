// Parity-grade fused forward of the generator trunk / critics: the SAME one-launch unit programs as dhaug_mlp.hip
// (activations never leave LDS), in fp32-grade arithmetic on the matrix cores.
//
// The reference's layers are fp32 nn.Linear (R/models_Fk_GAN/Fk_discriminator.py:180-201,253-266,
// R/models_Fk_GAN/Fk_generator.py:115-119) and the path's tolerance is 1e-4 relative on the logits; one bf16 pass misses it by
// two orders of magnitude.  Here every operand is carried as an fp16 PAIR  x = hi + lo  (hi = fp16(x), lo = fp16(x - hi):
// 22 mantissa bits) and a product is three v_mfma_f32_32x32x16_f16 terms accumulated in fp32:
//        W x  ~=  Whi Xhi + Whi Xlo + Wlo Xhi                       (the dropped Wlo Xlo is 2^-22 relative)
// fp16 products are exact in fp32 (11 x 11 bits), so the result differs from fp32 arithmetic by ~2^-21 per operand.  The F16
// MFMA runs at the BF16 rate, so this mode costs 3 matrix instructions per k-step instead of 1 -- its roofline is a third
// of the dense peak in ALGORITHMIC flops.  Range: |x| < 65 504 (fp16); values below 2^-14 keep an ABSOLUTE error of 2^-25.
//
// Structure (round 4).  The round-3 kernel -- 64-row tiles, two images, eight waves -- stood at 0.35 of the matrix peak because
// per 256 -> 256 layer and tile its three resources were of equal size: 6 144 clocks of matrix issue, 4 096 of the 64 B/clk
// vector-memory path for 256 KB of weight fragments, 4 096 of LDS fragment reads; what had to shrink was the fragment bytes
// PER ROW.  Now:
//   * 128-row batch tiles: a layer's hi + lo weight fragments (256 KB, streamed from L2) serve twice the rows -- the
//     vector-memory path drops to a third of the matrix time.  B = 65 536 is exactly two tiles per CU;
//   * ONE activation image per tile, updated IN PLACE: hi / lo planes [128][256] fp16 = 128 KB of the 160 KB (two images do
//     not fit).  A layer reads the image through its whole k loop, all waves meet at a barrier, then every lane writes the
//     elements it owns (16-byte chunks after one half-wave exchange: whole_chunk) and a second barrier opens the next layer;
//   * eight waves (two per SIMD, 256 registers; X3_NWAVES = 4 builds the one-wave-per-SIMD variant, measured slower:
//     363 us against 331 for the 3D critic): wave w owns feature slice w for all 128 rows = 64 accumulator registers,
//     12 matrix instructions per k-step for 2 KB of weights and 8 KB of LDS fragment reads, the k-step walked in two
//     row-tile pairs so that only 32 registers of fragments are in flight.  Narrow layers are dealt as (slice x row tiles)
//     blocks so that every wave has work: 1 x 2 (N <= 128), 1 x 1 (N <= 64);
//   * what a later layer adds as a RESIDUAL is not in LDS any more when it is needed (the image has been overwritten twice):
//     the lane that produced it is the lane that will add it, so it keeps the packed hi / lo pairs it has just stored in 64
//     REGISTERS (`stash`) across the layer in between.  (Through a global workspace -- the first form of this kernel, still
//     what the four-wave build does -- the same values cost 65 us of the 3D critic's 339: 4 MB per XCD of residuals evict
//     the weight fragments from the 4 MB L2 they are streamed from; ablation X3_ABL_NOWS.)
//   * a partial result that has to wait while ANOTHER branch uses the image and the stash (the 3D critic's KCS half of the
//     merge layer, fused.py `_d3_program`) is parked in a per-workgroup global workspace (region 1: fp32, written and read
//     back by the same lane, 64 KB per tile), no longer in a third LDS buffer;
//   * the virtual three-buffer programs of include/dhaug.h are kept: the host planner below proves that a program can run on
//     one image (every value is in the image, the stash or the workspace when it is read) and annotates its units;
//   * weights: pre-split and pre-packed in A-fragment order (dhaug_pack_wfrag_f16x2); MFMA issued swapped (A = weights,
//     B = activations): a lane owns one batch row and 4 consecutive features per register quad;
//   * per output element the arithmetic is the round-3 kernel's (k ascending, per k-step Wlo Xhi, Whi Xlo, Whi Xhi, bias as
//     the accumulator seed, the residual added last as hi + lo): the generator's head and the 2D critic's logits are the same
//     bits, the 3D critic's differ by the parked half (fp32 now, an fp16 pair then).
// Measured (MI355X, B = 65 536, D = 256, same box, tools/time_x3.py): G 152 us (184), D3 309 (384), D2 103 (128).  Where a
// 256 -> 256 layer's ~19 000 clocks per tile go (phase stamps, tools/stamp_x3.py): both waves' k loops 13 400 (12 288 of matrix
// issue), the next layer's first fragments 450, barrier 400, epilogue 5 000 -- its 128 KB of LDS stores, not its arithmetic
// (halving the VALU work, fencing or not fencing its phases, 8- or 16-byte stores: no change) -- barrier 150.
#include "dhaug_common.h"
#include "dhaug_fk_math.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_BM = 128;                                           // batch rows per tile
constexpr int X3_MT = X3_BM / 32;
#ifndef X3_NWAVES
#define X3_NWAVES 8
#endif
// both matrix-instruction shapes are built (eight waves); a program says which one its weights are packed for
// (DHAUG_MLP_F_T16 on its GEMM units: dhaug_pack_wfrag_f16x2_t16)
#define X3_SHAPE16 (X3_NWAVES == 8)
constexpr int X3_NW = X3_NWAVES;                                     // waves per workgroup: 4 (one per SIMD, 512 registers) or 8
constexpr int X3_THREADS = 64 * X3_NW;
constexpr int X3_MAX_UNITS = 32;
constexpr int PITCHB = 512;                                          // bytes per image row and plane (256 fp16)
constexpr int PLANE = X3_BM * PITCHB;                                // 65 536: the hi plane; the lo plane follows
constexpr int X3_LDS_BYTES = 2 * PLANE;                              // 131 072
constexpr int OUT_PITCH = 68;                                        // floats per row of the fp32 output staging image
#ifndef X3_SPREAD
#define X3_SPREAD 1
#endif
#ifndef X3_RING
#define X3_RING 3
#endif
constexpr int RING = X3_RING;                                              // weight k-steps in registers (two requested ahead)
constexpr int WS_FLOATS_PER_WAVE = 64 * 128 * 4 / X3_NW;              // 64 lanes x the wave's accumulator values (2 x 4 x 16 of four waves)
constexpr int WS_REGION_FLOATS = 256 * X3_NW * WS_FLOATS_PER_WAVE;   // one region: every workgroup's tile, 32 MB
static_assert(2LL * WS_REGION_FLOATS * 4 == DHAUG_MLP_X3_WORKSPACE_BYTES, "include/dhaug.h: DHAUG_MLP_X3_WORKSPACE_BYTES = two regions");

enum { U_LOAD_F32 = 0, U_GEMM = 3, U_LOAD_KCS = 5 };
enum { F_OUT_F32 = 4, F_T16 = 32 };
// what the planner found out about a GEMM unit (plan bits 20..)
enum { PF_ADD_R0 = 1,          // epilogue: + the values waiting in workspace region 0 (a residual)
       PF_ADD_R1 = 2,          // epilogue: + the values waiting in region 1 (a parked partial result)
       PF_COPY_R0 = 4,         // epilogue: the result also goes to region 0 (a later unit adds it as a residual)
       PF_TO_PARK = 8 };       // epilogue: the result goes to region 1 ONLY, the image stays
enum { MAP_2x4 = 0, MAP_1x4 = 1, MAP_1x2 = 2, MAP_1x1 = 3 };        // (feature slices x row tiles) per wave

struct Unit {
    int kind, plan, flags;
    int ksteps, N, act;
    float slope;
    int cols;
    long long ld;
    const void* g;
    const _Float16* w;
    const float* bias;
};
struct Program {
    int nunits, first_gemm;
    Unit u[X3_MAX_UNITS];
};
typedef const Unit __attribute__((address_space(4))) * UnitPtr;      // units are read from the kernarg segment (s_load)

__device__ __forceinline__ int chunk_off(int row, int c) { return row * PITCHB + ((c ^ (row & 15)) << 4); }
__device__ __forceinline__ float act_neg(int act, float slope) {
    return act == DHAUG_ACT_RELU ? 0.0f : (act == DHAUG_ACT_LRELU ? slope : 1.0f);
}
__device__ __forceinline__ float act_fn(float v, float neg) { return fmaxf(v, v * neg); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo);

#ifdef X3_TIMING
__device__ long long g_x3_stamps[8 * X3_MAX_UNITS + 8];
#ifndef X3_STAMP_TID
#define X3_STAMP_TID 0
#endif
#define X3_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == X3_STAMP_TID) g_x3_stamps[i] = (long long)__builtin_readcyclecounter();
#else
#define X3_STAMP(i)
#endif

typedef f16x8 WRing[RING][2][2];                  // [k-step % RING][slice of the wave][piece]
typedef f32x16 Seed[2];                           // bias of the wave's slice(s) in accumulator order
// Eight waves: a value that a later layer adds as a residual stays in REGISTERS (one slice x four row tiles = 64 per lane: the
// lane that produced it is the lane that adds it) -- through the workspace the same values cost 65 us of the 3D critic's 339
// (they fill the L2 the weight fragments are served from; ablation X3_ABL_NOWS).  Four waves have no room (128 + 128
// accumulators and residuals of 512: hipcc spills from ~330 live), they go through workspace region 0.
#ifndef X3_REG_STASH
#define X3_REG_STASH 1
#endif
constexpr bool REG_STASH = X3_NW == 8 && X3_REG_STASH;
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef u32x16 Stash[X3_MT];                      // per row tile: the 8 packed hi pairs, then the 8 packed lo pairs, of the lane's 16 values
constexpr int RQN = X3_NW == 8 ? 2 : 4;              // tiles of a workspace value in flight in the epilogue that adds it

__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const _Float16* w, int slice0, int kt) {
    // a slice holds kt k-steps of (hi, lo) 1 KB blocks
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w) + (long long)slice0 * kt * 1024, 0, 0x7fffffff, 0x27000);
}
__device__ __forceinline__ f16x8 load_frag(__amdgpu_buffer_rsrc_t rs, int lane16, int kt, int s, int k, int p) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, ((s * kt + k) * 2 + p) * 1024, 0));
}
// this wave's slot of the workspace (region 0: residual copies, region 1: parked partial sums): values lie in accumulator order,
// 16 bytes per lane and register quad -- written and read back by the same lane
__device__ __forceinline__ float* ws_base(const void* g, int wave, int lane, int region) {
    return static_cast<float*>(const_cast<void*>(g)) + (long long)region * WS_REGION_FLOATS +
           ((long long)blockIdx.x * X3_NW + wave) * WS_FLOATS_PER_WAVE + lane * 4;
}
#ifndef X3_WS_NT
#define X3_WS_NT 0           // (non-temporal workspace accesses: measured, 370 us against 344 -- off)
#endif
__device__ __forceinline__ f32x4 ws_load(const float* p) {
#if X3_WS_NT
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ void ws_store(float* p, f32x4 v) {
#if X3_WS_NT
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void load_ws_tile(const float* src, int t, f32x16& x) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (t * 4 + g) * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[4 * g + e] = v[e];
    }
}

// k-step 0 and the bias of GEMM unit `u` for this wave's slice(s) (shape known only at run time): requested ahead of the
// layer, i.e. before the previous layer's epilogue and barriers -- and no more than that: the CU's vector-memory path takes
// 64 B/clk, and the four waves ask at the same moment (two k-steps + bias = 96 KB: 1 500 clocks before the first wave got past
// its requests, phase stamps).  Slices beyond N are zero rows of the blob (it always holds 8 slices), so every wave may load.
__device__ __forceinline__ void prefetch_layer(UnitPtr u, int wave, int lane, WRing& ring, Seed& seed) {
    const int plan = u->plan;
    const int lg = (plan >> 8) & 15, map = (plan >> 12) & 15, kt = ((plan >> 16) & 15) * 4;
    const int sw = map == MAP_2x4 ? 2 : 1;
    const int slice0 = (wave & ((1 << lg) - 1)) * sw;
    const __amdgpu_buffer_rsrc_t rs = weight_rsrc(u->w, slice0, kt);
    const int lane16 = lane << 4, h = lane >> 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) ring[0][0][p] = load_frag(rs, lane16, kt, 0, 0, p);
    if (sw == 2) {                                                            // (wave-uniform)
#pragma unroll
        for (int p = 0; p < 2; ++p) ring[0][1][p] = load_frag(rs, lane16, kt, 1, 0, p);
    }
    const float* b = u->bias + 32 * slice0 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(b + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) seed[0][4 * g + e] = b4[e];
    }
    if (sw == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(b + 32 + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) seed[1][4 * g + e] = b4[e];
        }
    }
}

// v0, v1 -> packed fp16 pairs (hi, lo) with v = hi + lo to 22 bits.  lo = fp16(v - hi) in one instruction per element:
// v_fma_mix computes 1.0 * v - float(hi) in fp32 (exact: hi is v rounded to 11 bits) and rounds it to fp16.
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const f32x2 v = {x0, x1};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(x0), "v"(hi), "v"(x1));
}

// the lo halves of a pair whose hi halves are packed in `hi`
__device__ __forceinline__ uint32_t split_lo(float x0, float x1, uint32_t hi) {
    uint32_t lo;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=&v"(lo) : "v"(x0), "v"(hi), "v"(x1));
    return lo;
}

// Register quads g (even) and g + 1 of a lane are the first / second 8 bytes of chunk g for the lane's h = 0 partner and of
// chunk g + 1 for the h = 1 partner: one half exchange per dword (v_permlane32_swap: lanes 32..63 of the first operand swap
// with lanes 0..31 of the second) leaves every lane with ONE whole 16-byte chunk -- chunk g + h of its row.  The image is then
// written with ds_write_b128, which the chunk swizzle keeps conflict-free (8 lanes = 8 rows = 8 different chunks = 32 banks);
// the 8-byte stores hit every bank pair twice (chunks p and p + 8 of a 16-lane group) and took ~3 000 clocks per layer.
__device__ __forceinline__ uint4 whole_chunk(uint2 a, uint2 b) {
    const auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    return uint4{rx[0], ry[0], rx[1], ry[1]};
}

// (four waves: the epilogue's phases are fenced -- a lone wave must be kept from chaining dependent instructions and from copying
// every accumulator out of the AGPRs first; eight waves: the partner covers the stalls and the scheduler may interleave the LDS
// stores of one tile with the arithmetic of the next)
#ifndef X3_EPI_FENCE
#define X3_EPI_FENCE (X3_NWAVES == 4)
#endif
#if X3_EPI_FENCE
#define EPI_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define EPI_FENCE()
#endif

// What a later layer adds as a residual is the value the IMAGE holds, hi + lo (22 bits), as in every other use of an
// activation in this arithmetic -- and as the round-3 kernel added it, so the golden-test figures stay what they were.  The
// producer keeps the packed pairs it has just written (no instruction); the adding layer turns a pair into hi + lo with one
// v_fma_mix_f32 (float(hi) * 1.0 + float(lo), exact in fp32).
__device__ __forceinline__ void keep_tile(u32x16& st, const uint2 (&oh)[4], const uint2 (&ol)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        st[2 * g] = oh[g].x; st[2 * g + 1] = oh[g].y;
        st[8 + 2 * g] = ol[g].x; st[8 + 2 * g + 1] = ol[g].y;
    }
}
__device__ __forceinline__ float stash_value(const u32x16& st, int i) {           // value i (0..15) of the tile
    const uint32_t h = st[i >> 1], l = st[8 + (i >> 1)];
    float x;
    if (i & 1) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x) : "v"(h), "v"(l));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(x) : "v"(h), "v"(l));
    return x;
}

#ifndef X3_WRITE128
#define X3_WRITE128 1
#endif

// The epilogue of one layer for the wave's SW x RW accumulator tiles, one tile (16 values per lane) at a time:
//   v = act(acc [+ what waits in the workspace at `rin`, four tiles in flight]);  [v -> workspace at `rout`: a later layer's residual];
//   v -> (hi, lo) -> this lane's elements of the image  |  v -> the fp32 staging image of a network output  |  nothing (PARK)
enum { EP_IMAGE = 0, EP_OUT = 1, EP_PARK = 2 };
template <int SW, int RW, int MODE, bool COPY, int ADD, bool RELU, bool KEEP>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[SW][RW], f32x4 (&rq)[RQN][4], const float* rin, float* rout, unsigned char* img, int row0,
                                         int slice0, int h, float neg, Stash& stash) {
    constexpr int G = SW * RW;
    // the lane's image addresses: one per slice and register quad (row tiles and the lo plane are constants away)
    int oaddr[SW][2];                                // (chunk 2 j + h of the slice: see whole_chunk)
    if (MODE == EP_IMAGE) {
#pragma unroll
        for (int s = 0; s < SW; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) oaddr[s][j] = chunk_off(row0, 4 * (slice0 + s) + 2 * j + h);
    }
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const int s = t / RW, mt = t % RW;
        // a lone wave issues a dependent instruction ~8 clocks behind its producer but an independent one after 4: the tile's
        // 16 values go through every step TOGETHER (left alone the scheduler chains mul -> max -> cvt -> mix per value)
        f32x4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[g][e] = ADD == 1 ? acc[s][mt][4 * g + e] + rq[t % RQN][g][e] : (ADD == 2 ? acc[s][mt][4 * g + e] + stash_value(stash[t % X3_MT], 4 * g + e) : acc[s][mt][4 * g + e]);
        EPI_FENCE();
        if (ADD == 1 && t + RQN < G) {                 // the tile's residual registers are free: request tile t + 4 into them
#pragma unroll
            for (int g = 0; g < 4; ++g) rq[t % RQN][g] = ws_load(rin + ((t + RQN) * 4 + g) * 256);
        }
        if (RELU) {
            // ReLU as ONE instruction per value: a signed-integer max with 0 on the bit pattern (negative floats are negative
            // integers; +NaN passes, -NaN becomes 0 -- the fused bf16 kernels' convention).  The epilogue's arithmetic, not its
            // LDS stores, is what it takes time for: ~110 vector instructions per 16-value tile at ~4.5 clocks each
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = v[g][e];             // (a copy: bit_cast of a vector ELEMENT reads element 0)
                    const int b = __builtin_bit_cast(int, f);
                    v[g][e] = __builtin_bit_cast(float, b > 0 ? b : 0);
                }
        } else {
            f32x4 w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) w[g] = v[g] * neg;
            EPI_FENCE();
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], w[g][e]);
        }
        EPI_FENCE();
        if (COPY || MODE == EP_PARK) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ws_store(rout + (t * 4 + g) * 256, v[g]);
        }
        if (MODE == EP_OUT) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (slice0 + s < 2)                                              // (N <= 64)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(img) + (row0 + 32 * mt) * OUT_PITCH + 32 * (slice0 + s) + 4 * h + 8 * g) = v[g];
        } else if (MODE == EP_IMAGE) {
            uint2 oh[4], ol[4];
#ifdef X3_ABL_NOSPLIT
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                oh[g].x = __builtin_bit_cast(uint32_t, v[g][0]); oh[g].y = __builtin_bit_cast(uint32_t, v[g][1]);
                ol[g].x = __builtin_bit_cast(uint32_t, v[g][2]); ol[g].y = __builtin_bit_cast(uint32_t, v[g][3]);
            }
#else
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                oh[g].x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v[g][0], v[g][1]}, f16x2));
                oh[g].y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v[g][2], v[g][3]}, f16x2));
            }
            EPI_FENCE();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                ol[g].x = split_lo(v[g][0], v[g][1], oh[g].x);
                ol[g].y = split_lo(v[g][2], v[g][3], oh[g].y);
            }
#endif
            EPI_FENCE();
            if (REG_STASH && SW == 1 && KEEP) keep_tile(stash[t % X3_MT], oh, ol);
#ifdef X3_ABL_NOWRITE
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" :: "v"(oh[g].x), "v"(oh[g].y), "v"(ol[g].x), "v"(ol[g].y));
#else
#if X3_WRITE128
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                unsigned char* o = img + oaddr[s][g >> 1] + mt * 32 * PITCHB;
                *reinterpret_cast<uint4*>(o) = whole_chunk(oh[g], oh[g + 1]);
                *reinterpret_cast<uint4*>(o + PLANE) = whole_chunk(ol[g], ol[g + 1]);
            }
#else
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned char* o = img + chunk_off(row0, 4 * (slice0 + s) + g) + (h << 3) + mt * 32 * PITCHB;
                *reinterpret_cast<uint2*>(o) = oh[g];
                *reinterpret_cast<uint2*>(o + PLANE) = ol[g];
            }
#endif
#endif
        }
        // (one accumulator tile at a time: left alone the scheduler copies all 128 accumulators out of the AGPRs first)
        EPI_FENCE();
    }
}

// The image epilogue in two stages for the waves that finish their k loop FIRST (eight waves: the older wave of a SIMD pair
// takes the matrix pipe and is done ~6 000 clocks before its partner, phase stamps): stage 1 -- residual, activation, copy,
// hi / lo split into registers -- needs nothing from the other waves and runs under the partner's matrix instructions;
// stage 2, behind the layer's barrier, is the LDS stores alone.
template <int SW, int RW, bool COPY, int ADD>
__device__ __forceinline__ void epilogue_stage1(f32x16 (&acc)[SW][RW], f32x4 (&rq)[RQN][4], const float* rin, float* rout, float neg,
                                                Stash& stash, bool keep) {
    constexpr int G = SW * RW;
#pragma unroll
    for (int t = 0; t < G; ++t) {
        const int s = t / RW, mt = t % RW;
        f32x4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[g][e] = ADD == 1 ? acc[s][mt][4 * g + e] + rq[t % RQN][g][e] : (ADD == 2 ? acc[s][mt][4 * g + e] + stash_value(stash[t % X3_MT], 4 * g + e) : acc[s][mt][4 * g + e]);
        if (ADD == 1 && t + RQN < G) {
#pragma unroll
            for (int g = 0; g < 4; ++g) rq[t % RQN][g] = ws_load(rin + ((t + RQN) * 4 + g) * 256);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = v[g] * neg;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], w[e]);
        }
        if (COPY) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ws_store(rout + (t * 4 + g) * 256, v[g]);
        }
        uint2 oh[4], ol[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            oh[g].x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v[g][0], v[g][1]}, f16x2));
            oh[g].y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v[g][2], v[g][3]}, f16x2));
            ol[g].x = split_lo(v[g][0], v[g][1], oh[g].x);
            ol[g].y = split_lo(v[g][2], v[g][3], oh[g].y);
        }
        if (REG_STASH && SW == 1 && keep) keep_tile(stash[t % X3_MT], oh, ol);
        // the tile's 16 packed dwords take the place of its 16 accumulators (no second register array across the barrier)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint4 ch = whole_chunk(oh[2 * j], oh[2 * j + 1]), cl = whole_chunk(ol[2 * j], ol[2 * j + 1]);
            acc[s][mt][8 * j + 0] = __builtin_bit_cast(float, ch.x); acc[s][mt][8 * j + 1] = __builtin_bit_cast(float, ch.y);
            acc[s][mt][8 * j + 2] = __builtin_bit_cast(float, ch.z); acc[s][mt][8 * j + 3] = __builtin_bit_cast(float, ch.w);
            acc[s][mt][8 * j + 4] = __builtin_bit_cast(float, cl.x); acc[s][mt][8 * j + 5] = __builtin_bit_cast(float, cl.y);
            acc[s][mt][8 * j + 6] = __builtin_bit_cast(float, cl.z); acc[s][mt][8 * j + 7] = __builtin_bit_cast(float, cl.w);
        }
    }
}
template <int SW, int RW>
__device__ __forceinline__ void epilogue_stage2(const f32x16 (&acc)[SW][RW], unsigned char* img, int row0, int slice0, int h) {
#pragma unroll
    for (int t = 0; t < SW * RW; ++t) {
        const int s = t / RW, mt = t % RW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned char* o = img + chunk_off(row0, 4 * (slice0 + s) + 2 * j + h) + mt * 32 * PITCHB;
            const f32x16 a = acc[s][mt];
            *reinterpret_cast<f32x4*>(o) = f32x4{a[8 * j], a[8 * j + 1], a[8 * j + 2], a[8 * j + 3]};
            *reinterpret_cast<f32x4*>(o + PLANE) = f32x4{a[8 * j + 4], a[8 * j + 5], a[8 * j + 6], a[8 * j + 7]};
        }
    }
}

// Scheduling directive for one k-step: its NM matrix instructions with the ND LDS reads and NV weight loads of the coming
// k-steps dealt out between them, one request per group of matrix instructions.
template <int NM, int ND, int NV>
__device__ __forceinline__ void spread_requests() {
    constexpr int PER = NM / ND;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (i + ND < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if (NM - PER * ND > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * ND, 0);
}

// image = act(W image + bias [+ xin]), one layer, in place.  KT k-steps (a multiple of 4: the blob is padded with zeros);
// the wave computes SW feature slices x RW row tiles.  On entry the ring holds this layer's k-steps 0 and 1 and `seed` its
// bias; before the epilogue `next` (the GEMM unit that runs after this one, possibly of the next tile; nullptr: none) is
// prefetched the same way.  Waves without a block of this layer only take part in the prefetch and the barrier.
// ADD: the layer adds what waits in the workspace (a residual, a parked partial sum) in its epilogue, LAST, as the reference
// does ((W y + b) + x; starting the accumulators from x instead moved the generator's head error from 8.3e-7 to 9.5e-7 on the
// second golden set -- 1.55e-5 m of pose through the 10 tanh root); the first four tiles are requested under the last two
// k-steps, the others as registers come free.
template <int KT, int SW, int RW, int ADD, bool KEEP, bool EARLY>
__device__ __forceinline__ void gemm_layer(UnitPtr u, UnitPtr next, unsigned char* smem, int wave, int lane, WRing& ring, Seed& seed,
                                           Stash& stash, int ui) {
    asm volatile("" : "+v"(lane));                   // lane-derived constants are recomputed per unit, not parked across units
    const int r31 = lane & 31, h = lane >> 5, lane16 = lane << 4;
    const int plan = u->plan;
#ifdef X3_ABL_NOWS
    const int lg = (plan >> 8) & 15, pf = (plan >> 20) & PF_TO_PARK;
#else
    const int lg = (plan >> 8) & 15, pf = plan >> 20;
#endif
    const int sg = wave & ((1 << lg) - 1), rg = wave >> lg;
    const bool active = rg * RW < X3_MT;             // wave-uniform
    const int slice0 = sg * SW, row0 = rg * RW * 32 + r31;
    unsigned char* img = smem;
    if (!active) {                                   // no block of this layer: the prefetch and the layer's barrier, nothing else
        if (next != nullptr) prefetch_layer(next, wave, lane, ring, seed);
        if (!(pf & PF_TO_PARK)) lds_barrier();
        return;
    }
    f32x16 acc[SW][RW];
    const float* rin = ws_base(u->g, wave, lane, (pf & PF_ADD_R1) ? 1 : 0);
    f32x4 rq[RQN][4];                                // (ADD == 1) tiles of what the epilogue adds, RQN in flight
    {
        const __amdgpu_buffer_rsrc_t rs = weight_rsrc(u->w, slice0, KT);
        // activation fragments (hi, lo): the k-step is walked in row-tile groups of HR, the next group's fragments in flight while
        // this group's matrix instructions issue (two buffers of HR x 2 fragments: 32 registers at HR = 2 instead of the 64 of
        // whole k-steps -- what makes room for the residual stash beside the accumulators)
        constexpr int HR = (X3_NW == 8 && RW == 4) ? 2 : RW, NH = RW / HR;
        f16x8 fx[2][HR][2];
        // one address per k-step: row 32 mt + r has r's swizzle (32 % 16 == 0) and the lo plane is a constant away
        const int frag_row = row0 * PITCHB, frag_sw = r31 & 15;
        auto read_frags = [&](int step, f16x8 (&f)[HR][2]) {
            const int k = step / NH, hf = step % NH;
            const unsigned char* a = img + frag_row + (((2 * k + h) ^ frag_sw) << 4) + hf * HR * 32 * PITCHB;
#pragma unroll
            for (int mt = 0; mt < HR; ++mt) {
                f[mt][0] = *reinterpret_cast<const f16x8*>(a + mt * 32 * PITCHB);
                f[mt][1] = *reinterpret_cast<const f16x8*>(a + mt * 32 * PITCHB + PLANE);
            }
        };
        read_frags(0, fx[0]);
#pragma unroll
        for (int step = 0; step < KT * NH; ++step) {
            const int k = step / NH, hf = step % NH;
#ifndef X3_ABL_NOREAD
            if (step + 1 < KT * NH) read_frags(step + 1, fx[(step + 1) & 1]);
#endif
            bool wl0 = false, wl1 = false, rl = false;
            if (hf == 0) {
#ifndef X3_ABL_NOWLOAD
                if (k == 0) {                        // (the prefetch brought k-step 0 only)
#pragma unroll
                    for (int d = 1; d < RING - 1; ++d)
                        if (d < KT) {
#pragma unroll
                            for (int s = 0; s < SW; ++s)
#pragma unroll
                                for (int p = 0; p < 2; ++p) ring[d][s][p] = load_frag(rs, lane16, KT, s, d, p);
                        }
                    wl0 = true;
                }
                if (k + RING - 1 < KT) {             // RING - 1 k-steps ahead, into the entry k-step k - 1 just left
#pragma unroll
                    for (int s = 0; s < SW; ++s)
#pragma unroll
                        for (int p = 0; p < 2; ++p) ring[(k + RING - 1) % RING][s][p] = load_frag(rs, lane16, KT, s, k + RING - 1, p);
                    wl1 = true;
                }
#endif
                if (ADD == 1 && k == KT - 2) {       // what the epilogue adds: its first tiles travel under the last two k-steps
#pragma unroll
                    for (int t = 0; t < (SW * RW < RQN ? SW * RW : RQN); ++t)
#pragma unroll
                        for (int g = 0; g < 4; ++g) rq[t][g] = ws_load(rin + (t * 4 + g) * 256);
                    rl = true;
                }
            }
#if !X3_SPREAD
            __builtin_amdgcn_sched_barrier(0);
#endif
            // small terms first, then hi * hi (the order of the round-3 kernel)
#pragma unroll
            for (int s = 0; s < SW; ++s)
#pragma unroll
                for (int m = 0; m < HR; ++m)
                    acc[s][hf * HR + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[k % RING][s][1], fx[step & 1][m][0], k == 0 ? seed[s] : acc[s][hf * HR + m], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < SW; ++s)
#pragma unroll
                for (int m = 0; m < HR; ++m)
                    acc[s][hf * HR + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[k % RING][s][0], fx[step & 1][m][1], acc[s][hf * HR + m], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < SW; ++s)
#pragma unroll
                for (int m = 0; m < HR; ++m)
                    acc[s][hf * HR + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[k % RING][s][0], fx[step & 1][m][0], acc[s][hf * HR + m], 0, 0, 0);
#if X3_SPREAD
            // the step's requests are dealt out between its matrix instructions: a wave that computes right behind its own
            // read burst pays for the burst (MI355X_MICROARCH.md, two waves per SIMD, item 7)
            {
                constexpr int NM = 3 * SW * HR, ND = 2 * HR;
                const bool rd = step + 1 < KT * NH;
                if (rd && wl0 && wl1) spread_requests<NM, ND, 2 * SW * (RING - 1)>();
                else if (rd && wl1) spread_requests<NM, ND, 2 * SW>();
                else if (rd && rl) spread_requests<NM, ND, (SW * RW < RQN ? SW * RW : RQN) * 4>();
                else if (rd) spread_requests<NM, ND, 0>();
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (ui >= 0) { X3_STAMP(8 * ui + 1) }
    if (ADD == 1 && KT < 2) {
#pragma unroll
        for (int t = 0; t < (SW * RW < RQN ? SW * RW : RQN); ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) rq[t][g] = ws_load(rin + (t * 4 + g) * 256);
    }
    // the ring and the seed are dead: the next layer's first fragments travel during the epilogue and the barrier
    if (next != nullptr) prefetch_layer(next, wave, lane, ring, seed);
    if (ui >= 0) { X3_STAMP(8 * ui + 4) }
    const float neg = act_neg(u->act, u->slope);
    // this lane owns row (row0 + 32 mt), features 32 (slice0 + s) + 8 g + 4 h .. + 3 of the result
    float* rout = ws_base(u->g, wave, lane, (pf & PF_TO_PARK) ? 1 : 0);
    const bool copy = !REG_STASH && (pf & PF_COPY_R0) != 0;
    if (pf & PF_TO_PARK) {                           // (no barrier: the image is not touched)
        epilogue<SW, RW, EP_PARK, false, ADD, false, false>(acc, rq, rin, rout, img, row0, slice0, h, neg, stash);
        return;
    }
#ifndef X3_AB_SPLIT
#define X3_AB_SPLIT 0         // (measured: no gain -- the LDS stores, not the arithmetic, are what the layer ends with -- and 64 more live registers)
#endif
    if (EARLY && !(u->flags & F_OUT_F32)) {
        // the first-finishing half: everything but the stores happens BEFORE the barrier, under the partners' k loops
        if (copy) epilogue_stage1<SW, RW, true, ADD>(acc, rq, rin, rout, neg, stash, KEEP);
        else epilogue_stage1<SW, RW, false, ADD>(acc, rq, rin, rout, neg, stash, KEEP);
        lds_barrier();                               // every wave has read the image for the last time
        if (ui >= 0) { X3_STAMP(8 * ui + 5) }
        epilogue_stage2<SW, RW>(acc, img, row0, slice0, h);
        return;
    }
    // The waves that finish LAST (four waves: every wave -- one per SIMD, same work): the barrier comes first and the conversion
    // is fused with the stores behind it; nothing but the accumulators waits across the barrier.
    lds_barrier();                                   // every wave has read the image for the last time
    if (ui >= 0) { X3_STAMP(8 * ui + 5) }
    if (u->flags & F_OUT_F32) {                      // the image becomes the fp32 staging area of the network's output
        epilogue<SW, RW, EP_OUT, false, ADD, false, false>(acc, rq, rin, rout, img, row0, slice0, h, neg, stash);
        return;
    }
#ifdef X3_ABL_NOEPI
    if (u->slope != 12345.f) return;
#endif
    const bool relu = u->act == DHAUG_ACT_RELU;      // (wave-uniform)
    if (copy) epilogue<SW, RW, EP_IMAGE, true, ADD, false, false>(acc, rq, rin, rout, img, row0, slice0, h, neg, stash);
    else if (relu) epilogue<SW, RW, EP_IMAGE, false, ADD, true, KEEP>(acc, rq, rin, rout, img, row0, slice0, h, neg, stash);
    else epilogue<SW, RW, EP_IMAGE, false, ADD, false, KEEP>(acc, rq, rin, rout, img, row0, slice0, h, neg, stash);
}

#if X3_SHAPE16
// ======================================================================================================================
// The layer body on v_mfma_f32_16x16x32_f16.  Same image, same programs, same waves (eight: slice w x 128 rows); a wave's
// 32 x 128 block is 2 feature tiles x 8 row tiles of 16 x 16, a k-step covers 32 k.  Why: at equal cycles per flop the
// chip holds a higher clock on this shape -- tools/ubench/mfma_shape.hip, fragments re-read from LDS, two waves per SIMD, random
// data: 2 010 TFLOP/s against 1 690 for 32 x 32 x 16 (1.19 x; 1.93 against 1.62 GHz).
//   A operand (weights): lane l holds W[16 ft + (l & 15)][32 ks + 8 (l >> 4) + j];  blob [slice][ft][ks][piece][lane][8]
//   B operand (image):   lane l holds X[32 ks + 8 (l >> 4) + j][row 16 rt + (l & 15)] = chunk 4 ks + (l >> 4) of that row
//   accumulator tile (ft, rt): lane l, register e = feature 16 ft + 4 (l >> 4) + e of row 16 rt + (l & 15)
// A lane's register quad is 8 bytes of a chunk whose other half sits 16 lanes away: one v_permlane16_swap per dword over a PAIR
// of row tiles gives the even quarter-waves the whole chunk of the first tile, the odd ones that of the second.
typedef f16x8 WRing16[2][2][2];                   // [k-step & 1][feature tile][piece]
typedef f32x4 Seed16[2];                          // bias of the two feature tiles in accumulator order
typedef u32x16 Stash16[4];                        // tile t = ft * NT + rt: dwords 4 (t & 3) .. + 3 of [t >> 2] = hi pair, hi pair, lo pair, lo pair
constexpr int RQ16 = 4;                           // 16 x 16 tiles of a workspace value in flight

__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc16(const _Float16* w, int slice, int kt) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w) + (long long)slice * kt * 2048, 0, 0x7fffffff, 0x27000);
}
__device__ __forceinline__ f16x8 load_frag16(__amdgpu_buffer_rsrc_t rs, int lane16, int kt, int ft, int ks, int p) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, ((ft * kt + ks) * 2 + p) * 1024, 0));
}
__device__ __forceinline__ void prefetch_layer16(UnitPtr u, int wave, int lane, WRing16& ring, Seed16& seed) {
    const int plan = u->plan;
    const int lg = (plan >> 8) & 15, kt = ((plan >> 16) & 15) * 2;           // (32-k steps)
    const int slice = wave & ((1 << lg) - 1);
    const __amdgpu_buffer_rsrc_t rs = weight_rsrc16(u->w, slice, kt);
    const int lane16 = lane << 4, q = lane >> 4;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int p = 0; p < 2; ++p) ring[0][ft][p] = load_frag16(rs, lane16, kt, ft, 0, p);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) seed[ft] = *reinterpret_cast<const f32x4*>(u->bias + 32 * slice + 16 * ft + 4 * q);
}
__device__ __forceinline__ uint4 whole_chunk16(uint2 a, uint2 b) {
    const auto rx = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    return uint4{rx[0], ry[0], rx[1], ry[1]};
}
__device__ __forceinline__ float stash_value16(const Stash16& st, int t, int e) {
    const uint32_t h = st[t >> 2][4 * (t & 3) + (e >> 1)], l = st[t >> 2][4 * (t & 3) + 2 + (e >> 1)];
    float x;
    if (e & 1) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x) : "v"(h), "v"(l));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(x) : "v"(h), "v"(l));
    return x;
}

// NT = row tiles of 16 of the wave's block (8, 4 or 2)
template <int NT, int MODE, int ADD, bool RELU, bool KEEP>
__device__ __forceinline__ void epilogue16(f32x4 (&acc)[2][NT], f32x4 (&rq)[RQ16], const float* rin, float* rout, unsigned char* img, int row0,
                                           int slice, int q, float neg, Stash16& stash) {
    constexpr int G = 2 * NT;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int p = 0; p < NT / 2; ++p) {
            uint2 oh[2], ol[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rt = 2 * p + j, t = ft * NT + rt;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = ADD == 1 ? acc[ft][rt][e] + rq[t % RQ16][e] : (ADD == 2 ? acc[ft][rt][e] + stash_value16(stash, t, e) : acc[ft][rt][e]);
                if (ADD == 1 && t + RQ16 < G) rq[t % RQ16] = ws_load(rin + (t + RQ16) * 256);
                if (RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f = v[e];
                        const int b = __builtin_bit_cast(int, f);
                        v[e] = __builtin_bit_cast(float, b > 0 ? b : 0);
                    }
                } else {
                    const f32x4 w = v * neg;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], w[e]);
                }
                if (MODE == EP_PARK) ws_store(rout + t * 256, v);
                if (MODE == EP_OUT) {
                    if (slice < 2)                                                  // (N <= 64)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(img) + (row0 + 16 * rt) * OUT_PITCH + 32 * slice + 16 * ft + 4 * q) = v;
                } else if (MODE == EP_IMAGE) {
                    oh[j].x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v[0], v[1]}, f16x2));
                    oh[j].y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v[2], v[3]}, f16x2));
                    ol[j].x = split_lo(v[0], v[1], oh[j].x);
                    ol[j].y = split_lo(v[2], v[3], oh[j].y);
                    if (KEEP) {
                        stash[t >> 2][4 * (t & 3) + 0] = oh[j].x; stash[t >> 2][4 * (t & 3) + 1] = oh[j].y;
                        stash[t >> 2][4 * (t & 3) + 2] = ol[j].x; stash[t >> 2][4 * (t & 3) + 3] = ol[j].y;
                    }
                }
            }
            if (MODE == EP_IMAGE) {
                // even quarter-waves hold the chunk of row tile 2 p, odd ones that of 2 p + 1
                unsigned char* o = img + chunk_off(row0 + 16 * (2 * p + (q & 1)), 4 * slice + 2 * ft + (q >> 1));
                *reinterpret_cast<uint4*>(o) = whole_chunk16(oh[0], oh[1]);
                *reinterpret_cast<uint4*>(o + PLANE) = whole_chunk16(ol[0], ol[1]);
            }
        }
}

template <int NM, int ND, int NV>
__device__ __forceinline__ void spread_requests16() {
    constexpr int PER = NM / ND;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (i + ND < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if (NM - PER * ND > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * ND, 0);
}

// KT: 32-k steps (2, 4 or 8); RW: 32-row groups of the wave's block (4, 2 or 1)
template <int KT, int RW, int ADD, bool KEEP>
__device__ __forceinline__ void gemm_layer16(UnitPtr u, UnitPtr next, unsigned char* smem, int wave, int lane, WRing16& ring, Seed16& seed,
                                             Stash16& stash, int ui) {
    asm volatile("" : "+v"(lane));
    constexpr int NT = 2 * RW, NP = RW;              // row tiles of 16; PAIRS of row tiles = steps per k-step
    const int c = lane & 15, q = lane >> 4, lane16 = lane << 4;
    const int plan = u->plan;
#ifdef X3_ABL_NOWS
    const int lg = (plan >> 8) & 15, pf = (plan >> 20) & PF_TO_PARK;
#else
    const int lg = (plan >> 8) & 15, pf = plan >> 20;
#endif
    const int slice = wave & ((1 << lg) - 1), rg = wave >> lg;
    const bool active = rg * RW < X3_MT;
    const int row0 = rg * RW * 32 + c;
    unsigned char* img = smem;
    if (!active) {
        if (next != nullptr) prefetch_layer16(next, wave, lane, ring, seed);
        if (!(pf & PF_TO_PARK)) lds_barrier();
        return;
    }
    f32x4 acc[2][NT];
    const float* rin = ws_base(u->g, wave, lane, (pf & PF_ADD_R1) ? 1 : 0);
    f32x4 rq[RQ16];
    {
        const __amdgpu_buffer_rsrc_t rs = weight_rsrc16(u->w, slice, KT);
        f16x8 fx[2][2][2];                           // [buffer][row tile of the pair][plane]
        const int frag_row = row0 * PITCHB;
        auto read_frags = [&](int step, f16x8 (&f)[2][2]) {
            const int ks = step / NP, pr = step % NP;
            const unsigned char* a = img + frag_row + (((4 * ks + q) ^ c) << 4) + pr * 32 * PITCHB;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f[j][0] = *reinterpret_cast<const f16x8*>(a + j * 16 * PITCHB);
                f[j][1] = *reinterpret_cast<const f16x8*>(a + j * 16 * PITCHB + PLANE);
            }
        };
        read_frags(0, fx[0]);
#pragma unroll
        for (int step = 0; step < KT * NP; ++step) {
            const int ks = step / NP, pr = step % NP;
#ifndef X3_ABL_NOREAD
            if (step + 1 < KT * NP) read_frags(step + 1, fx[(step + 1) & 1]);
#endif
            bool wl = false, rl = false;
            if (pr == 0) {
#ifndef X3_ABL_NOWLOAD
                if (ks + 1 < KT) {                   // one 32-k step ahead, into the entry step ks - 1 just left
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                        for (int p = 0; p < 2; ++p) ring[(ks + 1) & 1][ft][p] = load_frag16(rs, lane16, KT, ft, ks + 1, p);
                    wl = true;
                }
#endif
                if (ADD == 1 && ks == KT - 1) {
#pragma unroll
                    for (int t = 0; t < (2 * NT < RQ16 ? 2 * NT : RQ16); ++t) rq[t] = ws_load(rin + t * 256);
                    rl = true;
                }
            }
            // small terms first, then hi * hi
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ft][2 * pr + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[ks & 1][ft][1], fx[step & 1][j][0], ks == 0 ? seed[ft] : acc[ft][2 * pr + j], 0, 0, 0);
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ft][2 * pr + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[ks & 1][ft][0], fx[step & 1][j][1], acc[ft][2 * pr + j], 0, 0, 0);
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ft][2 * pr + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[ks & 1][ft][0], fx[step & 1][j][0], acc[ft][2 * pr + j], 0, 0, 0);
#if X3_SPREAD
            {
                const bool rd = step + 1 < KT * NP;
                if (rd && wl) spread_requests16<12, 4, 4>();
                else if (rd && rl) spread_requests16<12, 4, (2 * NT < RQ16 ? 2 * NT : RQ16)>();
                else if (rd) spread_requests16<12, 4, 0>();
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (ui >= 0) { X3_STAMP(8 * ui + 1) }
    if (next != nullptr) prefetch_layer16(next, wave, lane, ring, seed);
    if (ui >= 0) { X3_STAMP(8 * ui + 4) }
    const float neg = act_neg(u->act, u->slope);
    float* rout = ws_base(u->g, wave, lane, 1);
    if (pf & PF_TO_PARK) {
        epilogue16<NT, EP_PARK, ADD, false, false>(acc, rq, rin, rout, img, row0, slice, q, neg, stash);
        return;
    }
    lds_barrier();                                   // every wave has read the image for the last time
    if (ui >= 0) { X3_STAMP(8 * ui + 5) }
    if (u->flags & F_OUT_F32) {
        epilogue16<NT, EP_OUT, ADD, false, false>(acc, rq, rin, rout, img, row0, slice, q, neg, stash);
        return;
    }
#ifdef X3_ABL_NOEPI
    if (u->slope != 12345.f) return;
#endif
    if (u->act == DHAUG_ACT_RELU) epilogue16<NT, EP_IMAGE, ADD, true, KEEP>(acc, rq, rin, rout, img, row0, slice, q, neg, stash);
    else epilogue16<NT, EP_IMAGE, ADD, false, KEEP>(acc, rq, rin, rout, img, row0, slice, q, neg, stash);
}
#endif  // X3_SHAPE16

// LOAD: global fp32 (M, ld) columns [0, cols) -> hi / lo planes of the image, zero-filled up to the next multiple of 64
// columns and below row M.  cols and ld even: a thread moves column pairs (8-byte loads, 4-byte LDS writes).
__device__ __forceinline__ void load_unit(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    asm volatile("" : "+v"(tid));
    const float* g = static_cast<const float*>(u->g);
    const long long ld = u->ld;
    const int cols = u->cols;
    const int pairs = ((cols + 63) & ~63) >> 1;                              // per row, zero fill included: 32, 64 or 128
    const int sh = pairs == 32 ? 5 : (pairs == 64 ? 6 : 7);
    const int total = X3_BM << sh;
    constexpr int NB = 8;                                                    // loads in flight per thread
    for (int i0 = tid; i0 < total; i0 += X3_THREADS * NB) {
        f32x2 v[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {                                        // (every access issued, at a clamped address, zeros by
            const int i = i0 + j * X3_THREADS, row = i >> sh, c2 = i & (pairs - 1);   // select: no branch per access -- csrc/dhaug_mlp.hip)
            const long long gm = m0 + row < M ? m0 + row : M - 1;
            const int cc = 2 * c2 < cols ? 2 * c2 : cols - 2;
            v[j] = *reinterpret_cast<const f32x2*>(g + gm * ld + cc);
            if (!(i < total && m0 + row < M && 2 * c2 < cols)) v[j] = f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i = i0 + j * X3_THREADS, row = i >> sh, c2 = i & (pairs - 1);
            uint32_t hi, lo;
            split2(v[j][0], v[j][1], hi, lo);
            if (i < total) {
                const int o = chunk_off(row, c2 >> 2) + ((c2 & 3) << 2);
                *reinterpret_cast<uint32_t*>(smem + o) = hi;
                *reinterpret_cast<uint32_t*>(smem + PLANE + o) = lo;
            }
        }
    }
}

// LOAD_KCS: the 30 KCS features of a tile's poses (global fp32 (M, ld >= 48), root-relative or not: bones are differences),
// computed here in the arithmetic of the stand-alone dhaug_kcs_forward (kcs_features, dhaug_fk_math.h: IEEE sqrt and divide) ->
// hi / lo planes of the image, columns 30..63 zero.  One lane per pose for the features (128 of the 256 threads), every
// thread for the zero fill.  Replaces a separate launch + 8 MB round trip in front of the 3D critic's parity program.
__device__ __forceinline__ void load_kcs_unit(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    asm volatile("" : "+v"(tid));
    const float* g = static_cast<const float*>(u->g);
    const long long ld = u->ld;
    // pairs 15 .. 31 of every row: zero
    for (int s = tid; s < X3_BM * 17; s += X3_THREADS) {
        const int row = s / 17, c2 = 15 + (s - row * 17);
        const int o = chunk_off(row, c2 >> 2) + ((c2 & 3) << 2);
        *reinterpret_cast<uint32_t*>(smem + o) = 0u;
        *reinterpret_cast<uint32_t*>(smem + PLANE + o) = 0u;
    }
    if (tid < X3_BM) {
        const int row = tid;
        float f[30];
        if (m0 + row < M) {
            const float* pr = g + (m0 + row) * ld;
            dhaug_fk::V3 p[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) p[j] = dhaug_fk::mk(pr[3 * j], pr[3 * j + 1], pr[3 * j + 2]);
            dhaug_fk::kcs_features(p, f);
        } else {
#pragma unroll
            for (int c = 0; c < 30; ++c) f[c] = 0.0f;
        }
#pragma unroll
        for (int c2 = 0; c2 < 15; ++c2) {
            uint32_t hi, lo;
            split2(f[2 * c2], f[2 * c2 + 1], hi, lo);
            const int o = chunk_off(row, c2 >> 2) + ((c2 & 3) << 2);
            *reinterpret_cast<uint32_t*>(smem + o) = hi;
            *reinterpret_cast<uint32_t*>(smem + PLANE + o) = lo;
        }
    }
}

__device__ __forceinline__ void store_output(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    const float* st = reinterpret_cast<const float*>(smem);
    float* out = static_cast<float*>(const_cast<void*>(u->g));
    const long long ld = u->ld;
    const int N = u->N;
    for (int i = tid; i < X3_BM * N; i += X3_THREADS) {
        const int row = i / N, c = i - row * N;
        if (m0 + row < M) out[(m0 + row) * ld + c] = st[row * OUT_PITCH + c];
    }
}

template <bool S16>
__global__ __launch_bounds__(X3_THREADS, X3_NW / 4) void fused_mlp_x3_kernel(Program prog, long long M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ntiles = (M + X3_BM - 1) / X3_BM;
    const unsigned char __attribute__((address_space(4)))* ka =
        (const unsigned char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    UnitPtr units = (UnitPtr)(ka + __builtin_offsetof(Program, u));
    const int nunits = *(const int __attribute__((address_space(4)))*)(ka + __builtin_offsetof(Program, nunits));
    const int first_gemm = *(const int __attribute__((address_space(4)))*)(ka + __builtin_offsetof(Program, first_gemm));
#if X3_SHAPE16
    WRing16 ring16;
    Seed16 seed16;
    Stash16 stash16;
#endif
    WRing ring;
    Seed seed;
    Stash stash;
    if ((long long)blockIdx.x < ntiles) {
#if X3_SHAPE16
        if (S16) prefetch_layer16(units + first_gemm, wave, lane, ring16, seed16);
        else
#endif
            prefetch_layer(units + first_gemm, wave, lane, ring, seed);
    }
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m0 = tile * X3_BM;
        const bool more = tile + gridDim.x < ntiles;
#pragma unroll 1
        for (int i = 0; i < nunits; ++i) {
            UnitPtr u = units + i;
            const int kind = u->kind, plan = u->plan;
            const int si = tile == blockIdx.x ? i : -1;
            if (si >= 0) { X3_STAMP(8 * i) }
            if (kind == U_LOAD_F32) {
                load_unit(u, smem, m0, M, tid);
            } else if (kind == U_LOAD_KCS) {
                load_kcs_unit(u, smem, m0, M, tid);
            } else {
                // the GEMM unit that runs after this one: plan bits 0..7 hold its index + 1 (0: this is the program's last;
                // the next tile then starts over at the first)
                const int nx = plan & 255;
                UnitPtr next = nx != 0 ? units + (nx - 1) : (more ? units + first_gemm : (UnitPtr) nullptr);
#if X3_SHAPE16
                if (S16) {
                // (chunks of 64 k, block map, what the layer adds: nothing | workspace | stash; whether its result is kept)
#define X3_B16(CH, MAP, RW) \
    case (CH) * 16 + (MAP): gemm_layer16<2 * (CH), RW, 0, false>(u, next, smem, wave, lane, ring16, seed16, stash16, si); break; \
    case 1024 + (CH) * 16 + (MAP): gemm_layer16<2 * (CH), RW, 0, true>(u, next, smem, wave, lane, ring16, seed16, stash16, si); break; \
    case 256 + (CH) * 16 + (MAP): gemm_layer16<2 * (CH), RW, 1, true>(u, next, smem, wave, lane, ring16, seed16, stash16, si); break; \
    case 512 + (CH) * 16 + (MAP): gemm_layer16<2 * (CH), RW, 2, true>(u, next, smem, wave, lane, ring16, seed16, stash16, si); break;
                int addsel = (plan & (PF_ADD_R1 << 20)) ? 256 : ((plan & (PF_ADD_R0 << 20)) ? 512 : 0);
                if (addsel == 0 && (plan & (PF_COPY_R0 << 20))) addsel = 1024;
#ifdef X3_ABL_NOWS
                addsel = 0;
#endif
                switch (((plan >> 12) & 255) | addsel) {
                    X3_B16(1, MAP_1x4, 4) X3_B16(2, MAP_1x4, 4) X3_B16(4, MAP_1x4, 4)
                    X3_B16(1, MAP_1x2, 2) X3_B16(2, MAP_1x2, 2) X3_B16(4, MAP_1x2, 2)
                    X3_B16(1, MAP_1x1, 1) X3_B16(2, MAP_1x1, 1) X3_B16(4, MAP_1x1, 1)
                    default: break;
                }
#undef X3_B16
                } else
#endif
                {
// (KEEP: the layer's result is a later layer's residual and stays in the register stash -- a template parameter, not a run-time
// select per value: layers that add something always keep (a result nobody adds later is overwritten by the next keeper))
#define X3_CASE(CH, MAP, SW, RW) \
    case (CH) * 16 + (MAP): gemm_layer<4 * (CH), SW, RW, 0, false, ROLE>(u, next, smem, wave, lane, ring, seed, stash, si); break; \
    case 1024 + (CH) * 16 + (MAP): gemm_layer<4 * (CH), SW, RW, 0, true, ROLE>(u, next, smem, wave, lane, ring, seed, stash, si); break;
#define X3_CASE_ADD(CH, MAP, SW, RW) \
    case 256 + (CH) * 16 + (MAP): gemm_layer<4 * (CH), SW, RW, 1, true, ROLE>(u, next, smem, wave, lane, ring, seed, stash, si); break;
#define X3_CASE_STASH(CH, MAP, SW, RW) \
    case 512 + (CH) * 16 + (MAP): gemm_layer<4 * (CH), SW, RW, 2, true, ROLE>(u, next, smem, wave, lane, ring, seed, stash, si); break;
                /* (chunks, map, what it adds: nothing | from the workspace | from the register stash): validated on the host */
                int addsel = (plan & (PF_ADD_R1 << 20)) ? 256 : ((plan & (PF_ADD_R0 << 20)) ? (REG_STASH ? 512 : 256) : 0);
                if (addsel == 0 && REG_STASH && (plan & (PF_COPY_R0 << 20))) addsel = 1024;
#if X3_AB_SPLIT && X3_NWAVES == 8
                // the waves that finish their k loop first (the older wave of every SIMD pair) run the bodies whose epilogue does its
                // arithmetic BEFORE the layer's barrier; the two roles are separate instantiations, so neither carries the other's
                // epilogue (as one body with a run-time role the register allocator spilled 160 registers)
                if (wave < 4) {
#define ROLE true
#ifdef X3_ABL_NOWS
                switch ((plan >> 12) & 255) {
#else
                switch (((plan >> 12) & 255) | addsel) {
#endif
#if X3_NWAVES == 4
                    X3_CASE(1, MAP_2x4, 2, 4) X3_CASE(2, MAP_2x4, 2, 4) X3_CASE(4, MAP_2x4, 2, 4)
                    X3_CASE_ADD(1, MAP_2x4, 2, 4) X3_CASE_ADD(2, MAP_2x4, 2, 4) X3_CASE_ADD(4, MAP_2x4, 2, 4)
#else
                    X3_CASE_STASH(1, MAP_1x4, 1, 4) X3_CASE_STASH(2, MAP_1x4, 1, 4) X3_CASE_STASH(4, MAP_1x4, 1, 4)
                    X3_CASE_STASH(1, MAP_1x2, 1, 2) X3_CASE_STASH(2, MAP_1x2, 1, 2) X3_CASE_STASH(4, MAP_1x2, 1, 2)
                    X3_CASE_STASH(1, MAP_1x1, 1, 1) X3_CASE_STASH(2, MAP_1x1, 1, 1) X3_CASE_STASH(4, MAP_1x1, 1, 1)
#endif
                    X3_CASE(1, MAP_1x4, 1, 4) X3_CASE(2, MAP_1x4, 1, 4) X3_CASE(4, MAP_1x4, 1, 4)
                    X3_CASE(1, MAP_1x2, 1, 2) X3_CASE(2, MAP_1x2, 1, 2) X3_CASE(4, MAP_1x2, 1, 2)
                    X3_CASE(1, MAP_1x1, 1, 1) X3_CASE(2, MAP_1x1, 1, 1) X3_CASE(4, MAP_1x1, 1, 1)
                    X3_CASE_ADD(1, MAP_1x4, 1, 4) X3_CASE_ADD(2, MAP_1x4, 1, 4) X3_CASE_ADD(4, MAP_1x4, 1, 4)
                    X3_CASE_ADD(1, MAP_1x2, 1, 2) X3_CASE_ADD(2, MAP_1x2, 1, 2) X3_CASE_ADD(4, MAP_1x2, 1, 2)
                    X3_CASE_ADD(1, MAP_1x1, 1, 1) X3_CASE_ADD(2, MAP_1x1, 1, 1) X3_CASE_ADD(4, MAP_1x1, 1, 1)
                    default: break;
                }
#undef ROLE
                } else {
#define ROLE false
#ifdef X3_ABL_NOWS
                switch ((plan >> 12) & 255) {
#else
                switch (((plan >> 12) & 255) | addsel) {
#endif
#if X3_NWAVES == 4
                    X3_CASE(1, MAP_2x4, 2, 4) X3_CASE(2, MAP_2x4, 2, 4) X3_CASE(4, MAP_2x4, 2, 4)
                    X3_CASE_ADD(1, MAP_2x4, 2, 4) X3_CASE_ADD(2, MAP_2x4, 2, 4) X3_CASE_ADD(4, MAP_2x4, 2, 4)
#else
                    X3_CASE_STASH(1, MAP_1x4, 1, 4) X3_CASE_STASH(2, MAP_1x4, 1, 4) X3_CASE_STASH(4, MAP_1x4, 1, 4)
                    X3_CASE_STASH(1, MAP_1x2, 1, 2) X3_CASE_STASH(2, MAP_1x2, 1, 2) X3_CASE_STASH(4, MAP_1x2, 1, 2)
                    X3_CASE_STASH(1, MAP_1x1, 1, 1) X3_CASE_STASH(2, MAP_1x1, 1, 1) X3_CASE_STASH(4, MAP_1x1, 1, 1)
#endif
                    X3_CASE(1, MAP_1x4, 1, 4) X3_CASE(2, MAP_1x4, 1, 4) X3_CASE(4, MAP_1x4, 1, 4)
                    X3_CASE(1, MAP_1x2, 1, 2) X3_CASE(2, MAP_1x2, 1, 2) X3_CASE(4, MAP_1x2, 1, 2)
                    X3_CASE(1, MAP_1x1, 1, 1) X3_CASE(2, MAP_1x1, 1, 1) X3_CASE(4, MAP_1x1, 1, 1)
                    X3_CASE_ADD(1, MAP_1x4, 1, 4) X3_CASE_ADD(2, MAP_1x4, 1, 4) X3_CASE_ADD(4, MAP_1x4, 1, 4)
                    X3_CASE_ADD(1, MAP_1x2, 1, 2) X3_CASE_ADD(2, MAP_1x2, 1, 2) X3_CASE_ADD(4, MAP_1x2, 1, 2)
                    X3_CASE_ADD(1, MAP_1x1, 1, 1) X3_CASE_ADD(2, MAP_1x1, 1, 1) X3_CASE_ADD(4, MAP_1x1, 1, 1)
                    default: break;
                }
#undef ROLE
                }
#else
#define ROLE false
#ifdef X3_ABL_NOWS
                switch ((plan >> 12) & 255) {
#else
                switch (((plan >> 12) & 255) | addsel) {
#endif
#if X3_NWAVES == 4
                    X3_CASE(1, MAP_2x4, 2, 4) X3_CASE(2, MAP_2x4, 2, 4) X3_CASE(4, MAP_2x4, 2, 4)
                    X3_CASE_ADD(1, MAP_2x4, 2, 4) X3_CASE_ADD(2, MAP_2x4, 2, 4) X3_CASE_ADD(4, MAP_2x4, 2, 4)
#else
                    X3_CASE_STASH(1, MAP_1x4, 1, 4) X3_CASE_STASH(2, MAP_1x4, 1, 4) X3_CASE_STASH(4, MAP_1x4, 1, 4)
                    X3_CASE_STASH(1, MAP_1x2, 1, 2) X3_CASE_STASH(2, MAP_1x2, 1, 2) X3_CASE_STASH(4, MAP_1x2, 1, 2)
                    X3_CASE_STASH(1, MAP_1x1, 1, 1) X3_CASE_STASH(2, MAP_1x1, 1, 1) X3_CASE_STASH(4, MAP_1x1, 1, 1)
#endif
                    X3_CASE(1, MAP_1x4, 1, 4) X3_CASE(2, MAP_1x4, 1, 4) X3_CASE(4, MAP_1x4, 1, 4)
                    X3_CASE(1, MAP_1x2, 1, 2) X3_CASE(2, MAP_1x2, 1, 2) X3_CASE(4, MAP_1x2, 1, 2)
                    X3_CASE(1, MAP_1x1, 1, 1) X3_CASE(2, MAP_1x1, 1, 1) X3_CASE(4, MAP_1x1, 1, 1)
                    X3_CASE_ADD(1, MAP_1x4, 1, 4) X3_CASE_ADD(2, MAP_1x4, 1, 4) X3_CASE_ADD(4, MAP_1x4, 1, 4)
                    X3_CASE_ADD(1, MAP_1x2, 1, 2) X3_CASE_ADD(2, MAP_1x2, 1, 2) X3_CASE_ADD(4, MAP_1x2, 1, 2)
                    X3_CASE_ADD(1, MAP_1x1, 1, 1) X3_CASE_ADD(2, MAP_1x1, 1, 1) X3_CASE_ADD(4, MAP_1x1, 1, 1)
                    default: break;
                }
#undef ROLE
#endif
#undef X3_CASE_STASH
#undef X3_CASE_ADD
#undef X3_CASE
                }
                if (u->flags & F_OUT_F32) {
                    lds_barrier();
                    store_output(u, smem, m0, M, tid);
                }
            }
            if (si >= 0) { X3_STAMP(8 * i + 2) }
            lds_barrier();
            if (si >= 0) { X3_STAMP(8 * i + 3) }
        }
    }
    (void)prog;
}

// weights -> hi / lo fp16 fragments.  dst[(((slice*ksteps + ks)*2 + piece)*64 + lane)*8 + j] =
//   piece(W[32 slice + (lane&31)][k0 + 16 ks + 8 (lane>>5) + j])
__global__ __launch_bounds__(256) void pack_wfrag_f16x2_kernel(const float* __restrict__ W, long long ldw, _Float16* __restrict__ dst,
                                                               int N, int K, int k0, int ksteps, int nslices, int t16) {
    const long long total = (long long)nslices * ksteps * 64 * 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long long blk = i >> 9;
        int n, k;
        if (t16) {
            // [slice][feature tile][32-k step][piece][lane][8]: blk = (slice * 2 + ft) * (ksteps / 2) + ks32
            const int kt = ksteps / 2, ks = (int)(blk % kt), ft = (int)((blk / kt) & 1), s = (int)(blk / (2 * kt));
            n = 32 * s + 16 * ft + (lane & 15); k = 32 * ks + 8 * (lane >> 4) + j;
        } else {
            const int ks = (int)(blk % ksteps), s = (int)(blk / ksteps);
            n = 32 * s + (lane & 31); k = 16 * ks + 8 * (lane >> 5) + j;
        }
        const float w = (n < N && k < K) ? W[(long long)n * ldw + k0 + k] : 0.0f;
        const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
        const long long o = ((blk * 2) * 64 + lane) * 8 + j;
        dst[o] = hi;
        dst[o + 512] = lo;
    }
}

}  // namespace

extern "C" {

static int pack_wfrag_f16x2_impl(const float* W, int64_t ldw, uint16_t* dst, int64_t N, int64_t K, int64_t k0, int t16, void* stream) {
    DHAUG_CHECK(N >= 1 && K >= 1 && k0 >= 0 && ldw >= k0 + K, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(W); DHAUG_CHECK_PTR(dst);
    DHAUG_CHECK(dhaug_aligned16(dst), DHAUG_EALIGN);
    const int ksteps = (int)((K + 63) / 64) * 4, nslices = 8;
    DHAUG_CHECK(ksteps <= 16 && N <= 256, DHAUG_EUNSUPPORTED);
    const long long total = (long long)nslices * ksteps * 512;
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_wfrag_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, (long long)ldw,
                       reinterpret_cast<_Float16*>(dst), (int)N, (int)K, (int)k0, ksteps, nslices, t16);
    return dhaug_launch_status();
}

/* see include/dhaug.h */
int dhaug_pack_wfrag_f16x2(const float* W, int64_t ldw, uint16_t* dst, int64_t N, int64_t K, int64_t k0, void* stream) {
    return pack_wfrag_f16x2_impl(W, ldw, dst, N, K, k0, 0, stream);
}
int dhaug_pack_wfrag_f16x2_t16(const float* W, int64_t ldw, uint16_t* dst, int64_t N, int64_t K, int64_t k0, void* stream) {
    return pack_wfrag_f16x2_impl(W, ldw, dst, N, K, k0, 1, stream);
}

/* The planner: the units address three virtual buffers (include/dhaug.h); the kernel has ONE image and a two-region global
 * workspace.  Walk the program, track where every buffer's current value lives, and annotate the GEMM units; a program in
 * which a value would be needed from a place it is not in is DHAUG_EUNSUPPORTED. */
int dhaug_mlp_forward_x3(const dhaug_mlp_unit* units, int nunits, int64_t M, void* stream) {
    DHAUG_CHECK(nunits >= 1 && nunits <= X3_MAX_UNITS && M >= 0, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(units);
    if (M == 0) return DHAUG_OK;
    Program prog;
    prog.nunits = nunits;
    auto okbuf = [](int b) { return b >= 0 && b <= 2; };
    auto is_out = [](const dhaug_mlp_unit& t) { return t.kind == U_GEMM && (t.flags & F_OUT_F32) != 0; };
    auto map_of = [](int n) {
        const int nsl = (n + 31) / 32;
        if (X3_NW == 8) return nsl > 4 ? MAP_1x4 : (nsl > 2 ? MAP_1x2 : MAP_1x1);
        return nsl > 4 ? MAP_2x4 : (nsl > 2 ? MAP_1x4 : (nsl == 2 ? MAP_1x2 : MAP_1x1));
    };
    // how buffer b's CURRENT value is read after unit i, until the buffer is written again: 1 as a source, 2 as a residual
    auto uses = [&](int b, int i) {
        int m = 0;
        for (int j = i + 1; j < nunits; ++j) {
            const dhaug_mlp_unit& t = units[j];
            if (t.kind == U_GEMM) {
                if (t.src == b) m |= 1;
                if (t.res == b) m |= 2;
                if (t.dst == b && !is_out(t)) break;
            } else if (t.dst == b) {
                break;
            }
        }
        return m;
    };
    const void* ws = nullptr;                                       /* the workspace: g of the GEMM units that are not outputs */
    for (int i = 0; i < nunits; ++i)
        if (units[i].kind == U_GEMM && !is_out(units[i]) && units[i].g != nullptr) { ws = units[i].g; break; }
    DHAUG_CHECK(ws == nullptr || dhaug_aligned16(ws), DHAUG_EALIGN);
    int img = -1;                                                   /* the virtual buffer whose value the image holds */
    int t16 = -1;                                                   /* which matrix instruction the program's weights are packed for */
    int r0 = -1, r0_map = -1, r1 = -1, r1_map = -1;                 /* ... region 0 (a copy for a residual), region 1 (parked) */
    for (int i = 0; i < nunits; ++i) {
        const dhaug_mlp_unit& s = units[i];
        Unit& u = prog.u[i];
        u.kind = s.kind; u.flags = s.flags; u.ksteps = s.ksteps; u.N = s.n; u.act = s.act; u.slope = s.slope; u.cols = s.cols;
        u.ld = s.ld; u.g = s.g; u.w = static_cast<const _Float16*>(s.w); u.bias = s.bias;
        u.plan = 0;
        // dhaug_set_nan_propagation(1): ReLU as LeakyReLU with slope 0 (max(v, v * 0): NaN / inf reach the logit) instead of
        // the one-instruction integer max
        if (dhaug_nan_propagation_ && s.kind == U_GEMM && s.act == DHAUG_ACT_RELU) { u.act = DHAUG_ACT_LRELU; u.slope = 0.0f; }
        DHAUG_CHECK(u.kind == U_LOAD_F32 || u.kind == U_GEMM || u.kind == U_LOAD_KCS, DHAUG_EUNSUPPORTED);
        if (u.kind == U_GEMM) {
            DHAUG_CHECK(okbuf(s.src) && u.ksteps >= 1 && u.ksteps <= 16 && u.N >= 1 && u.N <= 256, DHAUG_EUNSUPPORTED);
            DHAUG_CHECK(s.ksteps2 == 0, DHAUG_EUNSUPPORTED);              /* (a concatenation is two units: see fused.py) */
            DHAUG_CHECK(u.w != nullptr && dhaug_aligned16(u.w) && u.bias != nullptr && dhaug_aligned16(u.bias), DHAUG_EALIGN);
            DHAUG_CHECK((u.flags & ~(F_OUT_F32 | F_T16)) == 0, DHAUG_EUNSUPPORTED);
            if (t16 < 0) t16 = (u.flags & F_T16) ? 1 : 0;
            DHAUG_CHECK(t16 == ((u.flags & F_T16) ? 1 : 0), DHAUG_EINVAL);            /* one fragment layout per program */
            u.flags &= ~F_T16;
            DHAUG_CHECK(s.src == img, DHAUG_EUNSUPPORTED);                /* the source must be what the image holds */
            const int chunks = (u.ksteps + 3) / 4;
            DHAUG_CHECK(chunks == 1 || chunks == 2 || chunks == 4, DHAUG_EUNSUPPORTED);
            const int map = map_of(u.N);
            const int nsl_ = (u.N + 31) / 32;
            const int lg = X3_NW == 8 ? (nsl_ > 4 ? 3 : (nsl_ > 2 ? 2 : (nsl_ == 2 ? 1 : 0)))
                                      : (map == MAP_1x1 ? 0 : (map == MAP_1x2 ? 1 : 2));      /* log2(slice groups) */
            int pf = 0;
            if (s.res >= 0) {
                DHAUG_CHECK(okbuf(s.res) && s.res != s.src, DHAUG_EINVAL);
                if (s.res == r0) {
                    DHAUG_CHECK(r0_map == map, DHAUG_EUNSUPPORTED);      /* (written and read back lane by lane) */
                    pf |= PF_ADD_R0;
                } else {
                    DHAUG_CHECK(s.res == r1 && r1_map == map, DHAUG_EUNSUPPORTED);
                    pf |= PF_ADD_R1;
                    r1 = -1;
                }
            }
            const bool out = is_out(s);
            const int mdst = out ? 0 : uses(s.dst, i);
            const bool park = !out && (mdst & 2) && !(mdst & 1);
            if (!park) {
                /* the image is overwritten at this unit's epilogue: nobody may read the source from it afterwards, and
                   whoever adds it later must find it in the workspace */
                const int m = uses(s.src, i);
                DHAUG_CHECK(!(m & 1) && (!(m & 2) || s.src == r0 || s.src == r1), DHAUG_EUNSUPPORTED);
            }
            if (out) {
                DHAUG_CHECK(u.g != nullptr && u.ld >= u.N && u.N <= 64, DHAUG_EUNSUPPORTED);
                img = -1;                                                  /* the image becomes the staging area */
                if (pf) { DHAUG_CHECK(ws != nullptr, DHAUG_EINVAL); }
            } else {
                DHAUG_CHECK(okbuf(s.dst) && s.dst != s.src, DHAUG_EINVAL);
                const int m = mdst;
                if (park) {
                    /* nobody reads the result as a source before it is added somewhere: it waits in region 1 and the image
                       keeps the source */
                    DHAUG_CHECK(r1 < 0, DHAUG_EUNSUPPORTED);
                    pf |= PF_TO_PARK;
                    r1 = s.dst; r1_map = map;
                    if (r0 == s.dst) r0 = -1;
                } else {
                    if (m & 2) {
                        /* read as a source AND added later: image + a copy in region 0, whose old content must be dead */
                        DHAUG_CHECK(r0 < 0 || r0 == s.dst || !(uses(r0, i) & 2), DHAUG_EUNSUPPORTED);
                        pf |= PF_COPY_R0;
                        r0 = s.dst; r0_map = map;
                    } else if (r0 == s.dst) {
                        r0 = -1;                                           /* (the copy is of the old value) */
                    }
                    if (r1 == s.dst) r1 = -1;
                    img = s.dst;
                }
                u.g = ws;
                if (pf) { DHAUG_CHECK(ws != nullptr, DHAUG_EINVAL); }
            }
            if (out && pf) {
                /* an output that adds a residual reads the workspace through g, which holds the output: not supported */
                return DHAUG_EUNSUPPORTED;
            }
            u.plan = (lg << 8) | (map << 12) | (chunks << 16) | (pf << 20);   // bits 0..7: index + 1 of the next GEMM unit (below)
        } else {
            if (u.kind == U_LOAD_KCS) {
                DHAUG_CHECK(okbuf(s.dst) && u.g != nullptr && u.ld >= 48, DHAUG_EINVAL);
            } else {
                DHAUG_CHECK(okbuf(s.dst) && u.g != nullptr && u.cols >= 2 && ((u.cols + 63) & ~63) <= 256, DHAUG_EINVAL);
                DHAUG_CHECK(u.cols % 2 == 0 && u.ld % 2 == 0 && u.ld >= u.cols && (reinterpret_cast<uintptr_t>(u.g) & 7u) == 0, DHAUG_EALIGN);
            }
            if (img >= 0 && img != s.dst) {                                /* what the image held: dead, or safe in the workspace */
                const int m = uses(img, i);
                DHAUG_CHECK(!(m & 1) && (!(m & 2) || img == r0 || img == r1), DHAUG_EUNSUPPORTED);
            }
            if (r0 == s.dst) r0 = -1;
            if (r1 == s.dst) r1 = -1;
            img = s.dst;
        }
    }
    prog.first_gemm = -1;
    for (int i = nunits - 1, nx = 0; i >= 0; --i)
        if (prog.u[i].kind == U_GEMM) {
            prog.u[i].plan |= nx;
            nx = i + 1;
            prog.first_gemm = i;
        }
    DHAUG_CHECK(prog.first_gemm >= 0, DHAUG_EINVAL);
#if !X3_SHAPE16
    DHAUG_CHECK(t16 != 1, DHAUG_EUNSUPPORTED);
#endif
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_x3_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
#if X3_SHAPE16
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                X3_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
#endif
        configured = true;
    }
    const long long ntiles = (M + X3_BM - 1) / X3_BM;
    const unsigned grid = dhaug_persistent_grid(ntiles);           // one persistent workgroup per CU
#if X3_SHAPE16
    if (t16 == 1) hipLaunchKernelGGL(fused_mlp_x3_kernel<true>, dim3(grid), dim3(X3_THREADS), X3_LDS_BYTES, (hipStream_t)stream, prog, (long long)M);
    else
#endif
        hipLaunchKernelGGL(fused_mlp_x3_kernel<false>, dim3(grid), dim3(X3_THREADS), X3_LDS_BYTES, (hipStream_t)stream, prog, (long long)M);
    return dhaug_launch_status();
}

#ifdef X3_TIMING
int dhaug_debug_mlp_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3_stamps), sizeof(long long) * (n < 8 * X3_MAX_UNITS ? n : 8 * X3_MAX_UNITS));
}
#endif
}  // extern "C"
