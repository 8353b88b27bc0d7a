// Fused multi-layer forward of the generator trunk / critics: ONE launch per network, activations never leave LDS.
//
// Layer-by-layer execution of these 256-wide stacks is HBM/latency bound (AI ~ 128 FLOP/B per layer); fused, the
// compulsory traffic is the network input and its logits/head only and the stack becomes MFMA-bound.
//
// Structure (256 threads = 4 waves = one wave per SIMD with the 512-register budget; one persistent workgroup per CU
// walking 128-row batch tiles):
//   * activations: bf16 tiles [128 rows][256] in LDS (two 64 KB buffers + one 32 KB [128][128] buffer), 16-byte
//     chunks XOR-swizzled by (row & 15) -> conflict-free ds_read_b128 of MFMA B-operand fragments;
//   * weights: pre-packed on the host side in MFMA A-operand fragment order (dhaug_pack_wfrag): for every
//     32-feature slice and 16-wide k-step one contiguous 1 KB block = 64 lanes x 16 B, so a wave's weight load is
//     a perfectly coalesced global_load_dwordx4 served by L2; wave w owns feature slices w and w+4 of the layer.
//     Two execution paths share these images:
//       - gemm_stack: runs of full-width layers (the residual stacks, the narrow layer feeding them, the output
//         layer behind them) with register-resident weights and the epilogue riding the matrix pipe -- see there;
//       - gemm_layer: any other layer, one at a time, fragments through a 3-slot register ring (4 k-steps per slot,
//         two slots ahead of the MFMAs; each fragment feeds 8 MFMAs: 2 slices x 4 row tiles);
//   * MFMA issued swapped (A = weights, B = activations): a lane of the 32x32 accumulator owns one batch row and
//     4 consecutive features per register quad -> epilogue (bias, residual from LDS, ReLU/LeakyReLU, bf16 pack)
//     writes 8 bytes per lane straight into the next layer's operand image.  In-place residual layers
//     (dst == res) are safe: a lane reads and writes only its own elements.
//   * the network is a short "program" of units (load / gemm / store) passed by value; all control flow is
//     workgroup-uniform.
#include <cstdlib>
#include "dhaug_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_BM = 128;                                           // batch rows per tile
constexpr int MLP_MT = MLP_BM / 32;                                   // 32-row MFMA tiles per batch tile
constexpr int MLP_THREADS = 256;
constexpr int MLP_NS = 2;                                             // feature slices (of 32) per wave: w and w + 4
constexpr int MLP_MAX_UNITS = 32;
constexpr int MLP_MAX_KSTEPS = 16;
constexpr int BUF01_PITCH = 256, BUF2_PITCH = 128;                  // elements
constexpr int BUF01_BYTES = MLP_BM * BUF01_PITCH * 2;               // 65 536
constexpr int BUF2_BYTES = MLP_BM * BUF2_PITCH * 2;                 // 32 768
constexpr int MLP_LDS_BYTES = 2 * BUF01_BYTES + BUF2_BYTES;         // 163 840: all of the CU's LDS

enum { U_LOAD_F32 = 0, U_LOAD_BF16 = 1, U_STORE_BF16 = 2, U_GEMM = 3 };
// The forward-with-save translation unit (the training steps' sweep 1) propagates NaN through every activation: its ReLU is
// max(v, v * 0) in fp32 like the LeakyReLU layers' mul + max (NaN * 0 = NaN, max(NaN, NaN) = NaN), so a diverged network is
// REPORTED by D_cost as the reference's ATen ops report it.  The inference unit keeps the integer max on the packed bf16
// pair (v_pk_max_i16: +NaN passes, -NaN -- what the matrix pipe produces -- becomes 0).
#if (defined(DHAUG_MLP_SAVE_TU) || defined(DHAUG_MLP_NAN_SAFE)) && !defined(SAVE_ABL_NONANSAFE)
constexpr bool NAN_SAFE_TU = true;
#else
constexpr bool NAN_SAFE_TU = false;
#endif
enum { F_OUT_F32 = 4, F_DOT_OUT = 16 };

struct Unit {
    int kind;
    int plan;                   // how the kernel runs the unit, decided on the host (plan_unit): kind and plan are all the
                                // dispatch reads, and it reads them one unit ahead
    int flags;
    int src, dst, res;          // LDS buffer ids (0,1: pitch 256; 2: pitch 128), -1 = none
    int src2, ksteps2;          // optional second source (layers fed by a concatenation), ksteps2 = 0: none
    int ksteps, N, act;
    float slope;
    int cols;                   // LOAD / STORE: columns moved
    long long ld;               // leading dimension (elements) of the global tensor
    const void* g;              // LOAD source / STORE destination / fp32 output of the last layer
    const uint16_t* w;          // GEMM: packed fragments [slice][kstep][64 lanes][8]
    const uint16_t* w2;         // fragments of the second source
    const float* bias;          // GEMM: fp32 [32 * nslices] (zero padded)
    uint16_t* save;             // GEMM, optional: the layer's output image also goes to global memory (M, save_ld) bf16,
    long long save_ld;          // columns [0, ceil16(N)) -- the training step's forward-with-save
    long long save_rows;        // rows [0, save_rows) of the output are saved (the host stores M for "all", 0 for "none")
    uint32_t* bits;             // full-width layer inside a run, optional: one bit per output element, (y > 0) -- the mask
                                // act'(y) of the backward / tangent sweeps in 1/16 of the bytes (see DHAUG_MLP_BITS in dhaug.h)
};
// plan of a GEMM unit.  Stack: bits 0-3 lead k-steps, 4-9 run.  Single layer: bits 16-23 shape (chunks * 16 + chunks of
// source 1), 24-27 feature slices
enum { PLAN_STACK = 1 << 13, PLAN_LEAKY = 1 << 10, PLAN_ALT = 1 << 11, PLAN_TAIL = 1 << 12, PLAN_TAIL_BF16 = 1 << 14, PLAN_OUT = 1 << 28,
       PLAN_DOT = 1 << 29, PLAN_PAIR = 1 << 30 };                            // PAIR: this unit and the next run as one (gemm_pair)

struct Program {
    int nunits;
    int min_run;                                                             // shortest run of 256 -> 256 layers taken by gemm_stack
    Unit u[MLP_MAX_UNITS];
};

__device__ __forceinline__ unsigned char* buf_base(unsigned char* smem, int id) {
    return smem + (id == 0 ? 0 : (id == 1 ? BUF01_BYTES : 2 * BUF01_BYTES));
}
__device__ __forceinline__ int buf_pitch_bytes(int id) { return (id == 2 ? BUF2_PITCH : BUF01_PITCH) * 2; }
// byte offset of 16-byte chunk `c` of row `row`
__device__ __forceinline__ int chunk_off(int row, int c, int pitch_bytes) { return row * pitch_bytes + ((c ^ (row & 15)) << 4); }

// branch-free: `neg` is the slope applied to negative pre-activations (0: ReLU, slope: LeakyReLU, 1: identity)
__device__ __forceinline__ float act_neg(int act, float slope) {
    return act == DHAUG_ACT_RELU ? 0.0f : (act == DHAUG_ACT_LRELU ? slope : 1.0f);
}
// max(v, v*neg) == v for v > 0, v*neg otherwise (0 <= neg <= 1): one mul + one max, no compare/select
__device__ __forceinline__ float act_fn(float v, float neg) { return fmaxf(v, v * neg); }
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {          // v_cvt_pk_bf16_f32
    f32v2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16v2));
}

typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pk_relu(uint32_t packed, uint32_t lb) {     // v_pk_max_i16
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, packed), __builtin_bit_cast(s16x2_t, lb)));
}

constexpr int MLP_CH = 4;                                            // k-steps per ring slot (64 k)
constexpr int OUT_PITCH = 68;                                        // floats per row of the fp32 output staging image
typedef const Unit __attribute__((address_space(4))) * UnitPtr;      // units are read from the kernarg segment (s_load)

#ifdef DHAUG_MLP_TIMING
#ifndef DHAUG_STAMP_TID
#define DHAUG_STAMP_TID 0
#endif
// development aid (not built by default): shader-clock stamps of workgroup 0 at every unit boundary of its first tile
__device__ long long g_mlp_stamps[5 * MLP_MAX_UNITS + 68 + MLP_MAX_UNITS + 1];
#define DHAUG_STAMP(idx)                                                      \
    if (blockIdx.x == 0 && tid == 0) g_mlp_stamps[tile == 0 ? (idx) : 5 * MLP_MAX_UNITS + 68 + (idx)] = (long long)__builtin_readcyclecounter();
#ifdef DHAUG_MLP_TIMING_UNITS                                 /* unit boundaries only: the layer bodies compile as shipped */
#define DHAUG_LSTAMP(idx)
#else
#define DHAUG_LSTAMP(idx) \
    if (blockIdx.x == 0 && threadIdx.x == DHAUG_STAMP_TID) g_mlp_stamps[idx] = (long long)__builtin_readcyclecounter();
#endif
#else
#define DHAUG_STAMP(idx)
#define DHAUG_LSTAMP(idx)
#endif
// ends the basic block (a never-taken scalar branch the compiler cannot see through): the 32-row tiles of a stack layer
// are scheduled and register-allocated one at a time instead of as one 1 200-instruction block
#if defined(DHAUG_MLP_TIMING) && !defined(DHAUG_MLP_TIMING_UNITS)
#define DHAUG_BB_SPLIT()
#else
#define DHAUG_BB_SPLIT()                                  \
    {                                                     \
        int z_ = 0;                                       \
        asm volatile("" : "+s"(z_));                      \
        if (z_) asm volatile("s_nop 0");                  \
    }
#endif

// K is processed in chunks of 64 (4 k-steps); sources narrower than a multiple of 64 are zero-filled by their
// producer (LOAD units pad, GEMM epilogues write whole 32-feature slices) and the packed weights are zero there.
// Every wave always computes two feature slices (wave, wave + 4): the packed weights cover 8 slices (zero rows
// beyond N), which keeps ONE straight-line instruction stream per K shape.
__device__ __forceinline__ int chunks_of(int ksteps) { return (ksteps + MLP_CH - 1) / MLP_CH; }

// issue the global loads of chunk C (of the concatenated sources; the first source has NCH1 chunks) into a ring slot
template <int C, int NCH, int NCH1, int NS>
__device__ __forceinline__ void load_chunk(const uint16_t* w1, const uint16_t* w2, int wave, int lane,
                                           bf16x8 (&slot)[NS][MLP_CH]) {
    constexpr bool second = C >= NCH1;
    constexpr int kpad = (second ? NCH - NCH1 : NCH1) * MLP_CH;              // padded k-steps of this source's blob
    constexpr int k0 = (second ? C - NCH1 : C) * MLP_CH;
    const uint16_t* w = second ? w2 : w1;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const uint16_t* base = w + ((long long)(wave + 4 * t) * kpad * 64 + lane) * 8 + (long long)k0 * 512;
#pragma unroll
        for (int q = 0; q < MLP_CH; ++q) slot[t][q] = *reinterpret_cast<const bf16x8*>(base + q * 512);
    }
}

// One layer, self-contained (no register state crosses layers): NCH chunks of 64 k in total, the first NCH1 from
// source 1.  dst = act(W * src [+ W2 * src2] + bias + res).
//
// Software pipeline, pinned with sched_barrier because hipcc otherwise interleaves the weight loads with the MFMAs
// and then waits for the youngest load (vmcnt(0..2)) at every step:
//   k-step k :  [4 ds_read_b128 of the activation fragments of k+2]  [8 global loads of chunk c+2, once per chunk]
//               -- sched_barrier --   8 MFMAs (2 slices x 4 row tiles) on the fragments read during step k-2
// NS = feature slices this wave really owns in this layer (2: slices wave and wave+4; 1: only slice wave -- layers
// narrower than 160 features); waves with no slice skip the layer.
// everything a layer requests from global memory up front: the first two chunks of its weight fragments, its bias (the
// accumulators' start value), and -- DOT_OUT, see the epilogue -- the lane's 16 weights of the folded logit layer and its bias
template <int NCH, int NCH1, int NS>
__device__ __forceinline__ void layer_requests(UnitPtr u, int wave, int lane, bf16x8 (&ring)[3][NS][MLP_CH], f32x16 (&seed)[NS],
                                               f32x4 (&dv)[4], float& dbias, bool& dot_out) {
    const int h = lane >> 5;
    const uint16_t* w1 = u->w;
    const uint16_t* w2 = NCH1 < NCH ? u->w2 : u->w;
    load_chunk<0, NCH, NCH1, NS>(w1, w2, wave, lane, ring[0]);
    if constexpr (NCH > 1) load_chunk<1, NCH, NCH1, NS>(w1, w2, wave, lane, ring[1]);
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(u->bias + 32 * (wave + 4 * t) + 4 * h + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) seed[t][4 * g + e] = b4[e];
        }
    constexpr bool CAN_DOT = NS == 1 && NCH <= 2 && NCH1 == NCH;             // (shapes checked on the host)
    dot_out = CAN_DOT && (u->flags & F_DOT_OUT);
    dbias = 0.f;
    if (dot_out) {
        const float* v = reinterpret_cast<const float*>(u->w2);
#pragma unroll
        for (int g = 0; g < 4; ++g) dv[g] = *reinterpret_cast<const f32x4*>(v + 32 * wave + 8 * g + 4 * h);
        dbias = (wave == 0 && h == 0) ? v[256] : 0.f;
    }
}

template <int NCH, int NCH1, int NS>
__device__ __forceinline__ void layer_body(UnitPtr u, unsigned char* smem, int wave, int lane, int dbg, bf16x8 (&ring)[3][NS][MLP_CH],
                                           f32x16 (&seed)[NS], f32x4 (&dv)[4], float dbias, bool dot_out);

template <int NCH, int NCH1, int NS>
__device__ __forceinline__ void gemm_layer(UnitPtr u, unsigned char* smem, int wave, int lane, int dbg) {
    DHAUG_LSTAMP(dbg)
    // opaque: everything derived from the lane id below is recomputed per call (a few VALU) instead of being hoisted out
    // of the unit loop and parked on registers the stacks need
    asm volatile("" : "+v"(lane));
    bf16x8 ring[3][NS][MLP_CH];
    f32x16 seed[NS];
    f32x4 dv[4];
    float dbias;
    bool dot_out;
    layer_requests<NCH, NCH1, NS>(u, wave, lane, ring, seed, dv, dbias, dot_out);
    layer_body<NCH, NCH1, NS>(u, smem, wave, lane, dbg, ring, seed, dv, dbias, dot_out);
}

// Two consecutive narrow layers (two chunks of k, one feature slice per wave: the 100-wide top of the critics) as ONE unit: both
// layers' weights, biases and logit vector are requested before the first k loop, so the second layer starts on registers that
// are already there, and the pair costs one dispatch (3 900 + 5 700 + 2 x ~800 clocks of dispatch as two units of the 3D critic's
// 112 000-clock tile; phase stamps).  Same arithmetic as the two units.
__device__ __forceinline__ void lds_barrier();
__device__ __forceinline__ void gemm_pair(UnitPtr ua, UnitPtr ub, unsigned char* smem, int wave, int lane, int dbg) {
    DHAUG_LSTAMP(dbg)
    asm volatile("" : "+v"(lane));
    bf16x8 ra[3][1][MLP_CH], rb[3][1][MLP_CH];
    f32x16 sa[1], sb[1];
    f32x4 dva[4], dvb[4];
    float da, db;
    bool dota, dotb;
    layer_requests<2, 2, 1>(ua, wave, lane, ra, sa, dva, da, dota);
    layer_requests<2, 2, 1>(ub, wave, lane, rb, sb, dvb, db, dotb);
    layer_body<2, 2, 1>(ua, smem, wave, lane, dbg, ra, sa, dva, da, false);
    lds_barrier();
    DHAUG_LSTAMP(dbg + 2)
    layer_body<2, 2, 1>(ub, smem, wave, lane, dbg + 4, rb, sb, dvb, db, dotb);
}

template <int NCH, int NCH1, int NS>
__device__ __forceinline__ void layer_body(UnitPtr u, unsigned char* smem, int wave, int lane, int dbg, bf16x8 (&ring)[3][NS][MLP_CH],
                                           f32x16 (&seed)[NS], f32x4 (&dv)[4], float dbias, bool dot_out) {
    const int r31 = lane & 31, h = lane >> 5;
    const uint16_t* w1 = u->w;
    const uint16_t* w2 = NCH1 < NCH ? u->w2 : u->w;
    f32x16 acc[NS][MLP_MT];
    const unsigned char* src1 = buf_base(smem, u->src);
    const int pbs1 = buf_pitch_bytes(u->src);
    const unsigned char* src2 = NCH1 < NCH ? buf_base(smem, u->src2) : src1;
    const int pbs2 = NCH1 < NCH ? buf_pitch_bytes(u->src2) : pbs1;
    constexpr int KT = NCH * MLP_CH;                            // k-steps in total
    bf16x8 fx[3][MLP_MT];                                       // activation fragments: read two k-steps ahead (one wave per
                                                                // SIMD: nobody else hides the LDS latency)
    auto read_frags = [&](int k, bf16x8 (&f)[MLP_MT]) {              // k is a compile-time constant after unrolling
        const bool second = k >= NCH1 * MLP_CH;
        const unsigned char* src = second ? src2 : src1;
        const int pbs = second ? pbs2 : pbs1;
        const int kk = second ? k - NCH1 * MLP_CH : k;
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt)
            f[mt] = *reinterpret_cast<const bf16x8*>(src + chunk_off(32 * mt + r31, 2 * kk + h, pbs));
    };
    read_frags(0, fx[0]);
    if (KT > 1) read_frags(1, fx[1]);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int c = k / MLP_CH, q = k % MLP_CH;
        if (k + 2 < KT) read_frags(k + 2, fx[(k + 2) % 3]);
        if (q == 0) {
            if (c + 2 < NCH) {
                if (c + 2 == 2) load_chunk<2 < NCH ? 2 : 0, NCH, NCH1, NS>(w1, w2, wave, lane, ring[2]);
                if (c + 2 == 3) load_chunk<3 < NCH ? 3 : 0, NCH, NCH1, NS>(w1, w2, wave, lane, ring[0]);
                if (c + 2 == 4) load_chunk<4 < NCH ? 4 : 0, NCH, NCH1, NS>(w1, w2, wave, lane, ring[1]);
                if (c + 2 == 5) load_chunk<5 < NCH ? 5 : 0, NCH, NCH1, NS>(w1, w2, wave, lane, ring[2]);
                if (c + 2 == 6) load_chunk<6 < NCH ? 6 : 0, NCH, NCH1, NS>(w1, w2, wave, lane, ring[0]);
                if (c + 2 == 7) load_chunk<7 < NCH ? 7 : 0, NCH, NCH1, NS>(w1, w2, wave, lane, ring[1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt)
#pragma unroll
            for (int t = 0; t < NS; ++t)
                acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[c % 3][t][q], fx[k % 3][mt], k == 0 ? seed[t] : acc[t][mt],
                                                                    0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    DHAUG_LSTAMP(dbg + 1)
    const int nslices = (u->N + 31) >> 5;
    const bool to_global = (u->flags & F_OUT_F32) != 0;
    unsigned char* dst = buf_base(smem, u->dst);
    const int pbd = buf_pitch_bytes(u->dst);
    const int resid = u->res;
    if (resid >= 0 && !to_global) {
        // residual: two more k-steps against identity fragments (A[n][k'] = (n == 16 ks2 + k'), the lane's k' = 8h + j),
        // B = the residual image's chunks of this slice -- exact (1.0 * bf16 accumulated in fp32), no unpack on the VALU
        const unsigned char* res = buf_base(smem, resid);
        const int pbr = buf_pitch_bytes(resid);
        bf16x8 idf[2];
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const int dd = r31 - 16 * ks2 - 8 * h;
            u32x4_t v;
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) v[p2] = (dd == 2 * p2 ? 0x3F80u : 0u) | (dd == 2 * p2 + 1 ? 0x3F800000u : 0u);
            idf[ks2] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int slice = wave + 4 * t;
            if (slice >= nslices) continue;
            bf16x8 rf[MLP_MT][2];
#pragma unroll
            for (int mt = 0; mt < MLP_MT; ++mt)
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2)
                    rf[mt][ks2] = *reinterpret_cast<const bf16x8*>(res + chunk_off(32 * mt + r31, 4 * slice + 2 * ks2 + h, pbr));
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int mt = 0; mt < MLP_MT; ++mt)
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf[ks2], rf[mt][ks2], acc[t][mt], 0, 0, 0);
        }
    }
    const int act = u->act;
    if (dot_out) {
        // this layer feeds nothing but a 1-wide linear layer (a critic's logit): instead of storing the activation image and
        // running one more unit over it, every lane multiplies its 16 features per row (rounded to bf16 and activated
        // exactly as they would have been stored) with that layer's weights and leaves per-(wave, half) partial sums in
        // buffer dst as fp32 [8][128]; dot_output() adds them up.  w2 = fp32 [257]: the weights (bf16 values, zero beyond
        // N) and the bias at [256] (joins wave 0's partial sum).
        const float neg = act_neg(act, u->slope);
        const uint32_t lb = act == DHAUG_ACT_RELU ? 0u : 0x80008000u;
        // two partial sums per row tile (even / odd feature groups): eight independent FMA chains, not four serial ones
        float part[MLP_MT][2];
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt) { part[mt][0] = dbias; part[mt][1] = 0.f; }
        auto dot_pairs = [&](auto activate) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; e += 2)
#pragma unroll
                    for (int mt = 0; mt < MLP_MT; ++mt) {
                        const uint32_t pk = activate(acc[0][mt][4 * g + e], acc[0][mt][4 * g + e + 1]);
                        part[mt][g & 1] = fmaf(__builtin_bit_cast(float, pk << 16), dv[g][e], part[mt][g & 1]);
                        part[mt][g & 1] = fmaf(__builtin_bit_cast(float, pk & 0xffff0000u), dv[g][e + 1], part[mt][g & 1]);
                    }
        };
        if (wave < nslices) {
            if (act != DHAUG_ACT_LRELU && !NAN_SAFE_TU) dot_pairs([&](float a0, float a1) { return pk_relu(pack_bf16x2(a0, a1), lb); });
            else dot_pairs([&](float a0, float a1) { return pack_bf16x2(act_fn(a0, neg), act_fn(a1, neg)); });
        }
        float* st = reinterpret_cast<float*>(dst) + (2 * wave + h) * MLP_BM + r31;
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt) st[32 * mt] = part[mt][0] + part[mt][1];
        return;
    }
    // epilogue: this lane owns row (32 mt + r31), features 32*slice + 8g + 4h .. +3
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int slice = wave + 4 * t;
        if (slice >= nslices) continue;                        // wave-uniform: slices beyond N are never stored
        if (to_global) {
            // network output (<= 64 features): fp32 staging image [128][OUT_PITCH] in buffer dst, copied out
            // cooperatively by store_output() after the barrier
            const float neg = act_neg(act, u->slope);
            float* st = reinterpret_cast<float*>(dst);
#pragma unroll
            for (int mt = 0; mt < MLP_MT; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[t][mt][4 * g + e], neg);
                    *reinterpret_cast<f32x4*>(st + (32 * mt + r31) * OUT_PITCH + 32 * slice + 4 * h + 8 * g) = v;
                }
        } else if (act != DHAUG_ACT_LRELU && !NAN_SAFE_TU) {
            // ReLU on the packed pair: max as int16 against 0 (a negative bf16 is a negative int16); against INT16_MIN it
            // is the identity
            const uint32_t lb = act == DHAUG_ACT_RELU ? 0u : 0x80008000u;
#pragma unroll
            for (int mt = 0; mt < MLP_MT; ++mt) {
                const int row = 32 * mt + r31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o;
                    o.x = pk_relu(pack_bf16x2(acc[t][mt][4 * g + 0], acc[t][mt][4 * g + 1]), lb);
                    o.y = pk_relu(pack_bf16x2(acc[t][mt][4 * g + 2], acc[t][mt][4 * g + 3]), lb);
                    *reinterpret_cast<uint2*>(dst + chunk_off(row, 4 * slice + g, pbd) + (h << 3)) = o;
                }
            }
        } else {
            const float neg = act_neg(act, u->slope);
#pragma unroll
            for (int mt = 0; mt < MLP_MT; ++mt) {
                const int row = 32 * mt + r31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o;
                    o.x = pack_bf16x2(act_fn(acc[t][mt][4 * g + 0], neg), act_fn(acc[t][mt][4 * g + 1], neg));
                    o.y = pack_bf16x2(act_fn(acc[t][mt][4 * g + 2], neg), act_fn(acc[t][mt][4 * g + 3], neg));
                    *reinterpret_cast<uint2*>(dst + chunk_off(row, 4 * slice + g, pbd) + (h << 3)) = o;
                }
            }
        }
    }
}

template <int NCH, int NCH1, int NS>
__device__ __forceinline__ void gemm_single(UnitPtr u, unsigned char* smem, int wave, int lane, int dbg) {
    gemm_layer<NCH, NCH1, NS>(u, smem, wave, lane, dbg);
    DHAUG_LSTAMP(dbg + 2)
}

// ---------------------------------------------------------------------------------------------------------------
// Runs of consecutive full-width 256 -> 256 layers (the residual stacks: 6 of the generator's 8 layers, 12 of the 3D
// critic's 17).  At K = 256 the epilogue (bias, residual, activation, bf16 pack: ~5 VALU per element) costs about as
// many issue cycles as the layer's MFMAs, so it has to run BESIDE them:
//   * a wave holds the layer's complete weight fragments for its two feature slices in registers (128 VGPRs) and the
//     NEXT layer's are loaded into a second set while this layer computes (512-register budget, one wave per SIMD);
//   * the four 32-row tiles are processed one after the other; the epilogue of tile mt-1 sits in the same basic block
//     as the 32 MFMAs of tile mt, so the scheduler interleaves ~5 VALU per MFMA gap (separate pipes); only the last
//     tile's epilogue is exposed.
// ---------------------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, which would expose the latency
// of the weight fragments a wave has in flight for the next layer.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef bf16x8 WHalf[MLP_NS][MLP_MAX_KSTEPS / 2];                     // fragments of k-steps 0..7 or 8..15

// what a stack layer needs from its unit, fetched (scalar loads) one layer ahead so that no layer starts by waiting
// on the constant cache
struct StackDesc {
    int src, dst, res, act;
    float slope;
    const float* bias;
    const uint16_t* w;
    int narrow;                 // 1: the network's output layer (N <= 64): every wave holds slices 0 and 1
    int nslices;                // feature slices of the unit (tail_gemm: waves beyond them skip the layer)
};
__device__ __forceinline__ StackDesc stack_desc(UnitPtr u) {
    StackDesc d;
    d.src = u->src; d.dst = u->dst; d.res = u->res; d.act = u->act; d.slope = u->slope; d.bias = u->bias; d.w = u->w;
    d.narrow = (u->flags & F_OUT_F32) ? 1 : 0;
    d.nslices = (u->N + 31) >> 5;
    return d;
}

// a layer's fragments ([slice][KS k-steps][64 lanes][8]) are fetched with buffer loads: a scalar resource that starts at
// the wave's first slice, ONE per-lane offset (16 * lane) for everything, slice and k-step as scalar offset + immediate
// (measured beside MFMAs, tools/ubench/vmem_gap.hip: a buffer_load_dwordx4 costs ~4 issue cycles, a global_load_dwordx4 with
// its 64-bit per-lane address ~14 -- and eight lane-dependent base pointers would sit in registers for the whole kernel)
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int KS = MLP_MAX_KSTEPS>
struct WBase {
    __amdgpu_buffer_rsrc_t rs;
    int lane16, sstride;
    // the wave's slices are s0 and s0 + sstride (wave, wave + 4 in a full-width layer)
    __device__ __forceinline__ WBase(const uint16_t* w, int s0, int sstride_, int lane) {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(w) + (long long)s0 * KS * 512, 0, 0x7fffffff, 0x27000);
        lane16 = lane * 16;
        sstride = sstride_;
    }
    __device__ __forceinline__ bf16x8 frag(int t, int k) const {
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, (t * sstride * KS + k) * 1024, 0));
    }
};

__device__ __forceinline__ void load_seed(const float* bias, int s0, int sstride, int lane, f32x16 (&seed)[MLP_NS]) {
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < MLP_NS; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 32 * (s0 + sstride * t) + 4 * h + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) seed[t][4 * g + e] = b4[e];
        }
}

typedef short s16x2 __attribute__((ext_vector_type(2)));

// One layer of a stack.  On entry `w`/`seed` hold this layer's weight fragments and bias; on exit those of layer
// `nd` (the layer itself again for the last one): each fragment is re-requested right behind its last use in the
// layer's last tile, so the next layer's weights arrive during that tile and the drain, with no second register set.
// All images of a stack are the 512-byte-pitch buffers 0/1, so every LDS address is (per-lane constant) ^ (compile-
// time constant) + immediate.
// The wave has ONE issue port and a 32-cycle MFMA gap holds only ~6 VALU slots, fewer beside LDS instructions
// (tools/ubench/mfma_gap.hip), so everything that can ride the matrix pipe does: the bias seeds the accumulators and
// the residual is added by two extra k-steps against identity fragments (exact: 1.0 * bf16 in fp32).  What is left of
// the epilogue -- bf16 pack, ReLU on the packed pair (v_pk_max_i16 against 0: a negative bf16 is a negative int16;
// against INT16_MIN it is the identity), one ds_write_b64 per 4 elements -- is dealt out behind the MFMAs of the NEXT
// tile and pinned there (sched_barrier).  LEAKY stacks (the 2D critic) pay mul + max per element instead.
// One body per stack: a second instantiation inside the layer loop would make the register allocator shuffle the
// 128 weight registers between the bodies' assignments at every layer boundary.
// RESMODE 0: no residual, 1: residual, 2: decided at run time (the identity fragments are zeroed for a layer without).
// KS < 16: the narrow layer that feeds a stack (K = 16 KS <= 128, no residual); its own fragments sit in `wlo`, and
// BOTH halves of the following layer's are requested while it runs (tile 1: low, tile 2: high).
// SAVE (forward-with-save): the layer also writes its INPUT image to global memory (`sv`: buffer resource of the 128-row
// tile of the producer's save target, sv_ld bytes per row; zero records = nothing to save).  *(r5)* Wave w copies rows
// [32 w, 32 w + 32) of the source image as whole rows: every fourth k-step one ds_read_b128 in which consecutive lanes take
// consecutive 16-byte chunks (lane = 32 (row & 1) + chunk: two 512-byte rows per instruction), and two k-steps later one
// buffer store of the same shape -- sixteen fully coalesced 1 KB stores per wave, layer and tile, dealt out evenly over the
// layer's four row tiles (rows beyond M fall outside the resource and are dropped by the range check; the source image is
// read-only while the layer runs).  The first form stored the activation FRAGMENTS the wave reads for the matrix pipe (no
// LDS read): 16 bytes per lane of 32 different rows -- every quad of lanes four separate 16-byte accesses for the address
// coalescer -- bunched in ONE row tile per wave.  tools/ubench/store_issue.hip (this kernel's operating point: one wave per
// SIMD, 32 KB of weight fragments per wave and layer tile on the same vector-memory path, target resident in L2): the
// fragment pattern takes a layer tile from 2.00 to 4.08 us, whole rows bunched 3.72, whole rows spread 2.41.
// (tried, r5: v_maximum3_f32 v, 0, 0 -- gfx950's NaN-propagating maximum -- as the forward-with-save unit's ReLU in place of
// max(v, v * neg): ONE instruction per element instead of two, and the 3D critic's forward-with-save launch went from 510 to
// 717 us; the instruction is far from full rate on this part.  Removed.)
#ifdef SAVE_ABL_NOBITS                      /* timing only: no sign bits are computed or stored */
#define SAVE_BITS_DEFAULT false
#else
#define SAVE_BITS_DEFAULT SAVE
#endif
template <bool LEAKY, int RESMODE, int KS = MLP_MAX_KSTEPS, bool SAVE = false, bool BITS = SAVE_BITS_DEFAULT>
__device__ __forceinline__ void stack_layer(const StackDesc& d, const StackDesc& nd, unsigned char* smem, int wave, int lane,
                                            const WHalf& wlo, WHalf& nlo, WHalf& whi, f32x16 (&seed)[MLP_NS],
                                            const bf16x8 (&idf)[2], int dbg, __amdgpu_buffer_rsrc_t sv, int sv_ld,
                                            __amdgpu_buffer_rsrc_t brs) {
    DHAUG_LSTAMP(dbg)
    static_assert(KS == 16 || ((KS == 4 || KS == 8) && RESMODE == 0), "layer shape");
    constexpr bool LEAD = KS < MLP_MAX_KSTEPS;
    constexpr int PAIRS = 16 / KS;                                           // accumulator element pairs retired per k-step
    constexpr int TILE_BYTES = 32 * BUF01_PITCH * 2;                         // 16 384
    const int r31 = lane & 31, h = lane >> 5, x = lane & 15;
    f32x16 nseed[MLP_NS];                                                    // requested in the last tile: asked for at the layer's start
    const int ns0 = nd.narrow ? 0 : wave, nss = nd.narrow ? 1 : 4;           // they are parked right away, behind a full vmcnt drain
    const WBase<> nb(nd.w, ns0, nss, lane);
    const unsigned char* src = buf_base(smem, d.src);
    unsigned char* dst = buf_base(smem, d.dst);
    const bool has_res = d.res >= 0;
    const unsigned char* res = buf_base(smem, has_res ? d.res : 0);
    bf16x8 idl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bf16x8 zf = {0, 0, 0, 0, 0, 0, 0, 0};
        idl[i] = (RESMODE == 2 && !has_res) ? zf : idf[i];
    }
    const float neg = act_neg(d.act, d.slope);
    const uint32_t lb = d.act == DHAUG_ACT_RELU ? 0u : 0x80008000u;        // packed int16 lower bound
    const int lfx = r31 * (BUF01_PITCH * 2) | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4);      // ^ (k << 5): chunk 2k+h of row
    const int lrx = lfx ^ (wave << 6);                                                      // ^ (t << 8 | ks2 << 5): k-step 2(wave+4t)+ks2
    const int lep = r31 * (BUF01_PITCH * 2) | (((4 * wave) ^ x) << 4) | (h << 3);          // ^ ((16t+g) << 4): chunk 4(wave+4t)+g
    constexpr int FXD = 3;                                                   // fx prefetch distance in k-steps
    f32x16 acc[2][MLP_NS];                                                   // tile mt accumulates while tile mt-1 drains
    bf16x8 fx[4], rx[4];
    // the eight swizzled chunk addresses of a row's first 256 bytes, kept in registers for the layer (opaque, so that they
    // are not re-derived from parked constants with two VALU ops in front of every read); k-steps 8..15 and the tile are
    // immediate offsets
    typedef const unsigned char __attribute__((address_space(3))) * LdsPtr;
    uint32_t fxa[8];
    {
        const uint32_t sb = (uint32_t)(uintptr_t)(LdsPtr)src;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fxa[i] = sb + (uint32_t)(lfx ^ (i << 5));
            asm volatile("" : "+v"(fxa[i]));
        }
    }
    auto fx_load = [&](int s) {                                              // s = KS * tile + k-step
        const int k = s % KS;
        fx[s & 3] = *reinterpret_cast<const bf16x8 __attribute__((address_space(3)))*>((LdsPtr)(uintptr_t)fxa[k & 7] + ((k >> 3) << 8) +
                                                                                     (s / KS) * TILE_BYTES);
    };
#pragma unroll
    for (int s = 0; s < FXD; ++s) fx_load(s);
    // the epilogue write addresses (chunk 4(wave+4t)+g of the lane's row): four swizzle variants in registers, t and the
    // tile are immediate offsets
    uint32_t wa[4];
    {
        const uint32_t db = (uint32_t)(uintptr_t)(LdsPtr)dst;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            wa[g] = db + (uint32_t)(lep ^ (g << 4));
            asm volatile("" : "+v"(wa[g]));
        }
    }
    typedef unsigned char __attribute__((address_space(3))) * LdsWPtr;
    // elements j, j+1 (j even) of a tile held in `a`: j = 16t + 4g + e -> activation, packed bf16 pair
    auto pair_pack = [&](const f32x16 (&a)[MLP_NS], int j) -> uint32_t {
        const int t = j >> 4, g = (j >> 2) & 3, e = j & 3;
        const float v0 = a[t][4 * g + e], v1 = a[t][4 * g + e + 1];
        if (!LEAKY)
            return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16x2(v0, v1)),
                                                                          __builtin_bit_cast(s16x2, lb)));
        return pack_bf16x2(act_fn(v0, neg), act_fn(v1, neg));
    };
    // the four elements j .. j+3 (j a multiple of 4) of tile `mt`: one ds_write_b64
    auto quad_store = [&](int mt, int j, uint32_t p0, uint32_t p1) {
        const int t = j >> 4, g = (j >> 2) & 3;
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 oo = {p0, p1};
#ifdef ABL_NOWRITE
        asm volatile("" :: "v"(oo));
#else
        *reinterpret_cast<u32x2 __attribute__((address_space(3)))*>((LdsWPtr)(uintptr_t)wa[g] + (t << 8) + mt * TILE_BYTES) = oo;
#endif
    };
    // A wave has one issue port and every MFMA leaves a 28-cycle gap behind it; measured issue costs (tools/ubench/
    // mfma_gap.hip): VALU 4, ds_read_b128 ~26, ds_write_b64 ~28, a 16-byte global load ~16.  What exceeds a gap's 28 cycles
    // is lost on the matrix pipe, what stays below is not won back, so the filler is dealt out evenly and pinned
    // (sched_barrier on both sides of every MFMA).  Per two k-steps (4 MFMAs):
    //   after (k even, t0)  activation fragment of k+3                       ~30
    //   after (k even, t1)  ds_write_b64 of the two pairs packed one step ago ~28
    //   after (k odd,  t0)  activation fragment of k+3                       ~30
    //   after (k odd,  t1)  one fragment of the next layer's weights + pack/ReLU of two pairs of the previous tile ~32
    uint32_t st0 = 0, st1 = 0;                                               // packed pairs waiting for their store
    // SAVE: sign bits of the tile's output, (y > 0) per element: pair p = j / 2 of the lane's 32 elements puts its even element at
    // bit p and its odd element at bit 16 + p (so that a consumer turns pair p into a 16-bit lane mask with one shift and one
    // packed arithmetic shift); one dword per lane and 32-row tile at brs[((mt * 4 + wave) * 64 + lane) * 4]
    uint32_t bacc = 0;
    auto bits_or = [&](uint32_t packed, int pidx) {
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const s16x2 zero2 = {0, 0};
        const u16x2 one2 = {1, 1};
        const s16x2 pos = __builtin_elementwise_max(__builtin_bit_cast(s16x2, packed), zero2);       // negative -> 0
        const u16x2 m = __builtin_elementwise_min(__builtin_bit_cast(u16x2, pos), one2);              // positive -> 1
        bacc |= __builtin_bit_cast(uint32_t, m) << pidx;
    };
    auto bits_store = [&](int mt) {
        __builtin_amdgcn_raw_buffer_store_b32((int)bacc, brs, ((mt * 4 + wave) * 64 + lane) * 4, 0, 0);
        bacc = 0;
    };
    // SAVE: the in-layer copy of the source image (see above).  Row pair p (0..15) of the wave: rows 32 wave + 2 p + h', the lane's
    // chunk c = lane & 31, h' = lane >> 5.  The image swizzle c ^ (row & 15) = (c ^ h') ^ (2 p & 15): one XOR with a constant per read.
    uint32_t cp_lds = 0;
    int cp_goff = 0;
    i32x4 cpv = {0, 0, 0, 0};
    // No branch around the copy: a tile / layer with nothing to save (sv_ld = 0, zero records) still reads and issues stores that
    // the range check drops -- 32 wave-uniform branches per layer and tile in the matrix-instruction stream cost more than that
    // (SAVE_ABL_BRANCH builds the branching form for comparison).
#if defined(SAVE_ABL_NOSTORE)
    const bool sv_on = false;
#elif defined(SAVE_ABL_BRANCH)
    const bool sv_on = SAVE && !LEAD && sv_ld != 0;                          // (wave-uniform)
#else
    const bool sv_on = SAVE && !LEAD;
#endif
    if (SAVE && !LEAD) {
        const int c = lane & 31, row = 32 * wave + h;
        cp_lds = (uint32_t)(uintptr_t)(LdsPtr)src + (uint32_t)(row * (BUF01_PITCH * 2) + ((c ^ h) << 4));
        cp_goff = row * sv_ld + (c << 4);
    }
#pragma unroll
    for (int mt = 0; mt < MLP_MT; ++mt) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int s = mt * KS + k;
#pragma unroll
            for (int t = 0; t < MLP_NS; ++t) {
                acc[mt & 1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k < 8 ? wlo[t][k & 7] : whi[t][k & 7], fx[s & 3],
                                                                         k == 0 ? seed[t] : acc[mt & 1][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t == 0) {
                    if (mt == MLP_MT - 1 && k == 0) load_seed(nd.bias, ns0, nss, lane, nseed);
#ifndef ABL_NOREAD
                    if (s + FXD < MLP_MT * KS) fx_load(s + FXD);
#endif
                    if (RESMODE != 0 && k < 4)                               // this tile's residual fragments, used after k = 15
                        rx[k] = *reinterpret_cast<const bf16x8*>(res + (lrx ^ ((k >> 1) << 8 | (k & 1) << 5)) + mt * TILE_BYTES);
                }
                // the next layer's weights.  The vector-memory return path moves 64 B/clk: a layer's 128 KB of
                // fragments are half its MFMA time, so they must not bunch up.  k-steps 0..7 go to the second low
                // set, one fragment every fourth MFMA of tiles 1 and 2; k-steps 8..15 replace this layer's right
                // behind their last use in tile 3.
#ifndef ABL_NOWLOAD
                if (!LEAD) {
                    if ((mt == 1 || mt == 2) && (k & 1) == 1 && t == 1) {
                        const int f = (mt - 1) * 8 + (k >> 1);               // 0..15 -> (slice f >> 3, k-step f & 7)
                        nlo[f >> 3][f & 7] = nb.frag(f >> 3, f & 7);
                    }
                    if (mt == MLP_MT - 1 && k >= 8) whi[t][k & 7] = nb.frag(t, k);
                } else if (mt == 1 || mt == 2) {                             // 16 fragments over the tile's 2 KS MFMAs
#pragma unroll
                    for (int q = 0; q < 8 / KS; ++q) {
                        const int f = (MLP_NS * k + t) * (8 / KS) + q;       // 0..15
                        if (mt == 1) nlo[f >> 3][f & 7] = nb.frag(f >> 3, f & 7);
                        else whi[f >> 3][f & 7] = nb.frag(f >> 3, 8 + (f & 7));
                    }
                }
#endif
                if (SAVE && !LEAD && t == 0 && (k & 1) && sv_on) {           // row pair p = 4 mt + k / 4: read at k % 4 == 1, stored at k % 4 == 3
                    constexpr int ROWP = 2 * BUF01_PITCH * 2;                // bytes of a row pair in the image
                    const int p = 4 * mt + (k >> 2);
                    if ((k & 3) == 1)
                        cpv = *reinterpret_cast<const i32x4 __attribute__((address_space(3)))*>(
                            (LdsPtr)(uintptr_t)(cp_lds ^ (uint32_t)(((2 * p) & 15) << 4)) + p * ROWP);
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(cpv, sv, cp_goff, 2 * p * sv_ld, 0);
                }
                if (t == 1) {
                    if (!LEAD) {                                             // pairs 2(k-1), 2k are packed at odd k, stored at k+1
                        if (mt > 0 && (k & 1)) {
                            st0 = pair_pack(acc[(mt - 1) & 1], 2 * k - 2);
                            st1 = pair_pack(acc[(mt - 1) & 1], 2 * k);
                            if (BITS) {
                                bits_or(st0, k - 1);
                                bits_or(st1, k);
                                if (k == KS - 1) bits_store(mt - 1);         // all sixteen pairs of tile mt-1 are packed
                            }
                        } else if (mt > 0 && k >= 2) {
                            quad_store(mt - 1, 2 * k - 4, st0, st1);
                        } else if (mt > 1 && k == 0) {
                            quad_store(mt - 2, 28, st0, st1);
                        }
                    } else if (mt > 0) {
#pragma unroll
                        for (int q = 0; q < PAIRS; q += 2) {
                            const int j = 2 * (PAIRS * k + q);
                            const uint32_t p0 = pair_pack(acc[(mt - 1) & 1], j), p1 = pair_pack(acc[(mt - 1) & 1], j + 2);
                            quad_store(mt - 1, j, p0, p1);
                            if (BITS) {
                                bits_or(p0, j >> 1);
                                bits_or(p1, (j >> 1) + 1);
                            }
                        }
                        if (BITS && k == KS - 1) bits_store(mt - 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (RESMODE == 1 || (RESMODE == 2 && has_res)) {                   // (run-time case: a wave-uniform branch at the tile boundary)
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                    // (t, ks2) = (i & 1, i >> 1): alternate the accumulators
                acc[mt & 1][i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idl[i >> 1], rx[2 * (i & 1) + (i >> 1)], acc[mt & 1][i & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        DHAUG_LSTAMP(dbg + 1 + mt)
        DHAUG_BB_SPLIT()
    }
    if (!LEAD) quad_store(MLP_MT - 2, 28, st0, st1);
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
        const uint32_t p0 = pair_pack(acc[(MLP_MT - 1) & 1], j), p1 = pair_pack(acc[(MLP_MT - 1) & 1], j + 2);
        quad_store(MLP_MT - 1, j, p0, p1);
        if (BITS) {
            bits_or(p0, j >> 1);
            bits_or(p1, (j >> 1) + 1);
        }
    }
    if (BITS) bits_store(MLP_MT - 1);
    DHAUG_LSTAMP(dbg + 5)
#pragma unroll
    for (int t = 0; t < MLP_NS; ++t) seed[t] = nseed[t];
}

// The network's output layer (N <= 64, K = 256) behind a stack: wave w computes rows [32w, 32w+32) for slices 0 and 1
// with the fragments the last stack layer requested for it, and leaves act(.) as fp32 in the staging image
// [128][OUT_PITCH] of buffer dst (store_output copies it out).
__device__ __forceinline__ void tail_layer(const StackDesc& d, unsigned char* smem, int wave, int lane, const WHalf& wlo,
                                           const WHalf& whi, const f32x16 (&seed)[MLP_NS]) {
    constexpr int TILE_BYTES = 32 * BUF01_PITCH * 2;
    const int r31 = lane & 31, h = lane >> 5, x = lane & 15;
    const unsigned char* src = buf_base(smem, d.src) + wave * TILE_BYTES;
    float* st = reinterpret_cast<float*>(buf_base(smem, d.dst)) + (32 * wave + r31) * OUT_PITCH + 4 * h;
    const float neg = act_neg(d.act, d.slope);
    const int lfx = r31 * (BUF01_PITCH * 2) | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4);
    bf16x8 fx[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) fx[k] = *reinterpret_cast<const bf16x8*>(src + (lfx ^ (k << 5)));
    f32x16 acc[MLP_NS];
#pragma unroll
    for (int k = 0; k < MLP_MAX_KSTEPS; ++k) {
        if (k + 3 < MLP_MAX_KSTEPS) fx[(k + 3) & 3] = *reinterpret_cast<const bf16x8*>(src + (lfx ^ ((k + 3) << 5)));
#pragma unroll
        for (int t = 0; t < MLP_NS; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k < 8 ? wlo[t][k & 7] : whi[t][k & 7], fx[k & 3],
                                                             k == 0 ? seed[t] : acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < MLP_NS; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[t][4 * g + e], neg);
            *reinterpret_cast<f32x4*>(st + 32 * t + 8 * g) = v;
        }
}

// A layer of at most 128 features with K = 256 right behind a stack (the 3D critic's merge halves): the last stack layer
// requested its fragments like a next stack layer's (slices wave and wave + 4; only `wave` is used), so the unit starts
// with its weights and bias in registers instead of an L2 round trip.  dst = act(W src + bias [+ res]) as bf16 into any
// buffer; wave w computes slice w for the four 32-row tiles.
__device__ __forceinline__ void tail_gemm(const StackDesc& d, unsigned char* smem, int wave, int lane, const WHalf& wlo,
                                          const WHalf& whi, const f32x16 (&seed)[MLP_NS]) {
    if (wave >= d.nslices) return;
    const int r31 = lane & 31, h = lane >> 5;
    const unsigned char* src = buf_base(smem, d.src);
    const int pbs = buf_pitch_bytes(d.src);
    bf16x8 fx[3][MLP_MT];
    auto read_frags = [&](int k, bf16x8 (&f)[MLP_MT]) {
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt)
            f[mt] = *reinterpret_cast<const bf16x8*>(src + chunk_off(32 * mt + r31, 2 * k + h, pbs));
    };
    read_frags(0, fx[0]);
    read_frags(1, fx[1]);
    f32x16 acc[MLP_MT];
#pragma unroll
    for (int k = 0; k < MLP_MAX_KSTEPS; ++k) {
        if (k + 2 < MLP_MAX_KSTEPS) read_frags(k + 2, fx[(k + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k < 8 ? wlo[0][k & 7] : whi[0][k & 7], fx[k % 3][mt],
                                                              k == 0 ? seed[0] : acc[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned char* dst = buf_base(smem, d.dst);
    const int pbd = buf_pitch_bytes(d.dst);
    if (d.res >= 0) {                                       // residual: two k-steps against identity fragments (see gemm_layer)
        const unsigned char* res = buf_base(smem, d.res);
        const int pbr = buf_pitch_bytes(d.res);
        bf16x8 rf[MLP_MT][2];
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
                rf[mt][ks2] = *reinterpret_cast<const bf16x8*>(res + chunk_off(32 * mt + r31, 4 * wave + 2 * ks2 + h, pbr));
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const int dd = r31 - 16 * ks2 - 8 * h;
            u32x4_t v;
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) v[p2] = (dd == 2 * p2 ? 0x3F80u : 0u) | (dd == 2 * p2 + 1 ? 0x3F800000u : 0u);
            const bf16x8 idf = __builtin_bit_cast(bf16x8, v);
#pragma unroll
            for (int mt = 0; mt < MLP_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf, rf[mt][ks2], acc[mt], 0, 0, 0);
        }
    }
    if (d.act != DHAUG_ACT_LRELU && !NAN_SAFE_TU) {
        const uint32_t lb = d.act == DHAUG_ACT_RELU ? 0u : 0x80008000u;
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = pk_relu(pack_bf16x2(acc[mt][4 * g + 0], acc[mt][4 * g + 1]), lb);
                o.y = pk_relu(pack_bf16x2(acc[mt][4 * g + 2], acc[mt][4 * g + 3]), lb);
                *reinterpret_cast<uint2*>(dst + chunk_off(32 * mt + r31, 4 * wave + g, pbd) + (h << 3)) = o;
            }
    } else {
        const float neg = act_neg(d.act, d.slope);
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = pack_bf16x2(act_fn(acc[mt][4 * g + 0], neg), act_fn(acc[mt][4 * g + 1], neg));
                o.y = pack_bf16x2(act_fn(acc[mt][4 * g + 2], neg), act_fn(acc[mt][4 * g + 3], neg));
                *reinterpret_cast<uint2*>(dst + chunk_off(32 * mt + r31, 4 * wave + g, pbd) + (h << 3)) = o;
            }
    }
}

// data-movement units.  LOAD zero-fills columns [cols, ceil64(cols)) so that the consuming GEMM may read whole chunks.
// (row, col-group) walker for a workgroup-strided sweep of a [MLP_BM][q] grid: no division inside the loops
struct Sweep {
    int row, c, dr, dc, q;
    __device__ __forceinline__ Sweep(int tid, int q_) : q(q_) {
        row = tid / q_; c = tid - row * q_; dr = MLP_THREADS / q_; dc = MLP_THREADS - dr * q_;
    }
    __device__ __forceinline__ void next() {
        row += dr; c += dc;
        if (c >= q) { c -= q; ++row; }
    }
};

// forward-with-save: the bf16 image a layer left in buffer `id` (columns [0, cols), cols a multiple of 16) also goes to
// global memory, 16 bytes per thread and access.  Called behind the barrier that completes the image; the buffer is not
// written again before the next barrier (buffers alternate), and nobody waits for the stores.
__device__ __forceinline__ void save_image(int id, uint16_t* g, long long ld, int cols, unsigned char* smem, long long m0, long long M,
                                           int tid) {
    asm volatile("" : "+v"(tid));
#if defined(SAVE_ABL_NOSTORE)
    return;
#endif
    const unsigned char* img = buf_base(smem, id);
    const int pb = buf_pitch_bytes(id);
    // (tried: eight LDS reads in flight, then their stores -- slower inside the stack, 600 -> 830 us for the 3D critic at 3B
    // rows: the 40 extra registers spill.  Inside a run the layers save their INPUT from the fragments they read anyway, see
    // stack_layer; this copy is for the images no stack layer consumes.)
    for (Sweep sw(tid, cols >> 3); sw.row < MLP_BM; sw.next())
        if (m0 + sw.row < M)
            *reinterpret_cast<uint4*>(g + (m0 + sw.row) * ld + sw.c * 8) = *reinterpret_cast<const uint4*>(img + chunk_off(sw.row, sw.c, pb));
}

// A run of n full-width layers, optionally fed by one narrow layer (lead_ks = 4 or 8 k-steps, 0: none) and optionally
// followed by the network's output layer (`tail`).
// ALT: the layers alternate (no residual, residual) -- the myResNet blocks -- and n is even
// tail: 0 none, 1 the network's fp32 output layer (tail_layer), 2 a bf16 layer of <= 128 features (tail_gemm)
// SAVE (forward-with-save): every run layer is followed by a barrier and the copy of its image to global memory
template <bool LEAKY, bool ALT, bool SAVE>
__device__ __forceinline__ void gemm_stack(UnitPtr lead, int lead_ks, UnitPtr first, int n, int tail, unsigned char* smem,
                                           int wave, int lane, long long m0, long long M, int tid) {
    // forward-with-save: layer j of the run writes the image of its producer `su` (the unit in front of it; none for the
    // first layer behind a LOAD) to that unit's save target -- a buffer resource over the tile's rows; zero records: nothing.
    // (The unit is read from the kernarg segment where it is needed, not carried in the layer descriptors.)
    (void)tid;
    auto save_of = [&](UnitPtr su, bool on, int& ld_bytes) -> __amdgpu_buffer_rsrc_t {
        uint16_t* base = (SAVE && on) ? su->save : nullptr;
        long long ld = (SAVE && on) ? su->save_ld : 0;
        // rows [0, save_rows) only (the interpolated rows of a step's 3B-row batch leave nothing but their sign bits): a tile
        // beyond them gets ld = 0 -- stack_layer then issues no store instructions at all (a nullified store still costs its
        // issue on the vector-memory path: 426 us with every store dropped by the range check against 333 without the
        // instructions, 3D critic at 3B rows)
        const long long lim = (SAVE && on) ? su->save_rows : 0;
        long long rows = lim - m0 < MLP_BM ? lim - m0 : MLP_BM;
        if (rows <= 0) { rows = 0; ld = 0; base = nullptr; }
        ld_bytes = (int)(ld * 2);
#if defined(SAVE_ABL_NOSTORE) || defined(SAVE_ABL_NULLSTORES)          /* timing only: every store of the run falls outside its resource */
        base = nullptr;
#endif
        return __builtin_amdgcn_make_buffer_rsrc(base != nullptr ? base + m0 * ld : reinterpret_cast<uint16_t*>(smem), 0,
                                                 base != nullptr ? (int)(rows * ld * 2) : 0, 0x27000);
    };
    // the sign-bit array of a run layer's OWN output: 4 KB per 128-row tile (zero records: nothing asked for)
    auto bits_of = [&](UnitPtr bu) -> __amdgpu_buffer_rsrc_t {
        uint32_t* base = SAVE ? bu->bits : nullptr;
        return __builtin_amdgcn_make_buffer_rsrc(base != nullptr ? base + (m0 / MLP_BM) * 1024 : reinterpret_cast<uint32_t*>(smem), 0,
                                                 base != nullptr ? 4096 : 0, 0x27000);
    };
    // units first[0 .. nt): the n full-width layers and, if `tail`, the output layer behind them
    const int nt = n + (tail ? 1 : 0);
    StackDesc cur = stack_desc(first), nxt = stack_desc(first + (nt > 1 ? 1 : 0));
    WHalf loA, loB, hi;
    f32x16 seed[MLP_NS];
    // identity fragments: A[n][k'] = (n == 16 ks2 + k') for the lane's k' = 8h + j
    bf16x8 idf[2];
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
        const int dd = (lane & 31) - 16 * ks2 - 8 * (lane >> 5);
        u32x4 v;
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) v[p2] = (dd == 2 * p2 ? 0x3F80u : 0u) | (dd == 2 * p2 + 1 ? 0x3F800000u : 0u);
        idf[ks2] = __builtin_bit_cast(bf16x8, v);
    }
    int nold = 0;
    const __amdgpu_buffer_rsrc_t nosv = save_of(first, false, nold);
    if (lead_ks != 0) {
        const StackDesc ld = stack_desc(lead);
        load_seed(ld.bias, wave, 4, lane, seed);
        if (lead_ks == 8) {
            const WBase<8> b(ld.w, wave, 4, lane);
#pragma unroll
            for (int t = 0; t < MLP_NS; ++t)
#pragma unroll
                for (int k = 0; k < 8; ++k) loB[t][k] = b.frag(t, k);
            stack_layer<LEAKY, 0, 8, false, SAVE>(ld, cur, smem, wave, lane, loB, loA, hi, seed, idf, MLP_MAX_UNITS + 50, nosv, 0, bits_of(lead));
        } else {
            const WBase<4> b(ld.w, wave, 4, lane);
#pragma unroll
            for (int t = 0; t < MLP_NS; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) loB[t][k] = b.frag(t, k);
            stack_layer<LEAKY, 0, 4, false, SAVE>(ld, cur, smem, wave, lane, loB, loA, hi, seed, idf, MLP_MAX_UNITS + 50, nosv, 0, bits_of(lead));
        }
        lds_barrier();
        DHAUG_LSTAMP(MLP_MAX_UNITS + 56)
    } else {
        load_seed(cur.bias, wave, 4, lane, seed);
        const WBase<> b0(cur.w, wave, 4, lane);
#pragma unroll
        for (int t = 0; t < MLP_NS; ++t)
#pragma unroll
            for (int k = 0; k < MLP_MAX_KSTEPS / 2; ++k) {
                loA[t][k] = b0.frag(t, k);
                hi[t][k] = b0.frag(t, k + 8);
            }
    }
    // two layers per trip: the low weight sets swap roles without moving
    int l = 0;
#pragma unroll 1
    for (; l + 2 <= n; l += 2) {
        const StackDesc n2 = stack_desc(first + (l + 2 < nt ? l + 2 : nt - 1));         // arrive during the layers
        const StackDesc n3 = stack_desc(first + (l + 3 < nt ? l + 3 : nt - 1));
        const int dbg = MLP_MAX_UNITS + 2 + 8 * (l < 4 ? l : 4);
        int ldb0, ldb1;
        const __amdgpu_buffer_rsrc_t sv0 = save_of(l == 0 ? lead : first + (l - 1), l != 0 || lead_ks != 0, ldb0);
        const __amdgpu_buffer_rsrc_t sv1 = save_of(first + l, true, ldb1);
        stack_layer<LEAKY, ALT ? 0 : 2, MLP_MAX_KSTEPS, SAVE>(cur, nxt, smem, wave, lane, loA, loB, hi, seed, idf, dbg, sv0, ldb0,
                                                              bits_of(first + l));
        lds_barrier();
        DHAUG_LSTAMP(dbg + 6)
        stack_layer<LEAKY, ALT ? 1 : 2, MLP_MAX_KSTEPS, SAVE>(nxt, n2, smem, wave, lane, loB, loA, hi, seed, idf, dbg + 8, sv1, ldb1,
                                                              bits_of(first + l + 1));
        cur = n2;
        nxt = n3;
        if (l + 2 < nt) lds_barrier();
        DHAUG_LSTAMP(dbg + 14)
    }
    const bool odd = !ALT && l < n;
    if (odd) {
        int ldb;
        const __amdgpu_buffer_rsrc_t svo = save_of(l == 0 ? lead : first + (l - 1), l != 0 || lead_ks != 0, ldb);
        stack_layer<LEAKY, 2, MLP_MAX_KSTEPS, SAVE>(cur, nxt, smem, wave, lane, loA, loB, hi, seed, idf, MLP_MAX_UNITS + 2, svo, ldb,
                                                    bits_of(first + l));
        cur = nxt;
        if (tail) lds_barrier();
    }
    if (tail) {                                                              // one call site: the low set is moved, not re-instantiated
        if (odd) {
#pragma unroll
            for (int t = 0; t < MLP_NS; ++t)
#pragma unroll
                for (int k = 0; k < MLP_MAX_KSTEPS / 2; ++k) loA[t][k] = loB[t][k];
        }
        if (tail == 1) tail_layer(cur, smem, wave, lane, loA, hi, seed);
        else tail_gemm(cur, smem, wave, lane, loA, hi, seed);
    }
}

#ifdef MOVE_BATCH_OVERRIDE
constexpr int MOVE_BATCH = MOVE_BATCH_OVERRIDE;
#else
constexpr int MOVE_BATCH = 16;
#endif                                                // global accesses in flight per thread

// one pass of a LOAD: NB global accesses in flight per thread, then their conversions / LDS writes
template <int NB>
__device__ __forceinline__ void load_f32_pass(const float* g, long long ld, int cols, unsigned char* dst, int pb, long long m0,
                                              long long M, Sweep& sw) {
    // Every access is issued, at a clamped address (the tile's last row, the row's last group), and what lies outside becomes
    // zero afterwards: as `if (inside) v[i] = load` hipcc gave each access a branch of its own -- in the linear form below with a
    // wait right behind the access and ~50 register moves per slot (the whole array copied between two homes): 3 600 clocks for
    // six LDS reads per thread, and for global reads one memory round trip after the other.
    f32x4 v[NB];
    int off[NB];
    bool in[NB];
    const long long lastrow = M - 1;
    const int lastc = (cols >> 2) - 1;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const long long gm = m0 + sw.row;
        off[i] = sw.row < MLP_BM ? chunk_off(sw.row, sw.c >> 1, pb) + ((sw.c & 1) << 3) : -1;
        in[i] = sw.row < MLP_BM && gm < M && sw.c * 4 < cols;
        v[i] = *reinterpret_cast<const f32x4*>(g + (gm < M ? gm : lastrow) * ld + (sw.c < lastc ? sw.c : lastc) * 4);
        sw.next();
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        uint2 o;
        o.x = in[i] ? pack_bf16x2(v[i][0], v[i][1]) : 0u;
        o.y = in[i] ? pack_bf16x2(v[i][2], v[i][3]) : 0u;
        if (off[i] >= 0) *reinterpret_cast<uint2*>(dst + off[i]) = o;
    }
}
template <int NB>
__device__ __forceinline__ void load_bf16_pass(const uint16_t* g, long long ld, int cols, unsigned char* img, int pb, long long m0,
                                               long long M, Sweep& sw) {
    u32x4_t v[NB];                                                           // (clamped addresses, zeros afterwards: see load_f32_pass)
    int off[NB];
    const long long lastrow = M - 1;
    const int lastc = (cols >> 3) - 1;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const long long gm = m0 + sw.row;
        off[i] = sw.row < MLP_BM ? chunk_off(sw.row, sw.c, pb) : -1;
        const bool in = sw.row < MLP_BM && gm < M && sw.c * 8 < cols;
        v[i] = *reinterpret_cast<const u32x4_t*>(g + (gm < M ? gm : lastrow) * ld + (sw.c < lastc ? sw.c : lastc) * 8);
        if (!in) v[i] = u32x4_t{0u, 0u, 0u, 0u};
        sw.next();
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (off[i] >= 0) *reinterpret_cast<u32x4_t*>(img + off[i]) = v[i];
}

// fp32 LOAD when the row's 16-byte groups (Q4 = 16 or 32, zero fill included) divide the workgroup: a thread keeps its
// column group and walks rows with a constant stride, so the Q4 / 2 accesses of the tile need no per-slot index arithmetic
// (the generic pass spends ~25 VALU per slot on the sweep, the 64-bit address and the swizzle)
template <int Q4>
__device__ __forceinline__ void load_f32_fast(const float* g, long long ld, int cols, unsigned char* dst, int pb, long long m0,
                                              long long M, int tid) {
    constexpr int RS = MLP_THREADS / Q4, NB = MLP_BM / RS;            // rows per step, steps
    const int r0 = tid / Q4, c = tid % Q4;
    const bool col_live = c * 4 < cols;
    const float* p = g + (m0 + r0) * ld + (col_live ? c * 4 : 0);
    const long long stride = (long long)RS * ld;
    f32x4 v[NB];
    if (m0 + MLP_BM <= M) {                                                  // (workgroup-uniform) a whole tile: no access has a branch of its own
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + i * stride);
    } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m0 + r0 + i * RS < M) v[i] = *reinterpret_cast<const f32x4*>(p + i * stride);
        }
    }
    if (!col_live) {
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // LDS: row * pitch + (((c >> 1) ^ (row & 15)) << 4) + (c & 1) * 8; row & 15 is r0 & 15 (RS = 16) or alternates with ^ 8 (RS = 8)
    const int half = (c & 1) << 3, cc = c >> 1;
    const int o0 = r0 * pb + ((cc ^ (r0 & 15)) << 4) + half, o1 = r0 * pb + ((cc ^ ((r0 + RS) & 15)) << 4) + half;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        uint2 o;
        o.x = pack_bf16x2(v[i][0], v[i][1]);
        o.y = pack_bf16x2(v[i][2], v[i][3]);
        *reinterpret_cast<uint2*>(dst + ((RS == 16 || !(i & 1)) ? o0 : o1) + i * RS * pb) = o;
    }
}

// fp32 LOAD of a tile whose rows are contiguous in memory (ld == cols): the tile is one linear array of 16-byte groups,
// thread i takes groups i, i + 256, ... -- every lane live, every request a full contiguous 4 KB per wave -- and the
// (row, group) of each is recovered for the LDS image; the zero fill up to the next multiple of 64 columns is separate.
template <int NB>
__device__ __forceinline__ void load_f32_linear(const float* g, int cols, unsigned char* dst, int pb, long long m0, long long M,
                                                int tid) {
    const int gpr = cols >> 2, total = MLP_BM * gpr;                         // groups per row, per tile
    const long long live = (M - m0 < MLP_BM ? M - m0 : (long long)MLP_BM) * gpr;
    const float* p = g + m0 * cols;
    const float rc = __builtin_amdgcn_rcpf((float)gpr);                      // idx < 4096: floor((idx + 0.5) / gpr) is exact in fp32
    f32x4 v[NB];                                                             // (clamped addresses, zeros afterwards: see load_f32_pass)
    const int lastg = (int)live - 1;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + MLP_THREADS * i;
        v[i] = *reinterpret_cast<const f32x4*>(p + 4 * (idx < lastg ? idx : lastg));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + MLP_THREADS * i, row = (int)(((float)idx + 0.5f) * rc), c = idx - row * gpr;
        uint2 o;
        o.x = idx < live ? pack_bf16x2(v[i][0], v[i][1]) : 0u;
        o.y = idx < live ? pack_bf16x2(v[i][2], v[i][3]) : 0u;
        if (idx < total) *reinterpret_cast<uint2*>(dst + chunk_off(row, c >> 1, pb) + ((c & 1) << 3)) = o;
    }
    const int c0 = cols >> 2, np = (((cols + 63) & ~63) >> 2) - c0;          // zero fill, 8-byte groups
    const float rn = __builtin_amdgcn_rcpf((float)(np > 0 ? np : 1));
    for (int j = tid; j < MLP_BM * np; j += MLP_THREADS) {
        const int row = (int)(((float)j + 0.5f) * rn), c = c0 + j - row * np;
        *reinterpret_cast<uint2*>(dst + chunk_off(row, c >> 1, pb) + ((c & 1) << 3)) = make_uint2(0u, 0u);
    }
}

// LOADs of tiles whose rows are contiguous in memory with a row length known at compile time -- the critics' inputs: 48-column
// poses, 32-column projections (fp32, G groups of four per row), the 32-column KCS operand (bf16, Q chunks of eight per row).
// One wave per SIMD issues a vector instruction every four clocks at best and has nobody to hide behind: the generic passes' ~90
// instructions per slot (sweep, 64-bit addresses, float reciprocal, swizzle) were 2 000 - 3 000 clocks of a LOAD beside one
// memory round trip.  Here: clamped 32-bit offsets, division by a constant, all requests first, zeros by select.
template <int G>
__device__ __forceinline__ void load_f32_rows(const float* g, unsigned char* dst, int pb, long long m0, long long M, int tid) {
    constexpr int N = MLP_BM * G / MLP_THREADS, Z = (16 - G % 16) % 16;      // groups per thread; zero groups per row (to 64 columns)
    const long long left = M - m0;
    const int live = (left < MLP_BM ? (int)left : MLP_BM) * G, lastg = live - 1;
    const f32x4* p = reinterpret_cast<const f32x4*>(g + m0 * (4 * G));
    f32x4 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int idx = tid + MLP_THREADS * i;
        v[i] = p[idx < lastg ? idx : lastg];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int idx = tid + MLP_THREADS * i, row = idx / G, c = idx - row * G;
        uint2 o;
        o.x = idx < live ? pack_bf16x2(v[i][0], v[i][1]) : 0u;
        o.y = idx < live ? pack_bf16x2(v[i][2], v[i][3]) : 0u;
        *reinterpret_cast<uint2*>(dst + chunk_off(row, c >> 1, pb) + ((c & 1) << 3)) = o;
    }
#pragma unroll
    for (int i = 0; i < MLP_BM * Z / MLP_THREADS; ++i) {
        const int k = tid + MLP_THREADS * i, row = k / (Z > 0 ? Z : 1), c = G + k - row * Z;
        *reinterpret_cast<uint2*>(dst + chunk_off(row, c >> 1, pb) + ((c & 1) << 3)) = make_uint2(0u, 0u);
    }
}
template <int Q>
__device__ __forceinline__ void load_bf16_rows(const uint16_t* g, unsigned char* img, int pb, long long m0, long long M, int tid) {
    constexpr int N = MLP_BM * Q / MLP_THREADS, Z = 8 - Q;                   // chunks per thread; zero chunks per row (64 columns)
    const long long left = M - m0;
    const int live = (left < MLP_BM ? (int)left : MLP_BM) * Q, lastc = live - 1;
    const u32x4_t* p = reinterpret_cast<const u32x4_t*>(g + m0 * (8 * Q));
    u32x4_t v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int idx = tid + MLP_THREADS * i;
        v[i] = p[idx < lastc ? idx : lastc];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int idx = tid + MLP_THREADS * i, row = idx / Q, c = idx - row * Q;
        if (idx >= live) v[i] = u32x4_t{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4_t*>(img + chunk_off(row, c, pb)) = v[i];
    }
#pragma unroll
    for (int i = 0; i < MLP_BM * Z / MLP_THREADS; ++i) {
        const int k = tid + MLP_THREADS * i, row = k / (Z > 0 ? Z : 1), c = Q + k - row * Z;
        *reinterpret_cast<u32x4_t*>(img + chunk_off(row, c, pb)) = u32x4_t{0u, 0u, 0u, 0u};
    }
}

// The bf16 pass size follows the tile: q 16-byte groups per row -> q / 2 per thread (4 for the 32-column KCS operand: a
// fixed 16 made it walk 12 dead slots per thread, 4 400 -> 2 650 clocks).
__device__ __forceinline__ void move_unit(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    asm volatile("" : "+v"(tid));                                            // opaque (see gemm_layer): nothing derived from it is hoisted
    const int kind = u->kind, cols = u->cols;
    const long long ld = u->ld;
    if (kind == U_LOAD_F32) {
        const float* g = static_cast<const float*>(u->g);
        unsigned char* dst = buf_base(smem, u->dst);
        const int pb = buf_pitch_bytes(u->dst), q4 = ((cols + 63) & ~63) >> 2;
        // contiguous rows of a known length: the lean passes (3D critic: pose LOAD 4 700 -> 2 400 clocks, KCS operand 3 300 -> 1 200;
        // 112.1 -> 109.0 us at 65 536 poses with the branch-free generic passes).  The generator's 128-column noise tile is 64 KB per
        // workgroup, all workgroups at once -- 5 500 - 7 500 clocks whatever the pass: that one is the card's bandwidth.
#ifndef LOAD_ABL_NOROWS32
        if (ld == cols && cols == 128) { load_f32_rows<32>(g, dst, pb, m0, M, tid); return; }
#endif
        if (q4 == 32) { load_f32_fast<32>(g, ld, cols, dst, pb, m0, M, tid); return; }
        if (ld == cols && cols == 48) { load_f32_rows<12>(g, dst, pb, m0, M, tid); return; }
        if (ld == cols && cols == 32) { load_f32_rows<8>(g, dst, pb, m0, M, tid); return; }
        if (ld == cols && cols <= 32) { load_f32_linear<4>(g, cols, dst, pb, m0, M, tid); return; }
        Sweep sw(tid, q4);
        while (sw.row < MLP_BM) load_f32_pass<MOVE_BATCH>(g, ld, cols, dst, pb, m0, M, sw);
        return;
    }
    uint16_t* g = static_cast<uint16_t*>(const_cast<void*>(u->g));
    const int id = kind == U_LOAD_BF16 ? u->dst : u->src;
    unsigned char* img = buf_base(smem, id);
    const int pb = buf_pitch_bytes(id);
    const int q8 = (kind == U_LOAD_BF16 ? ((cols + 63) & ~63) : cols) >> 3;
    if (kind == U_LOAD_BF16 && ld == cols && cols == 32) { load_bf16_rows<4>(g, img, pb, m0, M, tid); return; }
    Sweep sw(tid, q8);
    if (kind == U_LOAD_BF16) {
        while (sw.row < MLP_BM) {
            if (q8 <= 8) load_bf16_pass<4>(g, ld, cols, img, pb, m0, M, sw);
            else load_bf16_pass<MOVE_BATCH>(g, ld, cols, img, pb, m0, M, sw);
        }
    } else {
        for (; sw.row < MLP_BM; sw.next())
            if (m0 + sw.row < M)
                *reinterpret_cast<uint4*>(g + (m0 + sw.row) * ld + sw.c * 8) = *reinterpret_cast<const uint4*>(img + chunk_off(sw.row, sw.c, pb));
    }
}

// logits of a layer flagged DOT_OUT: sum of the 2 x (waves that own a slice) partial sums + the bias
__device__ __forceinline__ void dot_output(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    if (tid >= MLP_BM || m0 + tid >= M) return;
    const float* st = reinterpret_cast<const float*>(buf_base(smem, u->dst));
    const int nw = min(4, (u->N + 31) >> 5);
    float s = 0.f;
    for (int i = 0; i < 2 * nw; ++i) s += st[i * MLP_BM + tid];
    static_cast<float*>(const_cast<void*>(u->g))[(m0 + tid) * u->ld] = s;
}

__device__ __forceinline__ void store_output(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    const float* st = reinterpret_cast<const float*>(buf_base(smem, u->dst));
    float* out = static_cast<float*>(const_cast<void*>(u->g));
    const long long ld = u->ld;
    for (Sweep sw(tid, u->N); sw.row < MLP_BM; sw.next())
        if (m0 + sw.row < M) out[(m0 + sw.row) * ld + sw.c] = st[sw.row * OUT_PITCH + sw.c];
}

template <bool SAVE>
__global__ __launch_bounds__(MLP_THREADS, 1) void fused_mlp_kernel(Program prog, long long M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ntiles = (M + MLP_BM - 1) / MLP_BM;
    // the program lives in the kernarg segment: index it there with scalar loads (a by-value struct indexed
    // dynamically would be copied to scratch)
    const unsigned char __attribute__((address_space(4)))* ka =
        (const unsigned char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    UnitPtr units = (UnitPtr)(ka + __builtin_offsetof(Program, u));
    const int nunits = *(const int __attribute__((address_space(4)))*)(ka + __builtin_offsetof(Program, nunits));
    int hk = units->kind, hp = units->plan;                                   // header of the next unit to run
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m0 = tile * MLP_BM;
        int i = 0;
#pragma unroll 1
        while (i < nunits) {
            UnitPtr u = units + i;
            const int kind = hk, plan = hp;
            // where the program continues, and that unit's header: requested now, used after this unit
            int ni = i + 1;
            if (kind == U_GEMM && (plan & PLAN_STACK)) ni = i + ((plan & 15) != 0) + ((plan >> 4) & 63) + ((plan & (PLAN_TAIL | PLAN_TAIL_BF16)) ? 1 : 0);
            if (kind == U_GEMM && (plan & PLAN_PAIR)) ni = i + 2;
            {
                UnitPtr nu = units + (ni < nunits ? ni : 0);
                hk = nu->kind;
                hp = nu->plan;
            }
            DHAUG_STAMP(i)
            DHAUG_LSTAMP(MLP_MAX_UNITS + 64 + 4 * i)
            const int ui = i;
            i = ni;
            (void)ui;
            if (kind != U_GEMM) {
                // a STORE is not waited for where it is issued (lds_barrier orders only its LDS reads); the LOAD that
                // reads the parked rows back drains the workgroup's stores first
                if (kind == U_LOAD_BF16 && (plan & 1)) __syncthreads();      // only in programs that park rows with a STORE
                move_unit(u, smem, m0, M, tid);
                if (kind == U_STORE_BF16) {
                    lds_barrier();
                    continue;
                }
            } else {
                if (plan & PLAN_STACK) {
                    // a run of consecutive plain 256 -> 256 layers, possibly fed by this (narrow) layer and followed by
                    // the output layer
                    const int lead_ks = plan & 15, run = (plan >> 4) & 63;
                    const int tail = (plan & PLAN_TAIL) ? 1 : ((plan & PLAN_TAIL_BF16) ? 2 : 0);
                    UnitPtr f0 = u + (lead_ks != 0), tu = f0 + run;
                    if ((plan & PLAN_LEAKY) || NAN_SAFE_TU) gemm_stack<true, false, SAVE>(u, lead_ks, f0, run, tail, smem, wave, lane, m0, M, tid);
                    else if (plan & PLAN_ALT) gemm_stack<false, true, SAVE>(u, lead_ks, f0, run, tail, smem, wave, lane, m0, M, tid);
                    else gemm_stack<false, false, SAVE>(u, lead_ks, f0, run, tail, smem, wave, lane, m0, M, tid);
                    lds_barrier();
                    if (SAVE) {                                              // (inside the run every layer saved its input)
                        UnitPtr lu = tu - 1;                                 // the run's last layer: nobody in the run read its image
                        if (lu->save != nullptr) save_image(lu->dst, lu->save, lu->save_ld, (lu->N + 15) & ~15, smem, m0, lu->save_rows, tid);
                        if (tail == 2 && tu->save != nullptr)
                            save_image(tu->dst, tu->save, tu->save_ld, (tu->N + 15) & ~15, smem, m0, tu->save_rows, tid);
                    }
                    if (tail == 1) {
                        store_output(tu, smem, m0, M, tid);
                        lds_barrier();                       // (not __syncthreads: nobody waits for the stores to be acknowledged)
                    }

                    continue;                                                // (stamps of the run's inner layers stay 0)
                }
                const int nslices = (plan >> 24) & 15, dbg = MLP_MAX_UNITS + 64 + 4 * ui;
                if (plan & PLAN_PAIR) {                                      // (both layers: <= 4 slices, one per wave)
                    gemm_pair(u, u + 1, smem, wave, lane, dbg);
                    if ((u + 1)->flags & F_DOT_OUT) {
                        lds_barrier();
                        dot_output(u + 1, smem, m0, M, tid);
                    }
                    lds_barrier();
                    DHAUG_LSTAMP(dbg + 7)
                    continue;
                }
#define DHAUG_SHAPES(NS)                                                                   \
    switch ((plan >> 16) & 255) {                              /* validated on the host */ \
        case 1 * 16 + 1: gemm_single<1, 1, NS>(u, smem, wave, lane, dbg); break;                 \
        case 2 * 16 + 2: gemm_single<2, 2, NS>(u, smem, wave, lane, dbg); break;                 \
        case 2 * 16 + 1: gemm_single<2, 1, NS>(u, smem, wave, lane, dbg); break;                 \
        case 4 * 16 + 4: gemm_single<4, 4, NS>(u, smem, wave, lane, dbg); break;                 \
        case 4 * 16 + 2: gemm_single<4, 2, NS>(u, smem, wave, lane, dbg); break;                 \
        case 8 * 16 + 4: gemm_single<8, 4, NS>(u, smem, wave, lane, dbg); break;                 \
        default: break;                                                                    \
    }
                if (wave + 4 < nslices) { DHAUG_SHAPES(2) }
                else if (wave < nslices) { DHAUG_SHAPES(1) }
#undef DHAUG_SHAPES
                if (plan & PLAN_OUT) {
                    lds_barrier();
                    store_output(u, smem, m0, M, tid);
                }
                if (plan & PLAN_DOT) {
                    lds_barrier();
                    dot_output(u, smem, m0, M, tid);
                }
            }
            lds_barrier();
            if (SAVE && kind == U_GEMM && !(plan & (PLAN_OUT | PLAN_DOT)) && u->save != nullptr)
                save_image(u->dst, u->save, u->save_ld, (u->N + 15) & ~15, smem, m0, u->save_rows, tid);
            DHAUG_LSTAMP(MLP_MAX_UNITS + 64 + 4 * ui + 3)
        }
        DHAUG_STAMP(nunits)
    }
    (void)prog;
}

#ifdef DHAUG_MLP_SAVE_TU
}  // namespace

// This translation unit (csrc/dhaug_mlp_save.hip) holds the forward-with-save instantiation only: the inference
// instantiation's code (register allocation, schedule) does not depend on it.  (Build note: while the layer descriptors
// carried the save pointers the instantiation spilled, and hipcc 7.2's "AMDGPU Rewrite AGPR-Copy-MFMA" pass crashes on
// spilled MFMA code under -amdgpu-mfma-vgpr-form; reading them from the kernarg segment at the save removed both.)
extern "C" __attribute__((visibility("hidden"))) int dhaug_mlp_launch_save_(const void* prog, long long M, unsigned grid, void* stream) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL(fused_mlp_kernel<true>, dim3(grid), dim3(MLP_THREADS), MLP_LDS_BYTES, (hipStream_t)stream,
                       *static_cast<const Program*>(prog), M);
    return dhaug_launch_status();
}
#else
// weights -> fragment order.  dst[((slice*ksteps + ks)*64 + lane)*8 + j] = W[32 slice + (lane&31)][k0 + 16 ks + 8 (lane>>5) + j]
__global__ __launch_bounds__(256) void pack_wfrag_kernel(const float* __restrict__ W, long long ldw, uint16_t* __restrict__ dst,
                                                         int N, int K, int k0, int ksteps, int nslices) {
    const long long total = (long long)nslices * ksteps * 64 * 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long long blk = i >> 9;
        const int ks = (int)(blk % ksteps), s = (int)(blk / ksteps);
        const int n = 32 * s + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
        dst[i] = (n < N && k < K) ? dhaug_f32_to_bf16(W[(long long)n * ldw + k0 + k]) : (uint16_t)0;
    }
}

// the same for every layer of a network in ONE launch (after an optimizer step): blockIdx.y = descriptor.  Also refreshes the
// zero-padded fp32 bias [256] and, for a logit layer folded into its producer, the DOT_OUT vector [257].
__global__ __launch_bounds__(256) void pack_wfrag_batch_kernel(const dhaug_wfrag_desc* __restrict__ descs) {
    const dhaug_wfrag_desc d = descs[blockIdx.y];
    const long long total = (long long)8 * d.ksteps * 64 * 8;
    uint16_t* dst = static_cast<uint16_t*>(d.dst);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long long blk = i >> 9;
        const int ks = (int)(blk % d.ksteps), sl = (int)(blk / d.ksteps);
        const int n = 32 * sl + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
        dst[i] = (n < d.N && k < d.K) ? dhaug_f32_to_bf16(d.W[(long long)n * d.ldw + d.k0 + k]) : (uint16_t)0;
    }
    if (blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (d.bias_dst != nullptr) d.bias_dst[t] = (d.bias != nullptr && t < d.N) ? d.bias[t] : 0.0f;
        if (d.dot_dst != nullptr) {                                          // (N == 1: the layer's weights as bf16 values, bias at [256])
            d.dot_dst[t] = t < d.K ? dhaug_bf16_to_f32(dhaug_f32_to_bf16(d.W[d.k0 + t])) : 0.0f;
            if (t == 0) d.dot_dst[256] = d.bias != nullptr ? d.bias[0] : 0.0f;
        }
    }
}

// how unit i is executed (see the dispatch in fused_mlp_kernel): a run of >= min_run full-width 256 -> 256 layers goes to
// gemm_stack, together with the narrow layer (K <= 128) feeding it and the <= 64-wide fp32 output layer behind it
int plan_unit(const Program& p, int i) {
    const Unit* U = p.u;
    const Unit& u = U[i];
    if (u.kind != U_GEMM) {
        // a bf16 LOAD may read back rows a STORE of this program parked in global memory: it then drains the workgroup's
        // stores first (full barrier); programs without a STORE skip that wait for write acknowledgements
        int parks = 0;
        for (int j = 0; j < p.nunits; ++j) parks |= U[j].kind == U_STORE_BF16;
        return u.kind == U_LOAD_BF16 ? parks : 0;
    }
    const bool wide = u.ksteps2 == 0 && u.N > 224 && !(u.flags & (F_OUT_F32 | F_DOT_OUT)) && u.src < 2 && u.dst < 2;
    const int ks4 = (u.ksteps + 3) & ~3;                                 // the fragment blob and the LOAD pad to whole chunks
    const int lead_ks = (wide && u.res < 0 && ks4 <= 8) ? ks4 : 0;
    const int i0 = i + (lead_ks != 0);
    int run = 0, leaky = lead_ks != 0 && u.act == DHAUG_ACT_LRELU, alt = 1;
    while (i0 + run < p.nunits && U[i0 + run].kind == U_GEMM && U[i0 + run].ksteps == 16 && U[i0 + run].ksteps2 == 0 &&
           U[i0 + run].N > 224 && !(U[i0 + run].flags & (F_OUT_F32 | F_DOT_OUT)) && U[i0 + run].src < 2 && U[i0 + run].dst < 2 &&
           U[i0 + run].res < 2) {
        leaky |= U[i0 + run].act == DHAUG_ACT_LRELU;
        alt &= (U[i0 + run].res >= 0) == ((run & 1) == 1);
        ++run;
    }
    if (run < p.min_run || run > 63) {
        const int c1 = (u.ksteps + 3) / 4, c2 = (u.ksteps2 + 3) / 4;
        // two narrow layers in a row (two chunks of k, at most 128 features, every wave owning a slice of both, nothing saved, the
        // first one an ordinary layer; the unit before must not have claimed this one as ITS second half) run as one unit
        auto narrow = [&](const Unit& v) {
            return v.kind == U_GEMM && v.ksteps2 == 0 && v.ksteps > 4 && v.ksteps <= 8 && v.N > 96 && v.N <= 128 && !(v.flags & F_OUT_F32) &&
                   v.save == nullptr;
        };
        const bool second = i > 0 && (U[i - 1].plan & PLAN_PAIR) && U[i - 1].kind == U_GEMM && !(U[i - 1].plan & PLAN_STACK);
        const bool pair = !second && i + 1 < p.nunits && narrow(u) && !(u.flags & F_DOT_OUT) && narrow(U[i + 1]) &&
                          !getenv("DHAUG_MLP_NOPAIR");
        return (((c1 + c2) * 16 + c1) << 16) | (((u.N + 31) >> 5) << 24) | ((u.flags & F_OUT_F32) ? PLAN_OUT : 0) |
               ((u.flags & F_DOT_OUT) ? PLAN_DOT : 0) | (pair ? PLAN_PAIR : 0);
    }
    const Unit* tu = U + i0 + run;
    const bool tail = i0 + run < p.nunits && tu->kind == U_GEMM && (tu->flags & F_OUT_F32) && tu->ksteps == 16 &&
                      tu->ksteps2 == 0 && tu->N <= 64 && tu->src < 2 && tu->dst < 2 && tu->res < 0;
    // a bf16 layer of at most 128 features with K = 256 behind the run starts with its weights in registers (tail_gemm)
    const bool tail2 = !tail && i0 + run < p.nunits && tu->kind == U_GEMM && !(tu->flags & (F_OUT_F32 | F_DOT_OUT)) &&
                       tu->ksteps == 16 && tu->ksteps2 == 0 && tu->N <= 128 && tu->src < 2 && tu->res != tu->src &&
                       !getenv("DHAUG_MLP_NOTAIL2");
    return PLAN_STACK | lead_ks | (run << 4) | (leaky ? PLAN_LEAKY : 0) | ((!leaky && alt && !(run & 1)) ? PLAN_ALT : 0) |
           (tail ? PLAN_TAIL : 0) | (tail2 ? PLAN_TAIL_BF16 : 0);
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) int dhaug_mlp_launch_save_(const void* prog, long long M, unsigned grid, void* stream);

extern "C" {

/* see include/dhaug.h */
int dhaug_pack_wfrag(const float* W, int64_t ldw, uint16_t* dst, int64_t N, int64_t K, int64_t k0, void* stream) {
    DHAUG_CHECK(N >= 1 && K >= 1 && k0 >= 0 && ldw >= k0 + K, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(W); DHAUG_CHECK_PTR(dst);
    DHAUG_CHECK(dhaug_aligned16(dst), DHAUG_EALIGN);
    // k-steps padded to whole chunks; always 8 feature slices (zero rows beyond N): every wave computes two slices
    const int ksteps = (int)((K + 63) / 64) * 4, nslices = 8;
    DHAUG_CHECK(ksteps <= MLP_MAX_KSTEPS && N <= 256, DHAUG_EUNSUPPORTED);
    const long long total = (long long)nslices * ksteps * 512;
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_wfrag_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, (long long)ldw, dst,
                       (int)N, (int)K, (int)k0, ksteps, nslices);
    return dhaug_launch_status();
}

int dhaug_pack_wfrag_batch(const dhaug_wfrag_desc* descs_device, int n, void* stream) {
    DHAUG_CHECK(n >= 0, DHAUG_EINVAL);
    if (n == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(descs_device);
    hipLaunchKernelGGL(pack_wfrag_batch_kernel, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream, descs_device);
    return dhaug_launch_status();
}

int dhaug_mlp_forward(const dhaug_mlp_unit* units, int nunits, int64_t M, void* stream) {
    DHAUG_CHECK(nunits >= 1 && nunits <= MLP_MAX_UNITS && M >= 0, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(units);
    if (M == 0) return DHAUG_OK;
    Program prog;
    prog.nunits = nunits;
    prog.min_run = getenv("DHAUG_MLP_NOSTACK") ? 1 << 20 : 2;                // debugging aid: layer-at-a-time path only
    bool any_save = false;
    for (int i = 0; i < nunits; ++i) {
        const dhaug_mlp_unit& s = units[i];
        Unit& u = prog.u[i];
        u.kind = s.kind; u.flags = s.flags; u.src = s.src; u.dst = s.dst; u.res = s.res;
        u.src2 = s.src2; u.ksteps2 = s.ksteps2;
        u.ksteps = s.ksteps; u.N = s.n; u.act = s.act; u.slope = s.slope; u.cols = s.cols; u.ld = s.ld;
        // dhaug_set_nan_propagation(1): ReLU runs as LeakyReLU with slope 0 -- max(v, v * 0) in fp32, the same value for every
        // finite v, NaN for NaN / inf (the forward-with-save unit always does this: NAN_SAFE_TU)
        if (dhaug_nan_propagation_ && s.kind == U_GEMM && s.act == DHAUG_ACT_RELU) { u.act = DHAUG_ACT_LRELU; u.slope = 0.0f; }
        u.g = s.g; u.w = static_cast<const uint16_t*>(s.w); u.w2 = static_cast<const uint16_t*>(s.w2); u.bias = s.bias;
        u.save = static_cast<uint16_t*>(s.save); u.save_ld = s.save_ld;
        u.save_rows = s.save_rows == 0 ? M : (s.save_rows < 0 ? 0 : (s.save_rows < M ? s.save_rows : M));
        u.bits = static_cast<uint32_t*>(s.bits);
        DHAUG_CHECK(u.kind >= U_LOAD_F32 && u.kind <= U_GEMM, DHAUG_EINVAL);
        if (u.bits != nullptr) {                                             // (written by full-width run layers only: checked below)
            DHAUG_CHECK(u.kind == U_GEMM && u.save != nullptr && u.N > 224, DHAUG_EINVAL);
            DHAUG_CHECK(dhaug_aligned16(u.bits), DHAUG_EALIGN);
        }
        if (u.save != nullptr) {
            DHAUG_CHECK(u.kind == U_GEMM && !(u.flags & (F_OUT_F32 | F_DOT_OUT)), DHAUG_EINVAL);
            DHAUG_CHECK(dhaug_aligned16(u.save) && u.save_ld % 8 == 0 && u.save_ld >= ((u.N + 15) & ~15), DHAUG_EALIGN);
            any_save = true;
        }
        auto okbuf = [](int b) { return b >= 0 && b <= 2; };
        auto pitch = [](int b) { return b == 2 ? BUF2_PITCH : BUF01_PITCH; };
        if (u.kind == U_GEMM) {
            DHAUG_CHECK(okbuf(u.src) && u.ksteps >= 1 && u.ksteps <= MLP_MAX_KSTEPS && u.N >= 1 && u.N <= 256, DHAUG_EUNSUPPORTED);
            DHAUG_CHECK(((u.ksteps + 3) / 4) * 64 <= pitch(u.src), DHAUG_EUNSUPPORTED);
            DHAUG_CHECK(u.w != nullptr && dhaug_aligned16(u.w), DHAUG_EALIGN);
            DHAUG_CHECK(u.bias != nullptr && dhaug_aligned16(u.bias), DHAUG_EALIGN);
            DHAUG_CHECK(u.ksteps2 >= 0 && u.ksteps2 <= MLP_MAX_KSTEPS, DHAUG_EUNSUPPORTED);
            if (u.ksteps2 > 0) {
                DHAUG_CHECK(okbuf(u.src2) && ((u.ksteps2 + 3) / 4) * 64 <= pitch(u.src2), DHAUG_EUNSUPPORTED);
                DHAUG_CHECK(u.ksteps % 4 == 0, DHAUG_EUNSUPPORTED);
                DHAUG_CHECK(u.w2 != nullptr && dhaug_aligned16(u.w2), DHAUG_EALIGN);
            }
            {
                const int c1 = (u.ksteps + 3) / 4, c2 = (u.ksteps2 + 3) / 4, sh = (c1 + c2) * 16 + c1;
                const bool ok = sh == 17 || sh == 34 || sh == 33 || sh == 68 || sh == 66 || sh == 132;
                DHAUG_CHECK(ok, DHAUG_EUNSUPPORTED);
            }
            if (u.flags & F_DOT_OUT) {
                DHAUG_CHECK(u.ksteps <= 8 && u.N <= 128, DHAUG_EUNSUPPORTED);     // the shapes that carry this epilogue
                DHAUG_CHECK(!(u.flags & F_OUT_F32) && u.ksteps2 == 0 && u.g != nullptr && u.ld >= 1 && u.w2 != nullptr &&
                            dhaug_aligned16(u.w2) && (u.dst == 0 || u.dst == 1) && u.dst != u.src && u.dst != u.res, DHAUG_EINVAL);
            }
            if (u.flags & F_OUT_F32) {
                DHAUG_CHECK(u.g != nullptr && u.ld >= u.N && u.N <= 64, DHAUG_EUNSUPPORTED);
                DHAUG_CHECK((u.dst == 0 || u.dst == 1) && u.dst != u.src && (u.ksteps2 == 0 || u.dst != u.src2), DHAUG_EINVAL);
            } else {
                DHAUG_CHECK(okbuf(u.dst) && u.dst != u.src && (u.ksteps2 == 0 || u.dst != u.src2), DHAUG_EINVAL);
                DHAUG_CHECK(((u.N + 31) / 32) * 32 <= pitch(u.dst), DHAUG_EUNSUPPORTED);
                DHAUG_CHECK(u.res < 0 || (okbuf(u.res) && u.res != u.src && (u.ksteps2 == 0 || u.res != u.src2)), DHAUG_EINVAL);
            }
        } else {
            const int b = u.kind == U_STORE_BF16 ? u.src : u.dst;
            DHAUG_CHECK(okbuf(b) && u.g != nullptr && u.cols >= 8 && ((u.cols + 63) & ~63) <= pitch(b), DHAUG_EINVAL);
            DHAUG_CHECK(u.cols % 8 == 0 && u.ld >= u.cols && dhaug_aligned16(u.g), DHAUG_EALIGN);
            DHAUG_CHECK(u.kind == U_LOAD_F32 ? (u.ld % 4 == 0) : (u.ld % 8 == 0), DHAUG_EALIGN);
        }
    }
    for (int i = 0; i < prog.nunits; ++i) prog.u[i].plan = plan_unit(prog, i);
    {   // sign bits are written by the layers of a run and by the narrow layer feeding it (not by layer-at-a-time units)
        bool in_run[MLP_MAX_UNITS] = {};
        for (int i = 0; i < prog.nunits;) {
            const int plan = prog.u[i].plan;
            if (prog.u[i].kind == U_GEMM && (plan & PLAN_STACK)) {
                const int lead = (plan & 15) != 0, run = (plan >> 4) & 63;
                for (int j = 0; j < lead + run; ++j) in_run[i + j] = true;
                i += lead + run + ((plan & (PLAN_TAIL | PLAN_TAIL_BF16)) ? 1 : 0);
            } else {
                ++i;
            }
        }
        for (int i = 0; i < prog.nunits; ++i) DHAUG_CHECK(prog.u[i].bits == nullptr || in_run[i], DHAUG_EUNSUPPORTED);
    }
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const long long ntiles = (M + MLP_BM - 1) / MLP_BM;
    const unsigned grid = dhaug_persistent_grid(ntiles);           // one persistent workgroup per CU
    // forward-with-save (a unit asks for its image in global memory) is a second instantiation: the inference kernel's
    // schedule is exactly what it was
    if (any_save) return dhaug_mlp_launch_save_(&prog, (long long)M, grid, stream);        // (csrc/dhaug_mlp_save.hip)
    hipLaunchKernelGGL(fused_mlp_kernel<false>, dim3(grid), dim3(MLP_THREADS), MLP_LDS_BYTES, (hipStream_t)stream, prog, (long long)M);
    return dhaug_launch_status();
}

}  // extern "C"

#endif  // DHAUG_MLP_SAVE_TU

#ifdef DHAUG_MLP_TIMING
extern "C" int dhaug_debug_mlp_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_stamps), sizeof(long long) * (n < 6 * MLP_MAX_UNITS + 69 ? n : 6 * MLP_MAX_UNITS + 69));
}
#endif
