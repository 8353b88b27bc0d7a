// Parity-grade fused forward of the generator trunk / critics: the SAME one-launch unit programs as dhaug_mlp.hip
// (activations never leave LDS), in fp32-grade arithmetic on the matrix cores.
//
// The reference's layers are fp32 nn.Linear (R/models_Fk_GAN/Fk_discriminator.py:180-201,253-266,
// R/models_Fk_GAN/Fk_generator.py:115-119) and the path's tolerance is 1e-4 relative on the logits; one bf16 pass misses it by
// two orders of magnitude.  Here every operand is carried as an fp16 PAIR  x = hi + lo  (hi = fp16(x), lo = fp16(x - hi):
// 22 mantissa bits) and a product is three v_mfma_f32_32x32x16_f16 terms accumulated in fp32:
//        W x  ~=  Whi Xhi + Whi Xlo + Wlo Xhi                       (the dropped Wlo Xlo is 2^-22 relative)
// fp16 products are exact in fp32 (11 x 11 bits), so the result differs from fp32 arithmetic by ~2^-21 per operand:
// measured against the reference's logits 2e-5 in the tests' strict relative metric, 1e-6 of the logit scale.  The F16
// MFMA runs at the BF16 rate, so this mode costs 3 matrix instructions per k-step instead of 1 -- its roofline is a third
// of the dense peak in ALGORITHMIC flops.  Range: |x| < 65 504 (fp16); values below 2^-14 keep an ABSOLUTE error of 2^-25.
//
// Structure (512 threads = 8 waves = TWO per SIMD, 256 registers each; one persistent workgroup per CU walking 64-row
// batch tiles).  Measured on the first version of this kernel (4 waves, one per SIMD; ablation builds, D3 at B = 65 536:
// 459 us as built, 411 without weight reloads, 352 without epilogue, 297 with MFMAs only against 145 us of pure MFMA issue):
// what a single wave per SIMD cannot hide is (a) the epilogue -- the fp32 -> hi/lo split is ~6 VALU per element pair -- and
// (b) the start of every layer, where the first weight fragments and the bias are requested and waited for.  Hence:
//   * wave w owns feature slice w (32 features) for all 64 rows: while one wave of a SIMD runs its epilogue or waits at the
//     layer boundary, the other one issues MFMAs;
//   * the NEXT layer's first two weight chunks and its bias are requested before the current layer's epilogue (the unit
//     program is scanned ahead; across tiles it wraps around to the first layer);
//   * activations: fp16 hi / lo planes [64 rows][256] per buffer (two 64 KB buffers + one 32 KB [64][128] buffer = all
//     160 KB of LDS), 16-byte chunks XOR-swizzled by (row & 15) -> conflict-free ds_read_b128 of MFMA B fragments;
//   * weights: pre-split and pre-packed in A-fragment order (dhaug_pack_wfrag_f16x2): per (32-feature slice, k-step)
//     two contiguous 1 KB blocks (hi, lo), streamed from L2 through a 2-slot register ring (4 k-steps per slot);
//   * MFMA issued swapped (A = weights, B = activations): a lane owns one batch row and 4 consecutive features per
//     register quad; the epilogue (bias = accumulator seed, residual from the LDS planes, activation, hi/lo split) writes
//     8 + 8 bytes per lane into the next layer's operand planes.  In-place residual layers are safe (own elements only).
#include "dhaug_common.h"
#include "dhaug_fk_math.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef X3_PRIO_SEL
#define X3_PRIO_SEL 2
#endif
#ifndef X3_STREAM
#define X3_STREAM 0          // (1: the per-k-step weight stream below -- measured: no gain, see its comment)
#endif
constexpr int X3_BM = 64;                                            // batch rows per tile
constexpr int X3_MT = X3_BM / 32;
constexpr int X3_THREADS = 512;
constexpr int X3_WAVES = X3_THREADS / 64;
constexpr int X3_CH = 4;                                             // k-steps per ring slot (64 k)
constexpr int X3_MAX_UNITS = 32;
constexpr int X3_MAX_KSTEPS = 16;
constexpr int P01 = 256, P2 = 128;                                   // pitch (elements) of buffers 0/1 and 2
constexpr int PLANE01 = X3_BM * P01 * 2, PLANE2 = X3_BM * P2 * 2;    // bytes of one fp16 plane
constexpr int BUF01 = 2 * PLANE01, BUF2 = 2 * PLANE2;                // hi plane, lo plane
constexpr int X3_LDS_BYTES = 2 * BUF01 + BUF2;                       // 163 840
constexpr int OUT_PITCH = 68;                                        // floats per row of the fp32 output staging image

enum { U_LOAD_F32 = 0, U_GEMM = 3, U_LOAD_KCS = 5 };
enum { F_OUT_F32 = 4 };

struct Unit {
    int kind, plan, flags;
    int src, dst, res, src2, ksteps2, ksteps, N, act;
    float slope;
    int cols;
    long long ld;
    const void* g;
    const _Float16* w;
    const _Float16* w2;
    const float* bias;
};
struct Program {
    int nunits, first_gemm;
    Unit u[X3_MAX_UNITS];
};
typedef const Unit __attribute__((address_space(4))) * UnitPtr;      // units are read from the kernarg segment (s_load)

__device__ __forceinline__ unsigned char* buf_base(unsigned char* smem, int id) {
    return smem + (id == 0 ? 0 : (id == 1 ? BUF01 : 2 * BUF01));
}
__device__ __forceinline__ int buf_pitch_bytes(int id) { return (id == 2 ? P2 : P01) * 2; }
__device__ __forceinline__ int buf_plane(int id) { return id == 2 ? PLANE2 : PLANE01; }
__device__ __forceinline__ int chunk_off(int row, int c, int pitch_bytes) { return row * pitch_bytes + ((c ^ (row & 15)) << 4); }
__device__ __forceinline__ float act_neg(int act, float slope) {
    return act == DHAUG_ACT_RELU ? 0.0f : (act == DHAUG_ACT_LRELU ? slope : 1.0f);
}
__device__ __forceinline__ float act_fn(float v, float neg) { return fmaxf(v, v * neg); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// x0, x1 -> packed fp16 pairs (hi, lo) with x = hi + lo to 22 bits
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const f32x2 v = {x0, x1};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(v - hf, f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

__device__ __forceinline__ int chunks_of(int ksteps) { return (ksteps + X3_CH - 1) / X3_CH; }

#ifdef X3_TIMING
__device__ long long g_x3_stamps[4 * X3_MAX_UNITS + 4];
#ifndef X3_STAMP_TID
#define X3_STAMP_TID 0
#endif
#define X3_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == X3_STAMP_TID) g_x3_stamps[i] = (long long)__builtin_readcyclecounter();
#else
#define X3_STAMP(i)
#endif
typedef f16x8 Ring[2][X3_CH][2];                  // [slot][k-step in chunk][piece]: chunk c lives in slot c & 1

// global loads of chunk C (of the concatenated sources; the first has NCH1 chunks) of this wave's slice into a ring slot
template <int C, int NCH, int NCH1>
__device__ __forceinline__ void load_chunk(const _Float16* w1, const _Float16* w2, int wave, int lane, f16x8 (&slot)[X3_CH][2]) {
    constexpr bool second = C >= NCH1;
    constexpr int kpad = (second ? NCH - NCH1 : NCH1) * X3_CH;
    constexpr int k0 = (second ? C - NCH1 : C) * X3_CH;
    const _Float16* base = (second ? w2 : w1) + ((long long)wave * kpad * 2 * 64 + lane) * 8 + (long long)k0 * 2 * 512;
#pragma unroll
    for (int q = 0; q < X3_CH; ++q)
#pragma unroll
        for (int p = 0; p < 2; ++p) slot[q][p] = *reinterpret_cast<const f16x8*>(base + (2 * q + p) * 512);
}

// chunks 0 and 1 and the bias of GEMM unit `u` for this wave's slice (shape known only at run time): requested ahead of the
// layer, i.e. before the previous layer's epilogue and barrier.  Slices beyond N are zero rows of the blob (it always holds
// 8 slices), so every wave may load.
__device__ __forceinline__ void prefetch_layer(UnitPtr u, int wave, int lane, Ring& ring, f32x16& seed) {
    const int kp1 = chunks_of(u->ksteps) * X3_CH, ks2 = u->ksteps2;
    const _Float16* b0 = u->w + ((long long)wave * kp1 * 2 * 64 + lane) * 8;
#pragma unroll
    for (int q = 0; q < X3_CH; ++q)
#pragma unroll
        for (int p = 0; p < 2; ++p) ring[0][q][p] = *reinterpret_cast<const f16x8*>(b0 + (2 * q + p) * 512);
    if (kp1 > X3_CH || ks2 > 0) {                                            // a second chunk exists (wave-uniform)
        const _Float16* b1 = kp1 > X3_CH ? b0 + X3_CH * 2 * 512
                                         : u->w2 + ((long long)wave * (chunks_of(ks2) * X3_CH) * 2 * 64 + lane) * 8;
#pragma unroll
        for (int q = 0; q < X3_CH; ++q)
#pragma unroll
            for (int p = 0; p < 2; ++p) ring[1][q][p] = *reinterpret_cast<const f16x8*>(b1 + (2 * q + p) * 512);
    }
    const int h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(u->bias + 32 * wave + 4 * h + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) seed[4 * g + e] = b4[e];
    }
}

// The weight stream (X3_STREAM): one k-step entry (hi + lo fragment, 2 KB per wave) is requested per k-step, eight k-steps
// ahead, into the ring entry the matrix instructions have just read -- and the stream runs on ACROSS the layer boundary: the
// last eight k-steps of a layer request the first eight of the next one.  Measured on the version that requested half a
// layer (two chunks + the bias, 20 KB per wave) behind the k loop: the eight waves' 160 KB take the CU's 64 B/clk
// vector-memory path 2 500 clocks to ACCEPT, and a wave's epilogue cannot start before its loads have issued (in-order):
// ~2 600 exposed clocks per 256 -> 256 layer (phase stamps of an MFMA-only build: 6 730 clocks of k loop + 2 780 of "nothing").
// RESULT (r3, D3 at B = 65 536, same box): 391 us with the stream against 386 us without.  The epilogue did shrink (4 000 -
// 5 300 -> 1 900 - 3 000 clocks) but the k loop grew by as much (6 400 - 6 900 -> 8 000 - 9 200): a wave issues in order, so
// a load the path cannot accept yet holds back the matrix instructions behind it wherever it stands.  The layer's 256 KB
// of fragments take 4 096 of the 6 144 matrix clocks on that path either way; only fewer bytes per row (a taller row tile,
// which LDS has no room for) would change it.  Kept as a build option (-DX3_STREAM=1), off.
struct NextW {                                       // the next GEMM unit's weight stream for this wave and lane
    const _Float16* a;                               // source 1: entry q at a + q * 1024 (hi), + 512 (lo)
    const _Float16* b;                               // source 2 (entries kp1 ..)
    int kp1, total;                                  // k-steps of source 1 (chunk-padded) / of both
};
__device__ __forceinline__ NextW next_stream(UnitPtr u, int wave, int lane) {
    NextW n;
    n.kp1 = chunks_of(u->ksteps) * X3_CH;
    const int ks2 = u->ksteps2, kp2 = ks2 > 0 ? chunks_of(ks2) * X3_CH : 0;
    n.total = n.kp1 + kp2;
    n.a = u->w + ((long long)wave * n.kp1 * 2 * 64 + lane) * 8;
    n.b = ks2 > 0 ? u->w2 + ((long long)wave * kp2 * 2 * 64 + lane) * 8 : n.a;
    return n;
}
__device__ __forceinline__ void load_next_entry(const NextW& n, int q, f16x8 (&e)[2]) {      // q < n.total (wave-uniform)
    const _Float16* base = q < n.kp1 ? n.a + (long long)q * 1024 : n.b + (long long)(q - n.kp1) * 1024;
    e[0] = *reinterpret_cast<const f16x8*>(base);
    e[1] = *reinterpret_cast<const f16x8*>(base + 512);
}
// entry K (compile time) of THIS layer's concatenated sources
template <int NCH, int NCH1>
__device__ __forceinline__ void load_own_entry(int K, const _Float16* w1, const _Float16* w2, int wave, int lane, f16x8 (&e)[2]) {
    const bool second = K >= NCH1 * X3_CH;                                    // (K: a constant once the k loop is unrolled)
    const int kpad = (second ? NCH - NCH1 : NCH1) * X3_CH;
    const int kk = second ? K - NCH1 * X3_CH : K;
    const _Float16* base = (second ? w2 : w1) + ((long long)wave * kpad * 2 * 64 + lane) * 8 + (long long)kk * 1024;
    e[0] = *reinterpret_cast<const f16x8*>(base);
    e[1] = *reinterpret_cast<const f16x8*>(base + 512);
}
__device__ __forceinline__ void load_bias_seed(UnitPtr u, int wave, int lane, f32x16& seed) {
    const int h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(u->bias + 32 * wave + 4 * h + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) seed[4 * g + e] = b4[e];
    }
}

// dst = act(W src [+ W2 src2] + bias + res), one layer.  NCH chunks of 64 k, the first NCH1 from source 1.  On entry the
// ring holds this layer's chunks 0 (and 1) and `seed` its bias; before the epilogue `next` (the GEMM unit that runs after
// this one, possibly of the next tile; nullptr: none) is prefetched the same way.  Waves whose slice lies beyond N only
// take part in the prefetch.
template <int NCH, int NCH1>
__device__ __forceinline__ void gemm_layer(UnitPtr u, UnitPtr next, unsigned char* smem, int wave, int lane, Ring& ring, f32x16& seed, int ui = 0) {
    asm volatile("" : "+v"(lane));                   // lane-derived constants are recomputed per unit, not parked across units
    const int r31 = lane & 31, h = lane >> 5;
    const int nslices = (u->N + 31) >> 5;
    if (wave >= nslices) {                           // wave-uniform
        if (next != nullptr) prefetch_layer(next, wave, lane, ring, seed);
        return;
    }
    const _Float16* w1 = u->w;
    const _Float16* w2 = NCH1 < NCH ? u->w2 : u->w;
    f32x16 acc[X3_MT];
    const unsigned char* src1 = buf_base(smem, u->src);
    const int pbs1 = buf_pitch_bytes(u->src), pl1 = buf_plane(u->src);
    const unsigned char* src2 = NCH1 < NCH ? buf_base(smem, u->src2) : src1;
    const int pbs2 = NCH1 < NCH ? buf_pitch_bytes(u->src2) : pbs1, pl2 = NCH1 < NCH ? buf_plane(u->src2) : pl1;
    constexpr int KT = NCH * X3_CH;
    f16x8 fx[2][X3_MT][2];                           // activation fragments (hi, lo), read one k-step ahead (the SIMD's other
                                                     // wave covers the LDS latency; three stages spill at 256 registers)
    // 256-wide sources live in buffers 0 / 1 (pitch and plane size are constants there): ONE address per k-step -- row
    // 32 + r has r's swizzle (32 % 16 == 0) and the lo plane is a constant away, so the four fragments of a k-step are
    // immediate offsets of it (2 VALU per k-step instead of 9 in front of the matrix instructions)
#ifdef X3_OLD_ADDR
    constexpr bool WIDE = false;
#else
    constexpr bool WIDE = NCH1 == 4 && (NCH == 4 || NCH == 8);
#endif
    const int wide_row = r31 * (P01 * 2), wide_sw = r31 & 15;
    auto read_frags = [&](int k, f16x8 (&f)[X3_MT][2]) {
        const bool second = k >= NCH1 * X3_CH;
        const unsigned char* src = second ? src2 : src1;
        const int kk = second ? k - NCH1 * X3_CH : k;
        if (WIDE) {
            const unsigned char* a = src + wide_row + (((2 * kk + h) ^ wide_sw) << 4);
#pragma unroll
            for (int mt = 0; mt < X3_MT; ++mt) {
                f[mt][0] = *reinterpret_cast<const f16x8*>(a + mt * 32 * (P01 * 2));
                f[mt][1] = *reinterpret_cast<const f16x8*>(a + mt * 32 * (P01 * 2) + PLANE01);
            }
            return;
        }
        const int pbs = second ? pbs2 : pbs1, pl = second ? pl2 : pl1;
#pragma unroll
        for (int mt = 0; mt < X3_MT; ++mt) {
            const int o = chunk_off(32 * mt + r31, 2 * kk + h, pbs);
            f[mt][0] = *reinterpret_cast<const f16x8*>(src + o);
            f[mt][1] = *reinterpret_cast<const f16x8*>(src + pl + o);
        }
    };
    read_frags(0, fx[0]);
#if X3_STREAM
    NextW nw = {};
    if (next != nullptr) nw = next_stream(next, wave, lane);
#endif
#if X3_PRIO_SEL
    // the two waves of a SIMD start every layer together (barrier) and would share the matrix pipe turn by turn, reaching
    // their epilogues together -- both exposed, and fighting over LDS.  The first wave of the pair takes the pipe (issue
    // priority) and runs its epilogue UNDER the second wave's matrix phase: one epilogue per layer is exposed, alone on
    // the SIMD.  (No arithmetic changes: results are the same bits.)
    if (X3_PRIO_SEL == 1 ? wave < 4 : (wave & 1) == 0) __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int c = k / X3_CH, q = k % X3_CH;
#ifndef X3_ABL_NOREAD
        if (k + 1 < KT) read_frags(k + 1, fx[(k + 1) & 1]);
#endif
#if !X3_STREAM
#ifdef X3_ABL_NOWLOAD
        if (false) {
#else
        // chunk c+1 is requested when chunk c starts, into the slot chunk c-1 just left (chunk 1 came with the prefetch)
        if (q == 0 && c >= 1 && c + 1 < NCH) {
#endif
            if (c + 1 == 2) load_chunk<2 < NCH ? 2 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[0]);
            if (c + 1 == 3) load_chunk<3 < NCH ? 3 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[1]);
            if (c + 1 == 4) load_chunk<4 < NCH ? 4 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[0]);
            if (c + 1 == 5) load_chunk<5 < NCH ? 5 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[1]);
            if (c + 1 == 6) load_chunk<6 < NCH ? 6 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[0]);
            if (c + 1 == 7) load_chunk<7 < NCH ? 7 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[1]);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // small terms first, then hi * hi
#pragma unroll
        for (int mt = 0; mt < X3_MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[c & 1][q][1], fx[k & 1][mt][0], k == 0 ? seed : acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < X3_MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[c & 1][q][0], fx[k & 1][mt][1], acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < X3_MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[c & 1][q][0], fx[k & 1][mt][0], acc[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#if X3_STREAM && !defined(X3_ABL_NOWLOAD)
        // the entry just read is free: request the one eight k-steps ahead into it -- this layer's, or the next layer's
        if (k + 8 < KT) {
            load_own_entry<NCH, NCH1>(k + 8, w1, w2, wave, lane, ring[c & 1][q]);
        } else if (next != nullptr) {
            if (KT >= 8) {
                if (k + 8 - KT < nw.total) load_next_entry(nw, k + 8 - KT, ring[c & 1][q]);
            } else {                                                      // a 4-k-step layer feeds two entries per step
                if (k < nw.total) load_next_entry(nw, k, ring[0][q]);
                if (k + 4 < nw.total) load_next_entry(nw, k + 4, ring[1][q]);
            }
        }
        if (k == KT - 1 && next != nullptr) load_bias_seed(next, wave, lane, seed);   // (16 registers: not live across the loop)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#if X3_PRIO_SEL
    __builtin_amdgcn_s_setprio(0);
#endif
    if (ui >= 0) { X3_STAMP(4 * ui + 1) }
#if !X3_STREAM
    // the ring and the seed are dead: the next layer's first fragments travel during the epilogue and the barrier
    if (next != nullptr) prefetch_layer(next, wave, lane, ring, seed);
#endif
    const bool to_global = (u->flags & F_OUT_F32) != 0;
    unsigned char* dst = buf_base(smem, u->dst);
    const int pbd = buf_pitch_bytes(u->dst), pld = buf_plane(u->dst);
    const int resid = to_global ? -1 : u->res;
    const unsigned char* res = buf_base(smem, resid >= 0 ? resid : 0);
    const int pbr = buf_pitch_bytes(resid >= 0 ? resid : 0), plr = buf_plane(resid >= 0 ? resid : 0);
    const float neg = act_neg(u->act, u->slope);
    const int slice = wave;
    // epilogue: this lane owns row (32 mt + r31), features 32*slice + 8g + 4h .. +3
    if (to_global) {
        float* st = reinterpret_cast<float*>(dst);
#pragma unroll
        for (int mt = 0; mt < X3_MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[mt][4 * g + e], neg);
                *reinterpret_cast<f32x4*>(st + (32 * mt + r31) * OUT_PITCH + 32 * slice + 4 * h + 8 * g) = v;
            }
        return;
    }
#ifdef X3_ABL_NOEPI
    if (u->slope != 12345.f) return;
#endif
    // all residual values first (16 reads in flight, one LDS round trip): read where they are used, every group's loads sat
    // behind the previous group's stores (dst may be res: they may alias, the compiler keeps the order) -- eight exposed LDS
    // round trips per epilogue under the other wave's fragment traffic (phase stamps: 4 - 5.6 k clocks per epilogue)
    f16x4 rhv[X3_MT][4], rlv[X3_MT][4];
    if (resid >= 0) {
#pragma unroll
        for (int mt = 0; mt < X3_MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ro = chunk_off(32 * mt + r31, 4 * slice + g, pbr) + (h << 3);
                rhv[mt][g] = *reinterpret_cast<const f16x4*>(res + ro);
                rlv[mt][g] = *reinterpret_cast<const f16x4*>(res + plr + ro);
            }
    }
#pragma unroll
    for (int mt = 0; mt < X3_MT; ++mt) {
        const int row = 32 * mt + r31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * g + e];
            if (resid >= 0) {
                const f16x4 rh = rhv[mt][g], rl = rlv[mt][g];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rh[e] + (float)rl[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fn(v[e], neg);
            uint2 oh, ol;
            split2(v[0], v[1], oh.x, ol.x);
            split2(v[2], v[3], oh.y, ol.y);
            const int o = chunk_off(row, 4 * slice + g, pbd) + (h << 3);
            *reinterpret_cast<uint2*>(dst + o) = oh;
            *reinterpret_cast<uint2*>(dst + pld + o) = ol;
        }
    }
}

// LOAD: global fp32 (M, ld) columns [0, cols) -> hi / lo planes of buffer dst, zero-filled up to the next multiple of 64
// columns and below row M.  cols and ld even: a thread moves column pairs (8-byte loads, 4-byte LDS writes).
__device__ __forceinline__ void load_unit(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    asm volatile("" : "+v"(tid));
    const float* g = static_cast<const float*>(u->g);
    const long long ld = u->ld;
    const int cols = u->cols, id = u->dst;
    unsigned char* dst = buf_base(smem, id);
    const int pb = buf_pitch_bytes(id), pl = buf_plane(id);
    const int pairs = ((cols + 63) & ~63) >> 1;                              // per row, zero fill included: 32, 64 or 128
    const int sh = pairs == 32 ? 5 : (pairs == 64 ? 6 : 7);
    const int total = X3_BM << sh;
    constexpr int NB = 8;                                                    // loads in flight per thread
    for (int i0 = tid; i0 < total; i0 += X3_THREADS * NB) {
        f32x2 v[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i = i0 + j * X3_THREADS, row = i >> sh, c2 = i & (pairs - 1);
            v[j] = f32x2{0.f, 0.f};
            if (i < total && m0 + row < M && 2 * c2 < cols) v[j] = *reinterpret_cast<const f32x2*>(g + (m0 + row) * ld + 2 * c2);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i = i0 + j * X3_THREADS, row = i >> sh, c2 = i & (pairs - 1);
            uint32_t hi, lo;
            split2(v[j][0], v[j][1], hi, lo);
            if (i < total) {
                const int o = chunk_off(row, c2 >> 2, pb) + ((c2 & 3) << 2);
                *reinterpret_cast<uint32_t*>(dst + o) = hi;
                *reinterpret_cast<uint32_t*>(dst + pl + o) = lo;
            }
        }
    }
}

// LOAD_KCS: the 30 KCS features of a tile's poses (global fp32 (M, ld >= 48), root-relative or not: bones are differences),
// computed here in the arithmetic of the stand-alone dhaug_kcs_forward (kcs_features, dhaug_fk_math.h: IEEE sqrt and divide) ->
// hi / lo planes of buffer dst, columns 30..63 zero.  One lane per pose for the features (64 of the 512 threads; the tile's
// 12 KB of poses are read with 16-byte loads), every thread for the zero fill.  Replaces a separate 9 us launch + 8 MB round
// trip in front of the 3D critic's parity program.
__device__ __forceinline__ void load_kcs_unit(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    asm volatile("" : "+v"(tid));
    const float* g = static_cast<const float*>(u->g);
    const long long ld = u->ld;
    const int id = u->dst;
    unsigned char* dst = buf_base(smem, id);
    const int pb = buf_pitch_bytes(id), pl = buf_plane(id);
    // pairs 15 .. 31 of every row: zero
    for (int s = tid; s < X3_BM * 17; s += X3_THREADS) {
        const int row = s / 17, c2 = 15 + (s - row * 17);
        const int o = chunk_off(row, c2 >> 2, pb) + ((c2 & 3) << 2);
        *reinterpret_cast<uint32_t*>(dst + o) = 0u;
        *reinterpret_cast<uint32_t*>(dst + pl + o) = 0u;
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
            const int o = chunk_off(row, c2 >> 2, pb) + ((c2 & 3) << 2);
            *reinterpret_cast<uint32_t*>(dst + o) = hi;
            *reinterpret_cast<uint32_t*>(dst + pl + o) = lo;
        }
    }
}

__device__ __forceinline__ void store_output(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    const float* st = reinterpret_cast<const float*>(buf_base(smem, u->dst));
    float* out = static_cast<float*>(const_cast<void*>(u->g));
    const long long ld = u->ld;
    const int N = u->N;
    for (int i = tid; i < X3_BM * N; i += X3_THREADS) {
        const int row = i / N, c = i - row * N;
        if (m0 + row < M) out[(m0 + row) * ld + c] = st[row * OUT_PITCH + c];
    }
}

__global__ __launch_bounds__(X3_THREADS, 1) void fused_mlp_x3_r3_kernel(Program prog, long long M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ntiles = (M + X3_BM - 1) / X3_BM;
    const unsigned char __attribute__((address_space(4)))* ka =
        (const unsigned char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    UnitPtr units = (UnitPtr)(ka + __builtin_offsetof(Program, u));
    const int nunits = *(const int __attribute__((address_space(4)))*)(ka + __builtin_offsetof(Program, nunits));
    const int first_gemm = *(const int __attribute__((address_space(4)))*)(ka + __builtin_offsetof(Program, first_gemm));
    Ring ring;
    f32x16 seed;
    if ((long long)blockIdx.x < ntiles) prefetch_layer(units + first_gemm, wave, lane, ring, seed);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m0 = tile * X3_BM;
        const bool more = tile + gridDim.x < ntiles;
#pragma unroll 1
        for (int i = 0; i < nunits; ++i) {
            UnitPtr u = units + i;
            const int kind = u->kind, plan = u->plan;
            if (tile == blockIdx.x) { X3_STAMP(4 * i) }
            if (kind == U_LOAD_F32) {
                load_unit(u, smem, m0, M, tid);
            } else if (kind == U_LOAD_KCS) {
                load_kcs_unit(u, smem, m0, M, tid);
            } else {
                // the GEMM unit that runs after this one: plan bits 0..7 hold its index + 1 (0: this is the program's last;
                // the next tile then starts over at the first)
                const int nx = plan & 255;
                UnitPtr next = nx != 0 ? units + (nx - 1) : (more ? units + first_gemm : (UnitPtr) nullptr);
                switch ((plan >> 16) & 255) {                          /* validated on the host */
                    case 1 * 16 + 1: gemm_layer<1, 1>(u, next, smem, wave, lane, ring, seed, tile == blockIdx.x ? i : -1); break;
                    case 2 * 16 + 2: gemm_layer<2, 2>(u, next, smem, wave, lane, ring, seed, tile == blockIdx.x ? i : -1); break;
                    case 2 * 16 + 1: gemm_layer<2, 1>(u, next, smem, wave, lane, ring, seed, tile == blockIdx.x ? i : -1); break;
                    case 4 * 16 + 4: gemm_layer<4, 4>(u, next, smem, wave, lane, ring, seed, tile == blockIdx.x ? i : -1); break;
                    case 4 * 16 + 2: gemm_layer<4, 2>(u, next, smem, wave, lane, ring, seed, tile == blockIdx.x ? i : -1); break;
                    case 8 * 16 + 4: gemm_layer<8, 4>(u, next, smem, wave, lane, ring, seed, tile == blockIdx.x ? i : -1); break;
                    default: break;
                }
                if (u->flags & F_OUT_F32) {
                    lds_barrier();
                    store_output(u, smem, m0, M, tid);
                }
            }
            if (tile == blockIdx.x) { X3_STAMP(4 * i + 2) }
            lds_barrier();
            if (tile == blockIdx.x) { X3_STAMP(4 * i + 3) }
        }
    }
    (void)prog;
}

}  // namespace

extern "C" {

int dhaug_mlp_forward_x3_r3(const dhaug_mlp_unit* units, int nunits, int64_t M, void* stream) {
    DHAUG_CHECK(nunits >= 1 && nunits <= X3_MAX_UNITS && M >= 0, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(units);
    if (M == 0) return DHAUG_OK;
    Program prog;
    prog.nunits = nunits;
    auto okbuf = [](int b) { return b >= 0 && b <= 2; };
    auto pitch = [](int b) { return b == 2 ? P2 : P01; };
    for (int i = 0; i < nunits; ++i) {
        const dhaug_mlp_unit& s = units[i];
        Unit& u = prog.u[i];
        u.kind = s.kind; u.flags = s.flags; u.src = s.src; u.dst = s.dst; u.res = s.res; u.src2 = s.src2; u.ksteps2 = s.ksteps2;
        u.ksteps = s.ksteps; u.N = s.n; u.act = s.act; u.slope = s.slope; u.cols = s.cols; u.ld = s.ld; u.g = s.g;
        u.w = static_cast<const _Float16*>(s.w); u.w2 = static_cast<const _Float16*>(s.w2); u.bias = s.bias;
        u.plan = 0;
        DHAUG_CHECK(u.kind == U_LOAD_F32 || u.kind == U_GEMM || u.kind == U_LOAD_KCS, DHAUG_EUNSUPPORTED);
        if (u.kind == U_GEMM) {
            DHAUG_CHECK(okbuf(u.src) && u.ksteps >= 1 && u.ksteps <= X3_MAX_KSTEPS && u.N >= 1 && u.N <= 256, DHAUG_EUNSUPPORTED);
            DHAUG_CHECK(((u.ksteps + 3) / 4) * 64 <= pitch(u.src), DHAUG_EUNSUPPORTED);
            DHAUG_CHECK(u.w != nullptr && dhaug_aligned16(u.w) && u.bias != nullptr && dhaug_aligned16(u.bias), DHAUG_EALIGN);
            DHAUG_CHECK(u.ksteps2 >= 0 && u.ksteps2 <= X3_MAX_KSTEPS, DHAUG_EUNSUPPORTED);
            DHAUG_CHECK((u.flags & ~F_OUT_F32) == 0, DHAUG_EUNSUPPORTED);
            if (u.ksteps2 > 0) {
                DHAUG_CHECK(okbuf(u.src2) && ((u.ksteps2 + 3) / 4) * 64 <= pitch(u.src2) && u.ksteps % 4 == 0, DHAUG_EUNSUPPORTED);
                DHAUG_CHECK(u.w2 != nullptr && dhaug_aligned16(u.w2), DHAUG_EALIGN);
            }
            const int c1 = (u.ksteps + 3) / 4, c2 = (u.ksteps2 + 3) / 4, sh = (c1 + c2) * 16 + c1;
            DHAUG_CHECK(sh == 17 || sh == 34 || sh == 33 || sh == 68 || sh == 66 || sh == 132, DHAUG_EUNSUPPORTED);
            if (u.flags & F_OUT_F32) {
                DHAUG_CHECK(u.g != nullptr && u.ld >= u.N && u.N <= 64, DHAUG_EUNSUPPORTED);
                DHAUG_CHECK((u.dst == 0 || u.dst == 1) && u.dst != u.src && (u.ksteps2 == 0 || u.dst != u.src2), DHAUG_EINVAL);
            } else {
                DHAUG_CHECK(okbuf(u.dst) && u.dst != u.src && (u.ksteps2 == 0 || u.dst != u.src2), DHAUG_EINVAL);
                DHAUG_CHECK(((u.N + 31) / 32) * 32 <= pitch(u.dst), DHAUG_EUNSUPPORTED);
                DHAUG_CHECK(u.res < 0 || (okbuf(u.res) && u.res != u.src && (u.ksteps2 == 0 || u.res != u.src2)), DHAUG_EINVAL);
                DHAUG_CHECK(u.res < 0 || ((u.N + 31) / 32) * 32 <= pitch(u.res), DHAUG_EUNSUPPORTED);
            }
            u.plan = (sh << 16) | (((u.N + 31) >> 5) << 24);                  // bits 0..7: index + 1 of the next GEMM unit (below)
        } else if (u.kind == U_LOAD_KCS) {
            DHAUG_CHECK(okbuf(u.dst) && u.g != nullptr && u.ld >= 48 && 64 <= pitch(u.dst), DHAUG_EINVAL);
        } else {
            DHAUG_CHECK(okbuf(u.dst) && u.g != nullptr && u.cols >= 2 && ((u.cols + 63) & ~63) <= pitch(u.dst), DHAUG_EINVAL);
            DHAUG_CHECK(u.cols % 2 == 0 && u.ld % 2 == 0 && u.ld >= u.cols && (reinterpret_cast<uintptr_t>(u.g) & 7u) == 0, DHAUG_EALIGN);
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
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_x3_r3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const long long ntiles = (M + X3_BM - 1) / X3_BM;
    const unsigned grid = dhaug_persistent_grid(ntiles);           // one persistent workgroup per CU
    hipLaunchKernelGGL(fused_mlp_x3_r3_kernel, dim3(grid), dim3(X3_THREADS), X3_LDS_BYTES, (hipStream_t)stream, prog, (long long)M);
    return dhaug_launch_status();
}

#if 0
int dhaug_debug_mlp_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3_stamps), sizeof(long long) * (n < 4 * X3_MAX_UNITS ? n : 4 * X3_MAX_UNITS));
}
#endif
}  // extern "C"
