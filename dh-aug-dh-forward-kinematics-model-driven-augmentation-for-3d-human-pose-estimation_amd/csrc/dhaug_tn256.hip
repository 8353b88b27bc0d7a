// Weight gradients of a whole critic step in ONE launch: for every layer i of a group,
//     C_i[N1_i, N2_i] (+)= A_i[M_i, N1_i]^T * B_i[M_i, N2_i],   colsum_i (+)= column sums of A_i over rows [0, cs_rows_i)
// with N1, N2 <= 256 (d/dW and d/db of the explicit critic step, sweep 4 of critic_step.py; replaces the parameter-gradient
// half of loss.backward() in R/models_Fk_GAN/model_fk_gan_train.py:191-214).
//
// Why grouped, and why whole-output tiles.  The 64 x 64-tile kernel (dhaug_gemm.hip, gemm_tn64_kernel) keeps the split-K
// volume small but every operand row crosses L2 -> LDS four times (805 MB per 3B x 256 x 256 layer for 201 MB of operands).
// A workgroup that owns the WHOLE 256 x 256 output over its slice of the batch reads every operand byte once and its loop
// runs at HBM speed (measured 32 us per layer, 6.3 TB/s) -- but it leaves one 256 KB partial result per workgroup, and
// merging 256 of them costs more than it saved: fp32 atomics retire at ~1 dword per L2 channel and clock (1 TB/s
// chip-wide; measured 64 us for the 64 MB of one layer, whether the lines are shared between the XCDs or private to one).
// The partial volume is (workgroups) x (tile), NOT per layer: so all layers of a step share one launch, the 256 workgroups
// are dealt out to the layers in proportion to their operand bytes (a 3B x 256 x 256 layer of the 3D critic gets ~20), each
// writes ONE partial with plain stores, and a second small launch sums a layer's partials into its gradient slot
// (64 MB of partials per STEP instead of per layer).
//   * 512 threads = 8 waves (two per SIMD); wave (w1, w2) accumulates C rows [64 w1, +64) x columns [128 w2, +128):
//     2 x 4 MFMA tiles of 32 x 32 = 128 accumulator registers (256 KB of fp32 per workgroup: only the register file holds it);
//     waves whose quadrant lies outside a narrow layer's N1 x N2 only move data;
//   * operand rows travel global -> LDS without registers (global_load_lds_dwordx4), 32-row stages of 32 KB (A | B), four in
//     the ring = three in flight per CU, exact vmcnt waits, LDS-only barriers; columns beyond a layer's width are copied from
//     16 zero bytes (every lane of every copy stays active: the vmcnt arithmetic is the same in every wave);
//   * fragments leave the row-major stage through the hardware transpose read (ds_read_b64_tr_b16).
// Where the operands are SHORT and the layers wide (a video step: 12 - 24 layers of 1000 x 1000 over 1 536 - 9 216 rows, every 256 x 256
// block one workgroup over all rows) the launch is bound by the stage body, not by memory.  Measured per 32-row stage on 12 such layers
// of 4 608 rows (tools/abl_tn_wide.sh, round 6): MFMAs alone 0.60 us (16 per wave, 20 where the column sums ride along, at the ~2 GHz
// the card holds under them: the matrix pipe's own time), fragment reads alone 0.25 us, copies alone 0.22 us; together 0.97 us -- the
// three mostly one after the other, 0.24 - 0.29 of the MFMA peak per launch.  Tried and not kept: all 24 fragment reads of a stage
// ahead of its first MFMA and the column-sum stages as a loop of their own (no branch between MFMAs): no change; the two waves of
// every SIMD a half stage apart (one reads while the other multiplies, two barriers per stage, the copies issued between the MFMAs):
// hides the reads (compute without copies 153 us against 157) but each copy then stalls the issuing wave's MFMA stream for ~150
// clocks -- 194 - 212 us against 180 for the launch, and 5.0 - 5.25 TB/s against 5.5 on the single-frame step's long operands.
// What did pay there: adding into the gradient slots with all of a tile's old values requested at once (see the epilogue), 245 -> 150 us
// for 24 layers of 1 536 rows; consecutive work items on one XCD (see the kernel's first lines), a further 5 %.
#include <cstdlib>
#include <algorithm>
#include "dhaug_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int T2_ROWS = 32;                       // contraction rows per stage
constexpr int T2_HALF = T2_ROWS * 512;            // one operand's stage image: 32 rows x 512 B
constexpr int T2_STG = 2 * T2_HALF;               // 32 768 B
constexpr int T2_NSTG = 4;
constexpr int T2_LDS = T2_NSTG * T2_STG;          // 131 072 B
constexpr int T2_WS_STRIDE = 256 * 256 + 256;     // floats per partial: C, then the column sums
constexpr int T2_MAX_WG = 256;
constexpr int T2_MAX_LAYERS = DHAUG_TN_GROUP_MAX;

struct TnLayer {
    const uint16_t* A; const uint16_t* B;
    float* C; float* colsum;
    long long lda, ldb, ldc;
    long long cs_rows;              // column sums over rows [0, cs_rows): a multiple of 32
    int nst;                        // 32-row stages of the layer (M / 32)
    int n1, n2;                     // whole layer: a layer wider than 256 is a grid of 256 x 256 blocks, (n2 + 255) / 256 per row
    int wg0, nwg;                   // workgroups [wg0, wg0 + nwg) work on this layer: `split` per block, block-major
    int accumulate;                 // bit 0: add into C / colsum; bits 8-9 / 10-11: A / B are the three PLANES of a split operand (see copy_stage)
    int split;                      // 1: the block's one workgroup adds its result into C / colsum itself (no partial, no sum)
    int ws0;                        // split > 1: the layer's partial results start at workspace slot ws0
};
struct TnGroup {
    int nlayers;
    int abl;                        // development: 1 no partial stores, 2 no fragment reads / MFMAs, 4 no copies (timing only);
                                    // bit 8 (set by the host for the two-phase form): a block's only workgroup writes a partial
                                    // result too -- in that form only the summing launch may touch the gradient slots
    float* ws;
    TnLayer L[T2_MAX_LAYERS];
};
typedef const TnLayer __attribute__((address_space(4))) * LayerPtr;

__device__ __forceinline__ void t2_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16 bytes per lane global -> LDS without registers; the wave's 1 KB lands at lds_wave_base + 16 * lane.  Issued as inline
// assembly ON PURPOSE: hipcc's waitcnt pass knows that the LDS-DMA builtin writes LDS and puts a full `s_waitcnt vmcnt(0)` in
// front of the next transpose read (an intrinsic without memory operand: "may alias") -- i.e. right behind the copies of
// the stage three ahead, which drains the ring at every stage.  The copies' completion is tracked by hand below (exact
// vmcnt: a wave's requests retire in order).  The pointer operand + "memory" tell the compiler that LDS is written.
typedef unsigned char __attribute__((address_space(3))) * t2_lds_ptr;
__device__ __forceinline__ void t2_copy16(const void* g, t2_lds_ptr lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_wave_base), "v"(g) : "memory");
}

// 16-byte chunk c of row r sits at position c ^ (4 * (r & 3)): rows are 512 B = 2 x 64 banks apart, so the four rows of a
// transpose read land in the four 64-byte quarters of the bank space
__device__ __forceinline__ int t2_sw(int row) { return (row & 3) << 2; }

// 8 consecutive contraction rows kbase .. kbase+7 of column col0 + (lane & 15): two hardware transpose reads
__device__ __forceinline__ bf16x8 t2_frag(const unsigned char* img, int kbase, int col0, int lane) {
    const int li = lane & 15, q = li >> 2, pp = li & 3;
    const int c = (col0 >> 3) + (pp >> 1), in = (pp & 1) << 3;
    const int r0 = kbase + q, r1 = r0 + 4;
    typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + r0 * 512 + ((c ^ t2_sw(r0)) << 4) + in));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + r1 * 512 + ((c ^ t2_sw(r1)) << 4) + in));
    const bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

__device__ uint4 g_t2_zero16 = {0u, 0u, 0u, 0u};

__global__ __launch_bounds__(512, 1) void gemm_tn_group_kernel(TnGroup grp_by_value) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];       // [4 stages][A 16 KB | B 16 KB]
    (void)grp_by_value;
    // the group lives in the kernarg segment: the layer of this workgroup is found with scalar loads (a by-value struct
    // indexed dynamically would be copied to scratch)
    const unsigned char __attribute__((address_space(4)))* ka =
        (const unsigned char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    const int nlayers = *(const int __attribute__((address_space(4)))*)(ka + __builtin_offsetof(TnGroup, nlayers));
    const int abl = *(const int __attribute__((address_space(4)))*)(ka + __builtin_offsetof(TnGroup, abl));
    float* ws = *(float* const __attribute__((address_space(4)))*)(ka + __builtin_offsetof(TnGroup, ws));
    LayerPtr layers = (LayerPtr)(ka + __builtin_offsetof(TnGroup, L));
    // Workgroup -> work.  A launch with layers wider than 256 (abl bit 16, grid a multiple of 8) deals CONSECUTIVE work items to ONE
    // XCD (workgroup b runs on XCD b % 8): the 4 x 4 blocks of a DenseDim-1000 layer share their operand slabs four ways, and only
    // blocks behind the same L2 read a slab from memory once.  Dealt round-robin, every block fetched its own copy: the video
    // step's launches moved 5 - 7 TB/s through the fabric whatever their shape, 3 000 clocks per 32-row stage with 211 workgroups
    // resident against 1 400 with 128.
    const int bx = (abl & 16) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int li = 0;
    for (int i = 1; i < nlayers; ++i)
        if (bx >= layers[i].wg0) li = i;
    LayerPtr L = layers + li;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w1 = wave >> 1, w2 = wave & 1;
    const int jw = bx - L->wg0;
    if (jw >= L->nwg) return;
    // the workgroup's 256 x 256 block of the layer, and its slice of the batch
    const int split = L->split, blk = jw / split, j = jw - blk * split;
    const int nbj = (L->n2 + 255) >> 8, bi = blk / nbj, bj = blk - bi * nbj;
    const int s0 = (int)((long long)L->nst * j / split), s1 = (int)((long long)L->nst * (j + 1) / split);
    const int nst = s1 - s0;                                      // >= 1: the host gives a block at most one workgroup per stage
    const long long ms = (long long)s0 * T2_ROWS;
    const uint16_t* A = L->A + 256 * bi;
    const uint16_t* B = L->B + 256 * bj;
    const long long lda = L->lda, ldb = L->ldb, cs_rows = L->cs_rows;
    const int n1 = L->n1 - 256 * bi < 256 ? L->n1 - 256 * bi : 256, n2 = L->n2 - 256 * bj < 256 ? L->n2 - 256 * bj : 256;
    const int ca = (n1 + 7) >> 3, cb = (n2 + 7) >> 3;             // live 16-byte chunks per row
    const t2_lds_ptr lds0 = (t2_lds_ptr)tsm;

    // An operand given as PLANES (dhaug_tn_layer.planes_a / _b: the three distinct bf16 pieces [hi | mid | lo] of a split fp32 tensor,
    // three rows per tensor row) is contracted over SIX virtual rows per tensor row, piece (0 0 1 1 0 2)[t] (order 1, the activation
    // side of dhaug_split_bf16) or (0 1 0 1 2 0)[t] (order 2, the weight side) for virtual row 6 m + t -- the rows of the six-segment
    // operand without their copies, in the same order (bit-identical sums): the repeated rows come out of L2, not out of HBM.
    const int flags = L->accumulate;
    const unsigned pmapA = ((flags >> 8) & 3) == 1 ? 0x850u : 0x244u, pmapB = ((flags >> 10) & 3) == 1 ? 0x850u : 0x244u;
    const bool planesA = ((flags >> 8) & 3) != 0, planesB = ((flags >> 10) & 3) != 0;
    auto real_row = [](long long rv, unsigned pmap) {
        const unsigned r = (unsigned)rv, q = __umulhi(r, 0xAAAAAAABu) >> 2, t = r - 6u * q;
        return (long long)(3u * q + ((pmap >> (2u * t)) & 3u));
    };
    auto copy_stage = [&](int st) {
        const long long m0 = ms + (long long)st * T2_ROWS;
        const t2_lds_ptr base = lds0 + (st % T2_NSTG) * T2_STG;
#pragma unroll
        for (int i = 0; i < 2; ++i) {                             // 32 chunks per row: two rows per wave instruction
            const int row0 = (wave * 2 + i) * 2, row = row0 + (lane >> 5), c = (lane & 31) ^ t2_sw(row);
            const long long ra = planesA ? real_row(m0 + row, pmapA) : m0 + row, rb = planesB ? real_row(m0 + row, pmapB) : m0 + row;
            t2_copy16(c < ca ? static_cast<const void*>(A + ra * lda + c * 8) : static_cast<const void*>(&g_t2_zero16),
                      base + row0 * 512);
            t2_copy16(c < cb ? static_cast<const void*>(B + rb * ldb + c * 8) : static_cast<const void*>(&g_t2_zero16),
                      base + T2_HALF + row0 * 512);
        }
    };
    f32x16 acc[2][4], accs[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accs[t][r] = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u][r] = 0.0f;
        }
    }
    const bool live = 64 * w1 < n1 && 128 * w2 < n2 && !(abl & 2);       // this wave's quadrant holds part of the result
    const bool do_cs = L->colsum != nullptr && bj == 0 && cs_rows > 0 && w2 == 0 && live;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const int grp = lane >> 4;                                    // 16-lane group: columns 16 (grp & 1), k half grp >> 1
#pragma unroll
    for (int st = 0; st < T2_NSTG - 1; ++st)
        if (st < nst && !(abl & 4)) copy_stage(st);
    for (int st = 0; st < nst; ++st) {
        const int younger = nst - 1 - st;                         // stages st+1, st+2 may still fly (4 copies per wave each)
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t2_lds_barrier();                                         // stage st is in LDS; every wave is done with stage st-1
        if (st + T2_NSTG - 1 < nst && !(abl & 4)) copy_stage(st + T2_NSTG - 1);
        if (!live) continue;                                      // (wave-uniform)
        const unsigned char* sa = tsm + (st % T2_NSTG) * T2_STG;
        const unsigned char* sb = sa + T2_HALF;
        const bool cs_stage = do_cs && ms + (long long)st * T2_ROWS < cs_rows;
#pragma unroll
        for (int ks = 0; ks < T2_ROWS / 16; ++ks) {
            const int kbase = 16 * ks + 8 * (grp >> 1);
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) fa[t] = t2_frag(sa, kbase, 64 * w1 + 32 * t + 16 * (grp & 1), lane);
#pragma unroll
            for (int u = 0; u < 4; ++u) fb[u] = t2_frag(sb, kbase, 128 * w2 + 32 * u + 16 * (grp & 1), lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t], fb[u], acc[t][u], 0, 0, 0);
                if (cs_stage) accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t], ones, accs[t], 0, 0, 0);
            }
        }
    }
    if (!live || (abl & 1)) return;
    // (D[n1][n2]: the lane owns column n2 = lane & 31 of its tile, rows n1 = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    if (split == 1 && !(abl & 8)) {
        // the block's only workgroup: its result IS the block's -- added into (or stored to) the gradient slot right here,
        // 128-byte row pieces; no partial result, nothing for the summing launch to do
        float* C = L->C + (long long)256 * bi * L->ldc + 256 * bj;
        const long long ldc = L->ldc;
        const int accm = L->accumulate & 1;
        float* csum = (L->colsum != nullptr && bj == 0) ? L->colsum + 256 * bi : nullptr;
        // Adding into the slot: ALL of a 32-row tile's old values are requested before the first is used (written element by
        // element, `*d = *d + acc`, every one of a wave's 128 read-modify-writes waited for its own load -- vmcnt(0), i.e. for the
        // stores before it as well: 128 memory round trips in a row, ~100 us per workgroup; that, not the contraction, was
        // what the video step's launches took: 1 536-row layers 245 us for 48 stages)
#pragma unroll
        for (int th = 0; th < 4; ++th) {                          // a 32-row tile in two halves of 8 register rows (register budget)
            const int t = th >> 1, rb = (th & 1) * 8;
            if (64 * w1 + 32 * t >= n1) continue;
            float old[8][4], olds[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = rb + q, r1 = 64 * w1 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c1 = 128 * w2 + 32 * u + (lane & 31);
                    old[q][u] = (accm && r1 < n1 && c1 < n2) ? C[r1 * ldc + c1] : 0.0f;
                }
                olds[q] = (accm && csum != nullptr && w2 == 0 && (lane & 31) == 0 && r1 < n1) ? csum[r1] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {                         // (the loads stay up there: values pinned before the first store)
                asm volatile("" : "+v"(olds[q]));
#pragma unroll
                for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(old[q][u]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = rb + q, r1 = 64 * w1 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (r1 >= n1) continue;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c1 = 128 * w2 + 32 * u + (lane & 31);
                    if (c1 < n2) C[r1 * ldc + c1] = old[q][u] + acc[t][u][r];
                }
                if (csum != nullptr && w2 == 0 && (lane & 31) == 0) csum[r1] = olds[q] + (cs_rows > 0 ? accs[t][r] : 0.0f);
            }
        }
        return;
    }
    // this workgroup's partial result: plain stores; only the tiles that hold part of the N1 x N2 result
    float* part = ws + (long long)(L->ws0 + jw) * T2_WS_STRIDE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (64 * w1 + 32 * t >= n1) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int r1 = 64 * w1 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (128 * w2 + 32 * u < n2) part[r1 * 256 + 128 * w2 + 32 * u + (lane & 31)] = acc[t][u][r];
            if (w2 == 0 && (lane & 31) == 0) part[256 * 256 + r1] = accs[t][r];   // (every column of A^T * ones is the sum)
        }
    }
}

// gradient slots (+)= the sum of a layer's partials.  grid (256, layers): block x sums row x of the 256 x 256 grid (and block 0
// also the 256 column sums)
__global__ __launch_bounds__(256) void tn_group_reduce_kernel(TnGroup g) {
    const unsigned char __attribute__((address_space(4)))* ka =
        (const unsigned char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    float* ws = *(float* const __attribute__((address_space(4)))*)(ka + __builtin_offsetof(TnGroup, ws));
    LayerPtr L = (LayerPtr)(ka + __builtin_offsetof(TnGroup, L)) + blockIdx.y;
    (void)g;
    const int abl = *(const int __attribute__((address_space(4)))*)(ka + __builtin_offsetof(TnGroup, abl));
    if (L->split == 1 && !(abl & 8)) return;                      // (its workgroups wrote the gradient slot themselves)
    const int n1 = L->n1, n2 = L->n2, nwg = L->nwg, acc = L->accumulate & 1;   // (split > 1: a single block, n1, n2 <= 256)
    const float* p0 = ws + (long long)L->ws0 * T2_WS_STRIDE;
    float* C = L->C;
    const long long ldc = L->ldc;
    // one element per thread, eight partials per trip: the partials are read once, from HBM / MALL -- a dependent chain of
    // nwg loads per element would cost nwg memory latencies
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if ((idx >> 8) < n1 && (idx & 255) < n2) {
        const float* q = p0 + idx;
        float s = 0.f;
        int w = 0;
        for (; w + 8 <= nwg; w += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = q[(long long)(w + i) * T2_WS_STRIDE];
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
        for (; w < nwg; ++w) s += q[(long long)w * T2_WS_STRIDE];
        float* d = C + (idx >> 8) * ldc + (idx & 255);
        *d = acc ? *d + s : s;
    }
    if (blockIdx.x == 0 && L->colsum != nullptr && (int)threadIdx.x < n1) {
        float s = 0.0f;
        if (L->cs_rows > 0)
            for (int w = 0; w < nwg; ++w) s += p0[(long long)w * T2_WS_STRIDE + 256 * 256 + threadIdx.x];
        float* d = L->colsum + threadIdx.x;
        *d = acc ? *d + s : s;
    }
}

}  // namespace

extern "C" {

static_assert(DHAUG_TN_GROUP_WORKSPACE_FLOATS == (long long)T2_MAX_WG * T2_WS_STRIDE, "workspace size");
static_assert(sizeof(TnGroup) <= 4096, "the group travels as a kernel argument");

/* see include/dhaug.h */
int dhaug_gemm_tn_group_bf16_phase(const dhaug_tn_layer* layers, int n, float* workspace, int phase, void* stream);

int dhaug_gemm_tn_group_bf16(const dhaug_tn_layer* layers, int n, float* workspace, void* stream) {
    return dhaug_gemm_tn_group_bf16_phase(layers, n, workspace, 0, stream);
}

int dhaug_gemm_tn_group_bf16_phase(const dhaug_tn_layer* layers, int n, float* workspace, int phase, void* stream) {
    DHAUG_CHECK(phase >= 0 && phase <= 2, DHAUG_EINVAL);
    DHAUG_CHECK(n >= 0 && n <= T2_MAX_LAYERS, DHAUG_EINVAL);
    if (n == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(layers); DHAUG_CHECK_PTR(workspace);
    DHAUG_CHECK(dhaug_aligned16(workspace), DHAUG_EALIGN);
    // the layers with the most blocks first (the order within a launch means nothing: distinct outputs): consecutive work items share
    // an XCD (kernel, abl bit 16), and a 16-block layer that starts at a multiple of 16 stays behind one L2
    dhaug_tn_layer ordered[T2_MAX_LAYERS];
    std::copy(layers, layers + n, ordered);
    auto blocks_of = [](const dhaug_tn_layer& s) { return ((s.N1 + 255) / 256) * ((s.N2 + 255) / 256); };
    std::stable_sort(ordered, ordered + n, [&](const dhaug_tn_layer& a, const dhaug_tn_layer& b) { return blocks_of(a) > blocks_of(b); });
    layers = ordered;
    TnGroup g;
    g.nlayers = n;
    g.abl = DHAUG_ABL_ENV("DHAUG_TN256_ABL");
    g.ws = workspace;
    double weight[T2_MAX_LAYERS], total = 0.0, bytes2 = 0.0;
    long long stages = 0, blocks = 0;
    int nblk[T2_MAX_LAYERS];
    bool wide = false;
    static const int floor_cols = getenv("DHAUG_TN_FLOOR") ? atoi(getenv("DHAUG_TN_FLOOR")) : 448;
    for (int i = 0; i < n; ++i) {
        const dhaug_tn_layer& s = layers[i];
        DHAUG_CHECK(s.M >= T2_ROWS && s.M % T2_ROWS == 0 && s.M / T2_ROWS < (1LL << 30), DHAUG_EUNSUPPORTED);
        DHAUG_CHECK(s.N1 >= 1 && s.N1 <= 4096 && s.N2 >= 1 && s.N2 <= 4096, DHAUG_EUNSUPPORTED);
        DHAUG_CHECK(s.colsum_rows >= 0 && s.colsum_rows <= s.M && s.colsum_rows % T2_ROWS == 0, DHAUG_EUNSUPPORTED);
        DHAUG_CHECK_PTR(s.A); DHAUG_CHECK_PTR(s.B); DHAUG_CHECK_PTR(s.C);
        DHAUG_CHECK(s.ldc >= s.N2, DHAUG_EINVAL);
        // columns are fetched in 16-byte chunks: the operand rows must be readable up to ceil8(N)
        DHAUG_CHECK(s.lda % 8 == 0 && s.ldb % 8 == 0 && s.lda >= ((s.N1 + 7) & ~7) && s.ldb >= ((s.N2 + 7) & ~7), DHAUG_EALIGN);
        DHAUG_CHECK(dhaug_aligned16(s.A) && dhaug_aligned16(s.B), DHAUG_EALIGN);
        // the layers of one launch are summed into their outputs concurrently (a block with one workgroup adds into C /
        // colsum_a with a plain read-modify-write): the outputs of a launch must be distinct (include/dhaug.h)
        for (int j = 0; j < i; ++j)
            DHAUG_CHECK(layers[j].C != s.C && (s.colsum_a == nullptr || layers[j].colsum_a != s.colsum_a), DHAUG_EINVAL);
        TnLayer& L = g.L[i];
        L.A = s.A; L.B = s.B; L.C = s.C; L.colsum = s.colsum_a;
        L.lda = s.lda; L.ldb = s.ldb; L.ldc = s.ldc; L.cs_rows = s.colsum_a != nullptr ? s.colsum_rows : 0;
        DHAUG_CHECK(s.planes_a >= 0 && s.planes_a <= 2 && s.planes_b >= 0 && s.planes_b <= 2, DHAUG_EINVAL);
        DHAUG_CHECK((s.planes_a == 0 && s.planes_b == 0) || (s.M % 6 == 0 && s.M < (1LL << 31)), DHAUG_EINVAL);   // six virtual rows per tensor row
        L.nst = (int)(s.M / T2_ROWS); L.n1 = s.N1; L.n2 = s.N2; L.accumulate = (s.accumulate ? 1 : 0) | (s.planes_a << 8) | (s.planes_b << 10);
        nblk[i] = ((s.N1 + 255) / 256) * ((s.N2 + 255) / 256);
        wide = wide || nblk[i] > 1;
        blocks += nblk[i];
        // (per 256 x 256 block of the layer)
        const int cols = ((std::min(s.N1, 256) + 7) & ~7) + ((std::min(s.N2, 256) + 7) & ~7);   // operand bytes / 2 per row
        bytes2 += (double)s.M * cols * nblk[i];
        // what a row COSTS its workgroup: its bytes at the workgroup's share of HBM (13 B/clk), but never less than the
        // stage's fixed work -- four LDS-DMA instructions per wave fill 32 KB at ~32 B/clk whatever the layer's width, plus
        // the barrier.  Measured (tools/time_tn_group.py, the 3D critic's 19 contractions, 3.16 GB): floor 0 (bytes alone) 1 027 us, 224 693, 320 649, 448 583 = 5.4 TB/s, 512+ (rows alone) 605.  Dealt by bytes alone, the few workgroups of
        // a NARROW layer (the 100 -> 1 logit layer: 128 columns, 6 144 stages over 4 workgroups) were the launch's long
        // pole: 704 us of stage floor against 356 us for a 256 x 256 layer's workgroups.
        weight[i] = (double)s.M * (cols > floor_cols ? cols : floor_cols);
        total += weight[i];
        stages += (long long)L.nst * nblk[i];
    }
    // deal the workgroups (one per CU) out in proportion to that cost: at least one, at most one per stage
    // (a short batch leaves little to read per layer: every workgroup costs a 256 KB partial result to write and to sum, so
    // the group gets about one workgroup per 768 KB of operands, at least one per layer, at most one per CU)
    static const double per_wg = getenv("DHAUG_TN_BYTES_PER_WG") ? atof(getenv("DHAUG_TN_BYTES_PER_WG")) : 768.0 * 1024.0;
    long long want = (long long)(bytes2 * 2.0 / per_wg) + 1;
    if (want < n) want = n;
    int cap = layers[0].max_workgroups > 0 && layers[0].max_workgroups < T2_MAX_WG ? layers[0].max_workgroups : T2_MAX_WG;
    if ((int)dhaug_persistent_grid(T2_MAX_WG) < cap) cap = (int)dhaug_persistent_grid(T2_MAX_WG);
    if (want > cap) want = cap;
    const int budget = (int)(stages < want ? stages : want);
    // every block of every layer one workgroup (`split` = 1: it adds its result into the gradient slot itself); then, while
    // workgroups are left, one more PER BLOCK to the layer whose slowest workgroup finishes last: its time is (stages per
    // workgroup, rounded UP) x (cost of a stage) -- the rounding matters, a layer's 6 144 stages over 17 or 18 workgroups differ
    // by a whole 6 %.  A group with more blocks than workgroups (the DenseDim-1000 layers of a video step: 16 blocks each) is
    // launched as it is: no block is split, nothing is written twice, the launch runs in waves of one workgroup per CU.
    long long used = 0;
    double stage_cost[T2_MAX_LAYERS];
    for (int i = 0; i < n; ++i) {
        g.L[i].split = 1;
        stage_cost[i] = weight[i] / g.L[i].nst;
        used += nblk[i];
    }
    DHAUG_CHECK(used <= 65535, DHAUG_EUNSUPPORTED);
    auto finish = [&](int i) { return (double)((g.L[i].nst + g.L[i].split - 1) / g.L[i].split) * stage_cost[i]; };
    while (used < budget) {
        int b = -1;
        for (int i = 0; i < n; ++i)
            if (nblk[i] == 1 && g.L[i].split < g.L[i].nst && used + 1 <= budget && (b < 0 || finish(i) > finish(b))) b = i;
        if (b < 0) break;
        ++g.L[b].split; used += nblk[b];
    }
    // (a wide layer that is split keeps one summing pass per BLOCK: not built -- its callers hand over wide layers only where
    // the group has blocks enough without splitting; a long batch of few wide layers goes block by block)
    int wg = 0, slot = 0;
    for (int i = 0; i < n; ++i) {
        DHAUG_CHECK(nblk[i] == 1 || phase == 0, DHAUG_EUNSUPPORTED);      // (a wide layer's blocks have no summing pass)
        g.L[i].wg0 = wg; g.L[i].nwg = nblk[i] * g.L[i].split; wg += g.L[i].nwg;
        g.L[i].ws0 = slot;
        if (g.L[i].split > 1 || phase != 0) slot += g.L[i].nwg;
    }
    if (phase != 0) g.abl |= 8;
    DHAUG_CHECK(slot <= T2_MAX_WG, DHAUG_EUNSUPPORTED);
    (void)blocks;
    static const bool xcd_deal = getenv("DHAUG_TN_NO_XCD_DEAL") == nullptr;
    if (wide && xcd_deal) { g.abl |= 16; wg = (wg + 7) & ~7; }     // (workgroups beyond the last layer's range return at once)
    hipStream_t s = (hipStream_t)stream;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_group_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    // phase 1: the partial results only; phase 2: only their sums into the gradient slots (the same dealing, recomputed from
    // the same descriptors); 0: both
    if (phase != 2) {
        hipLaunchKernelGGL(gemm_tn_group_kernel, dim3((unsigned)wg), dim3(512), T2_LDS, s, g);
        int rc = dhaug_launch_status();
        if (rc != DHAUG_OK || phase == 1) return rc;
    }
    // (nothing to sum where every block had one workgroup that added its result into the slot itself: a video step's wide layers)
    bool sums = phase != 0;
    for (int i = 0; i < n; ++i) sums = sums || g.L[i].split > 1;
    if (!sums) return DHAUG_OK;
    hipLaunchKernelGGL(tn_group_reduce_kernel, dim3(256, (unsigned)n), dim3(256), 0, s, g);
    return dhaug_launch_status();
}

}  // extern "C"
