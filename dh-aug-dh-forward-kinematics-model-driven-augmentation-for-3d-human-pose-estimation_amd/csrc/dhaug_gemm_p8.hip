// 256 x 256 x 64 tiles, eight waves in two groups that alternate between the matrix pipe and the memory path ("ping-pong"):
// the NT layer GEMM  C[M,N] = act(A[M,K] * B[N,K]^T + bias + residual) (* mask)  of the DenseDim-1000 layers (R/function_aug/config.py:
// 101-109: every *DenseDim* defaults to 1000; R/models_Fk_GAN/Fk_discriminator.py:149-201, 236-266, 381-587; Fk_generator.py:79-119) at
// row counts that fill the card with such tiles, and of the split-operand parity arithmetic's long 256-wide layers (K' = 3 K, 6 K).
//
// Why a layer GEMM and not a cross-layer fused unit at this width: a 1000 x 1000 layer has 500 flop per activation byte (2 M N K over
// 2 (M K + M N) bytes) -- above the card's ridge of ~310 -- so layer by layer it is bound by the matrix pipe, not by HBM (at 256 columns
// it is 128 flop per byte: that width needs the fused programs of dhaug_mlp.hip).  A unit that kept a 64-row tile's 1000-wide
// activations in LDS across layers would stream the layer's 2 MB of weights per 64 rows through the 64 B/clk vector-memory path: exactly
// the tile's matrix time (32 768 clocks each), i.e. at most half the pipe; 256-row tiles need the operand bytes a fourth as often.
//
// Structure (cdna_hip_programming.md section 5, "256^2 8-phase template"; MI355X_MICROARCH.md "Two waves per SIMD"):
//  * wave (wm, wn), wm = wave >> 2, wn = wave & 3, owns rows [128 wm, +128) x columns [64 wn, +64) of the tile: 8 x 4 tiles of
//    v_mfma_f32_16x16x32_bf16 = 128 accumulator registers; the matrix instruction is issued "swapped" (A operand = weight rows, B operand
//    = batch rows), so a lane owns one batch row and four consecutive features per accumulator quad (as in dhaug_gemm.hip).
//  * a K-tile (64 k) is four phases of 16 matrix instructions, one per QUADRANT of the wave's block (64 rows x 32 columns); a phase is
//    [fragment reads + one LDS-DMA half-tile request | barrier | 16 matrix instructions | barrier].  Waves 4-7 run ONE barrier behind
//    waves 0-3, so on every SIMD one wave is in its matrix segment while its partner reads fragments and issues copies.
//  * LDS: two K-tile buffers of four 16 KB regions: XL / XH = the first / second 64 rows of both wave groups' activation rows, WL / WH =
//    the first / second 32 weight rows of every wave column.  A region is read in ONE phase of its K-tile (XL, WL in phase 0, WH in 1, XH
//    in 2; WL's fragments stay in registers for phase 3), so it is free early: region by region the copies run up to five phases ahead
//    of their first read, inside a plain double buffer.  128-byte rows, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7): every
//    16-lane group of a ds_read_b128 covers all 64 banks; the swizzle is applied on the GLOBAL side of the copy (the LDS side of an
//    LDS-DMA is lane-linear).
//  * copies are inline assembly (global_load_lds_dwordx4, scalar base + 32-bit lane offset): hipcc's waitcnt pass does not see them, the
//    waits below are counted by hand -- a wave's vector-memory operations retire in order.  Schedule, in phases j = 4 t + q of K-tile t:
//        q = 0 requests WH(t+1), q = 1 XH(t+1), q = 2 XL(t+2), q = 3 WL(t+2)     (a region's previous content was last read >= 2 phases before)
//        q = 1 waits for XH(t) (read in q = 2): vmcnt(8);  q = 3 waits for XL, WL, WH(t+1) (read in the next two phases): vmcnt(6)
//    every wait sits in front of the phase's first barrier and the first read of what it retires is in the NEXT phase (the two wave
//    groups are one barrier apart: both have passed a barrier behind every wave's wait by then).
//  * K need not be a multiple of 64: the last K-tile's chunks beyond K are copied from 16 zero bytes (per-lane 64-bit addresses for
//    that one K-tile), so nothing is read beyond a row's K columns and the k loop has one form.
//  * epilogue: every wave transposes its block through an fp32 image of its own in LDS (no workgroup barrier), 64 rows at a time, and
//    writes whole 128-byte row pieces: bias, bf16 / fp32 residual, activation, bf16 / fp32 activation-backward mask, bf16 (+ zero pad
//    columns) and fp32 outputs -- the options of nt_store_tile.
#include "dhaug_gemm_args.h"
#include <stdlib.h>

namespace {

using namespace dhaug_gemm;

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned char __attribute__((address_space(3))) * lds_addr;

constexpr int P_BM = 256, P_BN = 256, P_BK = 64;
constexpr int P_HALF = 128 * P_BK * 2;                 // 16 384 bytes: one region
constexpr int P_BUF = 4 * P_HALF;                      // 65 536 bytes: one K-tile
constexpr int R_XL = 0, R_XH = 1, R_WL = 2, R_WH = 3;
constexpr int P_EPI_PITCH = (64 + 4) * 4;              // bytes per row of a wave's fp32 epilogue image
constexpr int P_EPI_WAVE = 64 * P_EPI_PITCH;           // 17 408
constexpr int P_LDS = 8 * P_EPI_WAVE;                  // 139 264
static_assert(P_LDS >= 2 * P_BUF, "the epilogue images reuse the K-tile buffers");

__device__ uint4 g_p8_zero16 = {0u, 0u, 0u, 0u};

// development switches (tools/build_p8_abl.sh; timing only, results wrong; a product build defines none: dhaug_common.h)
#if defined(P8_ABL_NOMMA) || defined(P8_ABL_NOREAD) || defined(P8_ABL_NOCOPY) || defined(P8_ABL_NOEPI) || defined(P8_ABL_NOSTAGGER) || \
    defined(P8_ABL_NOPRIO) || defined(P8_TIMING)
#if !defined(DHAUG_ABLATION_BUILD)
#error "a development / ablation switch is defined without -DDHAUG_ABLATION_BUILD"
#endif
#endif
#ifdef P8_TIMING
__device__ long long g_p8_stamps[2][160];
#define P8_STAMP(i) if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (i) < 160) g_p8_stamps[threadIdx.x >> 8][i] = (long long)__builtin_readcyclecounter();
#else
#define P8_STAMP(i)
#endif

__device__ __forceinline__ void p8_copy16_s(unsigned voff, const void* sbase, lds_addr lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void p8_copy16_v(const void* g, lds_addr lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_wave_base), "v"(g) : "memory");
}
__device__ __forceinline__ void p8_barrier() { asm volatile("s_barrier" ::: "memory"); }

// F16: the operands are IEEE half values (dhaug_gemm_f16x3); same tiles, same schedule, v_mfma_f32_16x16x32_f16
template <bool F16>
struct P8 {
    // per-lane byte offsets of the copies' sources, relative to the tile's first row of A / B: [region][i]
    unsigned voff[4][2];
    unsigned chunk[2];                                  // source chunk (16 bytes = 8 k) of copy i
    const uint16_t* baseA;                              // wave-uniform: A + m0 * lda, B + n0 * ldb
    const uint16_t* baseB;
    lds_addr sm;                                        // the workgroup's LDS
    const unsigned char* ax[2];                         // fragment addresses of k-step 0 / 1 (generic pointers into LDS): X rows, W rows
    const unsigned char* aw[2];
    int wave, nkt;
    long long K;
    bool tail;
    int xp_lg; unsigned xp_map; long long xp_kp;        // A as three planes of a split operand (GemmArgs): K-tile -> column offset in A
    f32x4 acc[8][4];
    bf16x8 xf[4][2], wlo[2][2], whi[2][2];

    template <int R>
    __device__ __forceinline__ void stage(int kt) {
        const lds_addr dst = sm + (kt & 1) * P_BUF + R * P_HALF + wave * 2048;
        long long koff = (long long)kt * P_BK;
        if (R < 2 && xp_lg != 0) {                      // (wave-uniform: scalar arithmetic)
            const int seg = kt >> (xp_lg - 1), within = kt & ((1 << (xp_lg - 1)) - 1);
            koff = (long long)((xp_map >> (2 * seg)) & 3u) * xp_kp + (long long)within * P_BK;
        }
        const uint16_t* base = (R < 2 ? baseA : baseB) + koff;
#ifdef P8_ABL_NOCOPY
        return;
#endif
        if (tail && kt == nkt - 1) {                    // the short last K-tile: chunks beyond K come from 16 zero bytes
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool in = (long long)kt * P_BK + chunk[i] * 8 < K;
                const unsigned char* src = in ? reinterpret_cast<const unsigned char*>(base) + voff[R][i]
                                              : reinterpret_cast<const unsigned char*>(&g_p8_zero16);
                p8_copy16_v(src, dst + i * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) p8_copy16_s(voff[R][i], base, dst + i * 1024);
        }
    }

    template <int B, int R, int NT>
    __device__ __forceinline__ void read_x(bf16x8 (&f)[4][2]) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#ifndef P8_ABL_NOREAD
                f[mt][ks] = *reinterpret_cast<const bf16x8*>(ax[ks] + B * P_BUF + R * P_HALF + mt * 2048);
#endif
#ifdef P8_ABL_NOMMA
                asm volatile("" ::"v"(f[mt][ks]));
#endif
            }
    }
    template <int B, int R>
    __device__ __forceinline__ void read_w(bf16x8 (&f)[2][2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#ifndef P8_ABL_NOREAD
                f[nt][ks] = *reinterpret_cast<const bf16x8*>(aw[ks] + B * P_BUF + R * P_HALF + nt * 2048);
#endif
#ifdef P8_ABL_NOMMA
                asm volatile("" ::"v"(f[nt][ks]));
#endif
            }
    }

    template <int MT0, int NT0>
    __device__ __forceinline__ void mma(const bf16x8 (&w)[2][2]) {
#ifdef P8_ABL_NOMMA
        return;
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    if (F16)
                        acc[MT0 + mt][NT0 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[nt][ks]), __builtin_bit_cast(f16x8, xf[mt][ks]),
                                                                                         acc[MT0 + mt][NT0 + nt], 0, 0, 0);
                    else
                        acc[MT0 + mt][NT0 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nt][ks], xf[mt][ks], acc[MT0 + mt][NT0 + nt], 0, 0, 0);
    }

    // phase Q of K-tile t (buffer B = t & 1)
    template <int Q, int B>
    __device__ __forceinline__ void phase(int t) {
        if (Q == 0) { read_w<B, R_WL>(wlo); read_x<B, R_XL, 4>(xf); }
        if (Q == 1) read_w<B, R_WH>(whi);
        if (Q == 2) read_x<B, R_XH, 4>(xf);
        if (Q == 0) { if (t + 1 < nkt) stage<R_WH>(t + 1); }
        if (Q == 1) { if (t + 1 < nkt) stage<R_XH>(t + 1); }
        if (Q == 2) { if (t + 2 < nkt) stage<R_XL>(t + 2); }
        if (Q == 3) { if (t + 2 < nkt) stage<R_WL>(t + 2); }
        if (Q == 1) {                                   // XH(t): younger are XL, WL, WH, XH of t + 1 where that K-tile exists
            if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (Q == 3 && t + 1 < nkt) {                    // XL, WL, WH(t+1): younger are XH(t+1) and XL, WL(t+2)
            if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        P8_STAMP(2 + 2 * (4 * t + Q))
        p8_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifndef P8_ABL_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        if (Q == 0) mma<0, 0>(wlo);
        if (Q == 1) mma<0, 2>(whi);
        if (Q == 2) mma<4, 2>(whi);
        if (Q == 3) mma<4, 0>(wlo);
#ifndef P8_ABL_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        P8_STAMP(2 + 2 * (4 * t + Q) + 1)
        p8_barrier();
    }
    template <int B>
    __device__ __forceinline__ void ktile(int t) {
        phase<0, B>(t); phase<1, B>(t); phase<2, B>(t); phase<3, B>(t);
    }
};

// Epilogue: the wave's 128 x 64 block through its own fp32 image in LDS (no workgroup barrier), 64 rows at a time; a lane of the read
// side owns EIGHT consecutive columns (x8) of rows r8, r8 + 8, ...: eight lanes write one 128-byte (bf16) piece of a row with one 16-byte
// store each (eight-byte stores took twice the time: what an epilogue's stores cost is their issue, cdna_hip_programming.md T21).
// Loads and stores share one in-order counter, so NO load may sit between the stores of a row group: every global operand of a group of
// rows (residual, mask) is requested first, the rows are computed, then stored; a layer without residual / mask (MODE 0) has no wait at
// all behind its stores.  (The first version loaded where it used: a full wait, i.e. a store round trip, per row group -- 21 000 clocks of
// the tile's 82 000.)  MODE 1: bf16 residual / mask; MODE 2: fp32 ones as well (the split-operand arithmetic): fewer rows per group.
template <int MODE, bool F16>
__device__ __forceinline__ void p8_epilogue(const GemmArgs& p, const f32x4 (&acc)[8][4], unsigned char* p8sm, int wave, int lane, long long m0,
                                            long long n0) {
    constexpr int CH = MODE == 0 ? 8 : (MODE == 1 ? 4 : 2);         // rows (per lane) whose operands travel together: registers
    const int wm = wave >> 2, wn = wave & 3;
    unsigned char* ew = p8sm + wave * P_EPI_WAVE;
    const int x = lane & 15, q4 = lane >> 4;                         // write side: the accumulator layout
    const int x8 = lane & 7, r8 = lane >> 3;                         // read side
    const long long ncol = n0 + 64 * wn + 8 * x8;
    const bool col_in = ncol < p.N, col_pad = !col_in && p.cb != nullptr && ncol < p.npad;
    const long long ncl = col_in ? ncol : 0;                         // (a column that exists, for the requests of lanes beyond N)
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
    if (p.bias != nullptr) { bias0 = *reinterpret_cast<const f32x4*>(p.bias + ncl); bias1 = *reinterpret_cast<const f32x4*>(p.bias + ncl + 4); }
    const float neg = p.act == DHAUG_ACT_RELU ? 0.0f : (p.act == DHAUG_ACT_LRELU ? p.slope : 1.0f);
    const bool has_res = MODE >= 1 && p.res != nullptr, has_msk = MODE >= 1 && p.dmask != nullptr, has_resf = MODE == 2 && p.resf != nullptr,
               has_mskf = MODE == 2 && p.dmaskf != nullptr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(ew + (16 * mt + x) * P_EPI_PITCH + (16 * nt + 4 * q4) * 4) = acc[4 * half + mt][nt];
#pragma unroll
        for (int c = 0; c < 8 / CH; ++c) {
            const long long mrow0 = m0 + 128 * wm + 64 * half + 8 * CH * c + r8;
            uint4 rr[CH], mm[CH];
            f32x4 rf[CH][2], mf[CH][2];
            if (MODE >= 1) {
#pragma unroll
                for (int it = 0; it < CH; ++it) {
                    rr[it] = make_uint4(0u, 0u, 0u, 0u); mm[it] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // (no mask: positive)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { rf[it][h] = f32x4{0.f, 0.f, 0.f, 0.f}; mf[it][h] = f32x4{1.f, 1.f, 1.f, 1.f}; }
                }
                long long gmc[CH];
#pragma unroll
                for (int it = 0; it < CH; ++it) { const long long gm = mrow0 + 8 * it; gmc[it] = gm < p.M ? gm : p.M - 1; }
                if (has_res) {
#pragma unroll
                    for (int it = 0; it < CH; ++it) rr[it] = *reinterpret_cast<const uint4*>(p.res + gmc[it] * p.ld_res + ncl);
                }
                if (has_msk) {
#pragma unroll
                    for (int it = 0; it < CH; ++it) mm[it] = *reinterpret_cast<const uint4*>(p.dmask + gmc[it] * p.ld_dmask + ncl);
                }
                if (has_resf) {
#pragma unroll
                    for (int it = 0; it < CH; ++it)
#pragma unroll
                        for (int h = 0; h < 2; ++h) rf[it][h] = *reinterpret_cast<const f32x4*>(p.resf + gmc[it] * p.ld_resf + ncl + 4 * h);
                }
                if (has_mskf) {
#pragma unroll
                    for (int it = 0; it < CH; ++it)
#pragma unroll
                        for (int h = 0; h < 2; ++h) mf[it][h] = *reinterpret_cast<const f32x4*>(p.dmaskf + gmc[it] * p.ld_dmaskf + ncl + 4 * h);
                }
            }
            f32x4 v[CH][2];
#pragma unroll
            for (int it = 0; it < CH; ++it) {
                const int row = 8 * CH * c + 8 * it + r8;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 t = *reinterpret_cast<const f32x4*>(ew + row * P_EPI_PITCH + x8 * 32 + 16 * h);
                    const f32x4 bb = h == 0 ? bias0 : bias1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] += bb[e];
                    if (MODE >= 1) {
                        const uint32_t w0 = h == 0 ? rr[it].x : rr[it].z, w1 = h == 0 ? rr[it].y : rr[it].w;
                        t[0] += __builtin_bit_cast(float, w0 << 16); t[1] += __builtin_bit_cast(float, w0 & 0xffff0000u);
                        t[2] += __builtin_bit_cast(float, w1 << 16); t[3] += __builtin_bit_cast(float, w1 & 0xffff0000u);
                    }
                    if (MODE == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] += rf[it][h][e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = t[e] > 0.0f ? t[e] : t[e] * neg;
                    if (MODE >= 1) {                                 // activation-backward mask of the producing layer (a positive bf16 is a positive int16)
                        const uint32_t k0 = h == 0 ? mm[it].x : mm[it].z, k1 = h == 0 ? mm[it].y : mm[it].w;
                        bool keep[4] = {(short)(k0 & 0xffffu) > 0, (short)(k0 >> 16) > 0, (short)(k1 & 0xffffu) > 0, (short)(k1 >> 16) > 0};
                        if (MODE == 2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) keep[e] = keep[e] && mf[it][h][e] > 0.0f;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = keep[e] ? t[e] : t[e] * p.dneg;
                    }
                    v[it][h] = t;
                }
            }
            // (everything loaded has been consumed: nothing below waits for the memory counter -- the values are made opaque HERE so that
            // neither the optimiser nor the scheduler sinks a row's arithmetic into its store's branch: every wait for a later row's operands
            // would then sit behind the earlier rows' stores)
#pragma unroll
            for (int it = 0; it < CH; ++it) { asm volatile("" : "+v"(v[it][0])); asm volatile("" : "+v"(v[it][1])); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < CH; ++it) {
                const long long gm = mrow0 + 8 * it;
                if (gm < p.M) {
                    if (col_in) {
                        if (p.cb != nullptr) {
                            uint4 o;
                            o.x = (uint32_t)dhaug_f32_to_bf16(v[it][0][0]) | ((uint32_t)dhaug_f32_to_bf16(v[it][0][1]) << 16);
                            o.y = (uint32_t)dhaug_f32_to_bf16(v[it][0][2]) | ((uint32_t)dhaug_f32_to_bf16(v[it][0][3]) << 16);
                            o.z = (uint32_t)dhaug_f32_to_bf16(v[it][1][0]) | ((uint32_t)dhaug_f32_to_bf16(v[it][1][1]) << 16);
                            o.w = (uint32_t)dhaug_f32_to_bf16(v[it][1][2]) | ((uint32_t)dhaug_f32_to_bf16(v[it][1][3]) << 16);
                            *reinterpret_cast<uint4*>(p.cb + gm * p.ldcb + ncol) = o;
                        }
                        if (p.cf != nullptr) {
                            *reinterpret_cast<f32x4*>(p.cf + gm * p.ldcf + ncol) = v[it][0];
                            *reinterpret_cast<f32x4*>(p.cf + gm * p.ldcf + ncol + 4) = v[it][1];
                        }
                        if (p.cp != nullptr) {                       // the pieces of the fp32 result (split_kernel's arithmetic, elem.hip)
                            uint32_t hi[4], mid[4], lo[4];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float xv = v[it][e >> 2][e & 3];
                                uint16_t h, m, l = 0;
                                if (F16) {
                                    const _Float16 hh = (_Float16)xv;
                                    const _Float16 mm2 = (_Float16)(xv - (float)hh);
                                    h = __builtin_bit_cast(uint16_t, hh); m = __builtin_bit_cast(uint16_t, mm2);
                                } else {
                                    h = dhaug_f32_to_bf16(xv);
                                    const float r1 = xv - dhaug_bf16_to_f32(h);
                                    m = dhaug_f32_to_bf16(r1);
                                    l = dhaug_f32_to_bf16(r1 - dhaug_bf16_to_f32(m));
                                }
                                if (e & 1) { hi[e >> 1] |= (uint32_t)h << 16; mid[e >> 1] |= (uint32_t)m << 16; lo[e >> 1] |= (uint32_t)l << 16; }
                                else { hi[e >> 1] = h; mid[e >> 1] = m; lo[e >> 1] = l; }
                            }
                            uint16_t* prow = p.cp + gm * p.ldcp + ncol;
                            *reinterpret_cast<uint4*>(prow) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                            *reinterpret_cast<uint4*>(prow + p.cp_kp) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
                            if (!F16) *reinterpret_cast<uint4*>(prow + 2 * p.cp_kp) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                        }
                    } else if (col_pad) {
                        *reinterpret_cast<uint4*>(p.cb + gm * p.ldcb + ncol) = make_uint4(0u, 0u, 0u, 0u);
                    }
                    if (!col_in && p.cp != nullptr && ncol < p.cp_kp) {  // the planes' pad columns: the next layer's K tail
                        uint16_t* prow = p.cp + gm * p.ldcp + ncol;
                        *reinterpret_cast<uint4*>(prow) = make_uint4(0u, 0u, 0u, 0u);
                        *reinterpret_cast<uint4*>(prow + p.cp_kp) = make_uint4(0u, 0u, 0u, 0u);
                        if (!F16) *reinterpret_cast<uint4*>(prow + 2 * p.cp_kp) = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            }
        }
    }
}

template <bool F16>
__device__ __forceinline__ void p8_body(const GemmArgs& p, long long mb, long long nb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char p8sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const long long m0 = mb * P_BM, n0 = nb * P_BN;

    P8<F16> s;
    s.wave = wave;
    s.K = p.K;
    s.nkt = (int)((p.K + P_BK - 1) / P_BK);
    s.tail = (p.K % P_BK) != 0;
    s.xp_lg = p.xp_lg; s.xp_map = p.xp_map; s.xp_kp = p.xp_kp;
    s.sm = (lds_addr)p8sm;
    // (a column tile that lies wholly in the zero-pad columns [N, n_pad) has no weight row of its own: it reads the last one and
    // stores zeros)
    const long long n0r = n0 < p.N ? n0 : p.N - 1;
    s.baseA = p.A + m0 * p.lda;
    s.baseB = p.B + n0r * p.ldb;
    {
        const long long rows_a = p.M - m0, rows_b = p.N - n0r;  // valid rows from the tile's first (>= 1); rows beyond read the last valid one
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rho = 16 * wave + 8 * i + (lane >> 3);                    // LDS row of the region
            const unsigned c = (unsigned)((lane & 7) ^ ((rho >> 1) & 7));       // source chunk
            s.chunk[i] = c;
            long long xr = 128 * (rho >> 6) + (rho & 63), wr = 64 * (rho >> 5) + (rho & 31);
            long long xl = xr, xh = xr + 64, wl = wr, wh = wr + 32;
            xl = xl < rows_a ? xl : rows_a - 1; xh = xh < rows_a ? xh : rows_a - 1;
            wl = wl < rows_b ? wl : rows_b - 1; wh = wh < rows_b ? wh : rows_b - 1;
            s.voff[R_XL][i] = (unsigned)(xl * p.lda * 2) + c * 16;
            s.voff[R_XH][i] = (unsigned)(xh * p.lda * 2) + c * 16;
            s.voff[R_WL][i] = (unsigned)(wl * p.ldb * 2) + c * 16;
            s.voff[R_WH][i] = (unsigned)(wh * p.ldb * 2) + c * 16;
        }
    }
    {
        const int x = lane & 15, q4 = lane >> 4;
        const int lo = x * 128 + ((q4 ^ (x >> 1)) << 4);
        s.ax[0] = p8sm + wm * (64 * 128) + lo; s.ax[1] = p8sm + wm * (64 * 128) + (lo ^ 64);
        s.aw[0] = p8sm + wn * (32 * 128) + lo; s.aw[1] = p8sm + wn * (32 * 128) + (lo ^ 64);
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s.acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    P8_STAMP(0)
    // prologue: K-tile 0 and the first two regions of K-tile 1 (nkt >= 2)
    s.template stage<R_XL>(0); s.template stage<R_WL>(0); s.template stage<R_WH>(0); s.template stage<R_XH>(0); s.template stage<R_XL>(1); s.template stage<R_WL>(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                // XL, WL, WH(0); XH(0) is waited for in phase 1
    p8_barrier();
#ifndef P8_ABL_NOSTAGGER
    if (wm == 1) p8_barrier();                                       // waves 4-7 run one barrier behind
#endif
    P8_STAMP(1)
    for (int t = 0; t < s.nkt; t += 2) {
        s.template ktile<0>(t);
        if (t + 1 < s.nkt) s.template ktile<1>(t + 1);
    }
#ifndef P8_ABL_NOSTAGGER
    if (wm == 0) p8_barrier();                                       // (every wave has passed the same number of barriers)
#endif
    p8_barrier();                                                    // every wave is done with the K-tile buffers
    P8_STAMP(150)
#ifdef P8_ABL_NOEPI
    return;
#endif

    if (p.resf != nullptr || p.dmaskf != nullptr) p8_epilogue<2, F16>(p, s.acc, p8sm, wave, lane, m0, n0);
    else if (p.res != nullptr || p.dmask != nullptr) p8_epilogue<1, F16>(p, s.acc, p8sm, wave, lane, m0, n0);
    else p8_epilogue<0, F16>(p, s.acc, p8sm, wave, lane, m0, n0);
    P8_STAMP(151)
}

// Workgroups go to the eight XCDs round-robin, and every XCD has its own L2: the column tiles of ONE row block are given to workgroups
// b, b + 8, b + 16 ... (same XCD, started together), so the row block's activation rows come from beyond L2 once, not once per column
// tile.  (Row blocks are padded to a multiple of eight: a workgroup beyond the batch leaves at once.)
template <bool F16>
__global__ __launch_bounds__(512, 2) void gemm_nt_p8_kernel(GemmArgs p) {
    const long long ntn = (p.W + P_BN - 1) / P_BN;
    const long long j = blockIdx.x >> 3;
    const long long mb = (j / ntn) * 8 + (blockIdx.x & 7), nb = j % ntn;
    if (mb * P_BM >= p.M) return;
    p8_body<F16>(p, mb, nb);
}

// The grouped form (members of one shape).  With 2, 4 or 8 members each member gets 8 / n XCDs of its own (an XCD's L2 then holds ONE
// member's weights: see gemm_nt_pipe2_group_kernel), and inside a member the column tiles of one row block stay on ONE of those XCDs:
// XCD j of the member takes row blocks j, j + per, ... with their column tiles one after the other.  (Dealt tile by tile over the
// member's XCDs, as the first version did, the four column tiles of a row block sat on four XCDs and each fetched the block's activation
// rows from beyond L2 for itself: 98 MB counted per launch of the video iteration's groups against 20 - 60 MB of operands.)
__global__ __launch_bounds__(512, 2) void gemm_nt_p8_group_kernel(GemmGroupArgs grp, int n, int tiles) {
    (void)grp;
    GemmArgs p;
    if (n == 2 || n == 4 || n == 8) {
        const int per = 8 / n, xcd = blockIdx.x & 7;
        const int member = xcd / per;
        load_group_member(p, member);
        const long long ntn = (p.W + P_BN - 1) / P_BN, q = blockIdx.x >> 3;
        const long long mb = (q / ntn) * per + (xcd % per), nb = q % ntn;
        if (mb * P_BM >= p.M) return;
        p8_body<false>(p, mb, nb);
    } else {
        const int member = blockIdx.x / tiles;
        const long long tile = blockIdx.x - (long long)member * tiles;
        if (member >= n) return;
        load_group_member(p, member);
        const long long ntn = (p.W + P_BN - 1) / P_BN;
        p8_body<false>(p, tile / ntn, tile % ntn);
    }
}

template <typename Kern>
int p8_configure(Kern kern) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
}

}  // namespace

bool dhaug_p8_supported(const dhaug_gemm::GemmArgs& p) {
    const long long ldmax = p.lda > p.ldb ? p.lda : p.ldb;
    const auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
    const bool planes = p.xp_lg != 0;
    // (six bf16 terms of three pieces, or -- the IEEE-half launcher -- three terms of two)
    if (planes && !(p.xp_lg >= 1 && p.xp_lg <= 8 && p.xp_kp == ((long long)P_BK << (p.xp_lg - 1)) &&
                    ((p.K == 6 * p.xp_kp && p.lda >= 3 * p.xp_kp) || (p.K == 3 * p.xp_kp && p.lda >= 2 * p.xp_kp))))
        return false;
    if (!(p.M > 0 && p.K >= 2 * P_BK && p.K % 8 == 0 && p.N % 8 == 0 && p.npad % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && (planes || p.lda >= p.K) &&
          p.ldb >= p.K && ldmax * 2 * P_BM < (1ll << 31) && p.dbits == nullptr && p.dbits2 == nullptr && al(p.A, 16) && al(p.B, 16)))
        return false;
    // the epilogue's vector accesses: eight columns per lane
    if (p.bias != nullptr && !al(p.bias, 16)) return false;
    if (p.cb != nullptr && !(p.ldcb % 8 == 0 && al(p.cb, 16))) return false;
    if (p.cf != nullptr && !(p.ldcf % 4 == 0 && al(p.cf, 16))) return false;
    if (p.res != nullptr && !(p.ld_res % 8 == 0 && al(p.res, 16))) return false;
    if (p.dmask != nullptr && !(p.ld_dmask % 8 == 0 && al(p.dmask, 16))) return false;
    if (p.resf != nullptr && !(p.ld_resf % 4 == 0 && al(p.resf, 16))) return false;
    if (p.dmaskf != nullptr && !(p.ld_dmaskf % 4 == 0 && al(p.dmaskf, 16))) return false;
    if (p.cp != nullptr && !(p.cf != nullptr && p.ldcp % 8 == 0 && p.cp_kp % 8 == 0 && p.cp_kp >= p.N && p.ldcp >= 2 * p.cp_kp && al(p.cp, 16))) return false;
    return true;
}

#ifdef P8_TIMING
extern "C" int dhaug_debug_p8_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p8_stamps), sizeof(long long) * (n < 320 ? n : 320));
}
#endif

template <bool F16>
static int p8_launch_single(hipStream_t s, const dhaug_gemm::GemmArgs& p) {
    static bool configured = false;
    if (!configured) {
        const int e = p8_configure(gemm_nt_p8_kernel<F16>);
        if (e != 0) return e;
        configured = true;
    }
    const long long grid = (((p.M + P_BM - 1) / P_BM + 7) / 8 * 8) * ((p.W + P_BN - 1) / P_BN);
    DHAUG_CHECK(grid <= 0x7fffffffLL, DHAUG_EUNSUPPORTED);
    hipLaunchKernelGGL(gemm_nt_p8_kernel<F16>, dim3((unsigned)grid), dim3(512), P_LDS, s, p);
    return dhaug_launch_status();
}
int dhaug_p8_launch(hipStream_t s, const dhaug_gemm::GemmArgs& p) { return p8_launch_single<false>(s, p); }
int dhaug_p8_launch_f16(hipStream_t s, const dhaug_gemm::GemmArgs& p) { return p8_launch_single<true>(s, p); }

int dhaug_p8_launch_group(hipStream_t s, const dhaug_gemm::GemmGroupArgs& g, int n) {
    static bool configured = false;
    if (!configured) {
        const int e = p8_configure(gemm_nt_p8_group_kernel);
        if (e != 0) return e;
        configured = true;
    }
    const dhaug_gemm::GemmArgs& p = g.g[0];
    const long long tiles = ((p.M + P_BM - 1) / P_BM) * ((p.W + P_BN - 1) / P_BN);
    long long grid = tiles * n;
    if (n == 2 || n == 4 || n == 8) {
        const long long per = 8 / n, mblocks = (p.M + P_BM - 1) / P_BM, ntn = (p.W + P_BN - 1) / P_BN;
        grid = ((mblocks + per - 1) / per) * ntn * 8;                 // (row blocks of a member dealt over its XCDs: see the kernel)
    }
    DHAUG_CHECK(grid <= 0x7fffffffLL && tiles <= 0x7fffffffLL, DHAUG_EUNSUPPORTED);
    hipLaunchKernelGGL(gemm_nt_p8_group_kernel, dim3((unsigned)grid), dim3(512), P_LDS, s, g, n, (int)tiles);
    return dhaug_launch_status();
}
