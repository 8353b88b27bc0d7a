// bf16 MFMA GEMMs for the generator / critic dense layers (gfx950: v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   gemm_nt : C[M,N] = act(A[M,K] * B[N,K]^T + bias + residual)     forward layers and input gradients
//             (nn.Linear weight layout [out,in]: both operands contraction-contiguous)
//   gemm_tn : C[N1,N2] += A[M,N1]^T * B[M,N2]                        weight gradients (contraction over the batch)
//
// Orientation.  The MFMA is issued "swapped": its A operand is the weight fragment (rows = output
// features n) and its B operand the activation fragment (columns = batch rows m), so a lane of the
// accumulator owns one batch row and 4 consecutive features per register quad.  That makes the epilogue's LDS
// write a conflict-free ds_write_b128 and keeps the layout that a fused multi-layer kernel can feed straight
// back as the next MFMA's B operand (cdna_hip_programming.md section 3, accumulator-as-operand).
//
// LDS images.  NT tiles are [rows][64 k] bf16 (128-B rows); the 16-byte chunk index is XOR-swizzled with
// (row >> 1) & 7 so that every ds_read_b128 lane group touches all 64 banks exactly once (two rows share a
// 256-B bank row; eight row pairs take eight different chunks).  TN tiles keep the global row-major layout
// [64 m][160] (320-B rows: consecutive rows shift by 16 banks) and are read with ds_read_b64_tr_b16, the
// hardware transpose read, because the contraction index m is the slow axis in memory.
#include <cstdlib>
#include "dhaug_common.h"
#include "dhaug_gemm_args.h"
#include <stdlib.h>

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

using namespace dhaug_gemm;        // GemmArgs, GemmGroupArgs, apply_act, ... (dhaug_gemm_args.h)

constexpr int BK = 64;

// coalesced epilogue of an fp32 C tile staged in LDS ([BM][BN + 4]): bias, residual, activation, bf16 / fp32 stores.
// A thread keeps ONE 8-column piece and walks the tile's rows (256 threads cover 256 / (BN / 8) rows per step): the bias of
// its columns is loaded once, and the residual / mask values of RB rows are requested together before the first of them is
// used.  (The first version re-loaded eight bias scalars and waited for its row's residual in every iteration: 16 dependent
// memory round trips per thread for a 128 x 256 tile -- 36 us of the big-tile kernel's 87 us at 13 824 x 1000 x 1000.)
template <int BM, int BN, int NT = 256>
__device__ __forceinline__ void nt_store_tile(const GemmArgs& p, const float* sC, long long m0, long long n0, int tid) {
    constexpr int CS = BN + 4;
    constexpr int PPR = BN / 8;                                  // pieces per row
    constexpr int RSTEP = NT / PPR;                              // rows per step of the workgroup (NT threads)
    constexpr int NIT = BM / RSTEP;                              // rows per thread
    constexpr int RB = NIT >= 4 ? 4 : NIT;                       // rows whose global operands travel together
    static_assert(NT % PPR == 0 && BM % RSTEP == 0 && NIT % RB == 0, "tile shape");
    const int pc = tid % PPR, r0 = tid / PPR;
    const long long n = n0 + pc * 8;
    const bool any_out = n < p.N || (p.cb != nullptr && n < p.npad);
    if (!any_out) return;
    const bool full = n + 8 <= p.N;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.0f;
    if (p.bias != nullptr) {
        if (full && (reinterpret_cast<uintptr_t>(p.bias + n) & 15) == 0) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bias[e] = b0[e]; bias[4 + e] = b1[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n + e < p.N) bias[e] = p.bias[n + e];
        }
    }
    const bool res_vec = p.res != nullptr && full && (p.ld_res & 7) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0;
    const bool msk_vec = p.dmask != nullptr && full && (p.ld_dmask & 7) == 0 && (reinterpret_cast<uintptr_t>(p.dmask) & 15) == 0;
    const bool rf_vec = p.resf != nullptr && full && (p.ld_resf & 3) == 0 && (reinterpret_cast<uintptr_t>(p.resf) & 15) == 0;
    for (int it0 = 0; it0 < NIT; it0 += RB) {
        uint4 rr[RB], mm[RB];
        f32x4 rf0[RB], rf1[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {                           // all global operands of RB rows first
            const long long gm = m0 + r0 + (it0 + j) * RSTEP;
            rr[j] = make_uint4(0, 0, 0, 0); mm[j] = make_uint4(0, 0, 0, 0);
            rf0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; rf1[j] = rf0[j];
            if (gm < p.M) {
                if (res_vec) rr[j] = *reinterpret_cast<const uint4*>(p.res + gm * p.ld_res + n);
                if (msk_vec) mm[j] = *reinterpret_cast<const uint4*>(p.dmask + gm * p.ld_dmask + n);
                if (rf_vec) { rf0[j] = *reinterpret_cast<const f32x4*>(p.resf + gm * p.ld_resf + n); rf1[j] = *reinterpret_cast<const f32x4*>(p.resf + gm * p.ld_resf + n + 4); }
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int row = r0 + (it0 + j) * RSTEP;
            const long long gm = m0 + row;
            if (gm >= p.M) continue;
            float v[8];
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(sC + row * CS + pc * 8);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(sC + row * CS + pc * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = c0[e] + bias[e]; v[4 + e] = c1[e] + bias[4 + e]; }
            if (p.resf != nullptr) {
                if (rf_vec) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += rf0[j][e]; v[4 + e] += rf1[j][e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += p.resf[gm * p.ld_resf + n + e];
                }
            }
            if (p.res != nullptr) {
                if (res_vec) {
                    const uint32_t w[4] = {rr[j].x, rr[j].y, rr[j].z, rr[j].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __builtin_bit_cast(float, w[e] << 16);
                        v[2 * e + 1] += __builtin_bit_cast(float, w[e] & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += dhaug_bf16_to_f32(p.res[gm * p.ld_res + n + e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (full || n + e < p.N) ? apply_act(v[e], p.act, p.slope) : 0.0f;
            if (p.dmask != nullptr) {                            // activation-backward mask of the producing layer
                if (msk_vec) {
                    const uint32_t w[4] = {mm[j].x, mm[j].y, mm[j].z, mm[j].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                // a positive bf16 is a positive int16
                        v[2 * e] = (short)(w[e] & 0xffffu) > 0 ? v[2 * e] : v[2 * e] * p.dneg;
                        v[2 * e + 1] = (short)(w[e] >> 16) > 0 ? v[2 * e + 1] : v[2 * e + 1] * p.dneg;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (full || n + e < p.N) v[e] = (short)p.dmask[gm * p.ld_dmask + n + e] > 0 ? v[e] : v[e] * p.dneg;
                }
            }
            if (p.dmaskf != nullptr) {
                const float* mrow = p.dmaskf + gm * p.ld_dmaskf + n;
                if (full && (p.ld_dmaskf & 3) == 0 && (reinterpret_cast<uintptr_t>(p.dmaskf) & 15) == 0) {
                    const f32x4 m0v = *reinterpret_cast<const f32x4*>(mrow), m1v = *reinterpret_cast<const f32x4*>(mrow + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = m0v[e] > 0.0f ? v[e] : v[e] * p.dneg;
                        v[4 + e] = m1v[e] > 0.0f ? v[4 + e] : v[4 + e] * p.dneg;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < p.N) v[e] = mrow[e] > 0.0f ? v[e] : v[e] * p.dneg;
                }
            }
            if (p.cb != nullptr) {
                if (n + 8 <= p.npad || full) {
                    uint4 o;
                    o.x = (uint32_t)dhaug_f32_to_bf16(v[0]) | ((uint32_t)dhaug_f32_to_bf16(v[1]) << 16);
                    o.y = (uint32_t)dhaug_f32_to_bf16(v[2]) | ((uint32_t)dhaug_f32_to_bf16(v[3]) << 16);
                    o.z = (uint32_t)dhaug_f32_to_bf16(v[4]) | ((uint32_t)dhaug_f32_to_bf16(v[5]) << 16);
                    o.w = (uint32_t)dhaug_f32_to_bf16(v[6]) | ((uint32_t)dhaug_f32_to_bf16(v[7]) << 16);
                    *reinterpret_cast<uint4*>(p.cb + gm * p.ldcb + n) = o;
                } else {
                    const long long lim = p.npad > p.N ? p.npad : p.N;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < lim) p.cb[gm * p.ldcb + n + e] = dhaug_f32_to_bf16(v[e]);
                }
            }
            if (p.cf != nullptr) {
                if (full && (p.ldcf & 3) == 0) {
                    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(p.cf + gm * p.ldcf + n) = o0;
                    *reinterpret_cast<f32x4*>(p.cf + gm * p.ldcf + n + 4) = o1;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < p.N) p.cf[gm * p.ldcf + n + e] = v[e];
                }
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_CH = BM * 8 / 256, B_CH = (BN * 8 + 255) / 256;
    constexpr int CS = BN + 4;                                  // fp32 C-tile row stride (16-byte multiple, 4 mod 32)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t* sA = reinterpret_cast<uint16_t*>(smem_raw);      // [2][BM*BK]
    uint16_t* sB = sA + 2 * BM * BK;                            // [2][BN*BK]
    float* sC = reinterpret_cast<float*>(smem_raw);             // [BM][CS], reuses the staging buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const long long ntn = (p.W + BN - 1) / BN;
    const long long m0 = (long long)(blockIdx.x / ntn) * BM;
    const long long n0 = (long long)(blockIdx.x % ntn) * BN;

    uint4 ra[A_CH], rb[B_CH];
    auto gload = [&](int kt) {
        const long long k0 = (long long)kt * BK;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int q = tid + 256 * i, row = q >> 3, c = q & 7;
            const long long gm = m0 + row, kk = k0 + c * 8;
            ra[i] = (gm < p.M && kk < p.K) ? *reinterpret_cast<const uint4*>(p.A + gm * p.lda + kk) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int q = tid + 256 * i, row = q >> 3, c = q & 7;
            const long long gn = n0 + row, kk = k0 + c * 8;
            rb[i] = (row < BN && gn < p.N && kk < p.K) ? *reinterpret_cast<const uint4*>(p.B + gn * p.ldb + kk)
                                                       : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int q = tid + 256 * i, row = q >> 3, c = q & 7;
            *reinterpret_cast<uint4*>(sA + buf * BM * BK + row * BK + ((c ^ ((row >> 1) & 7)) << 3)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int q = tid + 256 * i, row = q >> 3, c = q & 7;
            if (row < BN) *reinterpret_cast<uint4*>(sB + buf * BN * BK + row * BK + ((c ^ ((row >> 1) & 7)) << 3)) = rb[i];
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nkt = (int)((p.K + BK - 1) / BK);
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(kt + 1);
        const uint16_t* bufA = sA + (kt & 1) * BM * BK;
        const uint16_t* bufB = sB + (kt & 1) * BN * BK;
        const int ksteps = (int)(((p.K - (long long)kt * BK) < BK ? (p.K - (long long)kt * BK) : BK) >> 4);
        for (int ks = 0; ks < ksteps; ++ks) {
            const int chunk = 2 * ks + (lane >> 5);
            bf16x8 fw[TN], fx[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int row = wn * WN + 32 * i + (lane & 31);
                fw[i] = *reinterpret_cast<const bf16x8*>(bufB + row * BK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = wm * WM + 32 * j + (lane & 31);
                fx[j] = *reinterpret_cast<const bf16x8*>(bufA + row * BK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) lstore((kt + 1) & 1);
        __syncthreads();
    }

    // accumulators -> fp32 C tile in LDS.  D[n][m]: lane owns m = lane&31, register quad g owns
    // n = 8g + 4*(lane>>5) .. +3
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = wm * WM + 32 * j + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = wn * WN + 32 * i + 8 * g + 4 * (lane >> 5);
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(sC + m * CS + n) = v;
            }
        }
    __syncthreads();

    nt_store_tile<BM, BN>(p, sC, m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------------
// The same product for shapes the specialised kernels do not cover (DenseDim 1000 layers of the video configuration,
// concatenation layers, 100-wide blocks), with the global loads FOUR K-stages ahead.  gemm_nt_kernel above keeps one
// stage in flight and drains it at every __syncthreads(): 1.8 us per 64-wide K step whatever the tile does (29.7 us for
// a 512 x 1000 x 1000 layer, rocprof r02_video).  Here: 64 x 64 tiles (a 512-row layer still gives 128 workgroups),
// four register stages requested up front and refilled right after their LDS write, barriers that order LDS traffic
// only (vmcnt keeps counting across them), several workgroups per CU.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void p_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void p_copy16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// Operand stages travel global -> LDS without registers (global_load_lds_dwordx4, counted in vmcnt): register-staged
// loads behind control flow make hipcc wait with small vmcnt values right behind the requests (seen in the ISA of a
// first version), which serialises the stages again.  A stage is [64 rows][64 k] bf16 per operand, rows contiguous (the
// copy fixes a lane's LDS slot), 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) (applied on the global side).
// Rows beyond M / N read a valid row (their results are never stored); the K tail is cut in the k-step loop.
// *(r4, measured with `tools/time_nt_graph.py`: fifty launches replayed as one hipGraph -- the host issues a C-ABI call in ~10 us, a
// timing loop of these launches measures the host)*  1 536 x 1000 x 1000 (the motion critics' branch layers, 435 launches per video
// iteration): 13.3 us; without the epilogue 10.5, without fragment reads / matrix instructions 9.8, without copies 10.2, copies
// alone 7.1 -- the phases of a stage do not overlap inside a workgroup (wait for the stage, barrier, issue the copy three stages on,
// compute), and with 1.5 workgroups per CU little overlaps across them.  The template's other instantiation, 64 x 128 tiles with six
// stages (one workgroup per CU, all 192 resident at once), is SLOWER (17.5 us) and is not dispatched; requesting a stage's eight
// fragment reads before its four matrix instructions and alternating two accumulator sets (the whole-stage path below) bought 3 %.
#ifdef DHAUG_PIPE_TIMING
__device__ long long g_pipe_stamps[256];
#define PIPE_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 256) g_pipe_stamps[i] = (long long)__builtin_readcyclecounter();
#else
#define PIPE_STAMP(i)
#endif
template <int BN, int NSTG>
__global__ __launch_bounds__(256, BN == 64 ? 2 : 1) void gemm_nt_pipe_kernel(GemmArgs p) {
    constexpr int BM = 64;
    constexpr int TN = BN / 64;                                              // 32-column MFMA tiles per wave
    constexpr int CS = BN + 4;
    constexpr int STG = (BM + BN) * BK * 2;                                  // bytes per stage: 16 384 / 24 576
    constexpr int NCP = (BM + BN) / 32;                                      // copies per lane and stage: 4 / 6
    static_assert(NSTG * STG >= BM * CS * 4, "C tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sC = reinterpret_cast<float*>(smem_raw);             // [BM][CS], reuses the staging buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                    // 2 x 2 waves: rows [32 wm, +32), columns [BN / 2 wn, + BN / 2)
    const long long ntn = (p.W + BN - 1) / BN;
    const long long m0 = (long long)(blockIdx.x / ntn) * BM;
    const long long n0 = (long long)(blockIdx.x % ntn) * BN;
    const int nkt = (int)((p.K + BK - 1) / BK);

    // copy i of wave w moves rows [8 (4 i + w), +8) of the stage image (A rows first, then B rows): 8 lanes per row
    const uint16_t* pg[NCP];
    int rowoff[NCP];
#pragma unroll
    for (int i = 0; i < NCP; ++i) {
        const int row0 = (4 * i + wave) * 8, row = row0 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        if (i < BM / 32) {
            const long long gm = m0 + row;
            pg[i] = p.A + (gm < p.M ? gm : p.M - 1) * p.lda + c * 8;
        } else {
            const long long gn = n0 + row - BM;
            pg[i] = p.B + (gn < p.N ? gn : p.N - 1) * p.ldb + c * 8;
        }
        rowoff[i] = row0 * (BK * 2);
    }
    auto copy_stage = [&](int kt) {
        unsigned char* base = smem_raw + (kt % NSTG) * STG;
        long long k0 = (long long)kt * BK;
        if (k0 + BK > p.K) k0 = p.K - BK > 0 ? p.K - BK : 0;   // short last stage: re-read the last full window (K >= 64) ...
#pragma unroll                                                 // ... and skip its first columns in the k-step loop
        for (int i = 0; i < NCP; ++i) p_copy16(pg[i] + k0, base + rowoff[i]);
    };
    f32x16 acc[TN], acc2[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[u][r] = 0.0f; acc2[u][r] = 0.0f; }
#pragma unroll
    for (int s = 0; s < NSTG - 1; ++s)
        if (s < nkt && !(p.abl & 4)) copy_stage(s);
    const int rowa = wm * 32 + (lane & 31);
    PIPE_STAMP(0)
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt must have landed; younger: stages kt+1 .. kt+NSTG-2 (NCP copies each) where they exist
        const int younger = nkt - 1 - kt;
        PIPE_STAMP(4 + 4 * kt)
        if (NSTG >= 6 && younger >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NCP) : "memory");
        else if (NSTG >= 5 && younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NCP) : "memory");
        else if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCP) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIPE_STAMP(5 + 4 * kt)
        p_lds_barrier();                                        // everybody's copies of stage kt are in LDS, stage kt-1 is released
        PIPE_STAMP(6 + 4 * kt)
        if (kt + NSTG - 1 < nkt && !(p.abl & 4)) copy_stage(kt + NSTG - 1);
        PIPE_STAMP(7 + 4 * kt)
        if (p.abl & 2) continue;
        const unsigned char* bufA = smem_raw + (kt % NSTG) * STG;
        const unsigned char* bufB = bufA + BM * BK * 2;
        // k-steps of this stage: a short last stage was loaded as the last full 64-wide window, its first columns belong
        // to the previous stage
        const long long kbeg = (long long)kt * BK;
        int ks0 = 0;
        if (kbeg + BK > p.K && p.K >= BK) ks0 = (int)((kbeg - (p.K - BK)) >> 4);
        const int ks1 = p.K >= BK ? 4 : (int)(p.K >> 4);
        if (ks0 == 0 && ks1 == 4) {
            // a whole stage (all but a K tail): every fragment read is requested before the first matrix instruction, and the
            // k-steps alternate between two accumulator sets.  The loop below, with its run-time bounds, is compiled as read ->
            // wait -> MFMA per k-step on one accumulator: four LDS latencies plus four dependent matrix instructions per stage
            // -- *(measured, hipGraph replay, 512 x 1000 x 1000: one workgroup per CU)* 9.2 us of a 10.1 us launch with the copies
            // switched off.
            bf16x8 fx[4], fw[4][TN];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int chunk = 2 * ks + (lane >> 5);
                fx[ks] = *reinterpret_cast<const bf16x8*>(bufA + rowa * (BK * 2) + ((chunk ^ ((rowa >> 1) & 7)) << 4));
#pragma unroll
                for (int u = 0; u < TN; ++u) {
                    const int rowb = wn * (BN / 2) + 32 * u + (lane & 31);
                    fw[ks][u] = *reinterpret_cast<const bf16x8*>(bufB + rowb * (BK * 2) + ((chunk ^ ((rowb >> 1) & 7)) << 4));
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // (left alone the scheduler sinks every read next to its MFMA,
                                                                //  with a full lgkmcnt(0) in front of each)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int u = 0; u < TN; ++u) {
                    if (ks & 1) acc2[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][u], fx[ks], acc2[u], 0, 0, 0);
                    else acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][u], fx[ks], acc[u], 0, 0, 0);
                }
            continue;
        }
        for (int ks = ks0; ks < ks1; ++ks) {
            const int chunk = 2 * ks + (lane >> 5);
            const bf16x8 fx = *reinterpret_cast<const bf16x8*>(bufA + rowa * (BK * 2) + ((chunk ^ ((rowa >> 1) & 7)) << 4));
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                const int rowb = wn * (BN / 2) + 32 * u + (lane & 31);
                const bf16x8 fw = *reinterpret_cast<const bf16x8*>(bufB + rowb * (BK * 2) + ((chunk ^ ((rowb >> 1) & 7)) << 4));
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx, acc[u], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] += acc2[u][r];
    PIPE_STAMP(1)
    if (p.abl & 1) return;
    p_lds_barrier();                                            // the staging buffers become the C tile
    {
        const int m = wm * 32 + (lane & 31);
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = wn * (BN / 2) + 32 * u + 8 * g + 4 * (lane >> 5);
                f32x4 v = {acc[u][4 * g], acc[u][4 * g + 1], acc[u][4 * g + 2], acc[u][4 * g + 3]};
                *reinterpret_cast<f32x4*>(sC + m * CS + n) = v;
            }
    }
    p_lds_barrier();
    PIPE_STAMP(2)
    nt_store_tile<BM, BN>(p, sC, m0, n0, tid);
    PIPE_STAMP(3)
}

// *(r4)* The same 64 x 64 tiles with the fragment reads taken out of the stage's critical path.  Phase stamps of the kernel above
// (`tools/stamp_pipe.py`, one workgroup per CU): a stage is wait + barrier, ~300 clocks of copy issue, and ~600 of "read eight
// fragments, wait for them, four matrix instructions" -- nothing of it overlaps.  Here the fragments of stage kt + 1 are requested
// right behind the barrier that publishes them and travel under the matrix instructions of stage kt (two fragment sets, the k loop
// unrolled by two); a stage's buffer is free as soon as its fragments are in registers, so the copy issued behind the barrier is
// stage kt + 4's: FOUR stages buffered or in flight with the same four buffers.
__device__ __forceinline__ void nt_pipe2_body(const GemmArgs& p, long long tile) {
    constexpr int BM = 64, BN = 64, NSTG = 4, NCP = 4;
    constexpr int CS = BN + 4;
    constexpr int STG = (BM + BN) * BK * 2;                                  // 16 384 bytes per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sC = reinterpret_cast<float*>(smem_raw);             // [BM][CS], reuses the staging buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                    // 2 x 2 waves, one 32 x 32 MFMA tile each
    const long long ntn = (p.W + BN - 1) / BN;
    const long long m0 = (tile / ntn) * BM;
    const long long n0 = (tile % ntn) * BN;
    const int nkt = (int)((p.K + BK - 1) / BK);
    const uint16_t* pg[NCP];
    int rowoff[NCP];
#pragma unroll
    for (int i = 0; i < NCP; ++i) {
        const int row0 = (4 * i + wave) * 8, row = row0 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        if (i < BM / 32) {
            const long long gm = m0 + row;
            pg[i] = p.A + (gm < p.M ? gm : p.M - 1) * p.lda + c * 8;
        } else {
            const long long gn = n0 + row - BM;
            pg[i] = p.B + (gn < p.N ? gn : p.N - 1) * p.ldb + c * 8;
        }
        rowoff[i] = row0 * (BK * 2);
    }
    auto copy_stage = [&](int kt) {
        unsigned char* base = smem_raw + (kt % NSTG) * STG;
        long long k0 = (long long)kt * BK;
        if (k0 + BK > p.K) k0 = p.K - BK > 0 ? p.K - BK : 0;   // short last stage: the last full window (K >= 64)
#pragma unroll
        for (int i = 0; i < NCP; ++i) p_copy16(pg[i] + k0, base + rowoff[i]);
    };
    const int rowa = wm * 32 + (lane & 31), rowb = wn * 32 + (lane & 31), hh = lane >> 5;
    struct Frags { bf16x8 x[4], w[4]; };
    auto read_frags = [&](int kt, Frags& f) {
        const unsigned char* bufA = smem_raw + (kt % NSTG) * STG;
        const unsigned char* bufB = bufA + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = 2 * ks + hh;
            f.x[ks] = *reinterpret_cast<const bf16x8*>(bufA + rowa * (BK * 2) + ((chunk ^ ((rowa >> 1) & 7)) << 4));
            f.w[ks] = *reinterpret_cast<const bf16x8*>(bufB + rowb * (BK * 2) + ((chunk ^ ((rowb >> 1) & 7)) << 4));
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NSTG; ++s)
        if (s < nkt) copy_stage(s);
    {
        const int younger = nkt - 1 < 3 ? nkt - 1 : 3;           // stages 1 .. 3 may still fly
        if (younger == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p_lds_barrier();
    }
    Frags f0, f1;
    read_frags(0, f0);
    auto step = [&](int kt, const Frags& cur, Frags& nxt) {
        if (kt + 1 < nkt) {
            // stage kt + 1 must have landed; issued so far: stages up to min(kt + 3, nkt - 1)
            const int younger = nkt - 2 - kt < 2 ? nkt - 2 - kt : 2;
            if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            p_lds_barrier();                                    // stage kt + 1 is in LDS for everybody; everybody holds stage kt's fragments
            if (kt + NSTG < nkt) copy_stage(kt + NSTG);         // (into stage kt's buffer)
            read_frags(kt + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long long kbeg = (long long)kt * BK;
        int ks0 = 0;
        if (kbeg + BK > p.K && p.K >= BK) ks0 = (int)((kbeg - (p.K - BK)) >> 4);
        const int ks1 = p.K >= BK ? 4 : (int)(p.K >> 4);
        if (ks0 == 0 && ks1 == 4) {
            // (k ascending into ONE accumulator -- back-to-back matrix instructions on the same accumulator need no wait states on
            // this part; round 4 alternated two sets for 3 %.  One set is what lets the grouped launch's 128 x 128 tiles give a wave
            // 64 x 64 outputs within its registers and stay bit-identical to this kernel: gemm_nt_g128_group_kernel)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.w[0], cur.x[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.w[1], cur.x[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.w[2], cur.x[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.w[3], cur.x[3], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (ks >= ks0 && ks < ks1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.w[ks], cur.x[ks], acc, 0, 0, 0);
        }
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        step(kt, f0, f1);
        if (kt + 1 < nkt) step(kt + 1, f1, f0);
    }
    p_lds_barrier();                                            // the staging buffers become the C tile
    {
        const int m = wm * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = wn * 32 + 8 * g + 4 * (lane >> 5);
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            *reinterpret_cast<f32x4*>(sC + m * CS + n) = v;
        }
    }
    p_lds_barrier();
    nt_store_tile<BM, BN>(p, sC, m0, n0, tid);
}
__global__ __launch_bounds__(256, 2) void gemm_nt_pipe2_kernel(GemmArgs p) { nt_pipe2_body(p, blockIdx.x); }

// Up to eight independent GEMMs of ONE shape as one launch (blockIdx.y = member): the layers of a motion critic's four / two branches
// at the same depth (dhaug_gemm_bf16_group).  A 1 536 x 1000 x 1000 layer is 384 of these tiles -- one and a half waves of the card's
// 512 workgroup slots, behind a launch of its own; four of them are three full waves behind ONE launch.  The member's arguments are
// read from the kernarg segment with scalar loads (a by-value array indexed dynamically would be copied to scratch).
// Which member a workgroup takes: workgroups go to the eight XCDs round-robin and every XCD has its own 4 MB L2, so with 2, 4 or 8
// members each member gets XCDs of its own (member = XCD * n / 8): an XCD's L2 then holds ONE member's weights (2 MB at DenseDim
// 1000) beside the activation rows in flight.  *(measured, four 1 536 x 1000 x 1000 members)* dealt member by member the launch takes
// 37.9 us -- 393 MB of stage fills at 10.4 TB/s, the rate of the Infinity Cache: four weight matrices thrash every L2 -- against
// 12.7 us for one member alone.
__global__ __launch_bounds__(256, 2) void gemm_nt_pipe2_group_kernel(GemmGroupArgs grp, int n, int tiles) {
    (void)grp;
    int member;
    long long tile;
    if (n == 2 || n == 4 || n == 8) {
        const int per = 8 / n, xcd = blockIdx.x & 7;
        member = xcd / per;
        tile = (long long)(blockIdx.x >> 3) * per + (xcd % per);
    } else {
        member = blockIdx.x / tiles;
        tile = blockIdx.x - member * tiles;
    }
    if (tile >= tiles || member >= n) return;
    GemmArgs p;
    load_group_member(p, member);
    nt_pipe2_body(p, tile);
}

// *(r5)* The grouped launch on 128 x 128 tiles (`gemm_nt_g128_group_kernel`).  With 64 x 64 tiles the four branch layers of a motion
// critic are 1 536 workgroups -- three rounds of the card's 512 slots of this size, every slot of the card taken while the other
// three critics' streams wait (measured in round 4: 35 us alone against 4 x 12.7, but the video iteration SLOWER, 17.2 against
// 15.4 ms) -- and stage 393 MB.  A 128 x 128 tile stages half the bytes per output and the four layers are 384 workgroups: ONE
// round at two workgroups per CU (four 16 KB stages of 32 k: 64 KB), a quarter of the card's slot-time, and the stream's chain is one
// launch per depth instead of four.  Eight waves of 64 x 32 (four per SIMD, 128 registers; other forms: see the body); the stage mechanics of gemm_nt_pipe2_kernel (global -> LDS without registers, exact vmcnt waits, LDS-only
// barriers, the next stage's fragments under this stage's matrix instructions) with 64-byte stage rows, chunk c of row r at
// c ^ ((r >> 2) & 3) as in the 256 x 256-tile kernel; the epilogue through the fp32 C image in two passes of 64 rows.  The k-steps are
// summed in gemm_nt_pipe2_kernel's order (k ascending, one accumulator): the result is bit-identical to one launch per layer on that
// kernel, whatever the tiling (tests/test_gpu_loops.py, grouped against not; tests/test_gpu_kernels.py).
constexpr int Q_BM = 128, Q_BN = 128, Q_BK = 32, Q_NSTG = 4;
constexpr int Q_STG = (Q_BM + Q_BN) * Q_BK * 2;                              // 16 384 bytes per stage
constexpr int Q_LDS = Q_NSTG * Q_STG;                                        // 65 536: two workgroups per CU
constexpr int Q_CS = Q_BN + 4;
static_assert(64 * Q_CS * 4 <= Q_LDS, "C image of one pass");
__device__ __forceinline__ void nt_g128_body(const GemmArgs& p, long long tile) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sC = reinterpret_cast<float*>(smem_raw);             // [64][Q_CS], reuses the staging buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                    // 2 x 4 waves: rows [64 wm, +64), columns [32 wn, +32)
    const long long ntn = (p.W + Q_BN - 1) / Q_BN;
    const long long m0 = (tile / ntn) * Q_BM;
    const long long n0 = (tile % ntn) * Q_BN;
    const int nkt = (int)((p.K + Q_BK - 1) / Q_BK);
    constexpr int NCP = 2;                                      // copies per lane and stage: rows 16 (8 i + wave) .. + 16 of the image
    const uint16_t* pg[NCP];
    int rowoff[NCP];
#pragma unroll
    for (int i = 0; i < NCP; ++i) {
        const int row0 = (8 * i + wave) * 16, row = row0 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        if (i < 1) {
            const long long gm = m0 + row;
            pg[i] = p.A + (gm < p.M ? gm : p.M - 1) * p.lda + c * 8;
        } else {
            const long long gn = n0 + row - Q_BM;
            pg[i] = p.B + (gn < p.N ? gn : p.N - 1) * p.ldb + c * 8;
        }
        rowoff[i] = row0 * (Q_BK * 2);
    }
    auto copy_stage = [&](int kt) {
        unsigned char* base = smem_raw + (kt % Q_NSTG) * Q_STG;
        long long k0 = (long long)kt * Q_BK;
        if (k0 + Q_BK > p.K) k0 = p.K - Q_BK > 0 ? p.K - Q_BK : 0;       // short last stage: the last full window (K >= 32)
#pragma unroll
        for (int i = 0; i < NCP; ++i) p_copy16(pg[i] + k0, base + rowoff[i]);
    };
    // ONE accumulator per matrix tile, k ascending, like gemm_nt_pipe2_kernel.  Forms of this kernel *(measured, four 1 536 x 1000 x
    // 1000 members / four 512-row ones)*: eight waves of 64 x 32, fragments read behind the stage's barrier 28.9 / 16.5 us; the same
    // with the next stage's fragments under this stage's matrix instructions (this form) 28.2 / 16.8; five stages instead of four
    // 29.3 / 16.7 (six or eight, one workgroup per CU: 37.4 - 37.9 / 19.4); four waves of 64 x 64 (four fragment reads per four matrix instructions instead of three per two) 30.1 / 19.2; the
    // two accumulator sets of round 4's 64 x 64-tile kernel as the two waves of a pair 36.4 / 21.2.  All land within 10 % of each
    // other: what a stage costs is its 16 KB through the LDS-DMA path (~32 B/clk per CU: 512 of the ~620 clocks a stage takes), not
    // its fragment reads, its barrier or the depth of the prefetch -- fewer staged bytes per output is what would shorten it.
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < Q_NSTG; ++s2)
        if (s2 < nkt) copy_stage(s2);
    const int r31 = lane & 31, h = lane >> 5;
    int fxo[2], fwo;                                            // the lane's fragment rows: byte offset, swizzle in bits 4..5
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int rx = 64 * wm + 32 * a + r31;
        fxo[a] = rx * (Q_BK * 2) + (((rx >> 2) & 3) << 4);
    }
    {
        const int rw = 32 * wn + r31;
        fwo = Q_BM * Q_BK * 2 + rw * (Q_BK * 2) + (((rw >> 2) & 3) << 4);
    }
    // As in gemm_nt_pipe2_kernel, the fragments of stage kt + 1 are requested right behind the barrier that publishes them and travel
    // under stage kt's matrix instructions (two fragment sets); a stage's buffer is free once its fragments are in registers, so the
    // copy issued behind the barrier is stage kt + 4's: four stages buffered or in flight.
    struct Frags { bf16x8 x[2][2], w[2]; };                     // [k-step][row tile], [k-step]
    auto read_frags = [&](int kt, Frags& f) {
        const unsigned char* buf = smem_raw + (kt % Q_NSTG) * Q_STG;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cs = (2 * ks + h) << 4;                   // (chunk ^ swizzle) << 4 == (chunk << 4) ^ (swizzle << 4)
#pragma unroll
            for (int a = 0; a < 2; ++a) f.x[ks][a] = *reinterpret_cast<const bf16x8*>(buf + ((fxo[a] & ~63) | ((fxo[a] & 48) ^ cs)));
            f.w[ks] = *reinterpret_cast<const bf16x8*>(buf + ((fwo & ~63) | ((fwo & 48) ^ cs)));
        }
    };
    {
        const int younger = nkt - 1 < 3 ? nkt - 1 : 3;          // stages 1 .. 3 may still fly (2 copies each)
        if (younger == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p_lds_barrier();
    }
    Frags f0, f1;
    read_frags(0, f0);
    auto step = [&](int kt, const Frags& cur, Frags& nxt) {
        if (kt + 1 < nkt) {
            // stage kt + 1 must have landed; issued so far: stages up to min(kt + 3, nkt - 1)
            const int younger = nkt - 2 - kt < 2 ? nkt - 2 - kt : 2;
            if (younger == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            p_lds_barrier();                                    // stage kt + 1 is in LDS for everybody; everybody holds stage kt's fragments
            if (kt + Q_NSTG < nkt) copy_stage(kt + Q_NSTG);     // (into stage kt's buffer)
            read_frags(kt + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long long kbeg = (long long)kt * Q_BK;
        int ks0 = 0;
        if (kbeg + Q_BK > p.K && p.K >= Q_BK) ks0 = (int)((kbeg - (p.K - Q_BK)) >> 4);       // (the short last stage was copied as the last full window)
        const int ks1 = p.K >= Q_BK ? 2 : (int)(p.K >> 4);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            if (ks >= ks0 && ks < ks1) {
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.w[ks], cur.x[ks][a], acc[a], 0, 0, 0);
            }
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        step(kt, f0, f1);
        if (kt + 1 < nkt) step(kt + 1, f1, f0);
    }
    p_lds_barrier();                                            // the staging buffers become the C image
#pragma unroll
    for (int r = 0; r < 2; ++r) {                               // rows [64 r, +64) of the tile: the waves with wm == r hold them
        if (wm == r) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int m = 32 * a + r31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * wn + 8 * g + 4 * h;
                    f32x4 v = {acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(sC + m * Q_CS + n) = v;
                }
            }
        }
        p_lds_barrier();
        nt_store_tile<64, Q_BN, 512>(p, sC, m0 + 64 * r, n0, tid);
        if (r == 0) p_lds_barrier();
    }
}
__global__ __launch_bounds__(512, 4) void gemm_nt_g128_group_kernel(GemmGroupArgs grp, int n, int tiles) {
    (void)grp;
    int member;
    long long tile;
    if (n == 2 || n == 4 || n == 8) {                           // (members on XCDs of their own: see gemm_nt_pipe2_group_kernel)
        const int per = 8 / n, xcd = blockIdx.x & 7;
        member = xcd / per;
        tile = (long long)(blockIdx.x >> 3) * per + (xcd % per);
    } else {
        member = blockIdx.x / tiles;
        tile = blockIdx.x - member * tiles;
    }
    if (tile >= tiles || member >= n) return;
    GemmArgs p;
    load_group_member(p, member);
    nt_g128_body(p, tile);
}

// ---------------------------------------------------------------------------------------------------
// Long batches of the same generic shapes (the DenseDim-1000 layers of the frame critics' steps in the video configuration:
// 3 x 4 608 rows).  The 64 x 64-tile kernel above issues ONE matrix instruction per two fragment reads and moves
// (64 + 64) x K operand bytes per 64 x 64 outputs: measured 297 TFLOP/s at 13 824 x 1000 x 1000 whatever the batch -- LDS reads
// (2 KB per MFMA against 128 B/clk per CU) and L2 -> LDS traffic (32 flop per byte) bound it, not the matrix pipe.  Here a
// workgroup owns 128 x 256 outputs and each of its four waves (one per SIMD) a 64 x 128 block of them: 2 + 4 fragment reads feed
// 8 matrix instructions (0.75 KB per MFMA), 85 flop per L2 byte.  Same stage mechanics (global -> LDS without registers,
// exact vmcnt waits, LDS-only barriers, three 48 KB stages), same coalesced epilogue through an fp32 C tile in LDS.
// ---------------------------------------------------------------------------------------------------
constexpr int G_BM = 128, G_BN = 256, G_NSTG = 3;
constexpr int G_STG = (G_BM + G_BN) * BK * 2;                                // 49 152 bytes per stage
constexpr int G_CS = G_BN + 4;
constexpr int G_LDS = (G_NSTG * G_STG > G_BM * G_CS * 4) ? G_NSTG * G_STG : G_BM * G_CS * 4;   // 147 456

__global__ __launch_bounds__(256, 1) void gemm_nt_big_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    float* sC = reinterpret_cast<float*>(gsm);                  // [G_BM][G_CS], reuses the staging buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                    // 2 x 2 waves: rows [64 wm, +64), columns [128 wn, +128)
    const long long ntn = (p.W + G_BN - 1) / G_BN;
    // consecutive workgroups share the batch rows (the larger operand per tile) and walk the column tiles: the row block is
    // fetched from HBM once and re-read from L2
    const long long m0 = (long long)(blockIdx.x / ntn) * G_BM;
    const long long n0 = (long long)(blockIdx.x % ntn) * G_BN;
    const int nkt = (int)((p.K + BK - 1) / BK);
    constexpr int NCP = (G_BM + G_BN) * 8 / 256;                // copies per lane and stage: 12 (4 of A, 8 of B)
    const uint16_t* pg[NCP];
    int rowoff[NCP];
#pragma unroll
    for (int i = 0; i < NCP; ++i) {
        // copy i of wave w moves rows [8 (4 i + w), +8) of the stage image (A rows first, then B rows): 8 lanes per row
        const int row0 = (4 * i + wave) * 8, row = row0 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        if (i < G_BM / 32) {
            const long long gm = m0 + row;
            pg[i] = p.A + (gm < p.M ? gm : p.M - 1) * p.lda + c * 8;
        } else {
            const long long gn = n0 + row - G_BM;
            pg[i] = p.B + (gn < p.N ? gn : p.N - 1) * p.ldb + c * 8;
        }
        rowoff[i] = row0 * (BK * 2);
    }
    auto copy_stage = [&](int kt) {
        unsigned char* base = gsm + (kt % G_NSTG) * G_STG;
        long long k0 = (long long)kt * BK;
        if (k0 + BK > p.K) k0 = p.K - BK > 0 ? p.K - BK : 0;   // short last stage: the last full window (K >= 64)
#pragma unroll
        for (int i = 0; i < NCP; ++i) p_copy16(pg[i] + k0, base + rowoff[i]);
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < G_NSTG - 1; ++s2)
        if (s2 < nkt && !(p.abl & 4)) copy_stage(s2);
    const int r31 = lane & 31, h = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCP) : "memory");   // stage kt landed, kt+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p_lds_barrier();                                        // stage kt is in LDS for everybody, stage kt-1 is released
        if (kt + G_NSTG - 1 < nkt && !(p.abl & 4)) copy_stage(kt + G_NSTG - 1);
        if (p.abl & 2) continue;
        const unsigned char* bufA = gsm + (kt % G_NSTG) * G_STG;
        const unsigned char* bufB = bufA + G_BM * BK * 2;
        const long long kbeg = (long long)kt * BK;
        int ks0 = 0;
        if (kbeg + BK > p.K && p.K >= BK) ks0 = (int)((kbeg - (p.K - BK)) >> 4);
        const int ks1 = p.K >= BK ? 4 : (int)(p.K >> 4);
        for (int ks = ks0; ks < ks1; ++ks) {
            const int chunk = 2 * ks + h;
            bf16x8 fx[2], fw[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int row = 64 * wm + 32 * a + r31;
                fx[a] = *reinterpret_cast<const bf16x8*>(bufA + row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int row = 128 * wn + 32 * b + r31;
                fw[b] = *reinterpret_cast<const bf16x8*>(bufB + row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b], fx[a], acc[a][b], 0, 0, 0);
        }
    }
    if (p.abl & 1) return;
    p_lds_barrier();                                            // the staging buffers become the C tile
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int m = 64 * wm + 32 * a + r31;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 128 * wn + 32 * b + 8 * g + 4 * h;
                f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *reinterpret_cast<f32x4*>(sC + m * G_CS + n) = v;
            }
    }
    p_lds_barrier();
    nt_store_tile<G_BM, G_BN>(p, sC, m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------------
// 256 x 256 tiles for the wide layers (DenseDim 1000) *(r4)*.  The 128 x 256 kernel above runs ONE wave per SIMD: every k-step
// waits for its six fragment reads before its eight matrix instructions, and two 48 KB stages in flight feed it 48 KB per
// ~1 us: measured 31 us per workgroup of 128 x 256 x 1000 = 0.22 of the CU's matrix peak.  Here eight waves (two per SIMD: one
// wave's fragment reads run under the other's matrix instructions) own 128 x 64 each; a k-tile is 32 wide, so a stage is
// 32 KB and FIVE fit: four in flight = 128 KB per CU; 128 flop per staged byte instead of 85.  Same stage mechanics (global ->
// LDS without registers, exact vmcnt waits, LDS-only barriers); 64-byte stage rows, 16-byte chunk c of row r at position
// c ^ ((r >> 2) & 3): the sixteen rows a quarter-wave reads cover all 64 banks.  The epilogue goes through the fp32 C image in
// two passes of 128 rows (nt_store_tile: bias, residual, activation, mask, bf16 / fp32 stores -- coalesced).
// *(measured, 13 824 x 1000 x 1000, 216 workgroups)* 48 us against 62 (576 TFLOP/s): the k loop 32 us -- the copies alone take 22
// (1 MB per workgroup at 47 GB/s per CU = 10 TB/s of L2 -> LDS over the card: the LDS-DMA path's rate, not HBM's), the matrix
// instructions alone 22 --, the epilogue 10, the launch 5.  A workgroup's time does not shrink with the batch: below ~160
// workgroups the 128 x 256 kernel's smaller tiles win (4 608 rows: 31 us against 43).  (The copies' 47 GB/s per CU is this
// kernel's pipeline, not the path: tools/ubench/fill_rate.hip moves 145-150 GB/s per CU from L2 into LDS with the same row-strided
// requests, by LDS-DMA or through registers alike, once 48 KB per workgroup are in flight and nothing waits in between.  Tried:
// one dword load per wave and stage, eight stages ahead, to warm L2 -- 64 lines per instruction: 45 -> 61 us.)  Grouping the four 1 536-row branch
// layers of a motion critic into one launch of these tiles (96 workgroups, 45 us) buys nothing over four launches of 64 x 64 tiles
// (13 us each) -- that launch form was written, measured and removed.  So was a four-wave form of this kernel (128 x 128 per wave: 0.5 KB
// of fragment reads per matrix instruction instead of 0.75 -- LDS reads and the matrix pipe are both ~1 000 clocks per stage here):
// 256 accumulator registers + two fragment sets leave hipcc 96 spilled registers inside the k loop, and scratch reloads are entries of
// the memory counter the stage waits count.
// ---------------------------------------------------------------------------------------------------
constexpr int W_BM = 256, W_BN = 256, W_BK = 32, W_NSTG = 5;
constexpr int W_STG = (W_BM + W_BN) * W_BK * 2;                              // 32 768 bytes per stage
constexpr int W_LDS = W_NSTG * W_STG;                                        // 163 840: all of the CU's LDS
constexpr int W_CS = W_BN + 4;
static_assert(128 * W_CS * 4 <= W_LDS, "C image");

__global__ __launch_bounds__(512, 1) void gemm_nt_wide_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    float* sC = reinterpret_cast<float*>(wsm);                  // [128][W_CS], reuses the staging buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                    // 2 x 4 waves: rows [128 wm, +128), columns [64 wn, +64)
    const long long ntn = (p.W + W_BN - 1) / W_BN;
    // Workgroups go to the eight XCDs round-robin, and every XCD has its own L2: the column tiles of ONE row block are given to
    // workgroups b, b + 8, b + 16 ... (same XCD, started together), so the row block's operand rows come from beyond L2 once, not
    // once per column tile.  (Row blocks are padded to a multiple of eight: a workgroup beyond the batch leaves at once.)
#ifndef W_NO_XCD_MAP
    const long long j = blockIdx.x >> 3;
    const long long m0 = ((j / ntn) * 8 + (blockIdx.x & 7)) * W_BM;
    const long long n0 = (j % ntn) * W_BN;
    if (m0 >= p.M) return;
#else
    const long long m0 = (long long)(blockIdx.x / ntn) * W_BM;
    const long long n0 = (long long)(blockIdx.x % ntn) * W_BN;
#endif
    const int nkt = (int)((p.K + W_BK - 1) / W_BK);
    constexpr int NCP = 4;                                      // copies per lane and stage: rows 16 (8 i + wave) .. + 16 of the image
    const uint16_t* pg[NCP];
    int rowoff[NCP];
#pragma unroll
    for (int i = 0; i < NCP; ++i) {
        const int row0 = (8 * i + wave) * 16, row = row0 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        if (i < 2) {
            const long long gm = m0 + row;
            pg[i] = p.A + (gm < p.M ? gm : p.M - 1) * p.lda + c * 8;
        } else {
            const long long gn = n0 + row - W_BM;
            pg[i] = p.B + (gn < p.N ? gn : p.N - 1) * p.ldb + c * 8;
        }
        rowoff[i] = row0 * (W_BK * 2);
    }
    auto copy_stage = [&](int kt) {
        unsigned char* base = wsm + (kt % W_NSTG) * W_STG;
        long long k0 = (long long)kt * W_BK;
        if (k0 + W_BK > p.K) k0 = p.K - W_BK > 0 ? p.K - W_BK : 0;       // short last stage: the last full window (K >= 32)
#pragma unroll
        for (int i = 0; i < NCP; ++i) p_copy16(pg[i] + k0, base + rowoff[i]);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < W_NSTG - 1; ++s2)
        if (s2 < nkt && !(p.abl & 4)) copy_stage(s2);
    const int r31 = lane & 31, h = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt must have landed; younger: stages kt+1 .. kt+3 (4 copies each) where they exist
        const int younger = nkt - 1 - kt;
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p_lds_barrier();                                        // stage kt is in LDS for everybody, stage kt-1 is released
        if (kt + W_NSTG - 1 < nkt && !(p.abl & 4)) copy_stage(kt + W_NSTG - 1);
        if (p.abl & 2) continue;
        const unsigned char* bufA = wsm + (kt % W_NSTG) * W_STG;
        const unsigned char* bufB = bufA + W_BM * W_BK * 2;
        const long long kbeg = (long long)kt * W_BK;
        int ks0 = 0;
        if (kbeg + W_BK > p.K && p.K >= W_BK) ks0 = (int)((kbeg - (p.K - W_BK)) >> 4);
        const int ks1 = p.K >= W_BK ? 2 : (int)(p.K >> 4);
        for (int ks = ks0; ks < ks1; ++ks) {
            const int chunk = 2 * ks + h;
            bf16x8 fx[4], fw[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = 128 * wm + 32 * a + r31;
                fx[a] = *reinterpret_cast<const bf16x8*>(bufA + row * (W_BK * 2) + ((chunk ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int row = 64 * wn + 32 * b + r31;
                fw[b] = *reinterpret_cast<const bf16x8*>(bufB + row * (W_BK * 2) + ((chunk ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b], fx[a], acc[a][b], 0, 0, 0);
        }
    }
    if (p.abl & 1) return;
    p_lds_barrier();                                            // the staging buffers become the C image
#pragma unroll
    for (int r = 0; r < 2; ++r) {                               // rows [128 r, +128) of the tile: the waves with wm == r hold them
        if (wm == r) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int m = 32 * a + r31;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = 64 * wn + 32 * b + 8 * g + 4 * h;
                        f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(sC + m * W_CS + n) = v;
                    }
            }
        }
        p_lds_barrier();
        nt_store_tile<128, W_BN, 512>(p, sC, m0 + 128 * r, n0, tid);
        if (r == 0) p_lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------
// TN (weight gradient): C[N1,N2] += sum_m A[m,N1] * B[m,N2]
// ---------------------------------------------------------------------------------------------------
struct TnArgs {
    const uint16_t* A; long long lda;
    const uint16_t* B; long long ldb;
    float* C; long long ldc;
    float* colsum;                 // optional [N1]: column sums of A (bias gradient), accumulated by the n2-tile-0 blocks
    long long M, N1, N2;
    long long rows_per_split, ntiles, nsplits;
    long long cs_rows;             // the column sums cover rows [0, cs_rows) only (a multiple of the stage height)
};

constexpr int TN_BN = 64;           // output tile edge: small tiles keep the split-K partial sums (fp32 atomics, the
                                    // scarce resource: ~1.3 TB/s chip-wide) at splits x N1 x N2 x 4 B with few splits
constexpr int TN_LD = 96;           // LDS row stride (elements): 192 B = 48 banks, so the 4 rows of a transpose-read
                                    // block (64 B each) land on disjoint bank quarters

// 8 consecutive contraction rows k = kbase..kbase+7 of column (col0 + lane&15) : two hardware transpose reads.
__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* tile, int kbase, int col0, int lane) {
    const int li = lane & 15, q = li >> 2, pp = li & 3;
    const uint16_t* a0 = tile + (kbase + q) * TN_LD + col0 + 4 * pp;
    typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 4 * TN_LD));
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

// One workgroup = one 64 x 64 output tile over one slice of the batch.  The tiles of a slice are mapped to the same
// XCD (equal blockIdx % 8) so that the slice's rows are fetched from HBM once and re-read from that XCD's L2 by the
// other tiles (speed only; any placement is correct).
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs p) {
    __shared__ __attribute__((aligned(16))) uint16_t sA[2][BK * TN_LD];
    __shared__ __attribute__((aligned(16))) uint16_t sB[2][BK * TN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w1 = wave >> 1, w2 = wave & 1;                    // 2 x 2 waves, one 32 x 32 MFMA tile each
    const long long nt2 = (p.N2 + TN_BN - 1) / TN_BN;
    long long tile, split;
    if ((p.nsplits & 7) == 0) {
        const long long xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        tile = local % p.ntiles;
        split = xcd + 8 * (local / p.ntiles);
    } else {
        tile = blockIdx.x % p.ntiles;
        split = blockIdx.x / p.ntiles;
    }
    const long long n1_0 = (tile / nt2) * TN_BN, n2_0 = (tile % nt2) * TN_BN;
    const long long ms = split * p.rows_per_split;
    long long me = ms + p.rows_per_split;
    if (me > p.M) me = p.M;
    if (ms >= me) return;

    uint4 ra[2], rb[2];
    auto gload = [&](long long mrow0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + 256 * i, row = q >> 3, c = q & 7;
            const long long gm = mrow0 + row;
            const long long ca = n1_0 + c * 8, cb = n2_0 + c * 8;
            ra[i] = (gm < me && ca < p.N1) ? *reinterpret_cast<const uint4*>(p.A + gm * p.lda + ca) : make_uint4(0, 0, 0, 0);
            rb[i] = (gm < me && cb < p.N2) ? *reinterpret_cast<const uint4*>(p.B + gm * p.ldb + cb) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + 256 * i, row = q >> 3, c = q & 7;
            *reinterpret_cast<uint4*>(&sA[buf][row * TN_LD + c * 8]) = ra[i];
            *reinterpret_cast<uint4*>(&sB[buf][row * TN_LD + c * 8]) = rb[i];
        }
    };

    f32x16 acc, accs;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; accs[r] = 0.0f; }
    // bias gradient for free: A^T * ones on the matrix pipe (only the tiles of output-column block 0)
    const bool do_cs = p.colsum != nullptr && (tile % nt2) == 0 && w2 == 0;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

    const int nkt = (int)((me - ms + BK - 1) / BK);
    gload(ms);
    lstore(0);
    __syncthreads();
    const int grp = lane >> 4;                                  // 16-lane group: columns 16*(grp&1), k half grp>>1
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(ms + (long long)(kt + 1) * BK);
        const uint16_t* bufA = sA[kt & 1];
        const uint16_t* bufB = sB[kt & 1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kbase = 16 * ks + 8 * (grp >> 1);
            const bf16x8 fa = tr_frag(bufA, kbase, w1 * 32 + 16 * (grp & 1), lane);
            const bf16x8 fb = tr_frag(bufB, kbase, w2 * 32 + 16 * (grp & 1), lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
            if (do_cs && ms + (long long)kt * BK < p.cs_rows) accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, ones, accs, 0, 0, 0);
        }
        if (kt + 1 < nkt) lstore((kt + 1) & 1);
        __syncthreads();
    }
    // D[n1][n2]: lane owns column n2 = lane&31, rows n1 = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const long long n2 = n2_0 + w2 * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long n1 = n1_0 + w1 * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n1 < p.N1 && n2 < p.N2) atomicAdd(p.C + n1 * p.ldc + n2, acc[r]);
        if (do_cs && (lane & 31) == 0 && n1 < p.N1) atomicAdd(p.colsum + n1, accs[r]);   // every column of A^T*ones
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradients of the training path's regular shapes (N1, N2 multiples of 64, batch slice a multiple of 128 rows):
// same 64 x 64 output tiles and XCD-local slices as gemm_tn_kernel, but the operand rows travel global -> LDS without
// registers (global_load_lds_dwordx4), 128 rows per stage, two stages in flight, and the barriers order LDS traffic
// only -- the register-staged loop above drains its own prefetch at every __syncthreads().
// LDS stage: [128 rows][64 cols] bf16, rows contiguous (the copy fixes a lane's LDS slot), 16-byte chunk c of row r
// stored at position c ^ (4 * ((r >> 1) & 1)): the four rows x four column groups of a transpose-read then cover all
// 64 banks once (the copy applies the permutation on the global side).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void f_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t f_pack_bf16x2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}

__device__ __forceinline__ void f_copy16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// The same copy for the TN kernels below, as inline assembly ON PURPOSE: their fragments are read with the hardware
// transpose read (an intrinsic without memory operand), and hipcc's waitcnt pass -- which knows that the LDS-DMA builtin
// writes LDS -- puts a full `s_waitcnt vmcnt(0)` in front of the first such read of every stage: right behind the explicit
// vmcnt(N) that was meant to leave the younger stages in flight, so the ring was drained at every stage (visible in the ISA;
// found in round 3).  The asm copy is invisible to that pass; completion is tracked by the explicit waits alone.
typedef unsigned char __attribute__((address_space(3))) * tn_lds_ptr;
__device__ __forceinline__ void tn_copy16(const void* g, unsigned char* lds_wave_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"((tn_lds_ptr)lds_wave_base), "v"(g) : "memory");
}

constexpr int TF_ROWS = 128;                                    // contraction rows per stage

// fragment of 8 contraction rows x 16 columns from a stage whose rows hold CH 16-byte chunks (CH = 8: 64 columns,
// chunk c of row r at position c ^ (4 * ((r >> 1) & 1)); CH = 16: 128 columns, position c ^ (4 * (r & 3))): in both
// layouts the four rows x four column groups of a transpose-read cover all 64 banks once
template <int CH>
__device__ __forceinline__ int tf_sw(int row) { return CH == 8 ? ((row >> 1) & 1) << 2 : (row & 3) << 2; }

template <int CH>
__device__ __forceinline__ bf16x8 tf_frag(const unsigned char* stage, int kbase, int col0, int lane) {
    const int li = lane & 15, q = li >> 2, pp = li & 3;
    const int c = (col0 >> 3) + (pp >> 1), in = (pp & 1) << 3;
    const int r0 = kbase + q, r1 = r0 + 4;
    typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(stage + r0 * (CH * 16) + ((c ^ tf_sw<CH>(r0)) << 4) + in));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(stage + r1 * (CH * 16) + ((c ^ tf_sw<CH>(r1)) << 4) + in));
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}

// chunks beyond the operands' column counts (N1, N2 not multiples of the tile edge: the 100-wide and the input layers) are
// copied from these 16 zero bytes: every lane of every copy instruction stays active, so the vmcnt arithmetic is the same in
// every wave, and the tile's unused rows / columns contribute nothing
__device__ uint4 g_tn_zero16 = {0u, 0u, 0u, 0u};

// TM x 64 output tile (TM = 64 or 128 rows of C = columns of A), one batch slice.
template <int TM>
__global__ __launch_bounds__(256, TM == 64 ? 2 : 1) void gemm_tn64_kernel(TnArgs p) {
    constexpr int CHA = TM / 8, RT = TM / 64;                   // chunks per A row, MFMA row tiles per wave
    constexpr int SA = TF_ROWS * TM * 2, SB = TF_ROWS * 128;    // stage bytes
    constexpr int NCOPY = (TF_ROWS * CHA + TF_ROWS * 8) / 256;  // copies per lane and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];       // [2 stages][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w1 = wave >> 1, w2 = wave & 1;                    // 2 x 2 waves: rows [RT*32*w1, ...), columns [32*w2, +32)
    const long long nt2 = (p.N2 + TN_BN - 1) / TN_BN;
    long long tile, split;
    if ((p.nsplits & 7) == 0) {
        const long long xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        tile = local % p.ntiles;
        split = xcd + 8 * (local / p.ntiles);
    } else {
        tile = blockIdx.x % p.ntiles;
        split = blockIdx.x / p.ntiles;
    }
    const long long n1_0 = (tile / nt2) * TM, n2_0 = (tile % nt2) * TN_BN;
    const long long ms = split * p.rows_per_split;
    long long me = ms + p.rows_per_split;
    if (me > p.M) me = p.M;
    if (ms >= me) return;
    const int nst = (int)((me - ms) / TF_ROWS);                 // whole stages (checked on the host)
    const long long n1c = (p.N1 + 7) & ~7LL, n2c = (p.N2 + 7) & ~7LL;    // readable columns (the operands' rows are padded to 8)

    auto copy_stage = [&](int st, int buf) {                    // asynchronous: NCOPY copies per lane, counted in vmcnt
        const long long m0 = ms + (long long)st * TF_ROWS;
        unsigned char* base = tsm + buf * (SA + SB);
#pragma unroll
        for (int i = 0; i < TF_ROWS * CHA / 256; ++i) {
            constexpr int RW = 64 / CHA;                        // rows per wave instruction
            const int row0 = (wave * (TF_ROWS * CHA / 256) + i) * RW, row = row0 + lane / CHA, c = (lane % CHA) ^ tf_sw<CHA>(row);
            const bool in = n1_0 + c * 8 < n1c;
            tn_copy16(in ? static_cast<const void*>(p.A + (m0 + row) * p.lda + n1_0 + c * 8) : static_cast<const void*>(&g_tn_zero16),
                     base + row0 * (CHA * 16));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row0 = (wave * 4 + i) * 8, row = row0 + (lane >> 3), c = (lane & 7) ^ tf_sw<8>(row);
            const bool in = n2_0 + c * 8 < n2c;
            tn_copy16(in ? static_cast<const void*>(p.B + (m0 + row) * p.ldb + n2_0 + c * 8) : static_cast<const void*>(&g_tn_zero16),
                     base + SA + row0 * 128);
        }
    };
    f32x16 acc[RT], accs[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; accs[t][r] = 0.0f; }
    const bool do_cs = p.colsum != nullptr && (tile % nt2) == 0 && w2 == 0;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const int grp = lane >> 4;                                  // 16-lane group: columns 16*(grp&1), k half grp>>1

    copy_stage(0, 0);
    if (nst > 1) copy_stage(1, 1);
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) {                                     // this stage landed, the next may still fly
            if (NCOPY == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        f_lds_barrier();
        const unsigned char* sa = tsm + buf * (SA + SB);
        const unsigned char* sb = sa + SA;
        const bool cs_stage = do_cs && ms + (long long)st * TF_ROWS < p.cs_rows;       // wave-uniform
#pragma unroll
        for (int ks = 0; ks < TF_ROWS / 16; ++ks) {
            const int kbase = 16 * ks + 8 * (grp >> 1);
            const bf16x8 fb = tf_frag<8>(sb, kbase, w2 * 32 + 16 * (grp & 1), lane);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const bf16x8 fa = tf_frag<CHA>(sa, kbase, (w1 * RT + t) * 32 + 16 * (grp & 1), lane);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
                if (cs_stage) accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, ones, accs[t], 0, 0, 0);
            }
        }
        f_lds_barrier();                                        // every wave is done with this stage
        if (st + 2 < nst) copy_stage(st + 2, buf);
    }
    const long long n2 = n2_0 + w2 * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long n1 = n1_0 + (w1 * RT + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n1 < p.N1 && n2 < p.N2) atomicAdd(p.C + n1 * p.ldc + n2, acc[t][r]);
            if (do_cs && (lane & 31) == 0 && n1 < p.N1) atomicAdd(p.colsum + n1, accs[t][r]);
        }
}

// ---------------------------------------------------------------------------------------------------
// 128 x 128 output tiles for the long contractions of the explicit critic step (3B rows, N1 = N2 = 256).  With 64 x 64
// tiles every row of both operands crosses L2 -> LDS four times (16 tiles x 128 B of 512 B each): 805 MB per layer at
// 3B = 196 608 rows, and the measured 62 us per launch is 21 B/clk/CU of LDS-DMA -- the kernel is bound by the L2 -> CU
// path, not by HBM (33 us) or the matrix pipe (10 us).  128 x 128 tiles halve that traffic.  64-row stages of 32 KB
// (both operands), four in the ring (three in flight), one workgroup per CU; a wave owns a 64 x 64 quadrant.
// RESULT (measured): correct (tests/test_gpu_kernels.py::test_gemm_tn) but slower -- the GAN iteration takes 12.5 ms with
// it against 11.1 ms with the 64 x 64 tiles, with and without fragment reads one k-step ahead: one wave per SIMD hides
// less than two workgroups per CU do, and the atomic traffic doubles (64 slices x 256 KB).  Not used by default
// (DHAUG_TN_128=1 selects it).
// ---------------------------------------------------------------------------------------------------
constexpr int TB_ROWS = 64, TB_STG = 2 * TB_ROWS * 256, TB_NSTG = 4;           // 32 768 bytes per stage

__global__ __launch_bounds__(256, 1) void gemm_tn128_kernel(TnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];       // [4 stages][A 16 KB | B 16 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w1 = wave >> 1, w2 = wave & 1;                    // quadrant: C rows [64 w1, +64), columns [64 w2, +64)
    const long long nt2 = p.N2 / 128;
    long long tile, split;
    if ((p.nsplits & 7) == 0) {
        const long long xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        tile = local % p.ntiles;
        split = xcd + 8 * (local / p.ntiles);
    } else {
        tile = blockIdx.x % p.ntiles;
        split = blockIdx.x / p.ntiles;
    }
    const long long n1_0 = (tile / nt2) * 128, n2_0 = (tile % nt2) * 128;
    const long long ms = split * p.rows_per_split;
    long long me = ms + p.rows_per_split;
    if (me > p.M) me = p.M;
    if (ms >= me) return;
    const int nst = (int)((me - ms) / TB_ROWS);                 // whole stages (checked on the host)

    auto copy_stage = [&](int st) {                             // 8 copies per lane, counted in vmcnt
        const long long m0 = ms + (long long)st * TB_ROWS;
        unsigned char* base = tsm + (st % TB_NSTG) * TB_STG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                           // 16 chunks per row: 4 rows per wave instruction
            const int row0 = (wave * 4 + i) * 4, row = row0 + (lane >> 4), c = (lane & 15) ^ tf_sw<16>(row);
            tn_copy16(p.A + (m0 + row) * p.lda + n1_0 + c * 8, base + row0 * 256);
            tn_copy16(p.B + (m0 + row) * p.ldb + n2_0 + c * 8, base + TB_ROWS * 256 + row0 * 256);
        }
    };
    f32x16 acc[2][2], accs[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][0][r] = 0.0f; acc[t][1][r] = 0.0f; accs[t][r] = 0.0f; }
    }
    const bool do_cs = p.colsum != nullptr && (tile % nt2) == 0 && w2 == 0;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const int grp = lane >> 4;
#pragma unroll
    for (int st = 0; st < TB_NSTG - 1; ++st)
        if (st < nst) copy_stage(st);
    for (int st = 0; st < nst; ++st) {
        const int younger = nst - 1 - st;                       // stages st+1, st+2 may still fly (8 copies each)
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f_lds_barrier();                                        // stage st is in LDS, stage st-1 is released
        if (st + TB_NSTG - 1 < nst) copy_stage(st + TB_NSTG - 1);
        const unsigned char* sa = tsm + (st % TB_NSTG) * TB_STG;
        const unsigned char* sb = sa + TB_ROWS * 256;
        const bool cs_stage = do_cs && ms + (long long)st * TB_ROWS < p.cs_rows;
        // fragments one k-step ahead of their MFMAs (one wave per SIMD: nobody else covers the LDS latency)
        bf16x8 fa[2][2], fb[2][2];
        auto frags = [&](int ks, bf16x8 (&a)[2], bf16x8 (&b)[2]) {
            const int kbase = 16 * ks + 8 * (grp >> 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = tf_frag<16>(sa, kbase, w1 * 64 + 32 * t + 16 * (grp & 1), lane);
                b[t] = tf_frag<16>(sb, kbase, w2 * 64 + 32 * t + 16 * (grp & 1), lane);
            }
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < TB_ROWS / 16; ++ks) {
            if (ks + 1 < TB_ROWS / 16) frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][t], fb[ks & 1][u], acc[t][u], 0, 0, 0);
                if (cs_stage) accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][t], ones, accs[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long n2 = n2_0 + w2 * 64 + 32 * u + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long n1 = n1_0 + w1 * 64 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                atomicAdd(p.C + n1 * p.ldc + n2, acc[t][u][r]);
                if (u == 0 && do_cs && (lane & 31) == 0) atomicAdd(p.colsum + n1, accs[t][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// Weight-stationary NT kernel for the layer shapes of this path (M = batch, huge; N <= a few hundred; K <= 256).
//
// The first version above is bound by exposed global-load latency (its time is flat in K).  Here a workgroup
// owns 128 output features for the whole launch: each wave keeps ITS 32 weight rows as MFMA A-operand fragments in
// registers (K/16 x 4 VGPRs, loaded once), and the workgroup walks over 64-row batch tiles persistently:
//     top of iteration t : issue global loads of X(t+1) (and of the residual rows of tile t) into registers
//     compute tile t     : ds_read_b128 X fragments (swizzled image) -> 2 x K/16 MFMAs per wave
//     barrier            : tile t's image is dead -> reused as the fp32 C staging tile
//     stage C, write X(t+1) registers into the other buffer, barrier
//     coalesced epilogue : bias + residual + activation, 16-byte bf16 / fp32 stores
// so HBM requests stay in flight across the compute and epilogue phases; two workgroups per CU cover each
// other's waits.  LDS: 2 x 33 792 B.
// ---------------------------------------------------------------------------------------------------
constexpr int WS_BM = 64, WS_BN = 128, WS_CS = WS_BN + 4;
constexpr int WS_BUF_BYTES = WS_BM * WS_CS * 4;          // 33 792 >= 64 rows x 512 B

constexpr int ws_pow2_chunks(int s) { int p = 2; while (p < s) p <<= 1; return p; }

template <int SP> __device__ __forceinline__ int ws_swz(int row) {
    if (SP >= 16) return row & 15;
    if (SP == 8) return (row >> 1) & 7;
    if (SP == 4) return (row >> 2) & 3;
    return (row >> 3) & 1;
}

template <int KSTEPS>
__global__ __launch_bounds__(256, 2) void gemm_nt_ws_kernel(GemmArgs p) {
    constexpr int S = 2 * KSTEPS;                       // 16-byte chunks per operand row
    constexpr int SP = ws_pow2_chunks(S);               // LDS row pitch in chunks
    constexpr int XCH = (WS_BM * S + 255) / 256;        // X chunks per thread
    static_assert(WS_BM * SP * 16 <= WS_BUF_BYTES, "X tile must fit the staging buffer");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r31 = lane & 31, h = lane >> 5;
    const long long n0 = (long long)blockIdx.y * WS_BN;
    const long long mtiles = (p.M + WS_BM - 1) / WS_BM;

    // this wave's 32 weight rows, resident for the whole launch
    bf16x8 wf[KSTEPS];
    {
        const long long gn = n0 + 32 * wave + r31;
        const uint16_t* wrow = p.B + gn * p.ldb + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (gn < p.N) wf[ks] = *reinterpret_cast<const bf16x8*>(wrow + 16 * ks);
            else { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; wf[ks] = z; }
        }
    }

    uint4 rx[XCH];
    auto load_x = [&](long long mt) {
        const long long m0 = mt * WS_BM;
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int q = tid + 256 * i, row = q / S, c = q - row * S;
            const long long gm = m0 + row;
            rx[i] = (q < WS_BM * S && gm < p.M) ? *reinterpret_cast<const uint4*>(p.A + gm * p.lda + c * 8)
                                                : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_x = [&](int buf) {
        uint16_t* img = reinterpret_cast<uint16_t*>(smem_raw + buf * WS_BUF_BYTES);
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int q = tid + 256 * i, row = q / S, c = q - row * S;
            if (q < WS_BM * S) *reinterpret_cast<uint4*>(img + row * (SP * 8) + ((c ^ ws_swz<SP>(row)) << 3)) = rx[i];
        }
    };

    long long mt = blockIdx.x;
    if (mt >= mtiles) return;
    load_x(mt);
    store_x(0);
    __syncthreads();
    int buf = 0;
    for (; mt < mtiles; mt += gridDim.x, buf ^= 1) {
        const long long m0 = mt * WS_BM;
        const long long mt_next = mt + gridDim.x;
        const bool has_next = mt_next < mtiles;
        if (has_next) load_x(mt_next);
        // residual rows of this tile, 16 bytes per epilogue piece
        uint4 rres[4];
        const bool res_vec = p.res != nullptr;
        if (res_vec) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + 256 * i, row = q >> 4, pc = q & 15;
                const long long gm = m0 + row, n = n0 + pc * 8;
                // (a piece that straddles N -- columns 96..103 of a 100-wide layer -- is still read as one 16-byte load
                // where the row pitch covers it: element-wise loads in the epilogue cost the workgroup a memory round trip
                // per tile)
                rres[i] = (gm < p.M && n < p.N && n + 8 <= p.ld_res) ? *reinterpret_cast<const uint4*>(p.res + gm * p.ld_res + n)
                                                                     : make_uint4(0, 0, 0, 0);
            }
        }
        // ... and of the activation-backward mask (requested here, used in the epilogue: a load issued where its value is
        // needed costs one exposed memory round trip per piece -- 65 -> 4x us for the 100-wide backward / tangent layers)
        uint4 rmask[4];
        const bool mask_vec = p.dmask != nullptr && (p.ld_dmask & 7) == 0 && (reinterpret_cast<uintptr_t>(p.dmask) & 15) == 0;
        if (mask_vec) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + 256 * i, row = q >> 4, pc = q & 15;
                const long long gm = m0 + row, n = n0 + pc * 8;
                rmask[i] = (gm < p.M && n < p.N && n + 8 <= p.ld_dmask) ? *reinterpret_cast<const uint4*>(p.dmask + gm * p.ld_dmask + n)
                                                                        : make_uint4(0, 0, 0, 0);
            }
        }
        // ... or the same mask as sign bits (dhaug_mlp_unit.bits layout, one array per 256 output columns): a piece's 8 features
        // 8g .. 8g+7 of slice (wave_f + 4 t) are pairs 8t + 2g, +1 of the row's two lanes (h = 0: features +0..3, h = 1: +4..7)
        uint32_t rb0[4], rb1[4];
        const bool bits_mode = p.dbits != nullptr;
        if (bits_mode) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + 256 * i, row = q >> 4, pc = q & 15;
                const long long gm = m0 + row, n = n0 + pc * 8;
                rb0[i] = rb1[i] = 0u;
                if (gm < p.M && n + 8 <= p.N) {
                    const uint32_t* bb = n >= 256 ? p.dbits2 : p.dbits;
                    const int f = (int)(n & 255);
                    const long long w = ((gm >> 5) * 4 + ((f >> 5) & 3)) * 64 + (gm & 31);
                    rb0[i] = bb[w];
                    rb1[i] = bb[w + 32];
                }
            }
        }
        // compute
        const uint16_t* img = reinterpret_cast<const uint16_t*>(smem_raw + buf * WS_BUF_BYTES);
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 32 * j + r31;
                const bf16x8 fx = *reinterpret_cast<const bf16x8*>(img + row * (SP * 8) + (((2 * ks + h) ^ ws_swz<SP>(row)) << 3));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], fx, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();                                   // every wave is done reading this image
        float* sC = reinterpret_cast<float*>(smem_raw + buf * WS_BUF_BYTES);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = 32 * j + r31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(sC + m * WS_CS + 32 * wave + 8 * g + 4 * h) = v;
            }
        }
        if (has_next) store_x(buf ^ 1);
        __syncthreads();
        // epilogue: 1024 pieces of 8 features
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, row = q >> 4, pc = q & 15;
            const long long gm = m0 + row, n = n0 + pc * 8;
            if (gm >= p.M) continue;
            if (!(n < p.N || (p.cb != nullptr && n < p.npad))) continue;
            float v[8];
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(sC + row * WS_CS + pc * 8);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(sC + row * WS_CS + pc * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = c0[e]; v[4 + e] = c1[e]; }
            const bool full = n + 8 <= p.N;
            if (p.bias != nullptr) {
                if (full) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += p.bias[n + e];
                }
            }
            if (p.resf != nullptr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (full || n + e < p.N) v[e] += p.resf[gm * p.ld_resf + n + e];
            }
            if (res_vec) {
                if (full || n + 8 <= p.ld_res) {             // (elements beyond N are zeroed below)
                    const uint32_t w[4] = {rres[i].x, rres[i].y, rres[i].z, rres[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __builtin_bit_cast(float, w[e] << 16);
                        v[2 * e + 1] += __builtin_bit_cast(float, w[e] & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += dhaug_bf16_to_f32(p.res[gm * p.ld_res + n + e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (full || n + e < p.N) ? apply_act(v[e], p.act, p.slope) : 0.0f;
            if (bits_mode) {
                const int f = (int)(n & 255), p0 = 8 * (f >> 7) + 2 * ((f & 31) >> 3);
                const uint32_t w0 = rb0[i], w1 = rb1[i];
                v[0] = ((w0 >> p0) & 1u) ? v[0] : v[0] * p.dneg;
                v[1] = ((w0 >> (16 + p0)) & 1u) ? v[1] : v[1] * p.dneg;
                v[2] = ((w0 >> (p0 + 1)) & 1u) ? v[2] : v[2] * p.dneg;
                v[3] = ((w0 >> (17 + p0)) & 1u) ? v[3] : v[3] * p.dneg;
                v[4] = ((w1 >> p0) & 1u) ? v[4] : v[4] * p.dneg;
                v[5] = ((w1 >> (16 + p0)) & 1u) ? v[5] : v[5] * p.dneg;
                v[6] = ((w1 >> (p0 + 1)) & 1u) ? v[6] : v[6] * p.dneg;
                v[7] = ((w1 >> (17 + p0)) & 1u) ? v[7] : v[7] * p.dneg;
            }
            if (p.dmask != nullptr) {                        // activation-backward mask of the producing layer
                if (mask_vec && (full || n + 8 <= p.ld_dmask)) {
                    const uint4 mm = rmask[i];
                    const uint32_t w[4] = {mm.x, mm.y, mm.z, mm.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {            // a positive bf16 is a positive int16
                        v[2 * e] = (short)(w[e] & 0xffffu) > 0 ? v[2 * e] : v[2 * e] * p.dneg;
                        v[2 * e + 1] = (short)(w[e] >> 16) > 0 ? v[2 * e + 1] : v[2 * e + 1] * p.dneg;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (full || n + e < p.N) v[e] = (short)p.dmask[gm * p.ld_dmask + n + e] > 0 ? v[e] : v[e] * p.dneg;
                }
            }
            if (p.cb != nullptr) {
                if (n + 8 <= p.npad || full) {
                    uint4 o;
                    o.x = (uint32_t)dhaug_f32_to_bf16(v[0]) | ((uint32_t)dhaug_f32_to_bf16(v[1]) << 16);
                    o.y = (uint32_t)dhaug_f32_to_bf16(v[2]) | ((uint32_t)dhaug_f32_to_bf16(v[3]) << 16);
                    o.z = (uint32_t)dhaug_f32_to_bf16(v[4]) | ((uint32_t)dhaug_f32_to_bf16(v[5]) << 16);
                    o.w = (uint32_t)dhaug_f32_to_bf16(v[6]) | ((uint32_t)dhaug_f32_to_bf16(v[7]) << 16);
                    *reinterpret_cast<uint4*>(p.cb + gm * p.ldcb + n) = o;
                } else {
                    const long long lim = p.npad > p.N ? p.npad : p.N;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < lim) p.cb[gm * p.ldcb + n + e] = dhaug_f32_to_bf16(v[e]);
                }
            }
            if (p.cf != nullptr) {
                if (full && (p.ldcf & 3) == 0) {
                    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(p.cf + gm * p.ldcf + n) = o0;
                    *reinterpret_cast<f32x4*>(p.cf + gm * p.ldcf + n + 4) = o1;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (n + e < p.N) p.cf[gm * p.ldcf + n + e] = v[e];
                }
            }
        }
        // no barrier here: the next iteration's first LDS write (C staging) is behind its own barrier, and the
        // image it reads (buf^1) was completed before the second barrier above
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 256-feature layers of the training path (M x K -> M x 256, K = 128 or 256, bf16 in / bf16 out, optional bias, bf16
// residual, activation): HBM-bound (A, the residual and C once each), so the kernel is built to keep the memory
// system busy rather than the matrix pipe:
//   * one persistent workgroup per CU (4 waves, 512 registers each); wave w holds the weight rows of feature slices
//     w and w+4 for the whole launch (weight-stationary, 2*KS fragments);
//   * 64-row tiles, two LDS images each for the operand rows and the residual rows.  The rows of tile i+1 travel
//     global -> LDS without touching registers (global_load_lds_dwordx4) while tile i is computed; the swizzle is
//     applied on the global side (lane p of a row fetches chunk p ^ (row & 15); its LDS slot is fixed by the lane);
//   * the residual joins on the matrix pipe (two identity k-steps per slice against its LDS image), the bias seeds
//     the accumulators; the epilogue packs to bf16 into an LDS image that the workgroup then streams out as whole
//     512-byte rows;
//   * barriers order LDS traffic only; the one vmcnt(0) per tile sits after the tile's MFMAs, where the copy issued
//     before them has had the whole tile to land.
// Measured anatomy at 65536 x 256 x 256 (no residual, 23 us): 10.8 us of data movement and synchronisation + 9 us of MFMA
// phases + 7 us of store phases, simply added up -- one workgroup per CU runs its tile as a serial chain.  A 32-row /
// two-workgroups-per-CU variant overlaps them (K = 128: 16.3 -> 14.1 us) but spills at K = 256 (128 weight registers in a
// 256-register budget: 45 us); a third operand image and exact vmcnt accounting changed nothing.  Next step: split the
// tile's phases across two wave groups of one workgroup.
// LDS: 2 x 32 KB operand images + 2 x 32 KB residual images + 32 KB output image = 160 KB, 16-byte chunks XOR-swizzled
// by row & 15 (conflict-free ds_read_b128 fragment reads).
// ---------------------------------------------------------------------------------------------------------------
constexpr int F_BM = 64, F_PITCH = 512, F_IMG = F_BM * F_PITCH;             // 32 768 bytes per image
constexpr int F_LDS_BYTES = 5 * F_IMG;

// MODE 0: plain, 1: + bf16 residual, 2: result * act'(dmask) (the layer's output feeds an activation backward)
template <int KS, int MODE>
__global__ __launch_bounds__(256, 1) void gemm_nt256_kernel(GemmArgs p) {
    constexpr bool RES = MODE == 1, MASK = MODE == 2;
    static_assert(KS == 8 || KS == 16, "K = 128 or 256");
    constexpr int S = 2 * KS;                                                // 16-byte chunks per operand row
    constexpr int XP = S * 16;                                               // operand image pitch (bytes)
    constexpr int RCH = F_BM * 32 / 256;                                     // output chunks per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;                                                // [2][F_IMG]
    unsigned char* sR = smem + 2 * F_IMG;                                    // [2][F_IMG]
    unsigned char* sO = smem + 4 * F_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, h = lane >> 5, x = lane & 15;
    const long long mtiles = p.M / F_BM, g = gridDim.x;

    auto copy_tile = [&](long long mt, int buf) {                            // asynchronous: counted in vmcnt
        const long long m0 = (mt < mtiles ? mt : mtiles - 1) * F_BM;         // tiles past the end re-read the last one
        {
            constexpr int RW = 1024 / XP;                                    // rows per wave instruction
#pragma unroll
            for (int i = 0; i < S / 4; ++i) {
                const int row0 = (wave * (S / 4) + i) * RW, row = row0 + lane / S, c = (lane % S) ^ (row & 15);
                f_copy16(p.A + (m0 + row) * p.lda + c * 8, sX + buf * F_IMG + row0 * XP);
            }
        }
        if (RES || MASK) {                                       // the second image: residual rows or the mask source
            const uint16_t* src = RES ? p.res : p.dmask;
            const long long ld = RES ? p.ld_res : p.ld_dmask;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row0 = (wave * 8 + i) * 2, row = row0 + (lane >> 5), c = (lane & 31) ^ (row & 15);
                f_copy16(src + (m0 + row) * ld + c * 8, sR + buf * F_IMG + row0 * F_PITCH);
            }
        }
    };
    long long mt = blockIdx.x;
    if (mt >= mtiles) return;
    copy_tile(mt, 0);
    copy_tile(mt + g, 1);

    bf16x8 wf[2][KS];
    f32x16 seed[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint16_t* wrow = p.B + (long long)(32 * (wave + 4 * t) + r31) * p.ldb + 8 * h;
#pragma unroll
        for (int k = 0; k < KS; ++k) wf[t][k] = *reinterpret_cast<const bf16x8*>(wrow + 16 * k);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(p.bias + 32 * (wave + 4 * t) + 4 * h + 8 * gq);
#pragma unroll
            for (int e = 0; e < 4; ++e) seed[t][4 * gq + e] = b4[e];
        }
    }
    bf16x8 idf[2];                                                           // A[n][k'] = (n == 16 j + k'), k' = 8h + i
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int dd = r31 - 16 * j - 8 * h;
        unsigned v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (dd == 2 * q ? 0x3F80u : 0u) | (dd == 2 * q + 1 ? 0x3F800000u : 0u);
        const uint4 u = make_uint4(v[0], v[1], v[2], v[3]);
        idf[j] = __builtin_bit_cast(bf16x8, u);
    }
    const float neg = p.act == DHAUG_ACT_RELU ? 0.0f : (p.act == DHAUG_ACT_LRELU ? p.slope : 1.0f);
    const int lfx = r31 * XP | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4);                   // ^ (k << 5): chunk 2k+h of row
    const int lrx = (r31 * F_PITCH | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4)) ^ (wave << 6);   // ^ (t << 8 | j << 5)
    const int lep = r31 * F_PITCH | (((4 * wave) ^ x) << 4) | (h << 3);                  // ^ ((16t+g) << 4)
    constexpr int XHALF = 32 * XP, HALF = 32 * F_PITCH;                                   // second 32-row half of an image

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f_lds_barrier();
    int buf = 0;
    for (; mt < mtiles; mt += g, buf ^= 1) {
        const unsigned char* X = sX + buf * F_IMG;
        const unsigned char* R = sR + buf * F_IMG;
        f32x16 acc[2][2];
        bf16x8 fx[4];
        constexpr int NST = 2 * KS, D = 3;                                   // (half, k) steps; fragment prefetch distance
        auto fx_load = [&](int st) { fx[st & 3] = *reinterpret_cast<const bf16x8*>(X + (lfx ^ ((st % KS) << 5)) + (st / KS) * XHALF); };
#pragma unroll
        for (int st = 0; st < D; ++st) fx_load(st);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int hf = st / KS, k = st % KS;
            if (st + D < NST) fx_load(st + D);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[hf][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][k], fx[st & 3], k == 0 ? seed[t] : acc[hf][t], 0, 0, 0);
        }
        if (RES) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 rf = *reinterpret_cast<const bf16x8*>(R + (lrx ^ (t << 8 | j << 5)) + hf * HALF);
                        acc[hf][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf[j], rf, acc[hf][t], 0, 0, 0);
                    }
        }
        // epilogue -> output image
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = acc[hf][t][4 * gq + e];
                        v[e] = fmaxf(a, a * neg);
                    }
                    if (MASK) {                                              // this lane's 4 mask-source values (bf16)
                        const uint2 y = *reinterpret_cast<const uint2*>(R + (lep ^ ((16 * t + gq) << 4)) + hf * HALF);
                        const short y0 = (short)(y.x & 0xffffu), y1 = (short)(y.x >> 16), y2 = (short)(y.y & 0xffffu), y3 = (short)(y.y >> 16);
                        v[0] = y0 > 0 ? v[0] : v[0] * p.dneg;                // a positive bf16 is a positive int16
                        v[1] = y1 > 0 ? v[1] : v[1] * p.dneg;
                        v[2] = y2 > 0 ? v[2] : v[2] * p.dneg;
                        v[3] = y3 > 0 ? v[3] : v[3] * p.dneg;
                    }
                    uint2 o;
                    o.x = f_pack_bf16x2(v[0], v[1]);
                    o.y = f_pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(sO + (lep ^ ((16 * t + gq) << 4)) + hf * HALF) = o;
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the next tile's rows have landed
        f_lds_barrier();                                                     // output image complete; image buf is free
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 o[RCH];
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const int q = tid + 256 * i, row = q >> 5, c = q & 31;
            o[i] = *reinterpret_cast<const u32x4*>(sO + row * F_PITCH + ((c ^ (row & 15)) << 4));
        }
        copy_tile(mt + 2 * g, buf);
        f_lds_barrier();                                                     // output image free again
        const long long m0 = mt * F_BM;
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const int q = tid + 256 * i, row = q >> 5, c = q & 31;
            *reinterpret_cast<u32x4*>(p.cb + (m0 + row) * p.ldcb + c * 8) = o[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // no copy may land in LDS after the workgroup is gone
}

// ---------------------------------------------------------------------------------------------------------------
// The same layer with the tile's serial chain cut in two (gemm_nt256_kernel above: 10.8 us of movement + 9 us of MFMA
// phases + 7 us of store phases, added up).  512 threads, two roles, one LDS-only barrier per 32-row tile:
//   waves 0..3 (compute): MFMAs of tile i from its LDS images, epilogue (bias from an LDS copy, residual on the matrix
//                         pipe, activation, mask), bf16 result into output image i % 2;
//   waves 4..7 (movers) : request tile i+NX-1's rows into the image tile i-1 released (global_load_lds_dwordx4), stream
//                         tile i-1's output image out as whole 512-byte rows, and make sure tile i+1 has landed before
//                         the barrier.
// Two waves per SIMD: 256 registers each, so a compute wave keeps the weights (2 * KS fragments = 128 registers at
// K = 256) but no bias seeds.  Tile j lives in operand image j % NX (NX = 4 without a second operand, 3 with one).
// LDS: NX (x2 with a second operand) images of 16 KB + two output images + bias 1 KB = 97 / 129 KB.
// ---------------------------------------------------------------------------------------------------------------
// MODE bit 0: residual, bit 1: activation-backward mask; both (3): (A B^T + res) * act'(mask), the input-gradient step of a
// residual block (gz1 W1 + gz2) * relu'(x) and its tangent twin -- three operand streams, so the ring holds 2 tiles
template <int KS, int MODE>
__global__ __launch_bounds__(512, 1) void gemm_nt256s_kernel(GemmArgs p) {
    // MODE bit 2: the mask comes as a sign-bit array (one dword per lane and tile, read by the compute waves themselves: no
    // mask image, 1/16 of the bytes; the ring keeps the depth of the mode without a mask)
    constexpr bool RES = (MODE & 1) != 0, MASK = (MODE & 2) != 0, BITS = (MODE & 4) != 0;
    static_assert(!(MASK && BITS), "one mask source");
    constexpr int NSEC = (RES ? 1 : 0) + (MASK ? 1 : 0);
    constexpr bool SECOND = NSEC != 0;
    static_assert(KS == 8 || KS == 16, "K = 128 or 256");
    constexpr int BM = 32;
    constexpr int NX = 4 - NSEC;                                             // operand images in the ring: NX - 1 tiles ahead
    constexpr int S = 2 * KS, XP = S * 16;                                   // chunks per operand row, operand pitch (bytes)
    constexpr int IMG = BM * F_PITCH;                                        // bytes per image (operand images use XP <= 512)
    constexpr int NCR1 = BM * 32 / 256;                                      // copies per mover lane of ONE second operand
    constexpr int NCX = BM * S / 256, NCR = NSEC * NCR1;                     // copies per mover lane: operand / second operands
    constexpr int RCH = BM * 32 / 256;                                       // output chunks per mover lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;                                                // [NX][IMG]
    unsigned char* sR = smem + NX * IMG;                                     // [NX][IMG]  residual (or the mask when there is no residual)
    unsigned char* sM = RES && MASK ? smem + 2 * NX * IMG : sR;              // [NX][IMG]  mask
    unsigned char* sO = smem + (1 + NSEC) * NX * IMG;                        // [2][IMG]
    float* sBias = reinterpret_cast<float*>(sO + 2 * IMG);                   // [256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mover = wave >= 4;
    const int cw = wave & 3;                                                 // index within the role
    const int r31 = lane & 31, h = lane >> 5, x = lane & 15;
    const long long mtiles = p.M / BM, g = gridDim.x;
    const long long mt0 = blockIdx.x;
    if (mt0 >= mtiles) return;
    const int nt = (int)((mtiles - mt0 + g - 1) / g);                        // tiles of this workgroup

    if (mover) {
        auto copy_tile = [&](int i, int buf) {                               // asynchronous, counted in vmcnt
            const long long m0 = (mt0 + (long long)i * g) * BM;
            constexpr int RW = 1024 / XP;                                    // rows per wave instruction
#pragma unroll
            for (int q = 0; q < NCX; ++q) {
                const int row0 = (cw * NCX + q) * RW, row = row0 + lane / S, c = (lane % S) ^ (row & 15);
                f_copy16(p.A + (m0 + row) * p.lda + c * 8, sX + buf * IMG + row0 * XP);
            }
            if (RES) {
#pragma unroll
                for (int q = 0; q < NCR1; ++q) {
                    const int row0 = (cw * NCR1 + q) * 2, row = row0 + (lane >> 5), c = (lane & 31) ^ (row & 15);
                    f_copy16(p.res + (m0 + row) * p.ld_res + c * 8, sR + buf * IMG + row0 * F_PITCH);
                }
            }
            if (MASK) {
#pragma unroll
                for (int q = 0; q < NCR1; ++q) {
                    const int row0 = (cw * NCR1 + q) * 2, row = row0 + (lane >> 5), c = (lane & 31) ^ (row & 15);
                    f_copy16(p.dmask + (m0 + row) * p.ld_dmask + c * 8, sM + buf * IMG + row0 * F_PITCH);
                }
            }
        };
#pragma unroll
        for (int i = 0; i < NX - 1; ++i)
            if (i < nt) copy_tile(i, i);
        if (cw == 0) {                                                       // bias -> LDS (zeros without one)
#pragma unroll
            for (int q = 0; q < 4; ++q) sBias[lane + 64 * q] = p.bias != nullptr ? p.bias[lane + 64 * q] : 0.0f;
        }
        if (nt >= NX - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NX - 2) * (NCX + NCR)) : "memory");   // tile 0 landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f_lds_barrier();
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int mtid = tid - 256;
        for (int i = 0; i <= nt; ++i) {                                      // iteration i: tile i is being computed
            // tile j lives in image j % NX.  The image of tile i-1 was released at the last barrier (image NX-1 has not
            // been used yet in iteration 0): it takes tile i+NX-1, so NX-1 tiles are in flight or landed ahead of tile i
            if (i + NX - 1 < nt) copy_tile(i + NX - 1, (i + NX - 1) % NX);
            if (i >= 1) {                                                    // stream tile i-1 out
                const unsigned char* O = sO + ((i - 1) & 1) * IMG;
                const long long m0 = (mt0 + (long long)(i - 1) * g) * BM;
                u32x4 o[RCH];
#pragma unroll
                for (int q = 0; q < RCH; ++q) {
                    const int e = mtid + 256 * q, row = e >> 5, c = e & 31;
                    o[q] = *reinterpret_cast<const u32x4*>(O + row * F_PITCH + ((c ^ (row & 15)) << 4));
                }
#pragma unroll
                for (int q = 0; q < RCH; ++q) {
                    const int e = mtid + 256 * q, row = e >> 5, c = e & 31;
                    *reinterpret_cast<u32x4*>(p.cb + (m0 + row) * p.ldcb + c * 8) = o[q];
                }
            }
            if (i == nt) break;
            // tile i+1 must be in LDS when the compute waves leave the barrier.  vmcnt retires in issue order; younger
            // than tile i+1's copies are the copies of tiles i+2 .. i+NX-1 and min(i, NX-1) sets of row stores.  Near the
            // end, where some of those tiles do not exist, everything is drained instead.
            const int ks = i < NX - 1 ? i : NX - 1;                          // sets of row stores younger than tile i+1's copies
            if (i + NX - 1 >= nt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (ks == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NX - 2) * (NCX + NCR)) : "memory");
            else if (ks == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NX - 2) * (NCX + NCR) + RCH) : "memory");
            else if (ks == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NX - 2) * (NCX + NCR) + 2 * RCH) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NX - 2) * (NCX + NCR) + 3 * RCH) : "memory");
            f_lds_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- compute waves ----
    bf16x8 wf[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint16_t* wrow = p.B + (long long)(32 * (cw + 4 * t) + r31) * p.ldb + 8 * h;
#pragma unroll
        for (int k = 0; k < KS; ++k) wf[t][k] = *reinterpret_cast<const bf16x8*>(wrow + 16 * k);
    }
    bf16x8 idf[2];                                                           // A[n][k'] = (n == 16 j + k'), k' = 8h + i
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int dd = r31 - 16 * j - 8 * h;
        unsigned v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (dd == 2 * q ? 0x3F80u : 0u) | (dd == 2 * q + 1 ? 0x3F800000u : 0u);
        const uint4 u = make_uint4(v[0], v[1], v[2], v[3]);
        idf[j] = __builtin_bit_cast(bf16x8, u);
    }
    const bool leaky = p.act == DHAUG_ACT_LRELU;
    const float neg = p.slope;
    const uint32_t lb = p.act == DHAUG_ACT_RELU ? 0u : 0x80008000u;         // packed int16 lower bound
    const int lfx = r31 * XP | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4);                   // ^ (k << 5): chunk 2k+h of row
    const int lrx = (r31 * F_PITCH | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4)) ^ (cw << 6);     // ^ (t << 8 | j << 5)
    const int lep = r31 * F_PITCH | (((4 * cw) ^ x) << 4) | (h << 3);                    // ^ ((16t+g) << 4)
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // sign bits of this wave's two slices for the lane's row of tile j: word ((tile * 4 + cw) * 64 + lane), one tile ahead
    uint32_t bw_next = BITS ? p.dbits[((mt0 * 4) + cw) * 64 + lane] : 0u;
    f_lds_barrier();                                                         // tile 0 and the bias are in LDS
    for (int i = 0; i < nt; ++i) {
        const unsigned char* X = sX + (i % NX) * IMG;
        const unsigned char* R = sR + (i % NX) * IMG;
        const unsigned char* Mk = sM + (i % NX) * IMG;
        unsigned char* O = sO + (i & 1) * IMG;
        const uint32_t bw = bw_next;
        if (BITS && i + 1 < nt) bw_next = p.dbits[(((mt0 + (long long)(i + 1) * g) * 4) + cw) * 64 + lane];
        f32x16 acc[2];
        bf16x8 fx[4];
        constexpr int D = 3;
#pragma unroll
        for (int k = 0; k < D; ++k) fx[k] = *reinterpret_cast<const bf16x8*>(X + (lfx ^ (k << 5)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            // pinned: left alone the scheduler sinks each fragment read next to its MFMA and waits out the LDS latency
            // at every k-step
            if (k + D < KS) fx[(k + D) & 3] = *reinterpret_cast<const bf16x8*>(X + (lfx ^ ((k + D) << 5)));
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][k], fx[k & 3], k == 0 ? zero : acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (RES) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 rf = *reinterpret_cast<const bf16x8*>(R + (lrx ^ (t << 8 | j << 5)));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf[j], rf, acc[t], 0, 0, 0);
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias + 32 * (cw + 4 * t) + 4 * h + 8 * gq);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[t][4 * gq + e] + b4[e];
                if (leaky) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * neg);
                }
                if (MASK) {                                                  // this lane's 4 mask-source values (bf16)
                    const uint2 y = *reinterpret_cast<const uint2*>(Mk + (lep ^ ((16 * t + gq) << 4)));
                    const short y0 = (short)(y.x & 0xffffu), y1 = (short)(y.x >> 16), y2 = (short)(y.y & 0xffffu), y3 = (short)(y.y >> 16);
                    v[0] = y0 > 0 ? v[0] : v[0] * p.dneg;                    // a positive bf16 is a positive int16
                    v[1] = y1 > 0 ? v[1] : v[1] * p.dneg;
                    v[2] = y2 > 0 ? v[2] : v[2] * p.dneg;
                    v[3] = y3 > 0 ? v[3] : v[3] * p.dneg;
                }
                if (BITS) {                                                  // pairs 8t + 2gq, +1: even element at bit p, odd at 16 + p
                    const int p0 = 8 * t + 2 * gq;
                    v[0] = ((bw >> p0) & 1u) ? v[0] : v[0] * p.dneg;
                    v[1] = ((bw >> (16 + p0)) & 1u) ? v[1] : v[1] * p.dneg;
                    v[2] = ((bw >> (p0 + 1)) & 1u) ? v[2] : v[2] * p.dneg;
                    v[3] = ((bw >> (17 + p0)) & 1u) ? v[3] : v[3] * p.dneg;
                }
                // ReLU / identity on the packed pair: a negative bf16 is a negative int16 (lower bound 0 or INT16_MIN)
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                uint2 o;
                o.x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, f_pack_bf16x2(v[0], v[1])),
                                                                             __builtin_bit_cast(s16x2, lb)));
                o.y = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, f_pack_bf16x2(v[2], v[3])),
                                                                             __builtin_bit_cast(s16x2, lb)));
                *reinterpret_cast<uint2*>(O + (lep ^ ((16 * t + gq) << 4))) = o;
            }
        f_lds_barrier();                                                     // output image i complete, operand image i released
    }
}

// ---------------------------------------------------------------------------------------------------------------
// TWO 256 -> 256 layers of a residual block in one launch (r3):
//        Y1 = (X  W1^T) * act'(bits1)            Y2 = (Y1 W2^T + X) * act'(bits2)
// = the backward step through myResNet (gz1 = gz2 W_fc2 * relu'(h); gz_in = (gz1 W_fc1 + gz2) * relu'(x),
// R/models_Fk_GAN/special_operate.py:490-510 under autograd) and its tangent twin (uh = u W_fc1^T * relu'(h);
// u' = (uh W_fc2^T + u) * relu'(y)).  As two gemm_nt256s launches the pair moves 500 MB per 3B-row block (X in, Y1 out | Y1 in,
// X in as the skip, Y2 out); here Y1 stays in LDS between the layers and X is read once: 300 MB.
// 512 threads, one persistent workgroup per CU, 32-row tiles, and the two layers as a two-stage pipeline over the tiles:
//   waves 0..3 (stage A): tile i  : X image -> matrix pipe (W1 resident in registers) -> mask -> Y1 image (LDS)
//   waves 4..7 (stage B): tile i-1: Y1 image -> matrix pipe (W2 resident) + X image on the identity fragments -> mask -> Y2 image
// so a SIMD hosts one wave of each stage (256 registers each: 128 of weights).  There are no mover waves: every wave requests
// its share of tile i+2 (LDS-DMA, inline assembly: see tn_copy16) and streams its share of the finished images (Y1 of tile i-1,
// Y2 of tile i-2) out as whole rows at the top of an iteration, then computes; one LDS barrier per tile.  The sign bits of a
// tile (1 KB per layer) travel with the tile.  LDS: 4 X images + 2 Y1 + 2 Y2 (16 KB each) + 4 x 2 KB of bits + 1 KB = 137 KB.
// ---------------------------------------------------------------------------------------------------------------
struct Block2One {
    const uint16_t* W1; long long ldw1;
    const uint16_t* W2; long long ldw2;
    const uint32_t* bits1; const uint32_t* bits2;
    uint16_t* Y1; long long ldy1;
    uint16_t* Y2; long long ldy2;
};
// A STACK of blocks (<= DHAUG_BLOCK2_MAX): block b + 1 takes block b's Y2 as its X.  Row tiles are independent, and a
// workgroup owns the same tiles in every block, so it walks its tiles through block 0, reloads its weight registers, walks
// them through block 1 (reading back rows it wrote itself: drained + L1 invalidated in between) ... -- one launch, one
// pipeline fill and one set of launch latencies for the three blocks of a residual stack (~12 us each as separate launches).
struct Block2Args {
    int nb;
    const uint16_t* X; long long ldx;
    Block2One b[DHAUG_BLOCK2_MAX];
    long long M;
    float dneg;
};
constexpr int B2_BM = 32, B2_IMG = B2_BM * F_PITCH, B2_NX = 4;
constexpr int B2_LDS = (B2_NX + 4) * B2_IMG + B2_NX * 2048 + 1024;

#ifdef DHAUG_PIPE_TIMING
#define B2_STAMP(i) if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (i) < 128) g_pipe_stamps[(threadIdx.x >> 8) * 128 + (i)] = (long long)__builtin_readcyclecounter();
#else
#define B2_STAMP(i)
#endif
__global__ __launch_bounds__(512, 1) void gemm_block2_kernel(Block2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    B2_STAMP(0)
    unsigned char* sX = smem;                                                // [4][IMG]
    unsigned char* sG = smem + B2_NX * B2_IMG;                               // [2][IMG]  Y1
    unsigned char* sO = sG + 2 * B2_IMG;                                     // [2][IMG]  Y2
    unsigned char* sB = sO + 2 * B2_IMG;                                     // [4][2][1 KB] sign bits (layer 1 | layer 2)
    unsigned char* sScratch = sB + B2_NX * 2048;                             // 1 KB: landing zone of the copies that only keep the count uniform
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stageB = wave >= 4;
    const int cw = wave & 3;
    const int r31 = lane & 31, h = lane >> 5, x = lane & 15;
    const long long mtiles = p.M / B2_BM, g = gridDim.x;
    const long long mt0 = blockIdx.x;
    if (mt0 >= mtiles) return;
    const int nt = (int)((mtiles - mt0 + g - 1) / g);                        // tiles of this workgroup

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int lfx = r31 * F_PITCH | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4);                  // ^ (k << 5): chunk 2k+h of row
    const int lrx = (r31 * F_PITCH | ((x >> 1) << 5) | ((h ^ (x & 1)) << 4)) ^ (cw << 6);     // ^ (t << 8 | j << 5)
    const int lep = r31 * F_PITCH | (((4 * cw) ^ x) << 4) | (h << 3);                        // ^ ((16t+g) << 4)
    const unsigned char __attribute__((address_space(4)))* ka =
        (const unsigned char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    typedef const Block2One __attribute__((address_space(4)))* OnePtr;       // (indexed dynamically: read from the kernarg segment)
    const int nblk = p.nb;
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
    OnePtr P = (OnePtr)(ka + __builtin_offsetof(Block2Args, b)) + blk;
    const uint16_t* Xb = blk == 0 ? p.X : (const uint16_t*)P[-1].Y2;
    const long long ldxb = blk == 0 ? p.ldx : P[-1].ldy2;
    const uint32_t* bits1 = P->bits1;
    const uint32_t* bits2 = P->bits2;
    // three copies per wave and tile: 2 of the 16 KB operand tile, one of the bits (waves 0 / 1: layer 1 / 2; the others
    // repeat one into the scratch area so that every wave's vmcnt arithmetic is the same)
    auto copy_tile = [&](int i) {
        const long long tile = mt0 + (long long)i * g, m0 = tile * B2_BM;
        const int buf = i % B2_NX;
        int ln = lane;
        asm volatile("" : "+v"(ln));                 // lane-derived offsets are recomputed here, not parked in (spilled) registers
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row0 = (wave * 2 + q) * 2, row = row0 + (ln >> 5), c = (ln & 31) ^ (row & 15);
            tn_copy16(Xb + (m0 + row) * ldxb + c * 8, sX + buf * B2_IMG + row0 * F_PITCH);
        }
        const uint32_t* bsrc = ((wave & 1) ? bits2 : bits1) + tile * 256 + ln * 4;
        tn_copy16(bsrc, wave < 2 ? sB + buf * 2048 + wave * 1024 : sScratch);
    };
    // this stage's weights: feature slices cw and cw + 4, resident for the whole block.  *(r5)* They come THROUGH LDS: the eight
    // waves copy the 256 x 256 matrix as whole rows (LDS-DMA, two rows per instruction, the images' chunk swizzle) into the 128 KB
    // the tile images will use, the stage's waves read their fragments from there like activation fragments; W1 first, then W2.
    // Loaded straight from global memory a fragment is 16 bytes per lane of 32 different rows -- four separate accesses per quad for
    // the address coalescer: phase stamps (tools/stamp_block2.py) showed 9 600 (stage A) / 19 500 (stage B) clocks until the 32
    // loads per lane were ISSUED, 10 us of the 25 a block costs whatever its batch.  The store drain of the previous block of a
    // stack (its rows are this block's operand) waits behind the first copy instead of in front of it.
    bf16x8 wf[2][16];
#pragma unroll 1
    for (int ws = 0; ws < 2; ++ws) {                                         // ws = 0: W1 (stage A's waves read), 1: W2 (stage B's)
        const uint16_t* W = ws ? P->W2 : P->W1;
        const long long ldw = ws ? P->ldw2 : P->ldw1;
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row0 = (wave * 16 + q) * 2, row = row0 + (ln >> 5), c = (ln & 31) ^ (row & 15);
                tn_copy16(W + (long long)row * ldw + c * 8, smem + row0 * F_PITCH);
            }
        }
        if (ws == 0 && blk > 0) asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1" ::: "memory");   // (+ the previous block's row stores)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f_lds_barrier();
        if ((ws == 1) == stageB) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) wf[t][k] = *reinterpret_cast<const bf16x8*>(smem + 32 * (cw + 4 * t) * F_PITCH + (lfx ^ (k << 5)));
        }
        f_lds_barrier();                                                     // (fragments in registers: the area is free again)
    }
    if (0 < nt) copy_tile(0);
    if (1 < nt) copy_tile(1);
    B2_STAMP(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B2_STAMP(2)
    f_lds_barrier();                                                         // tiles 0 and 1 are in LDS
    B2_STAMP(3)

    for (int i = 0; i <= nt + 1; ++i) {
        B2_STAMP(8 + 8 * i)
        // ---- tile i+2's rows and bits into the image tile i-2 left (stage B was done with it at the last barrier).  (A fifth
        // image / two tiles in flight was measured: 80.4 against 77.6 us at 3B rows -- the reads are not what waits.)
        if (i + 2 < nt) copy_tile(i + 2);
        // ---- finished images out as whole 512-byte rows (four 16-byte chunks per thread): stage A's waves take Y1 of tile i-1
        // AFTER their matrix phase, stage B's take Y2 of tile i-2 BEFORE theirs -- the two waves of a SIMD are then in
        // different phases (one on the matrix pipe, the other on LDS / the vector-memory path) instead of queueing for the
        // same unit twice per tile.  (Giving ALL copies to stage A's waves and ALL row stores to stage B's -- so that a wait for
        // copies never waits for older stores in the same in-order queue -- was measured too: 88.5 us, stage B's chain becomes
        // the long one.)  Measured on the first version (every wave streaming first; phases switched off one by one
        // through a run-time flag, 3B rows): they simply added up -- 21.7 us of launch + loop, + 12 epilogue, + 23 matrix, + 17 row
        // stores, + 12 copies = 86; with the offset 77.6.
        auto stream_out = [&]() {
            const int jj = stageB ? i - 2 : i - 1;
            if (jj < 0 || jj >= nt) return;                                  // (wave-uniform)
            const unsigned char* I = (stageB ? sO : sG) + (jj & 1) * B2_IMG;
            uint16_t* Y = stageB ? P->Y2 : P->Y1;
            const long long ldy = stageB ? P->ldy2 : P->ldy1;
            int st = tid & 255;
            asm volatile("" : "+v"(st));             // (as in copy_tile)
#pragma unroll
            for (int q0 = 0; q0 < 4; q0 += 2) {                              // (two at a time: 8 registers, not 16)
                u32x4 o[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = st + 256 * (q0 + q), row = e >> 5, c = e & 31;
                    o[q] = *reinterpret_cast<const u32x4*>(I + row * F_PITCH + ((c ^ (row & 15)) << 4));
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = st + 256 * (q0 + q), row = e >> 5, c = e & 31;
                    *reinterpret_cast<u32x4*>(Y + ((mt0 + (long long)jj * g) * B2_BM + row) * ldy + c * 8) = o[q];
                }
            }
        };
        B2_STAMP(9 + 8 * i)
        if (stageB) stream_out();
        B2_STAMP(10 + 8 * i)
        // ---- compute: stage A tile i, stage B tile i-1
        const int j = stageB ? i - 1 : i;
        if (j >= 0 && j < nt) {                                              // (wave-uniform)
            const unsigned char* Xj = sX + (j % B2_NX) * B2_IMG;
            const unsigned char* S = stageB ? sG + (j & 1) * B2_IMG : Xj;    // this stage's operand image
            unsigned char* D = (stageB ? sO : sG) + (j & 1) * B2_IMG;        // ... and its result image
            const uint32_t bw = *reinterpret_cast<const uint32_t*>(sB + (j % B2_NX) * 2048 + (stageB ? 1024 : 0) + (cw * 64 + lane) * 4);
            f32x16 acc[2];
            bf16x8 fx[4];
            constexpr int DEP = 3;
#pragma unroll
            for (int k = 0; k < DEP; ++k) fx[k] = *reinterpret_cast<const bf16x8*>(S + (lfx ^ (k << 5)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k + DEP < 16) fx[(k + DEP) & 3] = *reinterpret_cast<const bf16x8*>(S + (lfx ^ ((k + DEP) << 5)));
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][k], fx[k & 3], k == 0 ? zero : acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (stageB) {                                                    // + X (the skip), exact on the matrix pipe
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    // identity fragment A[n][k'] = (n == 16 jj + k'), k' = 8h + i: built here (8 registers not kept across the tile)
                    const int dd = r31 - 16 * jj - 8 * h;
                    unsigned v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (dd == 2 * q ? 0x3F80u : 0u) | (dd == 2 * q + 1 ? 0x3F800000u : 0u);
                    const uint4 uu = make_uint4(v[0], v[1], v[2], v[3]);
                    const bf16x8 idf = __builtin_bit_cast(bf16x8, uu);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 rf = *reinterpret_cast<const bf16x8*>(Xj + (lrx ^ (t << 8 | jj << 5)));
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf, rf, acc[t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[t][4 * gq + e];
                    const int p0 = 8 * t + 2 * gq;                           // pairs 8t + 2gq, +1: even element at bit p, odd at 16 + p
                    v[0] = ((bw >> p0) & 1u) ? v[0] : v[0] * p.dneg;
                    v[1] = ((bw >> (16 + p0)) & 1u) ? v[1] : v[1] * p.dneg;
                    v[2] = ((bw >> (p0 + 1)) & 1u) ? v[2] : v[2] * p.dneg;
                    v[3] = ((bw >> (17 + p0)) & 1u) ? v[3] : v[3] * p.dneg;
                    uint2 o;
                    o.x = f_pack_bf16x2(v[0], v[1]);
                    o.y = f_pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(D + (lep ^ ((16 * t + gq) << 4))) = o;
                }
        }
        B2_STAMP(11 + 8 * i)
        if (!stageB) stream_out();
        B2_STAMP(12 + 8 * i)
        if (i > nt) break;
        // ---- tile i+1 must be in LDS when the barrier opens.  vmcnt retires in issue order; per wave and iteration the
        // order is [3 copies] ... [4 row stores]: younger than tile i+1's copies (issued at the top of iteration i-1) are that
        // iteration's 4 stores, this one's 3 copies and 4 stores.  The first iterations and the last ones (where some of those
        // do not exist) drain instead.
        // *(r5)* ... are counted too instead of drained: a drain waits for the acknowledgement of the row stores issued a moment
        // ago, a full memory round trip per iteration -- six of the ten iterations of a B-row tangent block (eight tiles per
        // workgroup), and most of the 25 us a block costs at ONE tile per workgroup (tools/time_block2_sweep.py).  Younger than tile
        // i+1's copies: the row stores of iteration i-1 (stage A: tile i-2's image, stage B: tile i-3's), this iteration's copies
        // (tile i+2) and row stores (stage A: tile i-1, stage B: tile i-2), each where the tile exists.  No tile i+1 (or i = 0: tiles
        // 0 and 1 were waited for in front of the loop): nothing to wait for -- the images are LDS traffic, ordered by the barrier.
        int young = -1;
        if (i >= 1 && i + 1 < nt) {
            const int sp = stageB ? i - 3 : i - 2, sc = stageB ? i - 2 : i - 1;
            young = ((sp >= 0 && sp < nt) ? 4 : 0) + ((i + 2 < nt) ? 3 : 0) + ((sc >= 0 && sc < nt) ? 4 : 0);
        }
        switch (young) {                                                     // (wave-uniform)
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            default: break;
        }
        B2_STAMP(13 + 8 * i)
        f_lds_barrier();
        B2_STAMP(14 + 8 * i)
    }
    B2_STAMP(4)
    // every row this workgroup stored is in L2 (its next block reads them back: through a clean L1), every image is free
    // (the rows this workgroup stored are the next block's operand: their drain -- and the L1 invalidate -- waits behind the next
    // block's first weight copy; the last block's stores are completed by the end of the kernel)
    B2_STAMP(5)
    f_lds_barrier();                                                         // every image is free
    B2_STAMP(6)
    }
}

template <int KS, int MODE>
int launch_nt256s_mode(hipStream_t s, const GemmArgs& p) {
    constexpr int NSEC = (MODE & 1) + ((MODE >> 1) & 1);                   // (bit 2, the sign-bit mask, has no image)
    constexpr int BM = 32, NX = 4 - NSEC;
    constexpr int LDS = ((1 + NSEC) * NX + 2) * BM * F_PITCH + 1024;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256s_kernel<KS, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const long long mtiles = p.M / BM;
    const unsigned grid = dhaug_persistent_grid(mtiles);
    hipLaunchKernelGGL((gemm_nt256s_kernel<KS, MODE>), dim3(grid), dim3(512), LDS, s, p);
    return dhaug_launch_status();
}

template <int KS>
int launch_nt256s(hipStream_t s, const GemmArgs& p) {
    if constexpr (KS == 16) {
        if (p.dbits != nullptr) return p.res != nullptr ? launch_nt256s_mode<KS, 5>(s, p) : launch_nt256s_mode<KS, 4>(s, p);
    }
    if (p.dmask != nullptr && p.res != nullptr) return launch_nt256s_mode<KS, 3>(s, p);
    if (p.dmask != nullptr) return launch_nt256s_mode<KS, 2>(s, p);
    if (p.res != nullptr) return launch_nt256s_mode<KS, 1>(s, p);
    return launch_nt256s_mode<KS, 0>(s, p);
}

template <int KS>
int launch_nt256(hipStream_t s, const GemmArgs& p) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256_kernel<KS, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256_kernel<KS, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256_kernel<KS, 2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const long long mtiles = p.M / F_BM;
    const unsigned grid = dhaug_persistent_grid(mtiles);
    if (p.dmask != nullptr) hipLaunchKernelGGL((gemm_nt256_kernel<KS, 2>), dim3(grid), dim3(256), F_LDS_BYTES, s, p);
    else if (p.res != nullptr) hipLaunchKernelGGL((gemm_nt256_kernel<KS, 1>), dim3(grid), dim3(256), F_LDS_BYTES, s, p);
    else hipLaunchKernelGGL((gemm_nt256_kernel<KS, 0>), dim3(grid), dim3(256), F_LDS_BYTES, s, p);
    return dhaug_launch_status();
}

template <int KSTEPS>
int launch_ws(hipStream_t s, const GemmArgs& p) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ws_kernel<KSTEPS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WS_BUF_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const long long ntiles = (p.W + WS_BN - 1) / WS_BN, mtiles = (p.M + WS_BM - 1) / WS_BM;
    long long gx = 2 * (long long)dhaug_persistent_grid(256) / ntiles;          // two workgroups per CU
    if (gx < 1) gx = 1;
    if (gx > mtiles) gx = mtiles;
    hipLaunchKernelGGL(gemm_nt_ws_kernel<KSTEPS>, dim3((unsigned)gx, (unsigned)ntiles), dim3(256), 2 * WS_BUF_BYTES, s, p);
    return dhaug_launch_status();
}

template <typename Kern>
int launch_nt(Kern kern, long long grid, size_t lds, hipStream_t s, const GemmArgs& p) {
    static bool configured = false;          // one attribute call per kernel instantiation
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, p);
    return dhaug_launch_status();
}

}  // namespace

extern "C" {

static int launch_wide(hipStream_t s, const GemmArgs& p) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const long long grid = (((p.M + W_BM - 1) / W_BM + 7) / 8 * 8) * ((p.W + W_BN - 1) / W_BN);     // (row blocks padded to eight: XCD map)
    hipLaunchKernelGGL(gemm_nt_wide_kernel, dim3((unsigned)grid), dim3(512), W_LDS, s, p);
    return dhaug_launch_status();
}

static int gemm_bf16_impl(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias,
                          const uint16_t* residual, int64_t ld_res, const float* residual_f32, int64_t ld_res_f32,
                          uint16_t* c_bf16, int64_t ldc_bf16, int64_t n_pad_zero,
                          float* c_f32, int64_t ldc_f32, int64_t M, int64_t N, int64_t K, int act, float slope,
                          const uint16_t* dmask, int64_t ld_dmask, float dneg, bool* mask_done, void* stream,
                          const float* dmaskf = nullptr, int64_t ld_dmaskf = 0, GemmArgs* collect = nullptr) {
    DHAUG_CHECK(M >= 0 && N >= 1 && K >= 16, DHAUG_EINVAL);
    DHAUG_CHECK(act >= DHAUG_ACT_NONE && act <= DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(A); DHAUG_CHECK_PTR(B);
    DHAUG_CHECK(c_bf16 != nullptr || c_f32 != nullptr, DHAUG_EINVAL);
    DHAUG_CHECK(K % 16 == 0, DHAUG_EUNSUPPORTED);
    DHAUG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K, DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(A) && dhaug_aligned16(B), DHAUG_EALIGN);
    if (residual) DHAUG_CHECK(ld_res % 8 == 0 && ld_res >= N && dhaug_aligned16(residual), DHAUG_EALIGN);
    if (residual_f32) DHAUG_CHECK(ld_res_f32 >= N, DHAUG_EINVAL);
    if (c_bf16) DHAUG_CHECK(ldc_bf16 % 8 == 0 && ldc_bf16 >= N && n_pad_zero <= ldc_bf16 && dhaug_aligned16(c_bf16), DHAUG_EALIGN);
    if (c_f32) DHAUG_CHECK(ldc_f32 >= N && ((ldc_f32 & 3) != 0 || dhaug_aligned16(c_f32)), DHAUG_EALIGN);
    GemmArgs p{A, lda, B, ldb, bias, residual, ld_res, residual_f32, ld_res_f32, c_bf16, ldc_bf16, c_bf16 ? (n_pad_zero > N ? n_pad_zero : N) : 0,
               c_f32, ldc_f32, M, N, K, N, act, slope, nullptr, 0, 1.0f, nullptr, nullptr, 0};
    hipStream_t s = (hipStream_t)stream;
    const long long width = (c_bf16 && p.npad > N) ? p.npad : N;
    p.W = width;
    if (collect != nullptr) {                                    // a member of a group (dhaug_gemm_bf16_group): checked, not launched
        DHAUG_CHECK(M > 0 && width > 64 && K >= 64 && lda >= 64 && ldb >= 64 && dmaskf == nullptr, DHAUG_EUNSUPPORTED);
        if (dmask != nullptr) { p.dmask = dmask; p.ld_dmask = ld_dmask; p.dneg = dneg; *mask_done = true; }
        *collect = p;
        return DHAUG_OK;
    }
    if (dmaskf != nullptr) {                                     // (fp32 mask: applied by nt_store_tile -- the kernels that have it)
        DHAUG_CHECK(dmask == nullptr && ld_dmaskf >= N, DHAUG_EINVAL);
        p.dmaskf = dmaskf; p.ld_dmaskf = ld_dmaskf; p.dneg = dneg;
    }
    // the training path's 256-wide layers
    if (dmaskf == nullptr && N == 256 && width == 256 && K <= 256 && M % F_BM == 0 && c_bf16 != nullptr && c_f32 == nullptr && residual_f32 == nullptr &&
        (bias == nullptr || dhaug_aligned16(bias)) && getenv("DHAUG_GEMM_GENERIC") == nullptr && getenv("DHAUG_GEMM_NO256") == nullptr) {
        if (getenv("DHAUG_NT256_SINGLE") == nullptr) {                        // two-role kernel (default): mask in its own LDS image
            if (dmask != nullptr && (K == 128 || K == 256)) {
                DHAUG_CHECK(ld_dmask % 8 == 0 && dhaug_aligned16(dmask), DHAUG_EALIGN);
                p.dmask = dmask; p.ld_dmask = ld_dmask; p.dneg = dneg;
                *mask_done = true;
            }
            if (K == 128) return launch_nt256s<8>(s, p);
            if (K == 256) return launch_nt256s<16>(s, p);
        } else if (dmask == nullptr || residual == nullptr) {                 // single-role kernel: mask OR residual
            if (dmask != nullptr && (K == 128 || K == 256)) {
                p.dmask = dmask; p.ld_dmask = ld_dmask; p.dneg = dneg;
                *mask_done = true;
            }
            switch (K / 16) {
                case 8: return launch_nt256<8>(s, p);
                case 16: return launch_nt256<16>(s, p);
                default: break;
            }
        }
    }
    // every other kernel applies the mask in its coalesced epilogue
    if (dmask != nullptr && !*mask_done) {
        p.dmask = dmask; p.ld_dmask = ld_dmask; p.dneg = dneg;
        *mask_done = true;
    }
    if (dmaskf == nullptr && width > 64 && K <= 256 && getenv("DHAUG_GEMM_GENERIC") == nullptr) {
        switch (K / 16) {
            case 1: return launch_ws<1>(s, p);
            case 2: return launch_ws<2>(s, p);
            case 3: return launch_ws<3>(s, p);
            case 4: return launch_ws<4>(s, p);
            case 7: return launch_ws<7>(s, p);
            case 8: return launch_ws<8>(s, p);
            case 16: return launch_ws<16>(s, p);
            default: break;
        }
    }
    // (DHAUG_GEMM_WIDE_MIN_TILES: a test switch -- the golden-vector tests reach these kernels with a few hundred rows)
    const long long wide_min_tiles = getenv("DHAUG_GEMM_WIDE_MIN_TILES") ? atoll(getenv("DHAUG_GEMM_WIDE_MIN_TILES")) : 160;
    if (width >= 256 && ((M + W_BM - 1) / W_BM) * ((width + W_BN - 1) / W_BN) >= wide_min_tiles && K >= 64 && lda >= 64 && ldb >= 64 &&
        getenv("DHAUG_GEMM_NOWIDE") == nullptr && getenv("DHAUG_GEMM_NOBIG") == nullptr && getenv("DHAUG_GEMM_NOPIPE") == nullptr) {
        // long batch, tiles enough for most of the card: 256 x 256 tiles, eight waves (the DenseDim-1000 layers of the frame
        // critics; the 256-wide layers of the split-operand parity arithmetic, K' = 3 K or 6 K)
        // (since round 6: the ping-pong kernel of dhaug_gemm_p8.hip; DHAUG_GEMM_NOP8=1 keeps the five-stage kernel below)
        if (getenv("DHAUG_GEMM_NOP8") == nullptr && dhaug_p8_supported(p)) return dhaug_p8_launch(s, p);
        p.abl = DHAUG_ABL_ENV("DHAUG_BIG_ABL");   // (development: timing only)
        return launch_wide(s, p);
    }
    if (width >= 512 && M >= 4096 && K >= 64 && lda >= 64 && ldb >= 64 && getenv("DHAUG_GEMM_NOBIG") == nullptr &&
        getenv("DHAUG_GEMM_NOPIPE") == nullptr) {
        // long batch, wide layer: 128 x 256 tiles, 64 x 128 per wave
        static bool configured = false;
        if (!configured) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_big_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
            if (e != hipSuccess) return (int)e;
            configured = true;
        }
        const long long grid = ((M + G_BM - 1) / G_BM) * ((width + G_BN - 1) / G_BN);
        p.abl = DHAUG_ABL_ENV("DHAUG_BIG_ABL");
        hipLaunchKernelGGL(gemm_nt_big_kernel, dim3((unsigned)grid), dim3(256), G_LDS, s, p);
        return dhaug_launch_status();
    }
    if (width > 64 && K >= 64 && lda >= 64 && ldb >= 64 && getenv("DHAUG_GEMM_NOPIPE") == nullptr) {
        const long long grid = ((M + 63) / 64) * ((width + 63) / 64);
        p.abl = DHAUG_ABL_ENV("DHAUG_BIG_ABL");   // (development: timing only)
        if (getenv("DHAUG_GEMM_PIPE1") == nullptr) {
            hipLaunchKernelGGL(gemm_nt_pipe2_kernel, dim3((unsigned)grid), dim3(256), 4 * (64 + 64) * BK * 2, s, p);
            return dhaug_launch_status();
        }
        hipLaunchKernelGGL((gemm_nt_pipe_kernel<64, 4>), dim3((unsigned)grid), dim3(256), 4 * (64 + 64) * BK * 2, s, p);
        return dhaug_launch_status();
    }
    if (width > 64) {
        constexpr int BM = 128, BN = 128;
        const long long grid = ((M + BM - 1) / BM) * ((width + BN - 1) / BN);
        size_t lds = (size_t)2 * (BM + BN) * BK * 2, ctile = (size_t)BM * (BN + 4) * 4;
        return launch_nt(gemm_nt_kernel<BM, BN, 2, 2>, grid, lds > ctile ? lds : ctile, s, p);
    } else if (width > 32) {
        constexpr int BM = 128, BN = 64;
        const long long grid = ((M + BM - 1) / BM);
        size_t lds = (size_t)2 * (BM + BN) * BK * 2, ctile = (size_t)BM * (BN + 4) * 4;
        return launch_nt(gemm_nt_kernel<BM, BN, 2, 2>, grid, lds > ctile ? lds : ctile, s, p);
    } else {
        constexpr int BM = 128, BN = 32;
        const long long grid = ((M + BM - 1) / BM);
        size_t lds = (size_t)2 * (BM + BN) * BK * 2, ctile = (size_t)BM * (BN + 4) * 4;
        return launch_nt(gemm_nt_kernel<BM, BN, 4, 1>, grid, lds > ctile ? lds : ctile, s, p);
    }
}

int dhaug_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias,
                    const uint16_t* residual, int64_t ld_res, const float* residual_f32, int64_t ld_res_f32,
                    uint16_t* c_bf16, int64_t ldc_bf16, int64_t n_pad_zero,
                    float* c_f32, int64_t ldc_f32, int64_t M, int64_t N, int64_t K, int act, float slope, void* stream) {
    bool done = false;
    return gemm_bf16_impl(A, lda, B, ldb, bias, residual, ld_res, residual_f32, ld_res_f32, c_bf16, ldc_bf16, n_pad_zero, c_f32,
                          ldc_f32, M, N, K, act, slope, nullptr, 0, 1.0f, &done, stream);
}

/* see include/dhaug.h */
int dhaug_gemm_f16x3(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias, const float* residual_f32,
                     int64_t ld_res_f32, float* c_f32, int64_t ldc_f32, int64_t M, int64_t N, int64_t K, int act, float slope, void* stream) {
    DHAUG_CHECK(M >= 0 && N >= 1 && K >= 16, DHAUG_EINVAL);
    DHAUG_CHECK(act >= DHAUG_ACT_NONE && act <= DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(A); DHAUG_CHECK_PTR(B); DHAUG_CHECK_PTR(c_f32);
    DHAUG_CHECK(K % 16 == 0, DHAUG_EUNSUPPORTED);
    DHAUG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && dhaug_aligned16(A) && dhaug_aligned16(B), DHAUG_EALIGN);
    if (residual_f32) DHAUG_CHECK(ld_res_f32 >= N, DHAUG_EINVAL);
    DHAUG_CHECK(ldc_f32 >= N, DHAUG_EINVAL);
    GemmArgs p{A, lda, B, ldb, bias, nullptr, 0, residual_f32, ld_res_f32, nullptr, 0, 0, c_f32, ldc_f32, M, N, K, N, act, slope,
               nullptr, 0, 1.0f, nullptr, nullptr, 0};
    DHAUG_CHECK(dhaug_p8_supported(p), DHAUG_EUNSUPPORTED);
    return dhaug_p8_launch_f16((hipStream_t)stream, p);
}

/* see include/dhaug.h */
int dhaug_gemm_f16x3_planes(const uint16_t* A, int64_t lda, int a_planes, const uint16_t* B, int64_t ldb, const float* bias,
                            const float* residual_f32, int64_t ld_res_f32, float* c_f32, int64_t ldc_f32, uint16_t* c_planes, int64_t ld_planes,
                            int64_t planes_kp, int64_t M, int64_t N, int64_t kp, int act, float slope, void* stream) {
    DHAUG_CHECK(M >= 0 && N >= 1 && kp >= 16 && kp % 8 == 0, DHAUG_EINVAL);
    DHAUG_CHECK(act >= DHAUG_ACT_NONE && act <= DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(A); DHAUG_CHECK_PTR(B); DHAUG_CHECK_PTR(c_f32);
    int lg = 0;
    while ((64ll << lg) < kp) ++lg;
    DHAUG_CHECK(!a_planes || (64ll << lg) == kp, DHAUG_EUNSUPPORTED);          // planes: 64, 128, 256, ... columns per piece
    DHAUG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= (a_planes ? 2 : 3) * kp && ldb >= 3 * kp && dhaug_aligned16(A) && dhaug_aligned16(B), DHAUG_EALIGN);
    if (residual_f32) DHAUG_CHECK(ld_res_f32 >= N, DHAUG_EINVAL);
    DHAUG_CHECK(ldc_f32 >= N, DHAUG_EINVAL);
    GemmArgs p{A, lda, B, ldb, bias, nullptr, 0, residual_f32, ld_res_f32, nullptr, 0, 0, c_f32, ldc_f32, M, N, 3 * kp, N, act, slope,
               nullptr, 0, 1.0f, nullptr, nullptr, 0};
    if (a_planes) { p.xp_lg = lg + 1; p.xp_map = 0x10u; p.xp_kp = kp; }        // dhaug_split_f16 mode 0 is [hi | hi | lo] = pieces 0 0 1
    if (c_planes != nullptr) {
        DHAUG_CHECK(planes_kp >= N && planes_kp % 8 == 0 && ld_planes >= 2 * planes_kp && ld_planes % 8 == 0 && dhaug_aligned16(c_planes), DHAUG_EALIGN);
        p.cp = c_planes; p.ldcp = ld_planes; p.cp_kp = planes_kp;
    }
    DHAUG_CHECK(dhaug_p8_supported(p), DHAUG_EUNSUPPORTED);
    return dhaug_p8_launch_f16((hipStream_t)stream, p);
}

/* see include/dhaug.h */
int dhaug_gemm_bf16x6_planes(const uint16_t* A_planes, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias,
                             const float* residual_f32, int64_t ld_res_f32, const float* dmask_f32, int64_t ld_dmask_f32, int dmask_act,
                             float dmask_slope, float* c_f32, int64_t ldc_f32, uint16_t* c_planes, int64_t ld_planes, int64_t M, int64_t N,
                             int64_t kp, int x_order, int act, float slope, void* stream) {
    DHAUG_CHECK(M >= 0 && N >= 1 && x_order >= 0 && x_order <= 2 && kp >= (x_order == 2 ? 16 : 64), DHAUG_EINVAL);
    DHAUG_CHECK(act >= DHAUG_ACT_NONE && act <= DHAUG_ACT_LRELU && dmask_act >= DHAUG_ACT_NONE && dmask_act <= DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(A_planes); DHAUG_CHECK_PTR(B); DHAUG_CHECK_PTR(c_f32);
    const bool six = x_order == 2;                                             // A is the ordinary six-segment operand (a narrow input layer)
    int lg = 0;
    while ((64ll << lg) < kp) ++lg;
    DHAUG_CHECK(six || (64ll << lg) == kp, DHAUG_EUNSUPPORTED);                // 64, 128, 256, ... columns per piece
    DHAUG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= (six ? 6 : 3) * kp && ldb >= 6 * kp && dhaug_aligned16(A_planes) && dhaug_aligned16(B), DHAUG_EALIGN);
    if (residual_f32) DHAUG_CHECK(ld_res_f32 >= N, DHAUG_EINVAL);
    if (dmask_act == DHAUG_ACT_NONE) dmask_f32 = nullptr;
    if (dmask_f32) DHAUG_CHECK(ld_dmask_f32 >= N, DHAUG_EINVAL);
    DHAUG_CHECK(ldc_f32 >= N, DHAUG_EINVAL);
    const float dneg = dmask_act == DHAUG_ACT_RELU ? 0.0f : dmask_slope;
    GemmArgs p{A_planes, lda, B, ldb, bias, nullptr, 0, residual_f32, ld_res_f32, nullptr, 0, 0, c_f32, ldc_f32, M, N, 6 * kp, N, act, slope,
               nullptr, 0, dneg, nullptr, nullptr, 0, dmask_f32, ld_dmask_f32};
    // dhaug_split_bf16 (terms 6): mode 0 is [hi | hi | mid | mid | hi | lo] = planes 0 0 1 1 0 2, mode 1 [hi | mid | hi | mid | lo | hi] = 0 1 0 1 2 0
    if (!six) { p.xp_lg = lg + 1; p.xp_map = x_order == 0 ? 0x850u : 0x244u; p.xp_kp = kp; }
    if (c_planes != nullptr) {
        DHAUG_CHECK(ld_planes >= 3 * N && ld_planes % 8 == 0 && dhaug_aligned16(c_planes), DHAUG_EALIGN);
        p.cp = c_planes; p.ldcp = ld_planes; p.cp_kp = N;
    }
    DHAUG_CHECK(dhaug_p8_supported(p), DHAUG_EUNSUPPORTED);
    return dhaug_p8_launch((hipStream_t)stream, p);
}

int dhaug_gemm_bf16_dmask_pad(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* residual,
                              int64_t ld_res, const uint16_t* dmask, int64_t ld_dmask, int dmask_act, float dmask_slope,
                              uint16_t* c_bf16, int64_t ldc_bf16, int64_t n_pad_zero, int64_t M, int64_t N, int64_t K, void* stream) {
    DHAUG_CHECK(dmask_act >= DHAUG_ACT_NONE && dmask_act <= DHAUG_ACT_LRELU, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(c_bf16);
    if (dmask_act == DHAUG_ACT_NONE) dmask = nullptr;
    if (dmask) DHAUG_CHECK(ld_dmask >= N && (reinterpret_cast<uintptr_t>(dmask) & 1u) == 0, DHAUG_EALIGN);
    const float dneg = dmask_act == DHAUG_ACT_RELU ? 0.0f : dmask_slope;
    bool done = false;
    return gemm_bf16_impl(A, lda, B, ldb, nullptr, residual, ld_res, nullptr, 0, c_bf16, ldc_bf16, n_pad_zero > N ? n_pad_zero : N,
                          nullptr, 0, M, N, K, DHAUG_ACT_NONE, 0.0f, dmask, ld_dmask, dneg, &done, stream);
}

/* see include/dhaug.h */
int dhaug_gemm_bf16_group(const dhaug_gemm_desc* d, int n, void* stream) {
    DHAUG_CHECK(n >= 1 && n <= NT_GROUP_MAX, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(d);
    GemmGroupArgs ga;
    for (int i = 0; i < n; ++i) {
        const dhaug_gemm_desc& e = d[i];
        DHAUG_CHECK(e.M == d[0].M && e.N == d[0].N && e.K == d[0].K && e.M > 0, DHAUG_EUNSUPPORTED);
        DHAUG_CHECK(e.dmask_act >= DHAUG_ACT_NONE && e.dmask_act <= DHAUG_ACT_LRELU, DHAUG_EINVAL);
        const uint16_t* dm = e.dmask_act == DHAUG_ACT_NONE ? nullptr : e.dmask;
        if (dm) DHAUG_CHECK(e.ld_dmask >= e.N && (reinterpret_cast<uintptr_t>(dm) & 1u) == 0, DHAUG_EALIGN);
        bool done = false;
        const int rc = gemm_bf16_impl(e.A, e.lda, e.B, e.ldb, e.bias, e.residual, e.ld_res, e.residual_f32, e.ld_res_f32, e.c_bf16,
                                      e.ldc_bf16, e.n_pad_zero, e.c_f32, e.ldc_f32, e.M, e.N, e.K, e.act, e.slope, dm, e.ld_dmask,
                                      e.dmask_act == DHAUG_ACT_RELU ? 0.0f : e.dmask_slope, &done, stream, nullptr, 0, &ga.g[i]);
        if (rc != DHAUG_OK) return rc;
        DHAUG_CHECK(ga.g[i].W == ga.g[0].W, DHAUG_EUNSUPPORTED);
    }
    const GemmArgs& p = ga.g[0];
    // 128 x 128 tiles (half the staged bytes, a quarter of the workgroups) unless the group is too small to fill the card with them
    // or DHAUG_NT_GROUP_TILE=64 asks for the 64 x 64 form
    // long members of wide layers: 256 x 256 ping-pong tiles (dhaug_gemm_p8.hip) -- fewer workgroups at a higher rate each, which leaves
    // CUs to the other critics' streams (DHAUG_NT_GROUP_P8_ROWS: the shortest member that takes them; 0 = never)
    const long long p8_rows = getenv("DHAUG_NT_GROUP_P8_ROWS") ? atoll(getenv("DHAUG_NT_GROUP_P8_ROWS")) : 1024;
    const long long p8_tiles = getenv("DHAUG_NT_GROUP_P8_TILES") ? atoll(getenv("DHAUG_NT_GROUP_P8_TILES")) : 96;
    if (p8_rows > 0 && p.M >= p8_rows && p.W >= 512 && n * ((p.M + 255) / 256) * ((p.W + 255) / 256) >= p8_tiles) {
        bool ok = true;
        for (int i = 0; i < n; ++i) ok = ok && dhaug_p8_supported(ga.g[i]);
        if (ok) return dhaug_p8_launch_group((hipStream_t)stream, ga, n);
    }
    static const int tile_env = getenv("DHAUG_NT_GROUP_TILE") ? atoi(getenv("DHAUG_NT_GROUP_TILE")) : 0;
    const bool big = tile_env != 64 && p.K >= 32 && p.M >= 128;
    const int T = big ? 128 : 64;
    const long long tiles = ((p.M + T - 1) / T) * ((p.W + T - 1) / T);
    long long grid = tiles * n;
    if (n == 2 || n == 4 || n == 8) grid = (tiles + 8 / n - 1) / (8 / n) * 8;       // (XCD map: see the kernels)
    DHAUG_CHECK(grid <= 0x7fffffffLL && tiles <= 0x7fffffffLL, DHAUG_EUNSUPPORTED);
    if (big) hipLaunchKernelGGL(gemm_nt_g128_group_kernel, dim3((unsigned)grid), dim3(512), Q_LDS, (hipStream_t)stream, ga, n, (int)tiles);
    else hipLaunchKernelGGL(gemm_nt_pipe2_group_kernel, dim3((unsigned)grid), dim3(256), 4 * (64 + 64) * BK * 2, (hipStream_t)stream, ga, n, (int)tiles);
    return dhaug_launch_status();
}

#ifdef DHAUG_PIPE_TIMING
int dhaug_debug_pipe_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pipe_stamps), sizeof(long long) * (n < 256 ? n : 256));
}
#endif
/* see include/dhaug.h */
int dhaug_gemm_bf16_dmask_f32(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* residual_f32, int64_t ld_res_f32,
                              const float* dmask, int64_t ld_dmask, int dmask_act, float dmask_slope, float* c_f32, int64_t ldc_f32,
                              int64_t M, int64_t N, int64_t K, void* stream) {
    DHAUG_CHECK(dmask_act == DHAUG_ACT_RELU || dmask_act == DHAUG_ACT_LRELU, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(c_f32); DHAUG_CHECK_PTR(dmask);
    bool done = false;
    return gemm_bf16_impl(A, lda, B, ldb, nullptr, nullptr, 0, residual_f32, ld_res_f32, nullptr, 0, 0, c_f32, ldc_f32, M, N, K,
                          DHAUG_ACT_NONE, 0.0f, nullptr, 0, dmask_act == DHAUG_ACT_RELU ? 0.0f : dmask_slope, &done, stream, dmask, ld_dmask);
}

int dhaug_gemm_bf16_dmask(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* residual,
                          int64_t ld_res, const uint16_t* dmask, int64_t ld_dmask, int dmask_act, float dmask_slope,
                          uint16_t* c_bf16, int64_t ldc_bf16, int64_t M, int64_t N, int64_t K, void* stream) {
    return dhaug_gemm_bf16_dmask_pad(A, lda, B, ldb, residual, ld_res, dmask, ld_dmask, dmask_act, dmask_slope, c_bf16, ldc_bf16,
                                     N, M, N, K, stream);
}

int dhaug_gemm_bf16_dbits(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* residual, int64_t ld_res,
                          const uint32_t* bits, int dmask_act, float dmask_slope, uint16_t* c_bf16, int64_t ldc_bf16, int64_t M,
                          void* stream) {
    DHAUG_CHECK(dmask_act == DHAUG_ACT_RELU || dmask_act == DHAUG_ACT_LRELU, DHAUG_EINVAL);
    DHAUG_CHECK(M >= 0 && M % 32 == 0, DHAUG_EUNSUPPORTED);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(A); DHAUG_CHECK_PTR(B); DHAUG_CHECK_PTR(bits); DHAUG_CHECK_PTR(c_bf16);
    DHAUG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= 256 && ldb >= 256 && ldc_bf16 % 8 == 0 && ldc_bf16 >= 256, DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(A) && dhaug_aligned16(B) && dhaug_aligned16(c_bf16) && dhaug_aligned16(bits), DHAUG_EALIGN);
    if (residual) DHAUG_CHECK(ld_res % 8 == 0 && ld_res >= 256 && dhaug_aligned16(residual), DHAUG_EALIGN);
    GemmArgs p{A, lda, B, ldb, nullptr, residual, ld_res, nullptr, 0, c_bf16, ldc_bf16, 256, nullptr, 0, M, 256, 256, 256,
               DHAUG_ACT_NONE, 0.0f, nullptr, 0, dmask_act == DHAUG_ACT_RELU ? 0.0f : dmask_slope, bits, nullptr, 0};
    return launch_nt256s<16>((hipStream_t)stream, p);
}

/* see include/dhaug.h */
int dhaug_gemm_bf16_dbits_wide(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint32_t* bits_lo,
                               const uint32_t* bits_hi, int dmask_act, float dmask_slope, uint16_t* c_bf16, int64_t ldc_bf16,
                               int64_t M, int64_t N, int64_t K, void* stream) {
    DHAUG_CHECK(dmask_act == DHAUG_ACT_RELU || dmask_act == DHAUG_ACT_LRELU, DHAUG_EINVAL);
    DHAUG_CHECK(M >= 0 && (N == 256 || N == 512) && K >= 16 && K <= 256 && K % 16 == 0, DHAUG_EUNSUPPORTED);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(A); DHAUG_CHECK_PTR(B); DHAUG_CHECK_PTR(bits_lo); DHAUG_CHECK_PTR(c_bf16);
    if (N == 512) DHAUG_CHECK_PTR(bits_hi);
    DHAUG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc_bf16 % 8 == 0 && ldc_bf16 >= N, DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(A) && dhaug_aligned16(B) && dhaug_aligned16(c_bf16) && dhaug_aligned16(bits_lo) &&
                (bits_hi == nullptr || dhaug_aligned16(bits_hi)), DHAUG_EALIGN);
    GemmArgs p{A, lda, B, ldb, nullptr, nullptr, 0, nullptr, 0, c_bf16, ldc_bf16, N, nullptr, 0, M, N, K, N,
               DHAUG_ACT_NONE, 0.0f, nullptr, 0, dmask_act == DHAUG_ACT_RELU ? 0.0f : dmask_slope, bits_lo, bits_hi, 0};
    hipStream_t s = (hipStream_t)stream;
    switch (K / 16) {
        case 1: return launch_ws<1>(s, p);
        case 2: return launch_ws<2>(s, p);
        case 3: return launch_ws<3>(s, p);
        case 4: return launch_ws<4>(s, p);
        case 7: return launch_ws<7>(s, p);
        case 8: return launch_ws<8>(s, p);
        case 16: return launch_ws<16>(s, p);
        default: return DHAUG_EUNSUPPORTED;
    }
}

/* see include/dhaug.h */
int dhaug_gemm_block2_stack_bf16(const uint16_t* X, int64_t ldx, const dhaug_block2* blocks, int nb, int mask_act, float mask_slope,
                                 int64_t M, void* stream) {
    DHAUG_CHECK(mask_act == DHAUG_ACT_RELU || mask_act == DHAUG_ACT_LRELU, DHAUG_EINVAL);
    DHAUG_CHECK(nb >= 1 && nb <= DHAUG_BLOCK2_MAX, DHAUG_EINVAL);
    DHAUG_CHECK(M >= 0 && M % B2_BM == 0, DHAUG_EUNSUPPORTED);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(X); DHAUG_CHECK_PTR(blocks);
    DHAUG_CHECK(ldx % 8 == 0 && ldx >= 256 && dhaug_aligned16(X), DHAUG_EALIGN);
    Block2Args a;
    a.nb = nb; a.X = X; a.ldx = ldx; a.M = M; a.dneg = mask_act == DHAUG_ACT_RELU ? 0.0f : mask_slope;
    const uint16_t* in = X;
    for (int i = 0; i < nb; ++i) {
        const dhaug_block2& b = blocks[i];
        DHAUG_CHECK_PTR(b.W1); DHAUG_CHECK_PTR(b.W2); DHAUG_CHECK_PTR(b.bits1); DHAUG_CHECK_PTR(b.bits2); DHAUG_CHECK_PTR(b.Y1); DHAUG_CHECK_PTR(b.Y2);
        DHAUG_CHECK(b.ldw1 % 8 == 0 && b.ldw2 % 8 == 0 && b.ldy1 % 8 == 0 && b.ldy2 % 8 == 0, DHAUG_EALIGN);
        DHAUG_CHECK(b.ldw1 >= 256 && b.ldw2 >= 256 && b.ldy1 >= 256 && b.ldy2 >= 256, DHAUG_EALIGN);
        DHAUG_CHECK(dhaug_aligned16(b.W1) && dhaug_aligned16(b.W2) && dhaug_aligned16(b.Y1) && dhaug_aligned16(b.Y2) &&
                    dhaug_aligned16(b.bits1) && dhaug_aligned16(b.bits2), DHAUG_EALIGN);
        // a block's Y1 / Y2 may not alias its input (read as the skip after Y1's rows are stored) or each other
        DHAUG_CHECK(b.Y1 != in && b.Y2 != in && b.Y1 != b.Y2, DHAUG_EINVAL);
        a.b[i] = Block2One{b.W1, b.ldw1, b.W2, b.ldw2, b.bits1, b.bits2, b.Y1, b.ldy1, b.Y2, b.ldy2};
        in = b.Y2;
    }
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_block2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, B2_LDS);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const long long mtiles = M / B2_BM;
    hipLaunchKernelGGL(gemm_block2_kernel, dim3(dhaug_persistent_grid(mtiles)), dim3(512), B2_LDS, (hipStream_t)stream, a);
    return dhaug_launch_status();
}

int dhaug_gemm_block2_bf16(const uint16_t* X, int64_t ldx, const uint16_t* W1, int64_t ldw1, const uint16_t* W2, int64_t ldw2,
                           const uint32_t* bits1, const uint32_t* bits2, int mask_act, float mask_slope,
                           uint16_t* Y1, int64_t ldy1, uint16_t* Y2, int64_t ldy2, int64_t M, void* stream) {
    const dhaug_block2 b{W1, ldw1, W2, ldw2, bits1, bits2, Y1, ldy1, Y2, ldy2};
    return dhaug_gemm_block2_stack_bf16(X, ldx, &b, 1, mask_act, mask_slope, M, stream);
}

int dhaug_gemm_tn_bf16_rows(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                            float* colsum_a, int64_t colsum_rows, int64_t M, int64_t N1, int64_t N2, int accumulate, void* stream);

int dhaug_gemm_tn_bf16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                       float* colsum_a, int64_t M, int64_t N1, int64_t N2, int accumulate, void* stream) {
    return dhaug_gemm_tn_bf16_rows(A, lda, B, ldb, C, ldc, colsum_a, M, M, N1, N2, accumulate, stream);
}

int dhaug_gemm_tn_bf16_rows(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                            float* colsum_a, int64_t colsum_rows, int64_t M, int64_t N1, int64_t N2, int accumulate, void* stream) {
    DHAUG_CHECK(colsum_rows >= 0 && colsum_rows <= M && (colsum_rows == M || colsum_rows % TF_ROWS == 0), DHAUG_EUNSUPPORTED);
    const long long cs_rows = colsum_rows == M ? (1LL << 62) : colsum_rows;
    DHAUG_CHECK(M >= 0 && N1 >= 1 && N2 >= 1, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(C);
    DHAUG_CHECK(ldc >= N2, DHAUG_EINVAL);
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N2 * 4, (size_t)N1, s);
        if (e != hipSuccess) return (int)e;
        if (colsum_a != nullptr) {
            e = hipMemsetAsync(colsum_a, 0, (size_t)N1 * 4, s);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(A); DHAUG_CHECK_PTR(B);
    // columns are fetched in 16-byte chunks: the operand rows must be readable up to ceil8(N)
    DHAUG_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda >= ((N1 + 7) & ~7LL) && ldb >= ((N2 + 7) & ~7LL), DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(A) && dhaug_aligned16(B), DHAUG_EALIGN);
    const long long tiles = ((N1 + TN_BN - 1) / TN_BN) * ((N2 + TN_BN - 1) / TN_BN);
    // about two resident workgroups per CU; few splits keep the atomic traffic (splits x N1 x N2 x 4 B) small
    long long splits = 768 / tiles;                              // three resident workgroups per CU (LDS 49 KB each)
    if (splits < 1) splits = 1;
    long long rows = (M + splits - 1) / splits;
    rows = (rows + BK - 1) / BK * BK;
    if (rows < 4 * BK) rows = 4 * BK;
    splits = (M + rows - 1) / rows;
    if (splits > 8) splits = (splits + 7) / 8 * 8;              // multiple of 8: XCD-local tile groups (empty slices exit)
    if (N1 % 128 == 0 && N2 % 128 == 0 && M % TB_ROWS == 0 && M >= 64 * 1024 && getenv("DHAUG_GEMM_GENERIC") == nullptr &&
        getenv("DHAUG_TN_128") != nullptr) {          // measured SLOWER than the 64 x 64 tiles (12.5 vs 11.1 ms per GAN iteration): kept selectable
        const long long tl = (N1 / 128) * (N2 / 128);
        long long sp = 256 / tl;                                 // one workgroup per CU
        if (sp < 1) sp = 1;
        long long r2 = ((M + sp - 1) / sp + TB_ROWS - 1) / TB_ROWS * TB_ROWS;
        if (r2 < 4 * TB_ROWS) r2 = 4 * TB_ROWS;
        sp = (M + r2 - 1) / r2;
        if (sp > 8) sp = (sp + 7) / 8 * 8;
        static bool configured128 = false;
        if (!configured128) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, TB_NSTG * TB_STG);
            if (e != hipSuccess) return (int)e;
            configured128 = true;
        }
        TnArgs pb{A, lda, B, ldb, C, ldc, colsum_a, M, N1, N2, r2, tl, sp, cs_rows};
        hipLaunchKernelGGL(gemm_tn128_kernel, dim3((unsigned)(tl * sp)), dim3(256), TB_NSTG * TB_STG, s, pb);
        return dhaug_launch_status();
    }
    static const bool tn_any = getenv("DHAUG_TN_NOPAD") == nullptr;     // (experiment switch: ragged N1 / N2 on the generic kernel)
    const bool whole = N1 % TN_BN == 0 && N2 % TN_BN == 0;
    if ((whole || (tn_any && M >= 16 * TF_ROWS)) && M % TF_ROWS == 0 && M >= 4 * TF_ROWS && getenv("DHAUG_GEMM_GENERIC") == nullptr) {
        // 128 x 64 tiles (one workgroup per CU) re-read less from L2 but measured slower (35 vs 27 us at 65536 x 256 x 256):
        // the loop is bound by requests in flight, not by L2 bandwidth.  Kept selectable for experiments.
        const bool wide = whole && N1 % 128 == 0 && getenv("DHAUG_TN_WIDE") != nullptr;
        const long long tl = wide ? tiles / 2 : tiles;
        long long sp = (wide ? 256 : 512) / tl;
        if (sp < 1) sp = 1;
        long long r2 = ((M + sp - 1) / sp + TF_ROWS - 1) / TF_ROWS * TF_ROWS;
        if (r2 < 2 * TF_ROWS) r2 = 2 * TF_ROWS;
        sp = (M + r2 - 1) / r2;
        if (sp > 8) sp = (sp + 7) / 8 * 8;
        const int lds = wide ? 2 * (TF_ROWS * 256 + TF_ROWS * 128) : 2 * (TF_ROWS * 128 + TF_ROWS * 128);
        static bool configured = false;
        if (!configured) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn64_kernel<64>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TF_ROWS * 128 + TF_ROWS * 128));
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn64_kernel<128>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TF_ROWS * 256 + TF_ROWS * 128));
            if (e != hipSuccess) return (int)e;
            configured = true;
        }
        TnArgs pf{A, lda, B, ldb, C, ldc, colsum_a, M, N1, N2, r2, tl, sp, cs_rows};
        if (wide) hipLaunchKernelGGL(gemm_tn64_kernel<128>, dim3((unsigned)(tl * sp)), dim3(256), lds, s, pf);
        else hipLaunchKernelGGL(gemm_tn64_kernel<64>, dim3((unsigned)(tl * sp)), dim3(256), lds, s, pf);
        return dhaug_launch_status();
    }
    TnArgs p{A, lda, B, ldb, C, ldc, colsum_a, M, N1, N2, rows, tiles, splits, cs_rows};
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(tiles * splits)), dim3(256), 0, s, p);
    return dhaug_launch_status();
}

}  // extern "C"
