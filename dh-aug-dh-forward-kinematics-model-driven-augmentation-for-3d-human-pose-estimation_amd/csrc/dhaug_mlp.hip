// Fused multi-layer forward of the generator trunk / critics: ONE launch per network, activations never leave LDS.
//
// Layer-by-layer execution of these 256-wide stacks is HBM/latency bound (AI ~ 128 FLOP/B per layer); fused, the
// compulsory traffic is the network input and its logits/head only and the stack becomes MFMA-bound.
//
// Structure (256 threads = 4 waves; TWO workgroups per CU, each persistent over its own 64-row batch tiles -- the two
// run out of phase, so one's epilogue VALU work overlaps the other's MFMAs on the same SIMDs):
//   * activations: bf16 tiles [64 rows][256] in LDS (two 32 KB buffers + one 16 KB [64][128] buffer), 16-byte
//     chunks XOR-swizzled by (row & 15) -> conflict-free ds_read_b128 of MFMA B-operand fragments;
//   * weights: pre-packed on the host side in MFMA A-operand fragment order (dhaug_pack_wfrag): for every
//     32-feature slice and 16-wide k-step one contiguous 1 KB block = 64 lanes x 16 B, so a wave's weight load is
//     a perfectly coalesced global_load_dwordx4 served by L2; wave w owns feature slices w and w+4 of the layer and
//     streams their fragments through a 3-slot register ring, 4 k-steps per slot, two slots ahead of the MFMAs
//     (each fragment feeds 8 MFMAs: 2 slices x 4 row tiles); the first two slots of the NEXT layer are requested
//     before the current layer's epilogue, so L2 latency hides behind MFMA work across layer boundaries too;
//   * MFMA issued swapped (A = weights, B = activations): a lane of the 32x32 accumulator owns one batch row and
//     4 consecutive features per register quad -> epilogue (bias, residual from LDS, ReLU/LeakyReLU, bf16 pack)
//     writes 8 bytes per lane straight into the next layer's operand image.  In-place residual layers
//     (dst == res) are safe: a lane reads and writes only its own elements.
//   * the network is a short "program" of units (load / gemm / store) passed by value; all control flow is
//     workgroup-uniform.
#include "dhaug_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_BM = 64;                                            // batch rows per tile
constexpr int MLP_MT = MLP_BM / 32;                                   // 32-row MFMA tiles per batch tile
constexpr int MLP_THREADS = 256;
constexpr int MLP_NS = 2;                                             // feature slices (of 32) per wave: w and w + 4
constexpr int MLP_MAX_UNITS = 32;
constexpr int MLP_MAX_KSTEPS = 16;
constexpr int BUF01_PITCH = 256, BUF2_PITCH = 128;                  // elements
constexpr int BUF01_BYTES = MLP_BM * BUF01_PITCH * 2;               // 32 768
constexpr int BUF2_BYTES = MLP_BM * BUF2_PITCH * 2;                 // 16 384
constexpr int MLP_LDS_BYTES = 2 * BUF01_BYTES + BUF2_BYTES;         // 81 920: two workgroups per CU

enum { U_LOAD_F32 = 0, U_LOAD_BF16 = 1, U_STORE_BF16 = 2, U_GEMM = 3 };
enum { F_OUT_F32 = 4 };

struct Unit {
    int kind, flags;
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
};

struct Program {
    int nunits;
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

constexpr int MLP_CH = 4;                                            // k-steps per ring slot (64 k)
constexpr int OUT_PITCH = 68;                                        // floats per row of the fp32 output staging image
typedef bf16x8 WRing[3][MLP_NS][MLP_CH];
typedef const Unit __attribute__((address_space(4))) * UnitPtr;      // units are read from the kernarg segment (s_load)

// K is processed in chunks of 64 (4 k-steps); sources narrower than a multiple of 64 are zero-filled by their
// producer (LOAD units pad, GEMM epilogues write whole 32-feature slices) and the packed weights are zero there.
// Every wave always computes two feature slices (wave, wave + 4): the packed weights cover 8 slices (zero rows
// beyond N), which keeps ONE straight-line instruction stream per K shape.
__device__ __forceinline__ int chunks_of(int ksteps) { return (ksteps + MLP_CH - 1) / MLP_CH; }

// issue the global loads of chunk C (of the concatenated sources; the first source has NCH1 chunks) into a ring slot
template <int C, int NCH, int NCH1>
__device__ __forceinline__ void load_chunk(const uint16_t* w1, const uint16_t* w2, int wave, int lane,
                                           bf16x8 (&slot)[MLP_NS][MLP_CH]) {
    constexpr bool second = C >= NCH1;
    constexpr int kpad = (second ? NCH - NCH1 : NCH1) * MLP_CH;              // padded k-steps of this source's blob
    constexpr int k0 = (second ? C - NCH1 : C) * MLP_CH;
    const uint16_t* w = second ? w2 : w1;
#pragma unroll
    for (int t = 0; t < MLP_NS; ++t) {
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
//   k-step k :  [4 ds_read_b128 of the activation fragments of k+1]  [8 global loads of chunk c+2, once per chunk]
//               -- sched_barrier --   8 MFMAs (2 slices x 4 row tiles) on the fragments read during step k-1
template <int NCH, int NCH1>
__device__ __forceinline__ void gemm_layer(UnitPtr u, unsigned char* smem, int wave, int lane) {
    const int r31 = lane & 31, h = lane >> 5;
    const uint16_t* w1 = u->w;
    const uint16_t* w2 = NCH1 < NCH ? u->w2 : u->w;
    WRing ring;
    load_chunk<0, NCH, NCH1>(w1, w2, wave, lane, ring[0]);
    if constexpr (NCH > 1) load_chunk<1, NCH, NCH1>(w1, w2, wave, lane, ring[1]);
    f32x16 acc[MLP_NS][MLP_MT];
    const unsigned char* src1 = buf_base(smem, u->src);
    const int pbs1 = buf_pitch_bytes(u->src);
    const unsigned char* src2 = NCH1 < NCH ? buf_base(smem, u->src2) : src1;
    const int pbs2 = NCH1 < NCH ? buf_pitch_bytes(u->src2) : pbs1;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int KT = NCH * MLP_CH;                            // k-steps in total
    bf16x8 fx[2][MLP_MT];
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
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int c = k / MLP_CH, q = k % MLP_CH;
        if (k + 1 < KT) read_frags(k + 1, fx[(k + 1) & 1]);
        if (q == 0) {
            if (c + 2 < NCH) {
                if (c + 2 == 2) load_chunk<2 < NCH ? 2 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[2]);
                if (c + 2 == 3) load_chunk<3 < NCH ? 3 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[0]);
                if (c + 2 == 4) load_chunk<4 < NCH ? 4 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[1]);
                if (c + 2 == 5) load_chunk<5 < NCH ? 5 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[2]);
                if (c + 2 == 6) load_chunk<6 < NCH ? 6 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[0]);
                if (c + 2 == 7) load_chunk<7 < NCH ? 7 : 0, NCH, NCH1>(w1, w2, wave, lane, ring[1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MLP_MT; ++mt)
#pragma unroll
            for (int t = 0; t < MLP_NS; ++t)
                acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[c % 3][t][q], fx[k & 1][mt], k == 0 ? zero : acc[t][mt],
                                                                    0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    const int nslices = (u->N + 31) >> 5;
    const bool to_global = (u->flags & F_OUT_F32) != 0;
    unsigned char* dst = buf_base(smem, u->dst);
    const int pbd = buf_pitch_bytes(u->dst);
    const int resid = u->res;
    const unsigned char* res = resid >= 0 ? buf_base(smem, resid) : nullptr;
    const int pbr = resid >= 0 ? buf_pitch_bytes(resid) : 0;
    const float neg = act_neg(u->act, u->slope);
    // epilogue: this lane owns row (32 mt + r31), features 32*slice + 8g + 4h .. +3
#pragma unroll
    for (int t = 0; t < MLP_NS; ++t) {
        const int slice = wave + 4 * t;
        if (slice >= nslices) continue;                        // wave-uniform: slices beyond N are never stored
        f32x4 bias[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(u->bias + 32 * slice + 4 * h + 8 * g);
        if (to_global) {
            // network output (<= 64 features): fp32 staging image [128][OUT_PITCH] in buffer dst, copied out
            // cooperatively by store_output() after the barrier
            float* st = reinterpret_cast<float*>(dst);
#pragma unroll
            for (int mt = 0; mt < MLP_MT; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[t][mt][4 * g + e] + bias[g][e], neg);
                    *reinterpret_cast<f32x4*>(st + (32 * mt + r31) * OUT_PITCH + 32 * slice + 4 * h + 8 * g) = v;
                }
        } else if (res != nullptr) {
#pragma unroll
            for (int mt = 0; mt < MLP_MT; ++mt) {
                const int row = 32 * mt + r31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 4 * slice + g;                         // 16-byte chunk = 8 features
                    const uint2 rr = *reinterpret_cast<const uint2*>(res + chunk_off(row, c, pbr) + (h << 3));
                    float v[4];
                    v[0] = acc[t][mt][4 * g + 0] + bias[g][0] + __builtin_bit_cast(float, rr.x << 16);
                    v[1] = acc[t][mt][4 * g + 1] + bias[g][1] + __builtin_bit_cast(float, rr.x & 0xffff0000u);
                    v[2] = acc[t][mt][4 * g + 2] + bias[g][2] + __builtin_bit_cast(float, rr.y << 16);
                    v[3] = acc[t][mt][4 * g + 3] + bias[g][3] + __builtin_bit_cast(float, rr.y & 0xffff0000u);
                    uint2 o;
                    o.x = pack_bf16x2(act_fn(v[0], neg), act_fn(v[1], neg));
                    o.y = pack_bf16x2(act_fn(v[2], neg), act_fn(v[3], neg));
                    *reinterpret_cast<uint2*>(dst + chunk_off(row, c, pbd) + (h << 3)) = o;
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MLP_MT; ++mt) {
                const int row = 32 * mt + r31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 4 * slice + g;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[t][mt][4 * g + e] + bias[g][e], neg);
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(dst + chunk_off(row, c, pbd) + (h << 3)) = o;
                }
            }
        }
    }
}

// fp32 staging image of the last layer -> global (M, ld), columns [0, N)
__device__ __forceinline__ void store_output(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    const float* st = reinterpret_cast<const float*>(buf_base(smem, u->dst));
    float* out = static_cast<float*>(const_cast<void*>(u->g));
    const int N = u->N, total = MLP_BM * N;
    const long long ld = u->ld;
    for (int e = tid; e < total; e += MLP_THREADS) {
        const int row = e / N, n = e - row * N;
        if (m0 + row < M) out[(m0 + row) * ld + n] = st[row * OUT_PITCH + n];
    }
}

// data-movement units.  LOAD zero-fills columns [cols, ceil64(cols)) so that the consuming GEMM may read whole chunks.
__device__ __forceinline__ void move_unit(UnitPtr u, unsigned char* smem, long long m0, long long M, int tid) {
    const int kind = u->kind, cols = u->cols;
    const long long ld = u->ld;
    if (kind == U_LOAD_F32) {
        const float* g = static_cast<const float*>(u->g);
        unsigned char* dst = buf_base(smem, u->dst);
        const int pb = buf_pitch_bytes(u->dst), q4 = ((cols + 63) & ~63) >> 2, total = MLP_BM * q4;
        for (int e = tid; e < total; e += MLP_THREADS) {
            const int row = e / q4, c4 = e - row * q4;
            const long long gm = m0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gm < M && c4 * 4 < cols) v = *reinterpret_cast<const f32x4*>(g + gm * ld + c4 * 4);
            uint2 o;
            o.x = (uint32_t)dhaug_f32_to_bf16(v[0]) | ((uint32_t)dhaug_f32_to_bf16(v[1]) << 16);
            o.y = (uint32_t)dhaug_f32_to_bf16(v[2]) | ((uint32_t)dhaug_f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<uint2*>(dst + chunk_off(row, c4 >> 1, pb) + ((c4 & 1) << 3)) = o;
        }
        return;
    }
    uint16_t* g = static_cast<uint16_t*>(const_cast<void*>(u->g));
    const int id = kind == U_LOAD_BF16 ? u->dst : u->src;
    unsigned char* img = buf_base(smem, id);
    const int pb = buf_pitch_bytes(id);
    const int q8 = (kind == U_LOAD_BF16 ? ((cols + 63) & ~63) : cols) >> 3, total = MLP_BM * q8;
    for (int e = tid; e < total; e += MLP_THREADS) {
        const int row = e / q8, c = e - row * q8;
        const long long gm = m0 + row;
        uint4* l = reinterpret_cast<uint4*>(img + chunk_off(row, c, pb));
        uint4* gg = reinterpret_cast<uint4*>(g + gm * ld + c * 8);
        if (kind == U_LOAD_BF16) *l = (gm < M && c * 8 < cols) ? *gg : make_uint4(0, 0, 0, 0);
        else if (gm < M) *gg = *l;
    }
}

__global__ __launch_bounds__(MLP_THREADS, 2) void fused_mlp_kernel(Program prog, long long M) {
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
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m0 = tile * MLP_BM;
#pragma unroll 1
        for (int i = 0; i < nunits; ++i) {
            UnitPtr u = units + i;
            if (u->kind != U_GEMM) {
                move_unit(u, smem, m0, M, tid);
            } else {
                const int nch1 = chunks_of(u->ksteps), nch = nch1 + chunks_of(u->ksteps2);
                switch (nch * 16 + nch1) {                         // validated on the host: one of these cases
                    case 1 * 16 + 1: gemm_layer<1, 1>(u, smem, wave, lane); break;
                    case 2 * 16 + 2: gemm_layer<2, 2>(u, smem, wave, lane); break;
                    case 2 * 16 + 1: gemm_layer<2, 1>(u, smem, wave, lane); break;
                    case 4 * 16 + 4: gemm_layer<4, 4>(u, smem, wave, lane); break;
                    case 4 * 16 + 2: gemm_layer<4, 2>(u, smem, wave, lane); break;
                    case 8 * 16 + 4: gemm_layer<8, 4>(u, smem, wave, lane); break;
                    default: break;
                }
                if (u->flags & F_OUT_F32) {
                    __syncthreads();
                    store_output(u, smem, m0, M, tid);
                }
            }
            __syncthreads();
        }
    }
    (void)prog;
}

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

}  // namespace

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

int dhaug_mlp_forward(const dhaug_mlp_unit* units, int nunits, int64_t M, void* stream) {
    DHAUG_CHECK(nunits >= 1 && nunits <= MLP_MAX_UNITS && M >= 0, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(units);
    if (M == 0) return DHAUG_OK;
    Program prog;
    prog.nunits = nunits;
    for (int i = 0; i < nunits; ++i) {
        const dhaug_mlp_unit& s = units[i];
        Unit& u = prog.u[i];
        u.kind = s.kind; u.flags = s.flags; u.src = s.src; u.dst = s.dst; u.res = s.res;
        u.src2 = s.src2; u.ksteps2 = s.ksteps2;
        u.ksteps = s.ksteps; u.N = s.n; u.act = s.act; u.slope = s.slope; u.cols = s.cols; u.ld = s.ld;
        u.g = s.g; u.w = static_cast<const uint16_t*>(s.w); u.w2 = static_cast<const uint16_t*>(s.w2); u.bias = s.bias;
        DHAUG_CHECK(u.kind >= U_LOAD_F32 && u.kind <= U_GEMM, DHAUG_EINVAL);
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
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const long long ntiles = (M + MLP_BM - 1) / MLP_BM;
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);           // two resident workgroups per CU
    hipLaunchKernelGGL(fused_mlp_kernel, dim3(grid), dim3(MLP_THREADS), MLP_LDS_BYTES, (hipStream_t)stream, prog, (long long)M);
    return dhaug_launch_status();
}

}  // extern "C"
