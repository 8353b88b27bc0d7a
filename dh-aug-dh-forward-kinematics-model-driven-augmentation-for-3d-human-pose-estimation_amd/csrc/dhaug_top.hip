// The TOP of a branch critic (merge layer -> 100-wide myResNet block -> logit layer: R/models_Fk_GAN/Fk_discriminator.py:149-201,
// merge_previous / merge_block1 / output) in the explicit training step, one launch per sweep instead of four / three.
//
//   backward chain (sweep 2)     gz_m2 = seed w_out         * act'(m1)          rows [0, M)      (rank-one: the logit layer)
//                                gz_m1 = (gz_m2 W_fc2)      * act'(mh)
//                                gz_m0 = (gz_m1 W_fc1 + gz_m2) * act'(m0)
//                                gcat  = (gz_m0 W_merge)    * act'(cat)         (M, 512): the branches' cotangents, masks = sign bits
//   tangent sweep  (sweep 3)     um0 = (ucat W_merge^T)     * act'(m0)          written over the rows of m0 it masks with (in place)
//                                umh = (um0 W_fc1^T)        * act'(mh)          over mh
//                                um1 = (umh W_fc2^T + um0)  * act'(m1)          over m1
//
// As separate launches (rank1_mask + 3 x gemm_nt_ws_kernel<7> | gemm_nt_pipe2 + 2 x gemm_nt_ws_kernel<7>) these took 194 / 111 us of a
// 2.5 ms 3D-critic step at B = 65 536 (tools/trace_step.py): every 100-wide tensor written, re-read as the next operand and re-read as
// a mask image, the 201 MB cotangent of the concatenation written at 2.5 TB/s.  Here a persistent workgroup (256 threads, one per CU)
// walks 64-row tiles: the 100-wide cotangents / tangents stay in LDS between the layers (bf16 images, rounded exactly where the
// separate launches round them), all weights sit in registers as MFMA fragments (staged through LDS once, as whole rows), the mask
// tiles of the NEXT tile travel in registers while this one is computed, and every result leaves as whole rows.
// Arithmetic per output element = the kernels replaced: k ascending into one fp32 accumulator (v_mfma_f32_32x32x16_bf16, issued
// swapped: A = weights), residual added in fp32, mask as v > 0 ? x : x * neg, one rounding to bf16 -- bit-identical results
// (tests/test_gpu_kernels.py).
#include "dhaug_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // a 16-byte chunk in registers (HIP's uint4 is a struct: copies of it
                                                                   // through references became memcpys into scratch objects)

constexpr int T_BM = 64;                    // rows per tile
constexpr int T_KS = 7;                     // k-steps of a 100-wide operand (112 columns)
constexpr int T_NC = 14;                    // 16-byte chunks of a 112-column row
constexpr int T_P = 240;                    // bytes per row of a 112-column LDS image (224 + 16: 16-byte aligned rows, 60 dwords apart)
constexpr int T_IMG = T_BM * T_P;           // 15 360
constexpr int T_OP = 1040;                  // bytes per row of the 512-column output image (1 024 + 16)
constexpr int T_LDS = 6 * T_IMG + T_BM * T_OP;   // three cotangent images, three mask images, the output image: 158 720

struct TopArgs {
    const uint16_t* seed; long long ld_seed;            // backward: (M, >= 1) bf16, column 0 = the logit cotangent of the row
    const uint16_t* wout; long long ld_wout;            // backward: the logit layer's weights, element stride ld_wout
    const uint16_t* x; long long ldx;                   // tangent: ucat (M, 512) bf16
    uint16_t* m1; uint16_t* mh; uint16_t* m0; long long ld_m;   // activations (masks) (M, ld_m >= 112); tangent: overwritten with the tangents
    const uint16_t* w2; long long ldw2;                 // fc2 operand: rows = this product's outputs (100), >= 112 columns
    const uint16_t* w1; long long ldw1;                 // fc1 operand
    const uint16_t* wm; long long ldwm;                 // merge operand: backward [512][>= 112], tangent [100][>= 512]
    const uint32_t* bits0; const uint32_t* bits1;       // backward: sign bits of the two 256-column blocks of the concatenation
    uint16_t* g2; uint16_t* g1; uint16_t* g0; long long ld_g;   // backward outputs (M, ld_g >= 112)
    uint16_t* gcat; long long ld_gcat;                  // backward output (M, 512)
    long long M; int n0; float dneg;
};

#ifdef DHAUG_TOP_TIMING
__device__ long long g_top_stamps[64];
#define TOP_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0 && tile == (long long)blockIdx.x + gridDim.x) g_top_stamps[i] = (long long)__builtin_readcyclecounter();
#else
#define TOP_STAMP(i)
#endif
__device__ __forceinline__ void t_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float bf2f(uint32_t lo16) { return __builtin_bit_cast(float, lo16 << 16); }

// rows [0, rows) x 14 chunks of a row-major bf16 matrix -> an LDS image of pitch T_P (rows beyond `rows` up to PAD_ROWS: zero).
// Up to 14 chunks per thread are requested before the first is written: as a plain loop (load, wait, write, 28 times for the merge
// operand, every workgroup after the same lines) the staging was 17 us of a launch -- as long as two of its twelve tiles.
template <int PAD_ROWS>
__device__ __forceinline__ void stage_rows(const uint16_t* W, long long ldw, int rows, unsigned char* img, int tid) {
    constexpr int N = PAD_ROWS * T_NC / 256, G = N < 14 ? N : 14;
    static_assert(PAD_ROWS * T_NC % 256 == 0 && N % G == 0, "whole passes");
#pragma unroll
    for (int j0 = 0; j0 < N; j0 += G) {
        u32x4 v[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int q = tid + 256 * (j0 + j), row = q / T_NC, c = q - row * T_NC;
            // (every request issued, at a clamped row, zeros by select: with `if (row < rows) v[j] = load` hipcc gives each request a
            // branch of its own and a full wait behind it -- 42 round trips one after the other)
            v[j] = *reinterpret_cast<const u32x4*>(W + (long long)(row < rows ? row : rows - 1) * ldw + c * 8);
            if (row >= rows) v[j] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int q = tid + 256 * (j0 + j), row = q / T_NC, c = q - row * T_NC;
            *reinterpret_cast<u32x4*>(img + row * T_P + c * 16) = v[j];
        }
    }
}
// fragment (slice of 32 rows, k-step ks) of a staged matrix: lane (r31, h) = row 32 slice + r31, columns 16 ks + 8 h ..
__device__ __forceinline__ bf16x8 frag(const unsigned char* img, int slice, int ks, int r31, int h) {
    return *reinterpret_cast<const bf16x8*>(img + (32 * slice + r31) * T_P + (2 * ks + h) * 16);
}

// one 64-row mask / activation tile (64 x 14 chunks = 896 16-byte chunks, 3.5 per thread) global -> registers, registers -> LDS image,
// LDS image -> global rows.  The thread's chunks (row, c) = (q / 14, q % 14), q = tid + 256 i, are computed ONCE (TileMap: the LDS
// offset and the row / column of each of its four chunks) -- as opaque per-use values they were 36 divisions by 14 per tile on a lone
// wave per SIMD, hoisted by hipcc they became ~40 loop-carried pointers and the kernel spilled.
struct TileRegs { u32x4 v0, v1, v2, v3; };
struct TileMap { int lds[4]; int row[4]; int col[4]; bool ok3; };
__device__ __forceinline__ TileMap tile_map(int tid) {
    TileMap m;
    m.ok3 = tid + 768 < T_BM * T_NC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (the fourth chunk exists for half the threads; the other half repeats chunk q - 128 on the way IN -- the same bytes to the same
        // LDS address, no branch, no conditional load for hipcc's wait counts to be pessimistic about -- and skips it on the way out)
        const int q = tid + 256 * i - (i == 3 && !m.ok3 ? 128 : 0), row = q / T_NC, c = q - row * T_NC;
        m.lds[i] = row * T_P + c * 16; m.row[i] = row; m.col[i] = c * 8;
    }
    return m;
}
__device__ __forceinline__ void tile_load(const uint16_t* m, long long ld, long long row0, const TileMap& tm, TileRegs& r) {
    r.v0 = *reinterpret_cast<const u32x4*>(m + (row0 + tm.row[0]) * ld + tm.col[0]);
    r.v1 = *reinterpret_cast<const u32x4*>(m + (row0 + tm.row[1]) * ld + tm.col[1]);
    r.v2 = *reinterpret_cast<const u32x4*>(m + (row0 + tm.row[2]) * ld + tm.col[2]);
    r.v3 = *reinterpret_cast<const u32x4*>(m + (row0 + tm.row[3]) * ld + tm.col[3]);
}
__device__ __forceinline__ void tile_to_lds(const TileRegs& r, unsigned char* img, const TileMap& tm) {
    *reinterpret_cast<u32x4*>(img + tm.lds[0]) = r.v0;
    *reinterpret_cast<u32x4*>(img + tm.lds[1]) = r.v1;
    *reinterpret_cast<u32x4*>(img + tm.lds[2]) = r.v2;
    *reinterpret_cast<u32x4*>(img + tm.lds[3]) = r.v3;
}
// (One wave per SIMD has nobody to hide an LDS round trip behind, and hipcc sinks every LDS read to its use: the reads of a phase are
// issued together and fenced off from their uses.)
__device__ __forceinline__ void tile_store(const unsigned char* img, uint16_t* g, long long ld, long long row0, const TileMap& tm) {
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x4*>(img + tm.lds[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < 3 || tm.ok3) *reinterpret_cast<u32x4*>(g + (row0 + tm.row[i]) * ld + tm.col[i]) = v[i];
}

// the lane's 4 consecutive values of register quad gq (features f0 .. f0 + 3 of its row): mask with the activation words `mk`
// (read from the mask image up front), round, store into the result image.  Slice 3 covers features 96 .. 127 and the images have 112
// columns + 16 pad bytes per row: the quads of features 112 .. 119 fall into the pad bytes as they are, those of 120 .. 127 are sent
// there too (qoff) -- no branch around a quad (a wave-uniform branch per quad was a third of the epilogues' time); what lands in the
// pad bytes is never read as data (fragments and row stores take the 14 chunks of a row).
__device__ __forceinline__ int qoff(int f0) { return (f0 >= 120 ? f0 - 8 : f0) * 2; }
__device__ __forceinline__ uint2 mask_words(const unsigned char* mimg, int row, int f0) {
    return *reinterpret_cast<const uint2*>(mimg + row * T_P + qoff(f0));
}
__device__ __forceinline__ void quad_out(float (&v)[4], uint2 mk, unsigned char* dimg, int row, int f0, float dneg) {
    const short y0 = (short)(mk.x & 0xffffu), y1 = (short)(mk.x >> 16), y2 = (short)(mk.y & 0xffffu), y3 = (short)(mk.y >> 16);
    v[0] = y0 > 0 ? v[0] : v[0] * dneg;                     // a positive bf16 is a positive int16
    v[1] = y1 > 0 ? v[1] : v[1] * dneg;
    v[2] = y2 > 0 ? v[2] : v[2] * dneg;
    v[3] = y3 > 0 ? v[3] : v[3] * dneg;
    uint2 o;
    o.x = (uint32_t)dhaug_f32_to_bf16(v[0]) | ((uint32_t)dhaug_f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)dhaug_f32_to_bf16(v[2]) | ((uint32_t)dhaug_f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2*>(dimg + row * T_P + qoff(f0)) = o;
}

// a 100 -> 100 layer on the tile: wave w = feature slice w, both 32-row tiles; src / res / mask / dst are LDS images
__device__ __forceinline__ void small_layer(const bf16x8 (&wf)[T_KS], const unsigned char* src, const unsigned char* res, const unsigned char* mimg,
                                            unsigned char* dst, int wave, int r31, int h, float dneg) {
    bf16x8 fx[4][2];                                                         // k-steps ks .. ks + 2 in flight
    uint2 mk[2][4], rr[2][4];
#define TOP_RD(F, S, ks) { _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_) F[(ks) & 3][a_] = *reinterpret_cast<const bf16x8*>((S) + (32 * a_ + r31) * T_P + (2 * (ks) + h) * 16); }
    TOP_RD(fx, src, 0) TOP_RD(fx, src, 1) TOP_RD(fx, src, 2)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int row = 32 * a + r31, f0 = 32 * wave + 8 * gq + 4 * h;
            mk[a][gq] = mask_words(mimg, row, f0);
            rr[a][gq] = res != nullptr ? mask_words(res, row, f0) : make_uint2(0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < T_KS; ++ks) {
        if (ks + 3 < T_KS) TOP_RD(fx, src, ks + 3)
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], fx[ks & 3][a], acc[a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int row = 32 * a + r31, f0 = 32 * wave + 8 * gq + 4 * h;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[a][4 * gq + e];
            if (res != nullptr) {
                v[0] += bf2f(rr[a][gq].x & 0xffffu); v[1] += __builtin_bit_cast(float, rr[a][gq].x & 0xffff0000u);
                v[2] += bf2f(rr[a][gq].y & 0xffffu); v[3] += __builtin_bit_cast(float, rr[a][gq].y & 0xffff0000u);
            }
            quad_out(v, mk[a][gq], dst, row, f0, dneg);
        }
}

__global__ __launch_bounds__(256, 1) void top_backward_kernel(TopArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sG2 = smem;
    unsigned char* sG1 = smem + T_IMG;
    unsigned char* sG0 = smem + 2 * T_IMG;
    unsigned char* sM1 = smem + 3 * T_IMG;
    unsigned char* sMH = smem + 4 * T_IMG;
    unsigned char* sM0 = smem + 5 * T_IMG;
    unsigned char* sO = smem + 6 * T_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const long long ntiles = p.M / T_BM;
    if ((long long)blockIdx.x >= ntiles) return;

    // Everything a tile reads from global memory -- its three mask tiles, its rows' seeds, its sign-bit words -- is requested TWO TILES
    // AHEAD, into registers, and moves to the LDS images one tile ahead: the memory counter retires in order, so a wait for a load
    // also waits for every store issued before it, and a load waited for a whole tile after its issue has nothing in front of it any
    // more.  (First form: the seeds and the sign bits loaded where they are used -- each of those waits drained the previous phase's row
    // stores, a memory round trip: 16 us per tile, 192 us for 3B = 196 608 rows against 184 for the four launches.)
    // WHERE in the tile they are requested matters as much: all 18 loads of a thread in one place -- 72 instructions, 576 cache lines
    // a CU, several round trips' worth of outstanding misses -- took 4 100 .. 4 900 clocks to ISSUE wherever they stood (phase stamps,
    // DHAUG_TOP_TIMING); one mask tile per layer boundary (4 loads a thread, right behind the move of its predecessor into the image
    // the layer just finished reading) issues in ~200.  131 -> 116 us at 196 608 rows with that and the staging below; a tile is then
    // 8.6 us for 154 KB, 4.6 TB/s over 256 CUs -- the card's copy rate is 5.3.
    const TileMap tm = tile_map(tid);
    TileRegs r1, rh, r0;
    uint32_t sd[2], bw[2][2];                                                // this tile: seed of the lane's row in row tile a; bits [block][a]
    // (a macro, not a lambda: tiles passed to a lambda by reference across the weight staging stayed in scratch)
    // (the small loads FIRST: the wait for the mask tiles at the end of the tile then covers them -- requested behind the masks, their
    // first use in the next iteration was a wait of its own, and that one drained the tile's row stores)
#define TOPB_SMALL(t, S2, B2) { const long long row0_ = (t) * T_BM; \
        _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_) { \
            S2[a_] = p.seed[(row0_ + 32 * a_ + r31) * p.ld_seed]; \
            const long long R_ = (row0_ >> 5) + a_; \
            B2[0][a_] = p.bits0[(R_ * 4 + wave) * 64 + lane]; \
            B2[1][a_] = p.bits1[(R_ * 4 + wave) * 64 + lane]; \
        } }
#define TOPB_REQUEST(t, Q1, QH, Q0, S2, B2) { TOPB_SMALL(t, S2, B2) tile_load(p.m1, p.ld_m, (t) * T_BM, tm, Q1); \
        tile_load(p.mh, p.ld_m, (t) * T_BM, tm, QH); tile_load(p.m0, p.ld_m, (t) * T_BM, tm, Q0); }
    uint32_t sd1[2], bw1[2][2];                                              // the next tile's seeds / bits (its masks: in r1, rh, r0)
    const long long last = ntiles - 1 - ((ntiles - 1 - blockIdx.x) % gridDim.x);   // this workgroup's last tile
    auto clampt = [&](long long t) { return t < ntiles ? t : last; };
    // The first two tiles are requested HERE, in front of the weight staging (whose round trips they share).
    TileRegs q1, qh, q0;
    TOPB_REQUEST((long long)blockIdx.x, r1, rh, r0, sd, bw)
    TOPB_REQUEST(clampt((long long)blockIdx.x + gridDim.x), q1, qh, q0, sd1, bw1)
    // ---- weights -> registers, through LDS (whole rows in, fragments out)
    bf16x8 w2f[T_KS], w1f[T_KS], wmf[2][2][T_KS];                           // merge: [block b][t]: slice 8 b + wave + 4 t of the 16
    stage_rows<128>(p.w2, p.ldw2, p.n0, smem, tid);
    stage_rows<128>(p.w1, p.ldw1, p.n0, smem + 128 * T_P, tid);
    t_barrier();
#pragma unroll
    for (int ks = 0; ks < T_KS; ++ks) { w2f[ks] = frag(smem, wave, ks, r31, h); w1f[ks] = frag(smem + 128 * T_P, wave, ks, r31, h); }
    t_barrier();
    stage_rows<512>(p.wm, p.ldwm, 512, smem, tid);                           // 512 x 240 = 122 880 bytes
    t_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < T_KS; ++ks) wmf[b][t][ks] = frag(smem, 8 * b + wave + 4 * t, ks, r31, h);
    // the logit layer's weights of the lane's 16 features (slice `wave`): f = 32 wave + 8 gq + 4 h + e
    float wo[16];
    {
        uint32_t raw[16];                                                    // (all sixteen requests first: one round trip, not sixteen)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = 32 * wave + 8 * (i >> 2) + 4 * h + (i & 3);
            raw[i] = p.wout[(long long)(f < p.n0 ? f : p.n0 - 1) * p.ld_wout];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = 32 * wave + 8 * (i >> 2) + 4 * h + (i & 3);
            wo[i] = f < p.n0 ? bf2f(raw[i]) : 0.0f;
        }
    }
    t_barrier();                                                             // (the staging area becomes the images)

    tile_to_lds(r1, sM1, tm); tile_to_lds(rh, sMH, tm); tile_to_lds(r0, sM0, tm);
    r1 = q1; rh = qh; r0 = q0;
    t_barrier();
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * T_BM;
        uint32_t sd2[2], bw2[2][2];
        TOP_STAMP(0)
        TOP_STAMP(1)
        // ---- the logit layer: gz_m2 = bf16(seed * w_out) * act'(m1)  (R/models_Fk_GAN/Fk_discriminator.py:199-201 under autograd)
        {
            uint2 mk[2][4];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) mk[a][gq] = mask_words(sM1, 32 * a + r31, 32 * wave + 8 * gq + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int row = 32 * a + r31;
                const float sv = bf2f(sd[a]);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf2f(dhaug_f32_to_bf16(sv * wo[4 * gq + e]));
                    quad_out(v, mk[a][gq], sG2, row, 32 * wave + 8 * gq + 4 * h, p.dneg);
                }
            }
        }
        TOP_STAMP(2)
        t_barrier();
        TOP_STAMP(3)
        tile_store(sG2, p.g2, p.ld_g, row0, tm);
        // (the m1 image has had its one reader: the next tile's goes in, the tile after that is requested)
        const long long t2 = clampt(tile + 2 * (long long)gridDim.x);
        tile_to_lds(r1, sM1, tm);
        TOPB_SMALL(t2, sd2, bw2)
        tile_load(p.m1, p.ld_m, t2 * T_BM, tm, r1);
        TOP_STAMP(4)
        // ---- merge_block1.fc2, fc1 (the skip joins before the mask)
        small_layer(w2f, sG2, nullptr, sMH, sG1, wave, r31, h, p.dneg);
        TOP_STAMP(5)
        t_barrier();
        tile_store(sG1, p.g1, p.ld_g, row0, tm);
        tile_to_lds(rh, sMH, tm);
        tile_load(p.mh, p.ld_m, t2 * T_BM, tm, rh);
        TOP_STAMP(6)
        small_layer(w1f, sG1, sG2, sM0, sG0, wave, r31, h, p.dneg);
        TOP_STAMP(7)
        t_barrier();                                                         // (every mask image of this tile has been read)
        TOP_STAMP(13)
        tile_store(sG0, p.g0, p.ld_g, row0, tm);
        TOP_STAMP(14)
        tile_to_lds(r0, sM0, tm);
        TOP_STAMP(15)
        tile_load(p.m0, p.ld_m, t2 * T_BM, tm, r0);
        TOP_STAMP(8)
        // ---- the merge layer: 512 outputs, masks = the sign bits of the branches' last layers (dhaug_mlp_unit.bits layout: the word of
        // lane (r31, h) of forward wave w and 32-row tile R holds the lane's elements of slices w (t = 0) and w + 4 (t = 1)); one
        // 256-column block at a time (64 accumulator registers)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            bf16x8 fxm[4][2];                                                // k-steps ks .. ks + 2 in flight
            TOP_RD(fxm, sG0, 0) TOP_RD(fxm, sG0, 1) TOP_RD(fxm, sG0, 2)
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[2][2];                                                // [t][row tile]
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][a][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < T_KS; ++ks) {
                if (ks + 3 < T_KS) TOP_RD(fxm, sG0, ks + 3)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        acc[t][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wmf[b][t][ks], fxm[ks & 3][a], acc[t][a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const uint32_t w = bw[b][a];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[t][a][4 * gq + e];
                        const int p0 = 8 * t + 2 * gq;                       // pairs 8 t + 2 gq, + 1: even element at bit p, odd at 16 + p
                        v[0] = ((w >> p0) & 1u) ? v[0] : v[0] * p.dneg;
                        v[1] = ((w >> (16 + p0)) & 1u) ? v[1] : v[1] * p.dneg;
                        v[2] = ((w >> (p0 + 1)) & 1u) ? v[2] : v[2] * p.dneg;
                        v[3] = ((w >> (17 + p0)) & 1u) ? v[3] : v[3] * p.dneg;
                        uint2 o;
                        o.x = (uint32_t)dhaug_f32_to_bf16(v[0]) | ((uint32_t)dhaug_f32_to_bf16(v[1]) << 16);
                        o.y = (uint32_t)dhaug_f32_to_bf16(v[2]) | ((uint32_t)dhaug_f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<uint2*>(sO + (32 * a + r31) * T_OP + (256 * b + 32 * (wave + 4 * t) + 8 * gq + 4 * h) * 2) = o;
                    }
            }
        }
        TOP_STAMP(9)
        t_barrier();                                                         // the output image is complete; every mask image has been read
        TOP_STAMP(10)
        {                                                                    // 64 rows x 64 chunks, whole 1 KB rows; eight reads, then their stores, twice
            int tq = tid;
            asm volatile("" : "+v"(tq));
            const int srow = tq >> 6, sc = tq & 63;                          // chunk q = tq + 256 i: row srow + 4 i, chunk sc
            const unsigned char* src = sO + srow * T_OP + sc * 16;
            uint16_t* dst = p.gcat + (row0 + srow) * p.ld_gcat + sc * 8;
            uint4 v0, v1, v2, v3, v4, v5, v6, v7;
#define TOP_LD8(base) v0 = *reinterpret_cast<const uint4*>(src + ((base) + 0) * 4 * T_OP); v1 = *reinterpret_cast<const uint4*>(src + ((base) + 1) * 4 * T_OP); \
                      v2 = *reinterpret_cast<const uint4*>(src + ((base) + 2) * 4 * T_OP); v3 = *reinterpret_cast<const uint4*>(src + ((base) + 3) * 4 * T_OP); \
                      v4 = *reinterpret_cast<const uint4*>(src + ((base) + 4) * 4 * T_OP); v5 = *reinterpret_cast<const uint4*>(src + ((base) + 5) * 4 * T_OP); \
                      v6 = *reinterpret_cast<const uint4*>(src + ((base) + 6) * 4 * T_OP); v7 = *reinterpret_cast<const uint4*>(src + ((base) + 7) * 4 * T_OP);
#define TOP_ST8(base) *reinterpret_cast<uint4*>(dst + ((base) + 0) * 4 * p.ld_gcat) = v0; *reinterpret_cast<uint4*>(dst + ((base) + 1) * 4 * p.ld_gcat) = v1; \
                      *reinterpret_cast<uint4*>(dst + ((base) + 2) * 4 * p.ld_gcat) = v2; *reinterpret_cast<uint4*>(dst + ((base) + 3) * 4 * p.ld_gcat) = v3; \
                      *reinterpret_cast<uint4*>(dst + ((base) + 4) * 4 * p.ld_gcat) = v4; *reinterpret_cast<uint4*>(dst + ((base) + 5) * 4 * p.ld_gcat) = v5; \
                      *reinterpret_cast<uint4*>(dst + ((base) + 6) * 4 * p.ld_gcat) = v6; *reinterpret_cast<uint4*>(dst + ((base) + 7) * 4 * p.ld_gcat) = v7;
            TOP_LD8(0)
            __builtin_amdgcn_sched_barrier(0);
            TOP_ST8(0)
            TOP_LD8(8)
            __builtin_amdgcn_sched_barrier(0);
            TOP_ST8(8)
        }
        TOP_STAMP(11)
        TOP_STAMP(12)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            sd[a] = sd1[a]; bw[0][a] = bw1[0][a]; bw[1][a] = bw1[1][a];
            sd1[a] = sd2[a]; bw1[0][a] = bw2[0][a]; bw1[1][a] = bw2[1][a];
        }
        t_barrier();
    }
}


// ---- the tangent sweep through the same three layers, in place over the interpolated rows of the saved activations
//        um0 = (x Wm^T) * act'(m0) -> m0;   umh = (um0 W1^T) * act'(mh) -> mh;   um1 = (umh W2^T + um0) * act'(m1) -> m1
// x (M, 512): the tangent of the concatenation; Wm [n0][>= 512], W1 / W2 [n0][>= 112] the "nt" operand copies.  Same tiles, same
// request-ahead scheme; the x tile (64 KB) travels in registers like the mask tiles.
// (the x tile's sixteen registers per thread are sixteen NAMED variables: as an array member of a struct hipcc kept them in scratch)
#define TOPT_X8A(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define TOPT_X8B(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define TOPT_X16(M) TOPT_X8A(M) TOPT_X8B(M)
__global__ __launch_bounds__(256, 1) void top_tangent_kernel(TopArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sU0 = smem;
    unsigned char* sUH = smem + T_IMG;
    unsigned char* sU1 = smem + 2 * T_IMG;
    unsigned char* sM1 = smem + 3 * T_IMG;
    unsigned char* sMH = smem + 4 * T_IMG;
    unsigned char* sM0 = smem + 5 * T_IMG;
    unsigned char* sX = smem + 6 * T_IMG;                                    // [64][T_OP]: the 512-column operand tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const long long ntiles = p.M / T_BM;
    if ((long long)blockIdx.x >= ntiles) return;

    bf16x8 w2f[T_KS], w1f[T_KS], wmf[32];                                    // merge: slice `wave`, 32 k-steps
    stage_rows<128>(p.w2, p.ldw2, p.n0, smem, tid);
    stage_rows<128>(p.w1, p.ldw1, p.n0, smem + 128 * T_P, tid);
    t_barrier();
#pragma unroll
    for (int ks = 0; ks < T_KS; ++ks) { w2f[ks] = frag(smem, wave, ks, r31, h); w1f[ks] = frag(smem + 128 * T_P, wave, ks, r31, h); }
    t_barrier();
#pragma unroll
    for (int j0 = 0; j0 < 32; j0 += 8) {                                   // Wm: 128 rows (zero beyond n0) x 64 chunks, pitch T_OP;
        u32x4 v[8];                                                        // eight requests, then their writes (see stage_rows)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = (tid >> 6) + 4 * (j0 + j), c = tid & 63;
            v[j] = *reinterpret_cast<const u32x4*>(p.wm + (long long)(row < p.n0 ? row : p.n0 - 1) * p.ldwm + c * 8);
            if (row >= p.n0) v[j] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(smem + ((tid >> 6) + 4 * (j0 + j)) * T_OP + (tid & 63) * 16) = v[j];
    }
    t_barrier();
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wmf[ks] = *reinterpret_cast<const bf16x8*>(smem + (32 * wave + r31) * T_OP + (2 * ks + h) * 16);
    t_barrier();

    const TileMap tm = tile_map(tid);
    const int xrow = tid >> 6, xc = tid & 63;                                // x chunk q = tid + 256 i: row xrow + 4 i, chunk xc
    TileRegs r1, rh, r0;
#define TOPT_DECL(i) uint4 rx##i;
    TOPT_X16(TOPT_DECL)
    const long long last = ntiles - 1 - ((ntiles - 1 - blockIdx.x) % gridDim.x);
    auto clampt = [&](long long t) { return t < ntiles ? t : last; };
#define TOPT_XLD(i) rx##i = *reinterpret_cast<const uint4*>(p.x + (row0_ + xrow + 4 * (i)) * p.ldx + xc * 8);
#define TOPT_XST(i) *reinterpret_cast<uint4*>(sX + (xrow + 4 * (i)) * T_OP + xc * 16) = rx##i;
    // requests in the order the loop issues them (a mask tile or half the x tile per layer boundary, see top_backward_kernel)
#define TOPT_REQUEST(t) { const long long row0_ = (t) * T_BM; \
        TOPT_X8A(TOPT_XLD) tile_load(p.m0, p.ld_m, row0_, tm, r0); TOPT_X8B(TOPT_XLD) tile_load(p.mh, p.ld_m, row0_, tm, rh); \
        tile_load(p.m1, p.ld_m, row0_, tm, r1); }
#define TOPT_TO_LDS() { tile_to_lds(r1, sM1, tm); tile_to_lds(rh, sMH, tm); tile_to_lds(r0, sM0, tm); TOPT_X16(TOPT_XST) }
    TOPT_REQUEST((long long)blockIdx.x)
    TOPT_TO_LDS()
    TOPT_REQUEST(clampt((long long)blockIdx.x + gridDim.x))
    t_barrier();
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * T_BM;
        // ---- the merge layer: K = 512, k ascending (as the generic kernel it replaces sums it)
        {
            bf16x8 fx[4][2];
            uint2 mk[2][4];
#define TOP_RDX(ks) { _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_) fx[(ks) & 3][a_] = *reinterpret_cast<const bf16x8*>(sX + (32 * a_ + r31) * T_OP + (2 * (ks) + h) * 16); }
            TOP_RDX(0) TOP_RDX(1) TOP_RDX(2)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) mk[a][gq] = mask_words(sM0, 32 * a + r31, 32 * wave + 8 * gq + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) {
                if (ks + 3 < 32) TOP_RDX(ks + 3)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wmf[ks], fx[ks & 3][a], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[a][4 * gq + e];
                    quad_out(v, mk[a][gq], sU0, 32 * a + r31, 32 * wave + 8 * gq + 4 * h, p.dneg);
                }
        }
        t_barrier();                                                         // (the x tile and the m0 tile have been read)
        tile_store(sU0, p.m0, p.ld_m, row0, tm);
        const long long row0_ = clampt(tile + 2 * (long long)gridDim.x) * T_BM;   // the tile requested during this one
        tile_to_lds(r0, sM0, tm); TOPT_X16(TOPT_XST)                         // the next tile's x and m0 go in, ...
        TOPT_X8A(TOPT_XLD) tile_load(p.m0, p.ld_m, row0_, tm, r0);           // ... the tile after that is requested: x rows 0 .. 31, m0
        small_layer(w1f, sU0, nullptr, sMH, sUH, wave, r31, h, p.dneg);
        t_barrier();
        tile_store(sUH, p.mh, p.ld_m, row0, tm);
        tile_to_lds(rh, sMH, tm);
        TOPT_X8B(TOPT_XLD) tile_load(p.mh, p.ld_m, row0_, tm, rh);
        small_layer(w2f, sUH, sU0, sM1, sU1, wave, r31, h, p.dneg);
        t_barrier();                                                         // (every mask image of this tile has been read)
        tile_store(sU1, p.m1, p.ld_m, row0, tm);
        tile_to_lds(r1, sM1, tm);
        tile_load(p.m1, p.ld_m, row0_, tm, r1);
        t_barrier();
    }
}

}  // namespace

extern "C" {
#ifdef DHAUG_TOP_TIMING
int dhaug_debug_top_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_top_stamps), sizeof(long long) * (n < 64 ? n : 64));
}
#endif

/* see include/dhaug.h */
int dhaug_critic_top_backward_bf16(const dhaug_top_desc* d, void* stream) {
    DHAUG_CHECK_PTR(d);
    DHAUG_CHECK(d->M >= 0 && d->M % T_BM == 0 && d->n0 >= 1 && d->n0 <= 112 && d->nc == 512, DHAUG_EUNSUPPORTED);
    DHAUG_CHECK(d->mask_act == DHAUG_ACT_RELU || d->mask_act == DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (d->M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(d->seed); DHAUG_CHECK_PTR(d->wout); DHAUG_CHECK_PTR(d->m1); DHAUG_CHECK_PTR(d->mh); DHAUG_CHECK_PTR(d->m0);
    DHAUG_CHECK_PTR(d->w2); DHAUG_CHECK_PTR(d->w1); DHAUG_CHECK_PTR(d->wm); DHAUG_CHECK_PTR(d->bits0); DHAUG_CHECK_PTR(d->bits1);
    DHAUG_CHECK_PTR(d->g2); DHAUG_CHECK_PTR(d->g1); DHAUG_CHECK_PTR(d->g0); DHAUG_CHECK_PTR(d->gcat);
    DHAUG_CHECK(d->ld_m >= 112 && d->ld_g >= 112 && d->ldw2 >= 112 && d->ldw1 >= 112 && d->ldwm >= 112 && d->ld_gcat >= 512 && d->ld_seed >= 1 &&
                d->ld_wout >= 1, DHAUG_EINVAL);
    DHAUG_CHECK(d->ld_m % 8 == 0 && d->ld_g % 8 == 0 && d->ldw2 % 8 == 0 && d->ldw1 % 8 == 0 && d->ldwm % 8 == 0 && d->ld_gcat % 8 == 0, DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(d->m1) && dhaug_aligned16(d->mh) && dhaug_aligned16(d->m0) && dhaug_aligned16(d->w2) && dhaug_aligned16(d->w1) &&
                dhaug_aligned16(d->wm) && dhaug_aligned16(d->g2) && dhaug_aligned16(d->g1) && dhaug_aligned16(d->g0) && dhaug_aligned16(d->gcat) &&
                dhaug_aligned16(d->bits0) && dhaug_aligned16(d->bits1), DHAUG_EALIGN);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(top_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    TopArgs a{};
    a.seed = d->seed; a.ld_seed = d->ld_seed; a.wout = d->wout; a.ld_wout = d->ld_wout;
    a.m1 = const_cast<uint16_t*>(d->m1); a.mh = const_cast<uint16_t*>(d->mh); a.m0 = const_cast<uint16_t*>(d->m0); a.ld_m = d->ld_m;
    a.w2 = d->w2; a.ldw2 = d->ldw2; a.w1 = d->w1; a.ldw1 = d->ldw1; a.wm = d->wm; a.ldwm = d->ldwm;
    a.bits0 = d->bits0; a.bits1 = d->bits1;
    a.g2 = d->g2; a.g1 = d->g1; a.g0 = d->g0; a.ld_g = d->ld_g; a.gcat = d->gcat; a.ld_gcat = d->ld_gcat;
    a.M = d->M; a.n0 = (int)d->n0; a.dneg = d->mask_act == DHAUG_ACT_RELU ? 0.0f : d->mask_slope;
    const long long tiles = d->M / T_BM;
    const unsigned grid = dhaug_persistent_grid(tiles);
    hipLaunchKernelGGL(top_backward_kernel, dim3(grid), dim3(256), T_LDS, (hipStream_t)stream, a);
    return dhaug_launch_status();
}

/* see include/dhaug.h */
int dhaug_critic_top_tangent_bf16(const dhaug_top_desc* d, void* stream) {
    DHAUG_CHECK_PTR(d);
    DHAUG_CHECK(d->M >= 0 && d->M % T_BM == 0 && d->n0 >= 1 && d->n0 <= 112 && d->nc == 512, DHAUG_EUNSUPPORTED);
    DHAUG_CHECK(d->mask_act == DHAUG_ACT_RELU || d->mask_act == DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (d->M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(d->x); DHAUG_CHECK_PTR(d->m1); DHAUG_CHECK_PTR(d->mh); DHAUG_CHECK_PTR(d->m0);
    DHAUG_CHECK_PTR(d->w2); DHAUG_CHECK_PTR(d->w1); DHAUG_CHECK_PTR(d->wm);
    DHAUG_CHECK(d->ld_m >= 112 && d->ldw2 >= 112 && d->ldw1 >= 112 && d->ldwm >= 512 && d->ldx >= 512, DHAUG_EINVAL);
    DHAUG_CHECK(d->ld_m % 8 == 0 && d->ldw2 % 8 == 0 && d->ldw1 % 8 == 0 && d->ldwm % 8 == 0 && d->ldx % 8 == 0, DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(d->m1) && dhaug_aligned16(d->mh) && dhaug_aligned16(d->m0) && dhaug_aligned16(d->w2) && dhaug_aligned16(d->w1) &&
                dhaug_aligned16(d->wm) && dhaug_aligned16(d->x), DHAUG_EALIGN);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(top_tangent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    TopArgs a{};
    a.x = d->x; a.ldx = d->ldx;
    a.m1 = const_cast<uint16_t*>(d->m1); a.mh = const_cast<uint16_t*>(d->mh); a.m0 = const_cast<uint16_t*>(d->m0); a.ld_m = d->ld_m;
    a.w2 = d->w2; a.ldw2 = d->ldw2; a.w1 = d->w1; a.ldw1 = d->ldw1; a.wm = d->wm; a.ldwm = d->ldwm;
    a.M = d->M; a.n0 = (int)d->n0; a.dneg = d->mask_act == DHAUG_ACT_RELU ? 0.0f : d->mask_slope;
    const unsigned grid = dhaug_persistent_grid(d->M / T_BM);
    hipLaunchKernelGGL(top_tangent_kernel, dim3(grid), dim3(256), T_LDS, (hipStream_t)stream, a);
    return dhaug_launch_status();
}

}  // extern "C"
