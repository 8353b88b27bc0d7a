// Fused DH forward-kinematics kernels (forward, generator-tail forward, and their reverse modes).
//
// Mapping: one wave (64 lanes) = one tile of 64 poses; one lane = one pose, whole skeleton in registers
// (dhaug_fk_math.h).  A workgroup is a single wave so that LDS (the occupancy limiter here) is allocated
// in small units: the forward kernels stage half tiles (6.3 KB per wave, 25 waves per CU).
//
// HBM <-> lane transposition goes through LDS: the tile's input rows are contiguous in memory
// (64 x 37 angles, 64 x 15 lengths, 64 x 3 root), so the wave copies them with 16-byte-per-lane coalesced
// loads into an LDS image with the same layout; lane i then reads row i (row strides 37 / 15 / 3 / 35 / 8
// dwords are odd -> conflict-free ds_read_b32).  Results are written to LDS with an odd row stride (49 / 97 /
// 37 ...) and leave as fully coalesced stores.  Algorithmic HBM traffic: 220 B in + 192 B out = 412 B/pose.
#include "dhaug_fk_math.h"

using namespace dhaug_fk;

namespace {

constexpr int TILE = 64;

// A workgroup is ONE wave, and a wave's LDS instructions execute in issue order: between the phases of the LDS
// transposition nothing has to be waited for except the LDS counter (and the compiler must not move accesses across).
// __syncthreads() here would also drain vmcnt, i.e. wait for every global store of the previous phase to be acknowledged
// (measured on the generator tail: 46 % of the wave's time, SQ_WAIT_ANY / SQ_WAVE_CYCLES).
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// coalesced copy of `count` floats global -> LDS (same linear layout); base of g is 16-byte aligned
__device__ __forceinline__ void stage_in(const float* __restrict__ g, float* __restrict__ l, int count, int lane) {
    const int n4 = count >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* l4 = reinterpret_cast<float4*>(l);
    for (int i = lane; i < n4; i += TILE) l4[i] = g4[i];
    const int rem = count & 3;
    if (lane < rem) l[(n4 << 2) + lane] = g[(n4 << 2) + lane];
}

// contiguous block of up to CMAX floats global -> LDS in two steps (see RowLoad): issue() puts every 16-byte request of
// the block in flight, commit() writes the LDS image -- linear, or with row stride LSTRIDE for rows of W floats
// (W a multiple of 4, so that a request never spans two rows)
template <int CMAX>
struct BlockLoad {
    static constexpr int K = ((CMAX + 3) / 4 + TILE - 1) / TILE;
    float4 t[K];
    float tail;
    __device__ __forceinline__ void issue(const float* __restrict__ g, int count, int lane) {
        const int n4 = count >> 2, rem = count & 3;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = lane + TILE * k;
            t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n4) t[k] = reinterpret_cast<const float4*>(g)[i];
        }
        tail = lane < rem ? g[(n4 << 2) + lane] : 0.f;
    }
    __device__ __forceinline__ void commit(float* __restrict__ l, int count, int lane) const {
        const int n4 = count >> 2, rem = count & 3;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = lane + TILE * k;
            if (i < n4) reinterpret_cast<float4*>(l)[i] = t[k];
        }
        if (lane < rem) l[(n4 << 2) + lane] = tail;
    }
    template <int W, int LSTRIDE>
    __device__ __forceinline__ void commit_strided(float* __restrict__ l, int count, int lane) const {
        static_assert(W % 4 == 0, "row width");
        const int n4 = count >> 2;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = lane + TILE * k, e = 4 * i, r = e / W, j = e - r * W;
            if (i < n4) {
                float* d = l + r * LSTRIDE + j;
                d[0] = t[k].x; d[1] = t[k].y; d[2] = t[k].z; d[3] = t[k].w;
            }
        }
    }
};

// LDS image with row stride LSTRIDE (odd) -> contiguous global rows of W floats.  One wave per workgroup has nobody to
// hide latency behind, so the copies are fully unrolled: all LDS reads of a pass are issued before the first store
// (a rolled loop pays one LDS round trip per dword), and rows of a multiple of 4 floats leave as 16-byte stores.
template <int W, int LSTRIDE, int ROWS = TILE / 2>
__device__ __forceinline__ void stage_out(const float* __restrict__ l, float* __restrict__ g, int rows, int lane) {
    if constexpr (W % 4 == 0) {
        constexpr int K = (ROWS * W / 4 + TILE - 1) / TILE;           // float4 per lane
        float4 t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = 4 * (lane + TILE * k), r = e / W, j = e - r * W;
            const float* src = l + r * LSTRIDE + j;
            if (r < ROWS) t[k] = make_float4(src[0], src[1], src[2], src[3]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = 4 * (lane + TILE * k), r = e / W;
            if (r < rows) *reinterpret_cast<float4*>(g + e) = t[k];
        }
    } else {
        constexpr int K = (ROWS * W + TILE - 1) / TILE;
        float t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = lane + TILE * k, r = e / W, j = e - r * W;
            if (r < ROWS) t[k] = l[r * LSTRIDE + j];
        }
        const int count = rows * W;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = lane + TILE * k;
            if (e < count) g[e] = t[k];
        }
    }
}
// (rolled form: the backward kernels sit at 232-244 registers, two waves per SIMD; the unrolled copy costs them a wave)
template <int W, int LSTRIDE>
__device__ __forceinline__ void stage_out_rolled(const float* __restrict__ l, float* __restrict__ g, int rows, int lane) {
    const int count = rows * W;
    for (int e = lane; e < count; e += TILE) {
        const int r = e / W, j = e - r * W;
        g[e] = l[r * LSTRIDE + j];
    }
}
template <int W, int LSTRIDE>
__device__ __forceinline__ void stage_in_strided(const float* __restrict__ g, float* __restrict__ l, int rows, int lane) {
    const int count = rows * W;
    for (int e = lane; e < count; e += TILE) {
        const int r = e / W, j = e - r * W;
        l[r * LSTRIDE + j] = g[e];
    }
}

// tanh = 1 - 2 / (exp(2|x|) + 1) on the hardware exp2 / rcp: |error| <= 1.2e-7 absolute (the quantity that matters: the
// result is scaled to degrees / metres), 7 instructions where the library routine takes ~30 and branches.  Forward and
// reverse mode use the same function.
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(fabsf(x) * 2.8853900817779268f);        // exp(2|x|); inf beyond 44: 1 - 0
    const float t = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    return copysignf(t, x);
}
__device__ __forceinline__ float tanh_acc(float x) { return tanh_fast(x); }

// MODE 0: in0 = angles (N,37), in2 = root (N,3).      MODE 1: in0 = head (N,35), in2 = scaler (N,8) or null.
template <int MODE> struct InLayout;
template <> struct InLayout<0> { static constexpr int W0 = 37, W2 = 3; };
template <> struct InLayout<1> { static constexpr int W0 = 35, W2 = 8; };

// Loads this lane's pose from the LDS image and produces ang[37], bl[15], root.
template <int MODE, bool PREANGLE>
__device__ __forceinline__ void load_pose(const float* __restrict__ l0, const float* __restrict__ l1,
                                          const float* __restrict__ l2, bool has2, int lane,
                                          float* __restrict__ ang, float* __restrict__ bl, V3& root,
                                          float* __restrict__ th /*35, MODE 1 only*/) {
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 37; ++j) ang[j] = l0[lane * 37 + j];
#pragma unroll
        for (int j = 0; j < 15; ++j) bl[j] = l1[lane * 15 + j];
        root = mk(l2[lane * 3 + 0], l2[lane * 3 + 1], l2[lane * 3 + 2]);
    } else {
#pragma unroll
        for (int j = 0; j < 35; ++j) th[j] = tanh_acc(l0[lane * 35 + j]);
        tail_angles<PREANGLE>(th, ang);
        root = mk(th[32] * 10.0f, th[33] * 10.0f, th[34] * 10.0f);
        float sc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[j] = has2 ? l2[lane * 8 + j] : 0.0f;
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            const float len = l1[lane * 15 + j];
            bl[j] = kJitterCol[j] < 0 ? len : len * (1.0f + sc[kJitterCol[j] < 0 ? 0 : kJitterCol[j]]);
        }
    }
}

constexpr int kTo32[16] = {0, 1, 2, 3, 6, 7, 8, 12, 13, 15, 17, 18, 19, 25, 26, 27};

// Global rows <-> lanes through a HALF-tile LDS image (32 rows): the two halves of the wave take turns, all 64 lanes
// copy.  Halving the image is what sets the occupancy of these kernels: 64 x 55 input floats + padding cost 14.8 KB per
// wave (10 waves per CU, VALU 55 % busy, 69 % of wave time waiting); 32 x 49 floats are 6.3 KB (25 waves per CU).
// global rows -> lanes in two steps, so that the loads of ALL inputs of a tile are in flight together: issue() requests
// both half tiles (16 bytes per lane and request, fully unrolled), commit() runs them through the half-tile LDS image.
template <int W>
struct RowLoad {
    static constexpr int N4 = (TILE / 2) * W / 4;                     // whole float4 per half tile
    static constexpr int REM = (TILE / 2) * W - 4 * N4;               // (< 4 floats: W odd)
    static constexpr int K = (N4 + TILE - 1) / TILE;
    float4 t[2][K];
    float tail[2];
    __device__ __forceinline__ void issue(const float* __restrict__ g, int rows, int lane) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int nr = rows - 32 * half < 0 ? 0 : (rows - 32 * half > 32 ? 32 : rows - 32 * half);
            const int n4 = (nr * W) >> 2, rem = (nr * W) & 3;
            const float* gh = g + 32 * half * W;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int i = lane + TILE * k;
                t[half][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n4) t[half][k] = reinterpret_cast<const float4*>(gh)[i];
            }
            tail[half] = 0.f;
            if (lane < rem) tail[half] = gh[(n4 << 2) + lane];
        }
    }
    __device__ __forceinline__ void commit(float* __restrict__ lds, int rows, int lane, float* __restrict__ v) const {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int nr = rows - 32 * half < 0 ? 0 : (rows - 32 * half > 32 ? 32 : rows - 32 * half);
            const int n4 = (nr * W) >> 2, rem = (nr * W) & 3;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int i = lane + TILE * k;
                if (i < n4) reinterpret_cast<float4*>(lds)[i] = t[half][k];
            }
            if (lane < rem) lds[(n4 << 2) + lane] = tail[half];
            wave_lds_sync();
            if ((lane >> 5) == half) {
                const int r = (lane & 31) < nr ? (lane & 31) : 0;      // idle lanes recompute a valid row (never stored)
#pragma unroll
                for (int j = 0; j < W; ++j) v[j] = lds[r * W + j];
            }
            wave_lds_sync();
        }
    }
};
template <int W>
__device__ __forceinline__ void rows_to_lanes(const float* __restrict__ g, float* __restrict__ lds, int rows, int lane,
                                              float* __restrict__ v) {
    RowLoad<W> ld;
    ld.issue(g, rows, lane);
    ld.commit(lds, rows, lane, v);
}
template <int W, int S>
__device__ __forceinline__ void lanes_to_rows(const float* __restrict__ v, float* __restrict__ lds, float* __restrict__ g,
                                              int rows, int lane) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int nr = rows - 32 * half < 0 ? 0 : (rows - 32 * half > 32 ? 32 : rows - 32 * half);
        if ((lane >> 5) == half) {
#pragma unroll
            for (int j = 0; j < W; ++j) lds[(lane & 31) * S + j] = v[j];
        }
        wave_lds_sync();
        stage_out<W, S>(lds, g + 32 * half * W, nr, lane);
        wave_lds_sync();
    }
}

// Optional outputs of the generator tail for the critics (all may be null): the root-relative pose and its bf16 KCS
// operand (3D critic), the H36M projection through one camera (2D critic) -- the joints are in registers here, a
// separate pass would read them back from HBM.
struct TailExtra {
    void* centered;             // (N,48) fp32, or bf16 if in_bf16
    uint16_t* kcs;              // (N,32) bf16: 15 cosines, 15 lengths, 2 zeros
    void* proj2d;               // (N,32) fp32, or bf16 if in_bf16
    int in_bf16;                // 1: centered / proj2d leave as bf16 (what the critics' LOAD units round them to anyway)
    float q[4], t[3], c[9];     // camera
    int draw;                   // 1: draw the bone-length jitter here (Philox4x32-10) instead of reading it
    unsigned long long seed, offset;
    float* scaler_out;          // optional (N,8): the jitter that was drawn
};

// Philox4x32-10 (Salmon et al., SC'11), counter = (pose index, offset), key = seed: 4 x 32 random bits per call
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                           unsigned (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

template <int MODE, int OUTJ, bool PREANGLE>
__global__ __launch_bounds__(TILE) __attribute__((amdgpu_waves_per_eu((MODE == 0 && OUTJ == 16) ? 3 : 1))) void fk_forward_kernel(const float* __restrict__ in0,
                                                          const float* __restrict__ bone_len,
                                                          const float* __restrict__ in2,
                                                          float* __restrict__ out, float* __restrict__ angles_out,
                                                          long long N) {
    using L = InLayout<MODE>;
    constexpr int OW = OUTJ * 3, OS = OW + 1;                  // output row width / odd LDS stride
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const long long ntiles = (N + TILE - 1) / TILE;
    const bool has2 = in2 != nullptr;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        const int rows = (int)((N - base) < TILE ? (N - base) : TILE);
        float v0[L::W0], bl[15], v2[L::W2];
        {
            RowLoad<L::W0> ld0;
            RowLoad<15> ld1;
            ld0.issue(in0 + base * L::W0, rows, lane);
            ld1.issue(bone_len + base * 15, rows, lane);
            ld0.commit(smem, rows, lane, v0);
            ld1.commit(smem, rows, lane, bl);
        }
#pragma unroll
        for (int j = 0; j < L::W2; ++j) v2[j] = 0.0f;
        if (has2) rows_to_lanes<L::W2>(in2 + base * L::W2, smem, rows, lane, v2);

        float ang[37];
        V3 root, p[16];
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 37; ++j) ang[j] = v0[j < L::W0 ? j : 0];
            root = mk(v2[0], v2[1], v2[2]);
        } else {
            float th[35];
#pragma unroll
            for (int j = 0; j < 35; ++j) th[j] = tanh_acc(v0[j < L::W0 ? j : 0]);
            tail_angles<PREANGLE>(th, ang);
            root = mk(th[32] * 10.0f, th[33] * 10.0f, th[34] * 10.0f);
#pragma unroll
            for (int j = 0; j < 15; ++j)
                bl[j] = kJitterCol[j] < 0 ? bl[j] : bl[j] * (1.0f + v2[kJitterCol[j] < 0 ? 0 : (kJitterCol[j] < L::W2 ? kJitterCol[j] : 0)]);
        }
        // MODE 0 keeps the range guard in front of every sincos (user angles).  (tried: one test per pose on max|angle| and an
        // unguarded body -- without the 37 branches the scheduler interleaves all the polynomials, the kernel spills at its
        // three-waves-per-SIMD register budget and N = 4 Mi takes 481 us instead of 349.)  The tail's angles are bounded.
        fk_pose<MODE == 0>(ang, bl, p);

        float o[OW];
        if (OUTJ == 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { o[3 * j] = p[j].x + root.x; o[3 * j + 1] = p[j].y + root.y; o[3 * j + 2] = p[j].z + root.z; }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) { o[3 * j] = root.x; o[3 * j + 1] = root.y; o[3 * j + 2] = root.z; }   // rows the reference never writes stay = root
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int sidx = kTo32[j];
                o[3 * sidx] = p[j].x + root.x; o[3 * sidx + 1] = p[j].y + root.y; o[3 * sidx + 2] = p[j].z + root.z;
            }
            // slot 14 ('Neck/Nose') = slot 15 ('Head'), forward_kinematics_DH_model.py:787-793
            o[42] = p[9].x + root.x; o[43] = p[9].y + root.y; o[44] = p[9].z + root.z;
        }
        lanes_to_rows<OW, OS>(o, smem, out + base * OW, rows, lane);
        if (MODE == 1 && angles_out != nullptr) lanes_to_rows<37, 37>(ang, smem, angles_out + base * 37, rows, lane);
    }
}

// ---------------------------------------------------------------------------------------------------
// Generator tail for the critics, FOUR waves per tile of 64 poses.
//
// One lane per pose and one wave per tile leaves a single wave on every SIMD at B = 65 536 (1 024 tiles, 1 024 SIMDs):
// nothing hides a load, an LDS round trip or a transcendental, and the whole skeleton (31 joint transforms, 35 tanh) is
// one dependency chain per lane.  Here a tile is a workgroup of four waves and the SKELETON is what is split: lane l of
// every wave works on pose l, wave role r on one group of kinematic chains (compile-time DH constants stay: the roles are
// wave-uniform code paths, not lane-divergent ones):
//     role 0  both legs                       role 2  right arm (starts at body frame 8)
//     role 1  body frames 0..8, then the head   role 3  left arm  (starts at body frame 8)
// Role 1 hands frame 8 to the arm waves through LDS (they have their sines / cosines ready by then); every role
// rebuilds the global rotation and the root it needs (3 + 3 tanh, 3 sincos: cheaper than a second hand-over).  The joints
// meet in an LDS pose image; the critics' operands are then dealt out again by role (KCS features | centred pose | camera
// projection of joints 0..7 | of joints 8..15) and everything leaves through coalesced 16-byte stores issued by all 256
// threads.  Roles rotate with the tile index so that a SIMD does not collect four waves of the heaviest role.
// tanh is 1 - 2 / (exp(2|x|) + 1) on the hardware exp2 / rcp (|error| <= 1.2e-7 absolute, 7 instructions; the library
// routine is ~30 and branches); the joint angles are bounded by construction, so sincos skips the large-argument guard.
// ---------------------------------------------------------------------------------------------------
constexpr int T4_THREADS = 256;
constexpr int T4_HEAD = 0, T4_BONE = TILE * 35, T4_FRAME = T4_BONE + TILE * 15;      // phase A: inputs, body frame 8 (stride 13)
constexpr int T4_CEN = 0, T4_KCS = TILE * 49;                                        // phase B (aliases phase A's region)
constexpr int T4_X = T4_KCS + TILE * 17;
constexpr int T4_POSE = T4_X;                                                        // 64 x 49
constexpr int T4_PROJ = T4_POSE + TILE * 49;                                         // 64 x 33 (fp32) or 64 x 17 (bf16 pairs)
constexpr int T4_RG = T4_PROJ;                                                       // phase A: global rotation + root, 64 x 13
constexpr int T4_SCAL = T4_PROJ + TILE * 33;                                         // 64 x 9
constexpr int T4_FLOATS = T4_SCAL + TILE * 9;                                        // 10 048 floats = 39.25 KB: four tiles per CU
static_assert(T4_FRAME + TILE * 13 <= T4_X, "phase A region");

__device__ __forceinline__ void t4_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// k / 1000 correctly rounded for the integers of the jitter draw (checked exhaustively on the host, tests/test_cpu_boundary.py)
__device__ __forceinline__ float div1000(float k) {
    const float q = k * 0.001f;
    return fmaf(fmaf(-q, 1000.0f, k), 0.001f, q);
}
// a / b within an ulp: reciprocal + one correction step each (the IEEE sequence is ~11 instructions)
__device__ __forceinline__ float rcp_nr(float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, r, 1.0f), r, r);
}
__device__ __forceinline__ float div_nr(float a, float rb, float b) {              // rb = rcp_nr(b)
    const float q = a * rb;
    return fmaf(fmaf(-b, q, a), rb, q);
}
__device__ __forceinline__ float pack2(float a, float b) {                          // two bf16 (RNE) in one dword
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 f = {a, b};
    return __builtin_bit_cast(float, __builtin_convertvector(f, bf2));
}

// the H36M projection of a camera-space point (second half of w2c_project, dhaug_fk_math.h), the two perspective
// divisions taken through one corrected reciprocal
__device__ __forceinline__ void project_nr(V3 xc, const float* __restrict__ c, float& ox, float& oy) {
    const float rz = rcp_nr(xc.z);
    const float u = dhaug_clamp_pm1(div_nr(xc.x, rz, xc.z)), v = dhaug_clamp_pm1(div_nr(xc.y, rz, xc.z));
    const float r2 = u * u + v * v;
    const float radial = 1.0f + (c[4] * r2 + c[5] * (r2 * r2) + c[6] * (r2 * r2 * r2));
    const float tan = c[7] * u + c[8] * v;
    ox = c[0] * (u * (radial + tan) + c[7] * r2) + c[2];
    oy = c[1] * (v * (radial + tan) + c[8] * r2) + c[3];
}
// theta0 (degrees) of leg slot i (0..3 right, 5..8 left) and of body frame i (0..8)
__device__ __forceinline__ constexpr float leg_theta0(int i) {
    constexpr float t[9] = {0.0f, -90.0f, 180.0f, 0.0f, 0.0f, 180.0f, -90.0f, 0.0f, 0.0f};
    return t[i];
}
__device__ __forceinline__ constexpr float body_theta0(int i) { return i == 0 ? 90.0f : -90.0f; }

// angle slot i of the pose whose head row is `h` (R/models_Fk_GAN/Fk_generator.py:121-168)
template <bool PREANGLE>
__device__ __forceinline__ float slot_angle(int i, const float* __restrict__ h) {
    const int col = kSlotCol[i];
    if (col < 0) return 0.0f;
    const float t = tanh_fast(h[col]);
    return PREANGLE ? (t * (kAngHi[i] - kAngLo[i])) * 0.5f + (kAngHi[i] + kAngLo[i]) * 0.5f : t * 180.0f;
}

// rows of W floats: LDS image with row stride S (odd) -> global, 16 bytes per thread and request
template <int W, int S>
__device__ __forceinline__ void t4_store_rows(const float* __restrict__ l, float* __restrict__ g, int rows, int tid) {
    static_assert(W % 4 == 0, "row width");
    constexpr int Q = W / 4, K = (TILE * Q + T4_THREADS - 1) / T4_THREADS;
    asm volatile("" : "+v"(tid));                              // (addresses are derived here, not hoisted out of the tile loop and spilled)
    float4 t[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + T4_THREADS * k, r = e / Q, j = (e - r * Q) * 4;
        const float* src = l + (r < TILE ? r : 0) * S + j;
        t[k] = make_float4(src[0], src[1], src[2], src[3]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + T4_THREADS * k, r = e / Q;
#ifdef T4_ABL_NOSTORE
        asm volatile("" :: "v"(t[k].x), "v"(t[k].y), "v"(t[k].z), "v"(t[k].w), "v"(g));
#else
        if (r < rows) reinterpret_cast<float4*>(g)[e] = t[k];
#endif
    }
}

template <bool PREANGLE>
__global__ __launch_bounds__(T4_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void gen_tail4_kernel(const float* __restrict__ head, const float* __restrict__ bone_len,
                                                               const float* __restrict__ scaler, float* __restrict__ out,
                                                               TailExtra ex, long long N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long ntiles = (N + TILE - 1) / TILE;
    const bool has2 = scaler != nullptr;
    float* pose = smem + T4_POSE;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        const int rows = (int)((N - base) < TILE ? (N - base) : TILE);
        const int role = (wave + (int)tile) & 3;
        {   // inputs of the tile -> LDS (linear images; row strides 35 / 15 are odd), all requests in flight together
            const float* gh = head + base * 35;
            const float* gb = bone_len + base * 15;
            const int n4h = (rows * 35) >> 2, n4b = (rows * 15) >> 2;
            float4 th[3], tb, ts = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = tid + T4_THREADS * k;
                th[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n4h) th[k] = reinterpret_cast<const float4*>(gh)[i];
            }
            tb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < n4b) tb = reinterpret_cast<const float4*>(gb)[tid];
            const bool rd_s = has2 && !ex.draw && tid < rows * 2;
            if (rd_s) ts = reinterpret_cast<const float4*>(scaler + base * 8)[tid];
            float remh = 0.f, remb = 0.f;
            const int rh = (rows * 35) & 3, rb = (rows * 15) & 3;
            if (tid < rh) remh = gh[4 * n4h + tid];
            if (tid < rb) remb = gb[4 * n4b + tid];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = tid + T4_THREADS * k;
                if (i < n4h) reinterpret_cast<float4*>(smem + T4_HEAD)[i] = th[k];
            }
            if (tid < n4b) reinterpret_cast<float4*>(smem + T4_BONE)[tid] = tb;
            if (tid < rh) smem[T4_HEAD + 4 * n4h + tid] = remh;
            if (tid < rb) smem[T4_BONE + 4 * n4b + tid] = remb;
            if (rd_s) {
                float* d = smem + T4_SCAL + (tid >> 1) * 9 + (tid & 1) * 4;
                d[0] = ts.x; d[1] = ts.y; d[2] = ts.z; d[3] = ts.w;
            }
        }
        t4_barrier();

#ifndef T4_ABL_NOCOMPUTE
        const int src = lane < rows ? lane : 0;                // idle lanes of a ragged tile recompute pose 0 (never stored)
        const float* hrow = smem + T4_HEAD + src * 35;
        const float* brow = smem + T4_BONE + src * 15;
        float* prow = pose + lane * 49;
        float* jrow = smem + T4_SCAL + lane * 9;               // the pose's jitter (8) ...
        float* grow = smem + T4_RG + lane * 13;                // ... and its global rotation (9) + root (3)
        float ang[37], bl[15];
        V3 p[16];
        ArmSC arm;
        // ---- before hand-over 0: sines / cosines of the role's chains; role 2 also builds the global rotation and the root,
        // roles 2 / 3 draw the bone-length jitter (Philox4x32-10, R/models_Fk_GAN/Fk_generator.py:196-203): columns 4..7 / 0..3
        auto draw_half = [&](int half) {
            const unsigned long long idx = (unsigned long long)(base + lane);
            unsigned r[4];
            philox4x32((unsigned)idx, (unsigned)(idx >> 32), (unsigned)ex.offset + half, (unsigned)(ex.offset >> 32),
                       (unsigned)ex.seed, (unsigned)(ex.seed >> 32), r);
#pragma unroll
            for (int q = 0; q < 4; ++q) jrow[4 * half + q] = div1000((float)((int)(r[q] % 400u) - 200));
        };
        // legs and body: sincos of slot i sits in sc[2i], sc[2i+1] once computed; the chains run after the hand-over
        float ssc[13], csc[13];
        if (role == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sincos_deg_t<false>(leg_theta0(i) + slot_angle<PREANGLE>(i, hrow), ssc[i], csc[i]);
                sincos_deg_t<false>(leg_theta0(5 + i) + slot_angle<PREANGLE>(5 + i, hrow), ssc[4 + i], csc[4 + i]);
            }
        } else if (role == 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) sincos_deg_t<false>(body_theta0(i) + slot_angle<PREANGLE>(10 + i, hrow), ssc[i], csc[i]);
        } else if (role == 2) {
            if (ex.draw) draw_half(1);
#pragma unroll
            for (int i = 34; i < 37; ++i) ang[i] = slot_angle<PREANGLE>(i, hrow);
            float sgx, cgx;
            const Rot R = fk_global<false>(ang, sgx, cgx);
            grow[0] = R.r0.x; grow[1] = R.r0.y; grow[2] = R.r0.z; grow[3] = R.r1.x; grow[4] = R.r1.y; grow[5] = R.r1.z;
            grow[6] = R.r2.x; grow[7] = R.r2.y; grow[8] = R.r2.z;
            grow[9] = tanh_fast(hrow[32]) * 10.0f; grow[10] = tanh_fast(hrow[33]) * 10.0f; grow[11] = tanh_fast(hrow[34]) * 10.0f;
#pragma unroll
            for (int i = 23; i < 27; ++i) ang[i] = slot_angle<PREANGLE>(i, hrow);
            arm = fk_right_arm_sc<false>(ang);
        } else {
            if (ex.draw) draw_half(0);
#pragma unroll
            for (int i = 28; i < 32; ++i) ang[i] = slot_angle<PREANGLE>(i, hrow);
            arm = fk_left_arm_sc<false>(ang);
        }
        t4_barrier();                                          // hand-over 0: jitter, global rotation, root
        Rot Rg;
        Rg.r0 = mk(grow[0], grow[1], grow[2]); Rg.r1 = mk(grow[3], grow[4], grow[5]); Rg.r2 = mk(grow[6], grow[7], grow[8]);
        const V3 root = mk(grow[9], grow[10], grow[11]);
        const bool jittered = ex.draw || has2;
        auto bone = [&](int b) -> float {
            const float len = brow[b];
            if (kJitterCol[b] < 0) return len;
            const float* jr = smem + T4_SCAL + (ex.draw ? lane : src) * 9;
            return len * (1.0f + (jittered ? jr[kJitterCol[b] < 0 ? 0 : kJitterCol[b]] : 0.0f));
        };
        auto put = [&](int q) {
            prow[3 * q] = p[q].x + root.x; prow[3 * q + 1] = p[q].y + root.y; prow[3 * q + 2] = p[q].z + root.z;
        };
        Frame B;
        if (role == 0) {
#pragma unroll
            for (int b = 0; b < 6; ++b) bl[b] = bone(b);
            p[0] = mk(0.0f, 0.0f, 0.0f);
            fk_right_leg_sc(ssc, csc, bl, Rg, p);
            fk_left_leg_sc(ssc + 4, csc + 4, bl, Rg, p);
#pragma unroll
            for (int q = 0; q < 7; ++q) put(q);
        } else if (role == 1) {
            bl[6] = bone(6); bl[7] = bone(7);
            B = fk_body_sc(ssc, csc, bl, Rg, p);
            float* f = smem + T4_FRAME + lane * 13;
            f[0] = B.c0.x; f[1] = B.c0.y; f[2] = B.c0.z; f[3] = B.c1.x; f[4] = B.c1.y; f[5] = B.c1.z;
            f[6] = B.c2.x; f[7] = B.c2.y; f[8] = B.c2.z; f[9] = B.t.x; f[10] = B.t.y; f[11] = B.t.z;
            put(7); put(8);
        }
        t4_barrier();                                          // hand-over 1: body frame 8
        if (role == 1) {
#pragma unroll
            for (int i = 19; i < 22; ++i) ang[i] = slot_angle<PREANGLE>(i, hrow);
            bl[14] = bone(14);
            fk_head<false>(ang, bl, Rg, B, p);
            put(9);
        } else if (role >= 2) {
            const float* f = smem + T4_FRAME + lane * 13;
            B.c0 = mk(f[0], f[1], f[2]); B.c1 = mk(f[3], f[4], f[5]); B.c2 = mk(f[6], f[7], f[8]); B.t = mk(f[9], f[10], f[11]);
            if (role == 2) {
                bl[9] = bone(9); bl[11] = bone(11); bl[13] = bone(13);
                fk_right_arm(arm, bl, Rg, B, p);
                put(13); put(14); put(15);
            } else {
                bl[8] = bone(8); bl[10] = bone(10); bl[12] = bone(12);
                fk_left_arm(arm, bl, Rg, B, p);
                put(10); put(11); put(12);
            }
        }
        t4_barrier();                                          // the pose image is complete; phase A's inputs are dead
        // the poses leave now: these stores are in flight while the critics' operands are computed
        t4_store_rows<48, 49>(pose, out + base * 48, rows, tid);

        if (role == 0) {
            if (ex.kcs != nullptr) {
                V3 pw[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) pw[q] = mk(prow[3 * q], prow[3 * q + 1], prow[3 * q + 2]);
                // 15 cosines between adjacent bones, 15 lengths (kcs_features), on the hardware sqrt / rcp: the operand
                // is rounded to bf16 (2^-9 relative) right below
                V3 b[15];
                float len[15];
#pragma unroll
                for (int i = 0; i < 15; ++i) { b[i] = pw[kcs_bone_c(i)] - pw[kcs_bone_p(i)]; len[i] = __builtin_amdgcn_sqrtf(dot(b[i], b[i])); }
                float f[32];
#pragma unroll
                for (int k = 0; k < 15; ++k) {
                    const float den = len[kcs_i(k)] * len[kcs_j(k)];
                    f[k] = div_nr(dot(b[kcs_i(k)], b[kcs_j(k)]), rcp_nr(den), den);
                }
#pragma unroll
                for (int i = 0; i < 15; ++i) f[15 + i] = len[i];
                f[30] = 0.0f; f[31] = 0.0f;
                float* d = smem + T4_KCS + lane * 17;
#pragma unroll
                for (int i = 0; i < 16; ++i) d[i] = pack2(f[2 * i], f[2 * i + 1]);
            }
        } else if (role == 1) {
            if (ex.centered != nullptr) {
                float o[48];
#pragma unroll
                for (int i = 0; i < 48; ++i) o[i] = prow[i];
                if (ex.in_bf16) {
                    float* d = smem + T4_CEN + lane * 25;
#pragma unroll
                    for (int i = 0; i < 24; ++i) d[i] = pack2(o[2 * i] - o[(2 * i) % 3], o[2 * i + 1] - o[(2 * i + 1) % 3]);
                } else {
                    float* d = smem + T4_CEN + lane * 49;
#pragma unroll
                    for (int i = 0; i < 48; ++i) d[i] = o[i] - o[i % 3];
                }
            }
        } else if (ex.proj2d != nullptr) {
            const int j0 = role == 2 ? 0 : 8;
            // the camera's rotation as a matrix (columns = the quaternion formula applied to the unit vectors: wave-uniform,
            // once per tile), then 9 multiply-adds per joint instead of two cross products
            const V3 qv = mk(-ex.q[1], -ex.q[2], -ex.q[3]);
            V3 mc[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const V3 e = mk(a == 0 ? 1.0f : 0.0f, a == 1 ? 1.0f : 0.0f, a == 2 ? 1.0f : 0.0f);
                const V3 uv = cross(qv, e);
                mc[a] = e + 2.0f * (ex.q[0] * uv + cross(qv, uv));
            }
            float pr[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float* w = prow + 3 * (j0 + q);
                const V3 x = mk(w[0] - ex.t[0], w[1] - ex.t[1], w[2] - ex.t[2]);
                const V3 xc = axpy(x.z, mc[2], axpy(x.y, mc[1], x.x * mc[0]));
                project_nr(xc, ex.c, pr[2 * q], pr[2 * q + 1]);
            }
            if (ex.in_bf16) {
                float* d = smem + T4_PROJ + lane * 17 + j0;
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] = pack2(pr[2 * i], pr[2 * i + 1]);
            } else {
                float* d = smem + T4_PROJ + lane * 33 + 2 * j0;
#pragma unroll
                for (int i = 0; i < 16; ++i) d[i] = pr[i];
            }
        }
#endif
        t4_barrier();

        if (ex.centered != nullptr) {
            if (ex.in_bf16) t4_store_rows<24, 25>(smem + T4_CEN, static_cast<float*>(ex.centered) + base * 24, rows, tid);
            else t4_store_rows<48, 49>(smem + T4_CEN, static_cast<float*>(ex.centered) + base * 48, rows, tid);
        }
        if (ex.kcs != nullptr) t4_store_rows<16, 17>(smem + T4_KCS, reinterpret_cast<float*>(ex.kcs) + base * 16, rows, tid);
        if (ex.proj2d != nullptr) {
            if (ex.in_bf16) t4_store_rows<16, 17>(smem + T4_PROJ, static_cast<float*>(ex.proj2d) + base * 16, rows, tid);
            else t4_store_rows<32, 33>(smem + T4_PROJ, static_cast<float*>(ex.proj2d) + base * 32, rows, tid);
        }
        if (ex.draw && ex.scaler_out != nullptr) t4_store_rows<8, 9>(smem + T4_SCAL, ex.scaler_out + base * 8, rows, tid);
        t4_barrier();                                          // (a workgroup that takes another tile overwrites the images)
    }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
// LDS: [in0 64xW0 | bone 64x15 | in2 64xW2]  reused for the gradient image, then [grad_out 64 x 49].
template <int MODE, bool PREANGLE>
__global__ __launch_bounds__(TILE) void fk_backward_kernel(const float* __restrict__ in0,
                                                           const float* __restrict__ bone_len,
                                                           const float* __restrict__ in2,
                                                           const float* __restrict__ grad_out,
                                                           float* __restrict__ g0 /*grad angles | grad head*/,
                                                           float* __restrict__ g1 /*grad bone_len (MODE 0)*/,
                                                           float* __restrict__ g2 /*grad root (MODE 0)*/,
                                                           long long N) {
    using L = InLayout<MODE>;
    constexpr int GS = 49;
    constexpr int A_FLOATS = TILE * 64;                        // region A: >= 64 x (37+15+8), 16-byte multiple
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const long long ntiles = (N + TILE - 1) / TILE;
    float* l0 = smem;
    float* l1 = l0 + TILE * L::W0 + ((TILE * L::W0) & 3 ? 4 - ((TILE * L::W0) & 3) : 0);
    float* l2 = l1 + TILE * 15;
    float* lg = smem + A_FLOATS;                               // 64 x 49
    const bool has2 = in2 != nullptr;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        const int rows = (int)((N - base) < TILE ? (N - base) : TILE);
        {                                                      // all four inputs of the tile in flight together
            BlockLoad<TILE * L::W0> b0;
            BlockLoad<TILE * 15> b1;
            BlockLoad<TILE * L::W2> b2;
            BlockLoad<TILE * 48> bg;
            b0.issue(in0 + base * L::W0, rows * L::W0, lane);
            b1.issue(bone_len + base * 15, rows * 15, lane);
            if (MODE == 1 && has2) b2.issue(in2 + base * L::W2, rows * L::W2, lane);
            bg.issue(grad_out + base * 48, rows * 48, lane);
            b0.commit(l0, rows * L::W0, lane);
            b1.commit(l1, rows * 15, lane);
            if (MODE == 1 && has2) b2.commit(l2, rows * L::W2, lane);
            bg.template commit_strided<48, GS>(lg, rows * 48, lane);
        }
        __syncthreads();

        float ang[37], bl[15], th[35], gang[37], gbl[15];
        V3 root, groot;
        const int src = lane < rows ? lane : 0;
        load_pose<MODE, PREANGLE>(l0, l1, l2, has2, src, ang, bl, root, th);
        const float* mine = lg + src * GS;
        fk_pose_backward(ang, bl, [&](int j) { return mk(mine[3 * j], mine[3 * j + 1], mine[3 * j + 2]); },
                         gang, gbl, groot);
        __syncthreads();
        if (MODE == 0) {
            float* o0 = smem;                                  // 64 x 37 | 64 x 15 | 64 x 3
            float* o1 = o0 + TILE * 37;
            float* o2 = o1 + TILE * 15;
#pragma unroll
            for (int j = 0; j < 37; ++j) o0[lane * 37 + j] = gang[j];
#pragma unroll
            for (int j = 0; j < 15; ++j) o1[lane * 15 + j] = gbl[j];
            o2[lane * 3 + 0] = groot.x; o2[lane * 3 + 1] = groot.y; o2[lane * 3 + 2] = groot.z;
            __syncthreads();
            stage_out_rolled<37, 37>(o0, g0 + base * 37, rows, lane);
            stage_out_rolled<15, 15>(o1, g1 + base * 15, rows, lane);
            stage_out_rolled<3, 3>(o2, g2 + base * 3, rows, lane);
        } else {
            float* o0 = smem;                                  // 64 x 35
            float gh[35];
#pragma unroll
            for (int c = 0; c < 35; ++c) gh[c] = 0.0f;
#pragma unroll
            for (int i = 0; i < 37; ++i) {
                const int col = kSlotCol[i];
                if (col >= 0) gh[col] = gang[i] * tail_scale<PREANGLE>(i);
            }
            gh[32] = groot.x * 10.0f; gh[33] = groot.y * 10.0f; gh[34] = groot.z * 10.0f;
#pragma unroll
            for (int c = 0; c < 35; ++c) o0[lane * 35 + c] = gh[c] * fmaf(-th[c], th[c], 1.0f);
            __syncthreads();
            stage_out_rolled<35, 35>(o0, g0 + base * 35, rows, lane);
        }
        __syncthreads();
    }
}

template <typename K, typename... Args>
int launch_tiles(K kernel, size_t lds_bytes, long long N, void* stream, Args... args) {
    if (N == 0) return DHAUG_OK;
    const long long ntiles = (N + TILE - 1) / TILE;
    const int grid = dhaug_stream_grid(ntiles, 1, 256 * 10 * 4);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(TILE), lds_bytes, (hipStream_t)stream, args..., N);
    return dhaug_launch_status();
}

constexpr size_t fwd_lds(int w_in, int w_out) {                                    // half-tile images (rows_to_lanes)
    size_t a = (size_t)(TILE / 2) * (w_in + 1) * 4, b = (size_t)(TILE / 2) * (w_out + 1) * 4;
    return a > b ? a : b;
}

}  // namespace

extern "C" {

int dhaug_fk_forward(const float* angles, const float* bone_len, const float* root, float* out, int64_t N,
                     int out_joints, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    DHAUG_CHECK(out_joints == 16 || out_joints == 32, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(angles); DHAUG_CHECK_PTR(bone_len); DHAUG_CHECK_PTR(out);
    DHAUG_CHECK(dhaug_aligned16(angles) && dhaug_aligned16(bone_len) && dhaug_aligned16(root) && dhaug_aligned16(out),
                DHAUG_EALIGN);
    if (out_joints == 16)
        return launch_tiles(fk_forward_kernel<0, 16, true>, fwd_lds(37, 48), N, stream, angles, bone_len, root, out,
                            (float*)nullptr);
    return launch_tiles(fk_forward_kernel<0, 32, true>, fwd_lds(37, 96), N, stream, angles, bone_len, root, out,
                        (float*)nullptr);
}

int dhaug_fk_backward(const float* angles, const float* bone_len, const float* grad_out16, float* grad_angles,
                      float* grad_bone_len, float* grad_root, int64_t N, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(angles); DHAUG_CHECK_PTR(bone_len); DHAUG_CHECK_PTR(grad_out16);
    DHAUG_CHECK_PTR(grad_angles); DHAUG_CHECK_PTR(grad_bone_len); DHAUG_CHECK_PTR(grad_root);
    DHAUG_CHECK(dhaug_aligned16(angles) && dhaug_aligned16(bone_len), DHAUG_EALIGN);
    return launch_tiles(fk_backward_kernel<0, true>, (size_t)TILE * (64 + 49) * 4, N, stream, angles, bone_len,
                        (const float*)nullptr, grad_out16, grad_angles, grad_bone_len, grad_root);
}

int dhaug_gen_tail_forward(const float* head, const float* bone_len, const float* scaler, float* fake16,
                           float* angles_out, int64_t N, int use_preangle, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(head); DHAUG_CHECK_PTR(bone_len); DHAUG_CHECK_PTR(fake16);
    DHAUG_CHECK(dhaug_aligned16(head) && dhaug_aligned16(bone_len) && dhaug_aligned16(scaler), DHAUG_EALIGN);
    if (use_preangle)
        return launch_tiles(fk_forward_kernel<1, 16, true>, fwd_lds(37, 48), N, stream, head, bone_len, scaler, fake16,
                            angles_out);
    return launch_tiles(fk_forward_kernel<1, 16, false>, fwd_lds(37, 48), N, stream, head, bone_len, scaler, fake16,
                        angles_out);
}

int dhaug_gen_tail_forward_critics(const float* head, const float* bone_len, const float* scaler, float* fake16,
                                   void* centered, uint16_t* kcs_bf16, const float* quat, const float* trans,
                                   const float* cam9, void* proj2d, int draw_scaler, uint64_t rng_seed,
                                   uint64_t rng_offset, float* scaler_out, int64_t N, int use_preangle, int inputs_bf16,
                                   void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(head); DHAUG_CHECK_PTR(bone_len); DHAUG_CHECK_PTR(fake16);
    DHAUG_CHECK(dhaug_aligned16(head) && dhaug_aligned16(bone_len) && dhaug_aligned16(scaler), DHAUG_EALIGN);
    DHAUG_CHECK(proj2d == nullptr || (quat != nullptr && trans != nullptr && cam9 != nullptr), DHAUG_EINVAL);
    DHAUG_CHECK(!(draw_scaler && scaler != nullptr), DHAUG_EINVAL);
    TailExtra ex{};
    ex.centered = centered; ex.kcs = kcs_bf16; ex.proj2d = proj2d; ex.in_bf16 = inputs_bf16 ? 1 : 0;
    ex.draw = draw_scaler ? 1 : 0; ex.seed = rng_seed; ex.offset = rng_offset; ex.scaler_out = draw_scaler ? scaler_out : nullptr;
    if (proj2d != nullptr) {
        for (int i = 0; i < 4; ++i) ex.q[i] = quat[i];
        for (int i = 0; i < 3; ++i) ex.t[i] = trans[i];
        for (int i = 0; i < 9; ++i) ex.c[i] = cam9[i];
    }
    const long long ntiles = (N + TILE - 1) / TILE;
    const int grid = dhaug_stream_grid(ntiles, 1, 256 * 4 * 4);
    if (use_preangle)
        hipLaunchKernelGGL(gen_tail4_kernel<true>, dim3(grid), dim3(T4_THREADS), T4_FLOATS * sizeof(float), (hipStream_t)stream,
                           head, bone_len, scaler, fake16, ex, (long long)N);
    else
        hipLaunchKernelGGL(gen_tail4_kernel<false>, dim3(grid), dim3(T4_THREADS), T4_FLOATS * sizeof(float), (hipStream_t)stream,
                           head, bone_len, scaler, fake16, ex, (long long)N);
    return dhaug_launch_status();
}

int dhaug_gen_tail_backward(const float* head, const float* bone_len, const float* scaler, const float* grad_fake16,
                            float* grad_head, int64_t N, int use_preangle, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(head); DHAUG_CHECK_PTR(bone_len); DHAUG_CHECK_PTR(grad_fake16); DHAUG_CHECK_PTR(grad_head);
    DHAUG_CHECK(dhaug_aligned16(head) && dhaug_aligned16(bone_len) && dhaug_aligned16(scaler), DHAUG_EALIGN);
    const size_t lds = (size_t)TILE * (64 + 49) * 4;
    if (use_preangle)
        return launch_tiles(fk_backward_kernel<1, true>, lds, N, stream, head, bone_len, scaler, grad_fake16, grad_head,
                            (float*)nullptr, (float*)nullptr);
    return launch_tiles(fk_backward_kernel<1, false>, lds, N, stream, head, bone_len, scaler, grad_fake16, grad_head,
                        (float*)nullptr, (float*)nullptr);
}

}  // extern "C"
