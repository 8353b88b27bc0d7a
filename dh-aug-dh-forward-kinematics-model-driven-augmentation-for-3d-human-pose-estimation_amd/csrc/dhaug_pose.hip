// Pose-feature and camera kernels around the FK output: bone lengths, KCS features (value, VJP, JVP),
// world<->camera quaternion transforms, H36M projection, root-centring and left/right flip.
// All are HBM-streaming elementwise work on (N,16,C) rows: one lane per pose (KCS) or per joint (camera),
// rows transposed through LDS with an odd stride so that global accesses stay fully coalesced.
#include "dhaug_common.h"

namespace {

constexpr int TILE = 64;
constexpr int PS = 49;      // LDS row stride for 48-float pose rows (odd -> conflict-free)

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// used_16key_15bone_len_table (R/models_Fk_GAN/forward_kinematics_DH_model.py:46-49): bone i = joint[c] - joint[p]
constexpr int kBoneP[15] = {5, 2, 4, 1, 0, 0, 0, 7, 8, 8, 10, 13, 11, 14, 8};
constexpr int kBoneC[15] = {6, 3, 5, 2, 4, 1, 7, 8, 10, 13, 11, 14, 12, 15, 9};
// KCS cosine pairs (R/models_Fk_GAN/Fk_discriminator.py:81-140)
constexpr int kKcsI[15] = {0, 1, 2, 3, 4, 4, 5, 6, 7, 7, 7, 8, 9, 10, 11};
constexpr int kKcsJ[15] = {2, 3, 4, 5, 5, 6, 6, 7, 14, 8, 9, 10, 11, 12, 13};

template <int W, int LSTRIDE>
__device__ __forceinline__ void rows_in(const float* __restrict__ g, float* __restrict__ l, int rows, int lane) {
    const int count = rows * W;
    for (int e = lane; e < count; e += TILE) { const int r = e / W; l[r * LSTRIDE + (e - r * W)] = g[e]; }
}
template <int W, int LSTRIDE>
__device__ __forceinline__ void rows_out(const float* __restrict__ l, float* __restrict__ g, int rows, int lane) {
    const int count = rows * W;
    for (int e = lane; e < count; e += TILE) { const int r = e / W; g[e] = l[r * LSTRIDE + (e - r * W)]; }
}

__device__ __forceinline__ void load_joints(const float* __restrict__ l, int row, V3* p) {
#pragma unroll
    for (int j = 0; j < 16; ++j) p[j] = mk(l[row * PS + 3 * j], l[row * PS + 3 * j + 1], l[row * PS + 3 * j + 2]);
}
__device__ __forceinline__ void bones(const V3* p, V3* b, float* len) {
#pragma unroll
    for (int i = 0; i < 15; ++i) { b[i] = p[kBoneC[i]] - p[kBoneP[i]]; len[i] = sqrtf(dot(b[i], b[i])); }
}

// MODE 0: bone lengths (N,15).  MODE 1: KCS features fp32 and/or bf16.
template <int MODE>
__global__ __launch_bounds__(TILE) void kcs_forward_kernel(const float* __restrict__ pose, float* __restrict__ out_f32,
                                                           uint16_t* __restrict__ out_bf16, long long ld_bf16,
                                                           int with_lengths, long long N, float* __restrict__ centered = nullptr) {
    __shared__ float lp[TILE * PS];
    __shared__ float lo[TILE * 31];
    const int lane = threadIdx.x;
    const long long ntiles = (N + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        const int rows = (int)((N - base) < TILE ? (N - base) : TILE);
        rows_in<48, PS>(pose + base * 48, lp, rows, lane);
        __syncthreads();
        V3 p[16], b[15];
        float len[15];
        load_joints(lp, lane < rows ? lane : 0, p);
        bones(p, b, len);
        if (centered != nullptr) {                             // root-relative copy (the 3D critic's input), same pass
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                lp[lane * PS + 3 * j] = p[j].x - p[0].x; lp[lane * PS + 3 * j + 1] = p[j].y - p[0].y; lp[lane * PS + 3 * j + 2] = p[j].z - p[0].z;
            }
            __syncthreads();
            rows_out<48, PS>(lp, centered + base * 48, rows, lane);
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 15; ++i) lo[lane * 15 + i] = len[i];
            __syncthreads();
            rows_out<15, 15>(lo, out_f32 + base * 15, rows, lane);
        } else {
            float f[30];
#pragma unroll
            for (int k = 0; k < 15; ++k) f[k] = dot(b[kKcsI[k]], b[kKcsJ[k]]) / (len[kKcsI[k]] * len[kKcsJ[k]]);
#pragma unroll
            for (int i = 0; i < 15; ++i) f[15 + i] = len[i];
            const int W = with_lengths ? 30 : 15;
#pragma unroll
            for (int c = 0; c < 30; ++c) lo[lane * 31 + c] = f[c];
            __syncthreads();
            if (out_bf16 != nullptr) {
                // rows of ld_bf16 (even) bf16: one packed pair per lane and step -> coalesced 4-byte stores
                const int ld = (int)ld_bf16, pairs = rows * ld / 2;
                uint32_t* dst = reinterpret_cast<uint32_t*>(out_bf16 + base * ld_bf16);
                for (int e = lane; e < pairs; e += TILE) {
                    const int r = (2 * e) / ld, c = 2 * e - r * ld;
                    const float a = c < W ? lo[r * 31 + c] : 0.0f;
                    const float d = (c + 1) < W ? lo[r * 31 + c + 1] : 0.0f;
                    dst[e] = (uint32_t)dhaug_f32_to_bf16(a) | ((uint32_t)dhaug_f32_to_bf16(d) << 16);
                }
            }
            if (out_f32 != nullptr) {
                if (with_lengths) rows_out<30, 31>(lo, out_f32 + base * 30, rows, lane);
                else              rows_out<15, 31>(lo, out_f32 + base * 15, rows, lane);
            }
        }
        __syncthreads();
    }
}

// JVP == false: VJP, aux = grad_feat (N,W) -> res = grad_pose (N,48).
// JVP == true : aux = tangent (N,48)       -> res = tan_feat (N,W).
template <bool JVP>
__global__ __launch_bounds__(TILE) void kcs_diff_kernel(const float* __restrict__ pose, const float* __restrict__ aux,
                                                        float* __restrict__ res, int with_lengths, long long N) {
    __shared__ float lp[TILE * PS];
    __shared__ float la[TILE * PS];
    const int lane = threadIdx.x;
    const int W = with_lengths ? 30 : 15;
    const long long ntiles = (N + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        const int rows = (int)((N - base) < TILE ? (N - base) : TILE);
        rows_in<48, PS>(pose + base * 48, lp, rows, lane);
        if (JVP) rows_in<48, PS>(aux + base * 48, la, rows, lane);
        else if (with_lengths) rows_in<30, 31>(aux + base * 30, la, rows, lane);
        else rows_in<15, 15>(aux + base * 15, la, rows, lane);
        __syncthreads();
        const int row = lane < rows ? lane : 0;
        V3 p[16], b[15];
        float len[15];
        load_joints(lp, row, p);
        bones(p, b, len);
        if (JVP) {
            V3 tp[16], tb[15];
            float dl[15], out[30];
            load_joints(la, row, tp);
#pragma unroll
            for (int i = 0; i < 15; ++i) { tb[i] = tp[kBoneC[i]] - tp[kBoneP[i]]; dl[i] = dot(b[i], tb[i]) / len[i]; }
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const int i = kKcsI[k], j = kKcsJ[k];
                const float inv = 1.0f / (len[i] * len[j]);
                const float c = dot(b[i], b[j]) * inv;
                out[k] = (dot(tb[i], b[j]) + dot(b[i], tb[j])) * inv - c * (dl[i] / len[i] + dl[j] / len[j]);
            }
#pragma unroll
            for (int i = 0; i < 15; ++i) out[15 + i] = dl[i];
            __syncthreads();
            if (with_lengths) {
#pragma unroll
                for (int c = 0; c < 30; ++c) la[lane * 31 + c] = out[c];
                __syncthreads();
                rows_out<30, 31>(la, res + base * 30, rows, lane);
            } else {
#pragma unroll
                for (int c = 0; c < 15; ++c) la[lane * 15 + c] = out[c];
                __syncthreads();
                rows_out<15, 15>(la, res + base * 15, rows, lane);
            }
        } else {
            const int stride = with_lengths ? 31 : 15;
            V3 gb[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                const float gl = with_lengths ? la[row * stride + 15 + i] : 0.0f;
                gb[i] = (gl / len[i]) * b[i];
            }
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const int i = kKcsI[k], j = kKcsJ[k];
                const float g = la[row * stride + k];
                const float inv = 1.0f / (len[i] * len[j]);
                const float c = dot(b[i], b[j]) * inv;
                gb[i] = gb[i] + g * (inv * b[j] - (c / (len[i] * len[i])) * b[i]);
                gb[j] = gb[j] + g * (inv * b[i] - (c / (len[j] * len[j])) * b[j]);
            }
            V3 gp[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) gp[j] = mk(0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 15; ++i) { gp[kBoneC[i]] = gp[kBoneC[i]] + gb[i]; gp[kBoneP[i]] = gp[kBoneP[i]] - gb[i]; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                lp[lane * PS + 3 * j] = gp[j].x; lp[lane * PS + 3 * j + 1] = gp[j].y; lp[lane * PS + 3 * j + 2] = gp[j].z;
            }
            __syncthreads();
            rows_out<48, PS>(lp, res + base * 48, rows, lane);
        }
        __syncthreads();
    }
}

// The gradient penalty of the 3D critic between the backward chain and the tangent sweep (R/models_Fk_GAN/Fk_discriminator.py:205-231
// through the KCS features, :81-140), one launch for what were six (kcs_diff<false>, add_f32, gp_penalty, kcs_diff<true>, two
// cast_pad: 122 us of a 2.3 ms step at B = 65 536, tools/trace_step.py) -- one lane per interpolated pose:
//     g   = KCS^T(x) gk + gp                       dD/dx_hat: the KCS branch's input cotangent pulled back + the pose branch's
//     n   = ||g||_2,  pen = (n - 1)^2,  v = coef (n - 1) / n g      (0 where n = 0: torch's norm subgradient)
//     tk  = dKCS(x)[v]                             the KCS branch's tangent input; the pose branch's is v
// and the two tangent inputs leave as the bf16 operands the tangent sweep and sweep 4 read ((N, 32) and (N, 48), zero pads).
// Same operations in the same order as the kernels replaced -- the row's sum of squares in gp_penalty_kernel's butterfly order --
// (fp32 results equal to the last bit or two: hipcc contracts a few multiply-adds differently per kernel).
__global__ __launch_bounds__(TILE) void d3_penalty_kernel(const float* __restrict__ pose, const float* __restrict__ gk,
                                                          const float* __restrict__ gp, float coef, uint16_t* __restrict__ tk_bf16,
                                                          uint16_t* __restrict__ v_bf16, float* __restrict__ pen, long long N) {
    __shared__ float lp[TILE * PS];
    __shared__ float la[TILE * PS];
    const int lane = threadIdx.x;
    const long long ntiles = (N + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        const int rows = (int)((N - base) < TILE ? (N - base) : TILE);
        rows_in<48, PS>(pose + base * 48, lp, rows, lane);
        rows_in<30, 31>(gk + base * 30, la, rows, lane);
        __syncthreads();
        const int row = lane < rows ? lane : 0;
        V3 p[16], b[15];
        float len[15];
        load_joints(lp, row, p);
        bones(p, b, len);
        // ---- VJP (kcs_diff_kernel<false>, with lengths)
        V3 gb[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const float gl = la[row * 31 + 15 + i];
            gb[i] = (gl / len[i]) * b[i];
        }
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            const int i = kKcsI[k], j = kKcsJ[k];
            const float g = la[row * 31 + k];
            const float inv = 1.0f / (len[i] * len[j]);
            const float c = dot(b[i], b[j]) * inv;
            gb[i] = gb[i] + g * (inv * b[j] - (c / (len[i] * len[i])) * b[i]);
            gb[j] = gb[j] + g * (inv * b[i] - (c / (len[j] * len[j])) * b[j]);
        }
        V3 gj[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) gj[j] = mk(0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 15; ++i) { gj[kBoneC[i]] = gj[kBoneC[i]] + gb[i]; gj[kBoneP[i]] = gj[kBoneP[i]] - gb[i]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(gj[j].x), "+v"(gj[j].y), "+v"(gj[j].z));
        __syncthreads();                                                      // (every lane has read its gk row)
        rows_in<48, PS>(gp + base * 48, la, rows, lane);
        __syncthreads();
        // ---- + the pose branch's cotangent (add_f32_kernel: a + b with a the pulled-back one), the penalty (gp_penalty_kernel)
        // (the values that the separate launches hand over through memory are made opaque here: with -ffp-contract=fast hipcc would
        // otherwise contract across the seams -- g * g + ..., k g_c - k g_p -- and the bits would differ)
        float g[48];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            g[3 * j] = gj[j].x + la[row * PS + 3 * j]; g[3 * j + 1] = gj[j].y + la[row * PS + 3 * j + 1]; g[3 * j + 2] = gj[j].z + la[row * PS + 3 * j + 2];
        }
#pragma unroll
        for (int c = 0; c < 48; ++c) asm volatile("" : "+v"(g[c]));
        float t[64];                                                          // the wave-wide butterfly of gp_penalty_kernel, in one lane
#pragma unroll
        for (int c = 0; c < 64; ++c) {
            t[c] = c < 48 ? g[c] * g[c] : 0.0f;
            asm volatile("" : "+v"(t[c]));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
            for (int c = 0; c < o; ++c) t[c] = t[c] + t[c + o];
        const float n = sqrtf(t[0]), d = n - 1.0f;
        const float kk = n > 0.0f ? coef * d / n : 0.0f;
        V3 tp[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            tp[j] = mk(kk * g[3 * j], kk * g[3 * j + 1], kk * g[3 * j + 2]);
            asm volatile("" : "+v"(tp[j].x), "+v"(tp[j].y), "+v"(tp[j].z));
        }
        // ---- JVP (kcs_diff_kernel<true>, with lengths) of v
        V3 tb[15];
        float dl[15], out[30];
#pragma unroll
        for (int i = 0; i < 15; ++i) { tb[i] = tp[kBoneC[i]] - tp[kBoneP[i]]; dl[i] = dot(b[i], tb[i]) / len[i]; }
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            const int i = kKcsI[k], j = kKcsJ[k];
            const float inv = 1.0f / (len[i] * len[j]);
            const float c = dot(b[i], b[j]) * inv;
            out[k] = (dot(tb[i], b[j]) + dot(b[i], tb[j])) * inv - c * (dl[i] / len[i] + dl[j] / len[j]);
        }
#pragma unroll
        for (int i = 0; i < 15; ++i) out[15 + i] = dl[i];
        __syncthreads();                                                      // (every lane has read its gp row)
        if (lane < rows) pen[base + lane] = d * d;
#pragma unroll
        for (int j = 0; j < 16; ++j) { lp[lane * PS + 3 * j] = tp[j].x; lp[lane * PS + 3 * j + 1] = tp[j].y; lp[lane * PS + 3 * j + 2] = tp[j].z; }
#pragma unroll
        for (int c = 0; c < 30; ++c) la[lane * 31 + c] = out[c];
        __syncthreads();
        {   // v: rows of 48 bf16, tk: rows of 32 bf16 (30 + 2 zeros): one packed pair per lane and step
            uint32_t* dv = reinterpret_cast<uint32_t*>(v_bf16 + base * 48);
            for (int e = lane; e < rows * 24; e += TILE) {
                const int r = e / 24, c = 2 * (e - r * 24);
                dv[e] = (uint32_t)dhaug_f32_to_bf16(lp[r * PS + c]) | ((uint32_t)dhaug_f32_to_bf16(lp[r * PS + c + 1]) << 16);
            }
            uint32_t* dk = reinterpret_cast<uint32_t*>(tk_bf16 + base * 32);
            for (int e = lane; e < rows * 16; e += TILE) {
                const int r = e >> 4, c = 2 * (e & 15);
                const float a = c < 30 ? la[r * 31 + c] : 0.0f, bb = (c + 1) < 30 ? la[r * 31 + c + 1] : 0.0f;
                dk[e] = (uint32_t)dhaug_f32_to_bf16(a) | ((uint32_t)dhaug_f32_to_bf16(bb) << 16);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// camera: one lane per joint
// ---------------------------------------------------------------------------------------------------
struct Cam { float q[4], t[3], c[9]; };

__device__ __forceinline__ V3 qrot(float w, V3 q, V3 v) {         // R/common/quaternion.py:6-24
    V3 uv = cross(q, v);
    V3 uuv = cross(q, uv);
    return v + 2.0f * (w * uv + uuv);
}

__global__ __launch_bounds__(256) void w2c_project_kernel(const float* __restrict__ pose, Cam cam,
                                                          float* __restrict__ cam3d, float* __restrict__ proj2d,
                                                          long long njoints) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < njoints; i += (long long)gridDim.x * 256) {
        V3 x = mk(pose[3 * i] - cam.t[0], pose[3 * i + 1] - cam.t[1], pose[3 * i + 2] - cam.t[2]);
        V3 xc = qrot(cam.q[0], mk(-cam.q[1], -cam.q[2], -cam.q[3]), x);      // qinverse: conjugate
        if (cam3d) { cam3d[3 * i] = xc.x; cam3d[3 * i + 1] = xc.y; cam3d[3 * i + 2] = xc.z; }
        if (proj2d) {                                                        // R/common/camera.py:82-94
            float u = dhaug_clamp_pm1(xc.x / xc.z), v = dhaug_clamp_pm1(xc.y / xc.z);
            float r2 = u * u + v * v;
            float radial = 1.0f + (cam.c[4] * r2 + cam.c[5] * (r2 * r2) + cam.c[6] * (r2 * r2 * r2));
            float tan = cam.c[7] * u + cam.c[8] * v;
            proj2d[2 * i] = cam.c[0] * (u * (radial + tan) + cam.c[7] * r2) + cam.c[2];
            proj2d[2 * i + 1] = cam.c[1] * (v * (radial + tan) + cam.c[8] * r2) + cam.c[3];
        }
    }
}

__global__ __launch_bounds__(256) void w2c_project_bwd_kernel(const float* __restrict__ pose, Cam cam,
                                                              const float* __restrict__ g3, const float* __restrict__ g2,
                                                              float* __restrict__ gpose, long long njoints) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < njoints; i += (long long)gridDim.x * 256) {
        V3 x = mk(pose[3 * i] - cam.t[0], pose[3 * i + 1] - cam.t[1], pose[3 * i + 2] - cam.t[2]);
        V3 qc = mk(-cam.q[1], -cam.q[2], -cam.q[3]);
        V3 xc = qrot(cam.q[0], qc, x);
        V3 g = g3 ? mk(g3[3 * i], g3[3 * i + 1], g3[3 * i + 2]) : mk(0.f, 0.f, 0.f);
        if (g2) {
            const float gx = g2[2 * i] * cam.c[0], gy = g2[2 * i + 1] * cam.c[1];
            const float ur = xc.x / xc.z, vr = xc.y / xc.z;
            const float u = dhaug_clamp_pm1(ur), v = dhaug_clamp_pm1(vr);
            const float r2 = u * u + v * v;
            const float k1 = cam.c[4], k2 = cam.c[5], k3 = cam.c[6], p1 = cam.c[7], p2 = cam.c[8];
            const float s = 1.0f + (k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2) + (p1 * u + p2 * v);
            const float ds_dr2 = k1 + 2.0f * k2 * r2 + 3.0f * k3 * r2 * r2;
            // out_x = u*s + p1*r2 ; out_y = v*s + p2*r2
            const float go_s = gx * u + gy * v;                       // d/ds
            const float go_r2 = go_s * ds_dr2 + gx * p1 + gy * p2;    // d/dr2
            float gu = gx * s + go_s * p1 + go_r2 * 2.0f * u;
            float gv = gy * s + go_s * p2 + go_r2 * 2.0f * v;
            if (ur < -1.0f || ur > 1.0f) gu = 0.0f;                   // clamp passes gradient only inside
            if (vr < -1.0f || vr > 1.0f) gv = 0.0f;
            const float iz = 1.0f / xc.z;
            g = g + mk(gu * iz, gv * iz, -(gu * ur + gv * vr) * iz);
        }
        // x_c = R(q^-1) (x - t): adjoint = rotate the gradient by q
        V3 gw = qrot(cam.q[0], mk(cam.q[1], cam.q[2], cam.q[3]), g);
        gpose[3 * i] = gw.x; gpose[3 * i + 1] = gw.y; gpose[3 * i + 2] = gw.z;
    }
}

// per-sample quaternion / translation, R/common/camera.py:53-59
__global__ __launch_bounds__(256) void c2w_kernel(const float* __restrict__ cam3d, const float* __restrict__ quat,
                                                  const float* __restrict__ trans, float* __restrict__ world,
                                                  long long njoints) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < njoints; i += (long long)gridDim.x * 256) {
        const long long n = i >> 4;
        V3 v = mk(cam3d[3 * i], cam3d[3 * i + 1], cam3d[3 * i + 2]);
        V3 r = qrot(quat[4 * n], mk(quat[4 * n + 1], quat[4 * n + 2], quat[4 * n + 3]), v);
        world[3 * i] = r.x + trans[3 * n]; world[3 * i + 1] = r.y + trans[3 * n + 1]; world[3 * i + 2] = r.z + trans[3 * n + 2];
    }
}

// out[n][j][c] = sgn_c * (in[n][perm j][c] - center * in[n][0][c]);  ADJ: the transpose of that linear map.
constexpr int kFlipPerm[16] = {0, 4, 5, 6, 1, 2, 3, 7, 8, 9, 13, 14, 15, 10, 11, 12};
template <int C, bool ADJ>
__global__ __launch_bounds__(256) void center_flip_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int center, int flip, long long N) {
    for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < N; n += (long long)gridDim.x * 256) {
        float v[16 * C];
#pragma unroll
        for (int e = 0; e < 16 * C; ++e) v[e] = in[n * 16 * C + e];
        float o[16 * C];
        if (!ADJ) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float x = (flip ? v[kFlipPerm[j] * C + c] : v[j * C + c]) - (center ? v[c] : 0.0f);
                    o[j * C + c] = (flip && c == 0) ? -x : x;
                }
        } else {
            float sum[C];
#pragma unroll
            for (int c = 0; c < C; ++c) sum[c] = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float g = flip ? v[kFlipPerm[j] * C + c] : v[j * C + c];   // perm is an involution
                    g = (flip && c == 0) ? -g : g;
                    o[j * C + c] = g;
                    sum[c] += g;
                }
            if (center) {
#pragma unroll
                for (int c = 0; c < C; ++c) o[c] -= sum[c];
            }
        }
#pragma unroll
        for (int e = 0; e < 16 * C; ++e) out[n * 16 * C + e] = o[e];
    }
}

// PoseAug bone order (R/utils/gan_utils.py:90-120): bone k = joint[P] - joint[C]
constexpr int kPaP[15] = {0, 1, 2, 0, 4, 5, 0, 7, 8, 8, 10, 11, 8, 13, 14};
constexpr int kPaC[15] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};

// random_bl_aug (R/function_aug/dataloader_update.py:18-40): keep every bone's direction, replace its length by
// new_len[k], rebuild the pose from the root down (get_pose3dbyBoneVec), add the root back.
__global__ __launch_bounds__(TILE) void bone_swap_kernel(const float* __restrict__ pose, const float* __restrict__ new_len,
                                                         float* __restrict__ out, long long N) {
    __shared__ float lp[TILE * PS];
    __shared__ float ll[TILE * 15];
    const int lane = threadIdx.x;
    const long long ntiles = (N + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        const int rows = (int)((N - base) < TILE ? (N - base) : TILE);
        rows_in<48, PS>(pose + base * 48, lp, rows, lane);
        rows_in<15, 15>(new_len + base * 15, ll, rows, lane);
        __syncthreads();
        const int row = lane < rows ? lane : 0;
        V3 p[16], o[16];
        load_joints(lp, row, p);
        o[0] = p[0];
#pragma unroll
        for (int k = 0; k < 15; ++k) {                       // parents always precede children in this order
            const V3 b = p[kPaP[k]] - p[kPaC[k]];
            const float s = ll[row * 15 + k] / sqrtf(dot(b, b));
            o[kPaC[k]] = o[kPaP[k]] - s * b;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) { lp[lane * PS + 3 * j] = o[j].x; lp[lane * PS + 3 * j + 1] = o[j].y; lp[lane * PS + 3 * j + 2] = o[j].z; }
        __syncthreads();
        rows_out<48, PS>(lp, out + base * 48, rows, lane);
        __syncthreads();
    }
}

// project_to_2d with per-sample intrinsics (N,9) on camera-space poses (R/common/camera.py:62-94)
__global__ __launch_bounds__(256) void project_batch_kernel(const float* __restrict__ cam3d, const float* __restrict__ cam9,
                                                            float* __restrict__ proj2d, long long njoints) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < njoints; i += (long long)gridDim.x * 256) {
        const float* c = cam9 + (i >> 4) * 9;
        const float x = cam3d[3 * i], y = cam3d[3 * i + 1], z = cam3d[3 * i + 2];
        const float u = dhaug_clamp_pm1(x / z), v = dhaug_clamp_pm1(y / z);
        const float r2 = u * u + v * v;
        const float radial = 1.0f + (c[4] * r2 + c[5] * (r2 * r2) + c[6] * (r2 * r2 * r2));
        const float tan = c[7] * u + c[8] * v;
        proj2d[2 * i] = c[0] * (u * (radial + tan) + c[7] * r2) + c[2];
        proj2d[2 * i + 1] = c[1] * (v * (radial + tan) + c[8] * r2) + c[3];
    }
}

Cam make_cam(const float* q, const float* t, const float* c9) {
    Cam c;
    for (int i = 0; i < 4; ++i) c.q[i] = q[i];
    for (int i = 0; i < 3; ++i) c.t[i] = t[i];
    for (int i = 0; i < 9; ++i) c.c[i] = c9 ? c9[i] : 0.0f;
    return c;
}

int grid1d(long long items, int block) {
    long long b = (items + block - 1) / block;
    if (b > 256 * 16) b = 256 * 16;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int dhaug_bone_length(const float* pose16, float* bone_len, int64_t N, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16); DHAUG_CHECK_PTR(bone_len);
    hipLaunchKernelGGL(kcs_forward_kernel<0>, dim3(grid1d((N + TILE - 1) / TILE, 1)), dim3(TILE), 0, (hipStream_t)stream,
                       pose16, bone_len, (uint16_t*)nullptr, 0LL, 1, (long long)N);
    return dhaug_launch_status();
}

int dhaug_kcs_forward(const float* pose16, float* out_f32, uint16_t* out_bf16, int64_t ld_bf16, int64_t N,
                      int with_lengths, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16);
    DHAUG_CHECK(out_f32 != nullptr || out_bf16 != nullptr, DHAUG_EINVAL);
    if (out_bf16) {
        DHAUG_CHECK(ld_bf16 >= (with_lengths ? 30 : 15) && ld_bf16 % 8 == 0 && dhaug_aligned16(out_bf16), DHAUG_EALIGN);
    }
    hipLaunchKernelGGL(kcs_forward_kernel<1>, dim3(grid1d((N + TILE - 1) / TILE, 1)), dim3(TILE), 0, (hipStream_t)stream,
                       pose16, out_f32, out_bf16, (long long)ld_bf16, with_lengths, (long long)N);
    return dhaug_launch_status();
}

int dhaug_center_kcs_forward(const float* pose16, float* centered, uint16_t* out_bf16, int64_t ld_bf16, int64_t N,
                             int with_lengths, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16); DHAUG_CHECK_PTR(centered); DHAUG_CHECK_PTR(out_bf16);
    DHAUG_CHECK(ld_bf16 >= (with_lengths ? 30 : 15) && ld_bf16 % 8 == 0 && dhaug_aligned16(out_bf16), DHAUG_EALIGN);
    hipLaunchKernelGGL(kcs_forward_kernel<1>, dim3(grid1d((N + TILE - 1) / TILE, 1)), dim3(TILE), 0, (hipStream_t)stream,
                       pose16, (float*)nullptr, out_bf16, (long long)ld_bf16, with_lengths, (long long)N, centered);
    return dhaug_launch_status();
}

int dhaug_kcs_backward(const float* pose16, const float* grad_feat, float* grad_pose16, int64_t N, int with_lengths,
                       void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16); DHAUG_CHECK_PTR(grad_feat); DHAUG_CHECK_PTR(grad_pose16);
    hipLaunchKernelGGL(kcs_diff_kernel<false>, dim3(grid1d((N + TILE - 1) / TILE, 1)), dim3(TILE), 0, (hipStream_t)stream,
                       pose16, grad_feat, grad_pose16, with_lengths, (long long)N);
    return dhaug_launch_status();
}

int dhaug_kcs_jvp(const float* pose16, const float* tangent, float* tan_feat, int64_t N, int with_lengths, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16); DHAUG_CHECK_PTR(tangent); DHAUG_CHECK_PTR(tan_feat);
    hipLaunchKernelGGL(kcs_diff_kernel<true>, dim3(grid1d((N + TILE - 1) / TILE, 1)), dim3(TILE), 0, (hipStream_t)stream,
                       pose16, tangent, tan_feat, with_lengths, (long long)N);
    return dhaug_launch_status();
}

/* see include/dhaug.h */
int dhaug_d3_penalty(const float* pose16, const float* grad_kcs, const float* grad_pose, float coef, uint16_t* tan_kcs_bf16,
                     uint16_t* tan_pose_bf16, float* pen, int64_t N, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16); DHAUG_CHECK_PTR(grad_kcs); DHAUG_CHECK_PTR(grad_pose); DHAUG_CHECK_PTR(tan_kcs_bf16); DHAUG_CHECK_PTR(tan_pose_bf16);
    DHAUG_CHECK_PTR(pen);
    DHAUG_CHECK((reinterpret_cast<uintptr_t>(tan_kcs_bf16) & 3) == 0 && (reinterpret_cast<uintptr_t>(tan_pose_bf16) & 3) == 0, DHAUG_EALIGN);
    hipLaunchKernelGGL(d3_penalty_kernel, dim3(grid1d((N + TILE - 1) / TILE, 1)), dim3(TILE), 0, (hipStream_t)stream, pose16, grad_kcs,
                       grad_pose, coef, tan_kcs_bf16, tan_pose_bf16, pen, (long long)N);
    return dhaug_launch_status();
}

int dhaug_world_to_camera_project(const float* pose16, const float* quat, const float* trans, const float* cam9,
                                  float* cam3d, float* proj2d, int64_t N, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16); DHAUG_CHECK_PTR(quat); DHAUG_CHECK_PTR(trans);
    DHAUG_CHECK(cam3d != nullptr || proj2d != nullptr, DHAUG_EINVAL);
    DHAUG_CHECK(proj2d == nullptr || cam9 != nullptr, DHAUG_EINVAL);
    hipLaunchKernelGGL(w2c_project_kernel, dim3(grid1d(N * 16, 256)), dim3(256), 0, (hipStream_t)stream, pose16,
                       make_cam(quat, trans, cam9), cam3d, proj2d, (long long)N * 16);
    return dhaug_launch_status();
}

int dhaug_world_to_camera_project_backward(const float* pose16, const float* quat, const float* trans, const float* cam9,
                                           const float* grad_cam3d, const float* grad_proj2d, float* grad_pose16,
                                           int64_t N, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16); DHAUG_CHECK_PTR(quat); DHAUG_CHECK_PTR(trans); DHAUG_CHECK_PTR(grad_pose16);
    DHAUG_CHECK(grad_proj2d == nullptr || cam9 != nullptr, DHAUG_EINVAL);
    hipLaunchKernelGGL(w2c_project_bwd_kernel, dim3(grid1d(N * 16, 256)), dim3(256), 0, (hipStream_t)stream, pose16,
                       make_cam(quat, trans, cam9), grad_cam3d, grad_proj2d, grad_pose16, (long long)N * 16);
    return dhaug_launch_status();
}

int dhaug_bone_length_swap(const float* pose16, const float* new_len, float* out, int64_t N, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(pose16); DHAUG_CHECK_PTR(new_len); DHAUG_CHECK_PTR(out);
    hipLaunchKernelGGL(bone_swap_kernel, dim3(grid1d((N + TILE - 1) / TILE, 1)), dim3(TILE), 0, (hipStream_t)stream, pose16,
                       new_len, out, (long long)N);
    return dhaug_launch_status();
}

int dhaug_project_to_2d(const float* cam3d, const float* cam9, float* proj2d, int64_t N, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(cam3d); DHAUG_CHECK_PTR(cam9); DHAUG_CHECK_PTR(proj2d);
    hipLaunchKernelGGL(project_batch_kernel, dim3(grid1d(N * 16, 256)), dim3(256), 0, (hipStream_t)stream, cam3d, cam9, proj2d,
                       (long long)N * 16);
    return dhaug_launch_status();
}

int dhaug_camera_to_world(const float* cam3d, const float* quat, const float* trans, float* world, int64_t N,
                          void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(cam3d); DHAUG_CHECK_PTR(quat); DHAUG_CHECK_PTR(trans); DHAUG_CHECK_PTR(world);
    hipLaunchKernelGGL(c2w_kernel, dim3(grid1d(N * 16, 256)), dim3(256), 0, (hipStream_t)stream, cam3d, quat, trans, world,
                       (long long)N * 16);
    return dhaug_launch_status();
}

static int center_flip_impl(const float* in, float* out, int64_t N, int C, int center, int flip, bool adj, void* stream) {
    DHAUG_CHECK(N >= 0, DHAUG_EINVAL);
    DHAUG_CHECK(C == 2 || C == 3, DHAUG_EINVAL);
    if (N == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(in); DHAUG_CHECK_PTR(out);
    dim3 g(grid1d(N, 256)), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (C == 3 && !adj) hipLaunchKernelGGL((center_flip_kernel<3, false>), g, b, 0, s, in, out, center, flip, (long long)N);
    if (C == 3 && adj)  hipLaunchKernelGGL((center_flip_kernel<3, true>), g, b, 0, s, in, out, center, flip, (long long)N);
    if (C == 2 && !adj) hipLaunchKernelGGL((center_flip_kernel<2, false>), g, b, 0, s, in, out, center, flip, (long long)N);
    if (C == 2 && adj)  hipLaunchKernelGGL((center_flip_kernel<2, true>), g, b, 0, s, in, out, center, flip, (long long)N);
    return dhaug_launch_status();
}

int dhaug_center_flip(const float* in, float* out, int64_t N, int C, int center, int flip, void* stream) {
    return center_flip_impl(in, out, N, C, center, flip, false, stream);
}
int dhaug_center_flip_backward(const float* grad_out, float* grad_in, int64_t N, int C, int center, int flip,
                               void* stream) {
    return center_flip_impl(grad_out, grad_in, N, C, center, flip, true, stream);
}

}  // extern "C"
