// Streaming helpers of the training step: operand packing (fp32 -> bf16, transposed, bf16x3 split), column
// sums (bias gradients), activation backward, fused Adam.  All HBM-bound; 16-byte accesses where the layout
// allows, grid-stride loops over >= 2048 workgroups.
#include <cstdlib>
#include "dhaug_common.h"

namespace {

int grid1d(long long items, int block) {
    long long b = (items + block - 1) / block;
    if (b > 256 * 16) b = 256 * 16;
    return (int)(b < 1 ? 1 : b);
}

// dst (rows, ld_dst) bf16 <- src (rows, cols) fp32, zero in [cols, pad_cols).  One thread per packed pair.
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, long long ld_src,
                                                       uint16_t* __restrict__ dst, long long ld_dst, long long rows,
                                                       long long cols, long long pad_cols) {
    const long long ppr = pad_cols >> 1;                       // pad_cols is even
    const long long total = rows * ppr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / ppr, c = (i - r * ppr) * 2;
        const float a = c < cols ? src[r * ld_src + c] : 0.0f;
        const float b = c + 1 < cols ? src[r * ld_src + c + 1] : 0.0f;
        *reinterpret_cast<uint32_t*>(dst + r * ld_dst + c) =
            (uint32_t)dhaug_f32_to_bf16(a) | ((uint32_t)dhaug_f32_to_bf16(b) << 16);
    }
}

// dst (cols, ld_dst) bf16 <- src (rows, cols)^T through a 32x33 LDS tile; zero in [rows, pad_cols).
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, long long ld_src,
                                                             uint16_t* __restrict__ dst, long long ld_dst,
                                                             long long rows, long long cols, long long pad_cols) {
    __shared__ float t[32][33];
    const long long tiles_r = (pad_cols + 31) / 32, tiles_c = (cols + 31) / 32;
    for (long long tile = blockIdx.x; tile < tiles_r * tiles_c; tile += gridDim.x) {
        const long long r0 = (tile / tiles_c) * 32, c0 = (tile % tiles_c) * 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int k = ty; k < 32; k += 8) {
            const long long r = r0 + k, c = c0 + tx;
            t[k][tx] = (r < rows && c < cols) ? src[r * ld_src + c] : 0.0f;
        }
        __syncthreads();
        for (int k = ty; k < 32; k += 8) {
            const long long c = c0 + k, r = r0 + tx;           // dst[c][r]
            if (c < cols && r < pad_cols) dst[c * ld_dst + r] = dhaug_f32_to_bf16(t[tx][k]);
        }
        __syncthreads();
    }
}

// Split of fp32 values into bf16 terms for fp32-grade products on the bf16 MFMA path.
//   terms 3 ("bf16x3", x = hi + lo):           activation side [hi|hi|lo],           weight side [hi|lo|hi]
//   terms 6 ("bf16x6", x = hi + mid + lo):     activation side [hi|hi|mid|mid|hi|lo], weight side [hi|mid|hi|mid|lo|hi]
// so that A' B'^T = hi*hi + hi*mid + mid*hi (+ mid*mid + hi*lo + lo*hi): every product term down to 2^-16 (2^-24)
// relative.  Each segment is pad_cols wide, zero padded.
// (eight columns per thread: one or two 16-byte loads, `terms` 16-byte stores -- the first version stored 2 bytes per lane and
// segment and ran at the store instruction rate, 66 us for a 196 608 x 256 operand; pad_cols is a multiple of 8)
// F16: the pieces are IEEE half values (x = hi + lo: 22 significant bits; terms = 3) for dhaug_gemm_f16x3 -- the fused parity
// programs' arithmetic (csrc/dhaug_mlp_x3.hip) as a layer GEMM.  |x| must stay below 65 504.
template <bool F16>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, long long ld_src,
                                                    uint16_t* __restrict__ dst, long long rows, long long cols,
                                                    long long pad_cols, int mode, int terms) {
    const long long ppr = pad_cols >> 3, total = rows * ppr;
    const bool vec = (ld_src & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / ppr, c = (i - r * ppr) * 8;
        float x[8];
        if (vec && c + 8 <= cols) {
            const float4 a = *reinterpret_cast<const float4*>(src + r * ld_src + c), b = *reinterpret_cast<const float4*>(src + r * ld_src + c + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = c + e < cols ? src[r * ld_src + c + e] : 0.0f;
        }
        uint32_t hi[4], mid[4], lo[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint16_t h, m, l;
            if (F16) {
                const _Float16 hh = (_Float16)x[e];
                const _Float16 mm = (_Float16)(x[e] - (float)hh);
                h = __builtin_bit_cast(uint16_t, hh); m = __builtin_bit_cast(uint16_t, mm); l = 0;
            } else {
                h = dhaug_f32_to_bf16(x[e]);
                const float r1 = x[e] - dhaug_bf16_to_f32(h);
                m = dhaug_f32_to_bf16(r1);
                l = dhaug_f32_to_bf16(r1 - dhaug_bf16_to_f32(m));
            }
            if (e & 1) { hi[e >> 1] |= (uint32_t)h << 16; mid[e >> 1] |= (uint32_t)m << 16; lo[e >> 1] |= (uint32_t)l << 16; }
            else { hi[e >> 1] = h; mid[e >> 1] = m; lo[e >> 1] = l; }
        }
        const uint4 H = make_uint4(hi[0], hi[1], hi[2], hi[3]), Mi = make_uint4(mid[0], mid[1], mid[2], mid[3]),
                    L = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        uint16_t* row = dst + r * (mode == 2 ? (F16 ? 2 : 3) : terms) * pad_cols + c;
        auto put = [&](int t, const uint4& v) { *reinterpret_cast<uint4*>(row + t * pad_cols) = v; };
        if (mode == 2) {                                           // the distinct pieces once: [hi | mid | lo] (dhaug_gemm_bf16x6_planes), [hi | lo] in IEEE half
            put(0, H); put(1, Mi);
            if (!F16) put(2, L);
        } else if (terms == 3) {
            put(0, H); put(1, mode == 0 ? H : Mi); put(2, mode == 0 ? Mi : H);
        } else if (mode == 0) {
            put(0, H); put(1, H); put(2, Mi); put(3, Mi); put(4, H); put(5, L);
        } else {
            put(0, H); put(1, Mi); put(2, H); put(3, Mi); put(4, L); put(5, H);
        }
    }
}

// column sums: block (x) covers 64 columns, (y) a slab of rows; 4 row-lanes per column, LDS combine, atomics out.
// Summation order: with an even row count, rows r and r + M/2 are added FIRST.  The cotangent of a WGAN critic's logits
// is -1/B on the real half of the batch and +1/B on the fake half; the reference accumulates the two halves in two
// identical passes, so its output-bias gradient is exactly 0 and Adam (lr * g / (|g| + eps)) leaves that bias alone.
// Pairing the halves keeps the zero exact for any B (with B = 72 an unpaired order left 1.2e-7, i.e. a full 1e-4 step).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ src, long long ld, float* __restrict__ dst,
                                                     long long M, long long N, long long rows_per_block, long long fold) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const long long col = (long long)blockIdx.x * 64 + cx;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;                                       // M = rows walked (half the rows when folding)
    float s = 0.0f;
    if (col < N) {
        for (long long r = r0 + ry; r < r1; r += 4) {
            if constexpr (sizeof(T) == 2) {
                float a = dhaug_bf16_to_f32((uint16_t)src[r * ld + col]);
                if (fold) a += dhaug_bf16_to_f32((uint16_t)src[(r + fold) * ld + col]);
                s += a;
            } else {
                float a = (float)src[r * ld + col];
                if (fold) a += (float)src[(r + fold) * ld + col];
                s += a;
            }
        }
    }
    part[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < N) atomicAdd(dst + col, part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
}

// the same sums for fp32 rows of whole 16-byte pieces (N % 4 == 0, ld % 4 == 0, 16-byte aligned base): a thread owns FOUR columns and
// requests four row pairs before it adds the first -- the scalar kernel above has two 4-byte loads in flight per thread and 1 024 slabs
// queueing their atomics on 256 addresses: a 196 608 x 256 cotangent at 3.3 TB/s (57 us; 49 launches = 3 ms of a parity-grade iteration;
// this one: 39 us).  Per column the additions are the scalar
// kernel's, in its order (rows r0 + ry, + 4, ...; the four row lanes combined 0 + 1 + 2 + 3): the same bits up to the order of the atomics.
__global__ __launch_bounds__(256) void colsum_f32x4_kernel(const float* __restrict__ src, long long ld, float* __restrict__ dst,
                                                           long long M, long long N, long long rows_per_block, long long fold) {
    __shared__ float4 part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const long long col = ((long long)blockIdx.x * 64 + cx) * 4;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < N) {
        const float* base = src + col;
        long long r = r0 + ry;
        for (; r + 12 < r1; r += 16) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = *reinterpret_cast<const float4*>(base + (r + 4 * u) * ld);
                b[u] = fold ? *reinterpret_cast<const float4*>(base + (r + 4 * u + fold) * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (fold) { s.x += a[u].x + b[u].x; s.y += a[u].y + b[u].y; s.z += a[u].z + b[u].z; s.w += a[u].w + b[u].w; }
                else { s.x += a[u].x; s.y += a[u].y; s.z += a[u].z; s.w += a[u].w; }
            }
        }
        for (; r < r1; r += 4) {
            float4 a = *reinterpret_cast<const float4*>(base + r * ld);
            if (fold) { const float4 b = *reinterpret_cast<const float4*>(base + (r + fold) * ld); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    part[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < N) {
        const float4 p0 = part[0][cx], p1 = part[1][cx], p2 = part[2][cx], p3 = part[3][cx];
        atomicAdd(dst + col, p0.x + p1.x + p2.x + p3.x);
        atomicAdd(dst + col + 1, p0.y + p1.y + p2.y + p3.y);
        atomicAdd(dst + col + 2, p0.z + p1.z + p2.z + p3.z);
        atomicAdd(dst + col + 3, p0.w + p1.w + p2.w + p3.w);
    }
}

__global__ __launch_bounds__(256) void act_backward_kernel(const uint16_t* __restrict__ g, long long ld_g,
                                                           const uint16_t* __restrict__ y, long long ld_y,
                                                           uint16_t* __restrict__ dst, long long ld_dst, long long M,
                                                           long long N8, int act, float slope) {
    // N8 = 16-byte chunks per row
    const long long total = M * N8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / N8, c = (i - r * N8) * 8;
        const uint4 gv = *reinterpret_cast<const uint4*>(g + r * ld_g + c);
        const uint4 yv = *reinterpret_cast<const uint4*>(y + r * ld_y + c);
        const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
        uint32_t ow[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g0 = __builtin_bit_cast(float, gw[e] << 16), g1 = __builtin_bit_cast(float, gw[e] & 0xffff0000u);
            const float y0 = __builtin_bit_cast(float, yw[e] << 16), y1 = __builtin_bit_cast(float, yw[e] & 0xffff0000u);
            const float neg = act == DHAUG_ACT_RELU ? 0.0f : (act == DHAUG_ACT_LRELU ? slope : 1.0f);
            g0 = y0 > 0.0f ? g0 : g0 * neg;
            g1 = y1 > 0.0f ? g1 : g1 * neg;
            ow[e] = (uint32_t)dhaug_f32_to_bf16(g0) | ((uint32_t)dhaug_f32_to_bf16(g1) << 16);
        }
        *reinterpret_cast<uint4*>(dst + r * ld_dst + c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

__global__ __launch_bounds__(256) void act_backward_f32_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                               float* __restrict__ dst, long long n, int act, float slope) {
    const float neg = act == DHAUG_ACT_RELU ? 0.0f : (act == DHAUG_ACT_LRELU ? slope : 1.0f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = y[i] > 0.0f ? g[i] : g[i] * neg;
}

// torch.optim.Adam (no amsgrad, no weight decay): m,v update, bias correction, p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);            // lerp form used by ATen
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

// every weight of a network -> its two bf16 operand copies, two launches per optimizer step (blockIdx.y = parameter)
__global__ __launch_bounds__(256) void repack_nt_kernel(const dhaug_repack_desc* __restrict__ descs) {
    const dhaug_repack_desc d = descs[blockIdx.y];
    const long long ppr = d.Kp >> 1, total = (long long)d.N * ppr;
    uint16_t* dst = static_cast<uint16_t*>(d.nt);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / ppr, c = (i - r * ppr) * 2;
        const float a = c < d.K ? d.W[r * d.K + c] : 0.0f;
        const float b = c + 1 < d.K ? d.W[r * d.K + c + 1] : 0.0f;
        *reinterpret_cast<uint32_t*>(dst + r * d.Kp + c) = (uint32_t)dhaug_f32_to_bf16(a) | ((uint32_t)dhaug_f32_to_bf16(b) << 16);
    }
}
__global__ __launch_bounds__(256) void repack_nn_kernel(const dhaug_repack_desc* __restrict__ descs) {
    __shared__ float t[32][33];
    const dhaug_repack_desc d = descs[blockIdx.y];
    uint16_t* dst = static_cast<uint16_t*>(d.nn);                // (K rows, Np columns) = W^T, zero beyond N
    const long long tiles_r = (d.Np + 31) / 32, tiles_c = (d.K + 31) / 32;
    for (long long tile = blockIdx.x; tile < tiles_r * tiles_c; tile += gridDim.x) {
        const long long r0 = (tile / tiles_c) * 32, c0 = (tile % tiles_c) * 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int k = ty; k < 32; k += 8) {
            const long long r = r0 + k, c = c0 + tx;
            t[k][tx] = (r < d.N && c < d.K) ? d.W[r * d.K + c] : 0.0f;
        }
        __syncthreads();
        for (int k = ty; k < 32; k += 8) {
            const long long c = c0 + k, r = r0 + tx;
            if (c < d.K && r < d.Np) dst[c * d.Np + r] = dhaug_f32_to_bf16(t[tx][k]);
        }
        __syncthreads();
    }
}

// frame differences of clips: x (rows, R * in_w) -> out (rows, (R-1) * w), out[r][f][c] = x[r][f+1][c] - x[r][f][c] for the first
// w columns of every frame (the motion critics' diff branches, R/models_Fk_GAN/Fk_discriminator.py:458-460,489-492,570-573);
// adjoint: g (rows, (R-1) * w) -> out (rows, R * in_w), out[r][f][c] = g[r][f-1][c] - g[r][f][c] (0 outside, 0 for c >= w)
__global__ __launch_bounds__(256) void frame_diff_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows,
                                                         int R, int in_w, int w, int adjoint) {
    if (!adjoint) {
        const long long per = (long long)(R - 1) * w, total = rows * per;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long r = i / per, j = i - r * per;
            const int f = (int)(j / w), c = (int)(j - (long long)f * w);
            const float* row = x + r * (long long)R * in_w;
            out[i] = row[(f + 1) * in_w + c] - row[f * in_w + c];
        }
    } else {
        const long long per = (long long)R * in_w, total = rows * per;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long r = i / per, j = i - r * per;
            const int f = (int)(j / in_w), c = (int)(j - (long long)f * in_w);
            float v = 0.0f;
            if (c < w) {
                const float* row = x + r * (long long)(R - 1) * w;
                if (f > 0) v += row[(f - 1) * w + c];
                if (f < R - 1) v -= row[f * w + c];
            }
            out[i] = v;
        }
    }
}

// same step with the step count read from device memory (a hipGraph replays the launch; the count must not be baked in)
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                       float b1, float b2, float eps, const int* __restrict__ step_dev,
                                                       float gscale) {
    const int step = *step_dev;
    const float bc1 = (float)(1.0 - pow((double)b1, (double)step)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}
__global__ void counter_add_kernel(int* p, int v) { *p += v; }

// The optimizer step of a whole network as TWO streaming launches (before: counter_add + adam_dev + repack_nt + repack_nn, the
// copies re-reading every fp32 parameter twice, in 128-byte pieces):
//   adam_nt_kernel   count, Adam, and the bf16 copy nt (N, Kp) of every weight written from the updated values while they are in
//                    registers.  nt has W's row order, so the launch walks nt's index space in 4 096-element pieces -- every
//                    access of a wave is one contiguous run (element e of nt = W[e / Kp][e % Kp], zero for e % Kp >= K);
//                    everything that is not a weight (biases) in 4 096-element pieces of the flat vector.
//   nn_from_nt_kernel  nn (K, Np) = nt^T through 64 x 64 LDS tiles: reads 2 bytes per parameter instead of 4.
// (A single launch that also transposed -- 64 x 64 tiles of W, Adam on the tile, both copies from LDS -- was written first:
// bit-identical, and 450 us against 275 for the four launches on the 25 M parameters of a DenseDim-1000 motion critic: reading
// four fp32 arrays in 256-byte row pieces ran at 1.4 TB/s.)
// The step count: adam_nt_kernel reads *state and uses *state + 1; the launch behind it (nn_from_nt_kernel, or the counter kernel
// for a network without weights) stores the new count -- stream order is the synchronisation.  (A "last workgroup out" ticket
// counter was measured first: 8 192 returning atomics on ONE address execute one after the other at the memory side, ~65 ns
// each -- the launch took 540 us for 140 us of traffic.)
__global__ __launch_bounds__(256) void adam_nt_kernel(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ Mo,
                                                      float* __restrict__ Vo, float lr, float b1, float b2, float eps,
                                                      const int* __restrict__ state, float gscale,
                                                      const dhaug_adam_desc* __restrict__ descs, int ndesc, long long nitems) {
    const int tid = threadIdx.x;
    const int step = state[0] + 1;
    __shared__ float s_bc[2];                                   // the bias corrections: double-precision pow, once per workgroup
    if (tid == 0) {
        s_bc[0] = (float)(1.0 - pow((double)b1, (double)step));
        s_bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)step));
    }
    __syncthreads();
    const float bc1 = s_bc[0], bc2_sqrt = s_bc[1];
    auto adam = [&](float pi, float graw, float mo, float vo, float& mi, float& vi) -> float {   // (adam_dev_kernel's arithmetic,
        const float gi = graw * gscale;                                                          //  expression for expression)
        mi = mo + (gi - mo) * (1.0f - b1);
        vi = vo * b2 + gi * gi * (1.0f - b2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        return pi - (lr / bc1) * (mi / denom);
    };
    for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
        int lo = 0, hi = ndesc - 1;                                           // the last descriptor with item0 <= item
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].item0 <= item) lo = mid; else hi = mid - 1;
        }
        const dhaug_adam_desc d = descs[lo];
        const long long e0 = (item - d.item0) * 4096;
        const bool mat = d.N != 0;
        const long long total = mat ? (long long)d.N * d.Kp : d.len;
        // (a pointer read from a descriptor is a generic one: stores through it are FLAT instructions, and the compiler drains
        // the memory counter around every one of them -- as a global pointer the copy's stores travel with the others)
        typedef uint32_t __attribute__((address_space(1))) * GlobalU32;
        const GlobalU32 nt32 = (GlobalU32)(uintptr_t)d.nt;
        const unsigned kp = mat ? (unsigned)d.Kp : 1u;
        // sixteen elements per thread, four at a time: their sixteen loads are requested before the first value is used
        // (unconditional, from a clamped index: a load behind a branch is waited for right there)
#pragma unroll 1
        for (int b4 = 0; b4 < 4; ++b4) {
            float pv[4], gv[4], mv[4], vv[4];
            long long idx[4];
            bool live[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long e = e0 + (b4 * 4 + q) * 256 + tid;
                long long i = e;                                              // index into the flat vectors, relative to d.off
                live[q] = e < total;
                if (mat) {
                    const unsigned r = (unsigned)(e < total ? e : 0) / kp, c = (unsigned)(e < total ? e : 0) - r * kp;
                    live[q] = live[q] && c < (unsigned)d.K;
                    i = (long long)r * d.K + c;
                }
                idx[q] = d.off + (live[q] ? i : 0);
                pv[q] = P[idx[q]]; gv[q] = G[idx[q]]; mv[q] = Mo[idx[q]]; vv[q] = Vo[idx[q]];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long e = e0 + (b4 * 4 + q) * 256 + tid;
                float a = 0.0f;
                if (live[q]) {
                    float mi, vi;
                    a = adam(pv[q], gv[q], mv[q], vv[q], mi, vi);
                    Mo[idx[q]] = mi; Vo[idx[q]] = vi; P[idx[q]] = a;
                }
                if (mat) {                                                    // (wave-uniform) even lanes store their pair: e is even there
                    const unsigned h = dhaug_f32_to_bf16(a);
                    const unsigned hn = (unsigned)__shfl_xor((int)h, 1);
                    if (!(tid & 1) && e < total) nt32[e >> 1] = h | (hn << 16);
                }
            }
        }
    }
}

// nn (K, Np) = nt^T (nt: (N, Kp) bf16), zero for columns >= N; blockIdx.y = weight.  One thread advances the step count.
__global__ __launch_bounds__(256) void nn_from_nt_kernel(const dhaug_repack_desc* __restrict__ descs, int* __restrict__ state) {
    __shared__ uint16_t t[64][66];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *state += 1;
    const dhaug_repack_desc d = descs[blockIdx.y];
    typedef uint32_t __attribute__((address_space(1))) * GlobalU32;          // (global, not generic: see adam_nt_kernel)
    const GlobalU32 nt32 = (GlobalU32)(uintptr_t)d.nt, nn32 = (GlobalU32)(uintptr_t)d.nn;
    const int tid = threadIdx.x;
    const int tiles_r = (d.Np + 63) >> 6, tiles_c = (d.K + 63) >> 6;
    for (int tile = blockIdx.x; tile < tiles_r * tiles_c; tile += gridDim.x) {
        const int r0 = (tile / tiles_c) * 64, c0 = (tile % tiles_c) * 64;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {                                // rows of nt, two columns per lane
            const int rr = pass * 8 + (tid >> 5), c2 = 2 * (tid & 31);
            const int r = r0 + rr, c = c0 + c2;
            const uint32_t v = (r < d.N && c < d.Kp) ? nt32[((long long)r * d.Kp + c) >> 1] : 0u;
            *reinterpret_cast<uint32_t*>(&t[rr][c2]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {                                // rows of nn = columns of nt, two per lane
            const int cc = pass * 8 + (tid >> 5), rr = 2 * (tid & 31);
            const int c = c0 + cc, r = r0 + rr;
            if (c < d.K && r < d.Np)
                nn32[((long long)c * d.Np + r) >> 1] = (uint32_t)t[rr][cc] | ((uint32_t)t[rr + 1][cc] << 16);
        }
        __syncthreads();
    }
}

// rows [0,B) = real, [B,2B) = fake, [2B,3B) = alpha*real + (1-alpha)*fake: the batch one critic step scores
// (R/models_Fk_GAN/model_fk_gan_train.py:186-198, R/models_Fk_GAN/Fk_discriminator.py:210-216)
// outb (optional): the real / fake rows once more as bf16 (2B, ldb) -- the operand sweep 4 contracts the input layer's cotangent with;
// as a cast launch of its own it re-read what this kernel has in registers
__global__ __launch_bounds__(256) void gp_assemble_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                          const float* __restrict__ alpha, float* __restrict__ out,
                                                          long long B, long long W, uint16_t* __restrict__ outb, long long ldb) {
    const long long total = B * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / W;
        const float r = real[i], f = fake[i], a = alpha[b];
        out[i] = r;
        out[total + i] = f;
        out[2 * total + i] = a * r + ((1.0f - a) * f);
        if (outb != nullptr) {
            const long long c = i - b * W;
            outb[b * ldb + c] = dhaug_f32_to_bf16(r);
            outb[(B + b) * ldb + c] = dhaug_f32_to_bf16(f);
        }
    }
}

// one wave per row: n = ||g_b||_2, pen[b] = (n - 1)^2, v_b = coef * (n - 1) / n * g_b  (= d/dg of coef/2 * (n-1)^2)
__global__ __launch_bounds__(256) void gp_penalty_kernel(const float* __restrict__ g, float* __restrict__ v,
                                                         float* __restrict__ pen, long long B, int W, float coef,
                                                         uint16_t* __restrict__ vb, long long ldvb) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    for (long long b = wave; b < B; b += nw) {
        const float* row = g + b * W;
        float ss = 0.0f;
        for (int c = lane; c < W; c += 64) ss = fmaf(row[c], row[c], ss);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float n = sqrtf(ss), d = n - 1.0f;
        if (lane == 0) pen[b] = d * d;
        // n == 0 (every unit of the critic dead on this row): torch's norm has subgradient 0 there (the reference's
        // gradients.norm(2, dim=1) back-propagates 0, not -inf * 0 = NaN)
        const float k = n > 0.0f ? coef * d / n : 0.0f;
        for (int c = lane; c < W; c += 64) {
            const float vv = k * row[c];
            v[b * W + c] = vv;
            if (vb != nullptr) vb[b * ldvb + c] = dhaug_f32_to_bf16(vv);   // (the tangent sweep's first operand: see gp_assemble_kernel)
        }
    }
}

// out[0..4] = D_real, D_fake, GP, Wasserstein_D = D_real - D_fake, D_cost = D_fake - D_real + GP
// logits (3B) fp32: rows [0,B) real, [B,2B) fake; pen (B).  One workgroup.
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                      long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = a[i] + b[i];
}

// stage 1: partial sums of the real logits, the fake logits and the penalties: block b -> part[3 b + k]; 64 blocks keep enough
// requests in flight (one workgroup took 24 us for 0.8 MB)
constexpr int CS_BLOCKS = 64;
__global__ __launch_bounds__(256) void critic_scalars_partial_kernel(const float* __restrict__ logits, long long ld,
                                                                      const float* __restrict__ pen, long long B, long long P,
                                                                      float* __restrict__ part) {
    __shared__ float red[3][4];
    float s[3] = {0.f, 0.f, 0.f};
    const long long stride = (long long)CS_BLOCKS * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < B; i += stride) {
        s[0] += logits[i * ld];
        s[1] += logits[(B + i) * ld];
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += stride) s[2] += pen[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) part[3 * blockIdx.x + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// stage 2: out[0..4] = D_real, D_fake, GP, Wasserstein_D = D_real - D_fake, D_cost = D_fake - D_real + GP (fixed summation order)
__global__ __launch_bounds__(64) void critic_scalars_final_kernel(const float* __restrict__ part, long long B, long long P, float lambda,
                                                                   float* __restrict__ out) {
    float s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s[k] = threadIdx.x < CS_BLOCKS ? part[3 * threadIdx.x + k] : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
    }
    if (threadIdx.x == 0) {
        const float dr = s[0] / (float)B, df = s[1] / (float)B, gp = lambda * (s[2] / (float)P);
        out[0] = dr; out[1] = df; out[2] = gp; out[3] = dr - df; out[4] = df - dr + gp;
    }
}

// (2) cotangent of a 1-wide linear layer's input behind an activation: out[r][c] = bf16(seed[r] * w[c]) * act'(mask[r][c]),
// columns [N, pad) zero -- the rank-one first step of a critic's backward chain (the logit layer), which as a GEMM with
// K = 1 spent 43 us on 44 MB
constexpr int DHAUG_RANK1_MAX_N = 1024;
__global__ __launch_bounds__(256) void rank1_mask_kernel(const uint16_t* __restrict__ seed, long long ld_seed,
                                                         const uint16_t* __restrict__ w, long long ld_w,
                                                         const uint16_t* __restrict__ mask, long long ld_mask,
                                                         uint16_t* __restrict__ out, long long ld_out, long long M, int N, int pad,
                                                         float dneg, const uint32_t* __restrict__ bits) {
    // the weight row once per workgroup (it is a strided column of the packed matrix: read per element in the loop, every
    // thread waited for eight 2-byte gathers -- 54 us for the 88 MB of the 3D critic's logit layer, 1.6 TB/s)
    __shared__ float sw[DHAUG_RANK1_MAX_N];
    for (int c = threadIdx.x; c < pad; c += 256) sw[c] = c < N ? dhaug_bf16_to_f32(w[(long long)c * ld_w]) : 0.0f;
    __syncthreads();
    const int cpr = pad >> 3;                                 // 16-byte chunks per row
    const long long total = M * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cpr;
        const int c0 = (int)(i - r * cpr) * 8;
        const float sv = dhaug_bf16_to_f32(seed[r * ld_seed]);
        uint32_t mw[4];                                        // per element pair: (y > 0) of the even / odd element in bits 0 / 16
        if (bits != nullptr) {
            // the mask as sign bits (dhaug_mlp_unit.bits, a 256-wide layer): features c0 .. c0+3 sit in the word of the row's
            // lane h = 0, c0+4 .. c0+7 in the word of lane h = 1, pairs 8t + 2g (+1) at bit p / 16 + p
            const long long wi = ((r >> 5) * 4 + ((c0 >> 5) & 3)) * 64 + (r & 31);
            const uint32_t w0 = bits[wi], w1 = bits[wi + 32];
            const int p0 = 8 * (c0 >> 7) + 2 * ((c0 & 31) >> 3);
            mw[0] = ((w0 >> p0) & 1u) | (((w0 >> (16 + p0)) & 1u) << 16);
            mw[1] = ((w0 >> (p0 + 1)) & 1u) | (((w0 >> (17 + p0)) & 1u) << 16);
            mw[2] = ((w1 >> p0) & 1u) | (((w1 >> (16 + p0)) & 1u) << 16);
            mw[3] = ((w1 >> (p0 + 1)) & 1u) | (((w1 >> (17 + p0)) & 1u) << 16);
        } else {
            const uint4 mk = *reinterpret_cast<const uint4*>(mask + r * ld_mask + c0);
            mw[0] = mk.x; mw[1] = mk.y; mw[2] = mk.z; mw[3] = mk.w;
        }
        uint32_t ow[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = c0 + 2 * e + q;
                const short y = (short)(q ? (mw[e] >> 16) : (mw[e] & 0xffffu));
                const float g = dhaug_bf16_to_f32(dhaug_f32_to_bf16(sv * sw[c]));              // (0 beyond N)
                v[q] = y > 0 ? g : g * dneg;
            }
            ow[e] = (uint32_t)dhaug_f32_to_bf16(v[0]) | ((uint32_t)dhaug_f32_to_bf16(v[1]) << 16);
        }
        *reinterpret_cast<uint4*>(out + r * ld_out + c0) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// out[r][f][:] = x[r][R-1-f][:] for clips (rows, R*w): its own inverse and its own transpose
__global__ __launch_bounds__(256) void frame_reverse_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows,
                                                            int R, int w) {
    const long long total = rows * (long long)R * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / ((long long)R * w);
        const int rem = (int)(i - row * R * w), f = rem / w, c = rem - f * w;
        out[i] = x[row * R * w + (long long)(R - 1 - f) * w + c];
    }
}

// out[0] = sum_i weight_i * mean(x_i[0 .. n_i)): the generator loss from the critics' logits (one workgroup; the arrays are
// read with several requests in flight per thread)
struct WMeans {
    int n;
    const float* x[DHAUG_WEIGHTED_MEANS_MAX];
    long long count[DHAUG_WEIGHTED_MEANS_MAX];
    float weight[DHAUG_WEIGHTED_MEANS_MAX];
};
__global__ __launch_bounds__(1024) void weighted_means_kernel(WMeans a, float* __restrict__ out) {
    __shared__ float red[16];
    float total = 0.f;
    for (int i = 0; i < a.n; ++i) {
        const float* x = a.x[i];
        const long long n = a.count[i];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        long long j = threadIdx.x;
        for (; j + 3 * 1024 < n; j += 4 * 1024) {
            const float v0 = x[j], v1 = x[j + 1024], v2 = x[j + 2048], v3 = x[j + 3072];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; j < n; j += 1024) s0 += x[j];
        float s = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < 16; ++w) t += red[w];
            total += a.weight[i] * (t / (float)n);
        }
    }
    if (threadIdx.x == 0) out[0] = total;
}

}  // namespace

template <typename T>
static int colsum_impl(const T* src, int64_t ld, float* dst, int64_t M, int64_t N, int accumulate, void* stream) {
    DHAUG_CHECK(M >= 0 && N >= 1 && ld >= N, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(dst);
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dst, 0, (size_t)N * 4, s);
        if (e != hipSuccess) return (int)e;
    }
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(src);
    const long long fold = (M % 2 == 0) ? M / 2 : 0, Mw = fold ? fold : M;           // rows walked
    if constexpr (sizeof(T) == 4) {
        if (N % 4 == 0 && ld % 4 == 0 && dhaug_aligned16(src) && Mw >= 4096) {
            const long long cb = (N / 4 + 63) / 64;
            // (every slab ends in one atomic per column, and those queue up on the column's address: 196 608 x 256 with 1 024 slabs 114 us,
            // 512 72, 256 48, 128 39 = 5.2 TB/s, 96 41, 64 53, 32 92 -- eight 16-byte loads in flight per thread keep few blocks fed)
            static const long long blocks = getenv("DHAUG_COLSUM_BLOCKS") ? atoll(getenv("DHAUG_COLSUM_BLOCKS")) : 128;
            long long sl = blocks / cb;
            if (sl < 1) sl = 1;
            long long rows = (Mw + sl - 1) / sl;
            if (rows < 64) rows = 64;
            sl = (Mw + rows - 1) / rows;
            hipLaunchKernelGGL(colsum_f32x4_kernel, dim3((unsigned)cb, (unsigned)sl), dim3(256), 0, s, reinterpret_cast<const float*>(src),
                               (long long)ld, dst, Mw, (long long)N, rows, fold);
            return dhaug_launch_status();
        }
    }
    const long long col_blocks = (N + 63) / 64;
    long long slabs = 1024 / col_blocks;
    if (slabs < 1) slabs = 1;
    long long rpb = (Mw + slabs - 1) / slabs;
    if (rpb < 64) rpb = 64;
    slabs = (Mw + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_kernel<T>, dim3((unsigned)col_blocks, (unsigned)slabs), dim3(256), 0, s, src, (long long)ld,
                       dst, Mw, (long long)N, rpb, fold);
    return dhaug_launch_status();
}

extern "C" {

int dhaug_version(void) { return DHAUG_VERSION; }
const char* dhaug_arch(void) { return "gfx950"; }

int dhaug_cast_pad_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                        int64_t pad_cols, void* stream) {
    DHAUG_CHECK(rows >= 0 && cols >= 1 && pad_cols >= cols && ld_src >= cols && ld_dst >= pad_cols, DHAUG_EINVAL);
    if (rows == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(src); DHAUG_CHECK_PTR(dst);
    DHAUG_CHECK((pad_cols & 1) == 0 && (ld_dst & 1) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0, DHAUG_EALIGN);
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid1d(rows * (pad_cols / 2), 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)ld_src, dst, (long long)ld_dst, (long long)rows, (long long)cols, (long long)pad_cols);
    return dhaug_launch_status();
}

int dhaug_cast_transpose_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows,
                              int64_t cols, int64_t pad_cols, void* stream) {
    DHAUG_CHECK(rows >= 1 && cols >= 1 && pad_cols >= rows && ld_src >= cols && ld_dst >= pad_cols, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(src); DHAUG_CHECK_PTR(dst);
    const long long tiles = ((pad_cols + 31) / 32) * ((cols + 31) / 32);
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(grid1d(tiles, 1)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)ld_src, dst, (long long)ld_dst, (long long)rows, (long long)cols, (long long)pad_cols);
    return dhaug_launch_status();
}

int dhaug_split_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t rows, int64_t cols, int64_t pad_cols,
                     int mode, int terms, void* stream) {
    DHAUG_CHECK(rows >= 0 && cols >= 1 && pad_cols >= cols && ld_src >= cols && (mode == 0 || mode == 1 || (mode == 2 && terms == 6)), DHAUG_EINVAL);
    DHAUG_CHECK(terms == 3 || terms == 6, DHAUG_EINVAL);
    if (rows == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(src); DHAUG_CHECK_PTR(dst);
    DHAUG_CHECK(pad_cols % 8 == 0 && dhaug_aligned16(dst), DHAUG_EALIGN);
    hipLaunchKernelGGL(split_kernel<false>, dim3(grid1d(rows * (pad_cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)ld_src, dst, (long long)rows, (long long)cols, (long long)pad_cols, mode, terms);
    return dhaug_launch_status();
}

/* see include/dhaug.h */
int dhaug_split_f16(const float* src, int64_t ld_src, uint16_t* dst, int64_t rows, int64_t cols, int64_t pad_cols, int mode,
                    void* stream) {
    DHAUG_CHECK(rows >= 0 && cols >= 1 && pad_cols >= cols && ld_src >= cols && (mode == 0 || mode == 1 || mode == 2), DHAUG_EINVAL);
    if (rows == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(src); DHAUG_CHECK_PTR(dst);
    DHAUG_CHECK(pad_cols % 8 == 0 && dhaug_aligned16(dst), DHAUG_EALIGN);
    hipLaunchKernelGGL(split_kernel<true>, dim3(grid1d(rows * (pad_cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)ld_src, dst, (long long)rows, (long long)cols, (long long)pad_cols, mode, 3);
    return dhaug_launch_status();
}

int dhaug_colsum_f32(const float* src, int64_t ld, float* dst, int64_t M, int64_t N, int accumulate, void* stream) {
    return colsum_impl<float>(src, ld, dst, M, N, accumulate, stream);
}
int dhaug_colsum_bf16(const uint16_t* src, int64_t ld, float* dst, int64_t M, int64_t N, int accumulate, void* stream) {
    return colsum_impl<uint16_t>(src, ld, dst, M, N, accumulate, stream);
}

int dhaug_act_backward_bf16(const uint16_t* g, int64_t ld_g, const uint16_t* y, int64_t ld_y, uint16_t* dst,
                            int64_t ld_dst, int64_t M, int64_t N, int act, float slope, void* stream) {
    DHAUG_CHECK(M >= 0 && N >= 8, DHAUG_EINVAL);
    DHAUG_CHECK(act >= DHAUG_ACT_NONE && act <= DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(g); DHAUG_CHECK_PTR(y); DHAUG_CHECK_PTR(dst);
    DHAUG_CHECK(N % 8 == 0 && ld_g % 8 == 0 && ld_y % 8 == 0 && ld_dst % 8 == 0, DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(g) && dhaug_aligned16(y) && dhaug_aligned16(dst), DHAUG_EALIGN);
    hipLaunchKernelGGL(act_backward_kernel, dim3(grid1d(M * (N / 8), 256)), dim3(256), 0, (hipStream_t)stream, g,
                       (long long)ld_g, y, (long long)ld_y, dst, (long long)ld_dst, (long long)M, (long long)(N / 8), act,
                       slope);
    return dhaug_launch_status();
}

int dhaug_act_backward_f32(const float* g, const float* y, float* dst, int64_t n, int act, float slope, void* stream) {
    DHAUG_CHECK(n >= 0 && act >= DHAUG_ACT_NONE && act <= DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (n == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(g); DHAUG_CHECK_PTR(y); DHAUG_CHECK_PTR(dst);
    hipLaunchKernelGGL(act_backward_f32_kernel, dim3(grid1d(n, 256)), dim3(256), 0, (hipStream_t)stream, g, y, dst,
                       (long long)n, act, slope);
    return dhaug_launch_status();
}

int dhaug_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, int step, float grad_scale, void* stream) {
    DHAUG_CHECK(n >= 0 && step >= 1, DHAUG_EINVAL);
    if (n == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(param); DHAUG_CHECK_PTR(grad); DHAUG_CHECK_PTR(exp_avg); DHAUG_CHECK_PTR(exp_avg_sq);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (long long)n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
    return dhaug_launch_status();
}

int dhaug_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                        float beta2, float eps, const int* step_dev, float grad_scale, void* stream) {
    DHAUG_CHECK(n >= 0, DHAUG_EINVAL);
    if (n == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(param); DHAUG_CHECK_PTR(grad); DHAUG_CHECK_PTR(exp_avg); DHAUG_CHECK_PTR(exp_avg_sq); DHAUG_CHECK_PTR(step_dev);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid1d(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (long long)n, lr, beta1, beta2, eps, step_dev, grad_scale);
    return dhaug_launch_status();
}

int dhaug_adam_repack_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                           float eps, int* state, float grad_scale, const dhaug_adam_desc* descs_device, int ndesc, int64_t nitems,
                           const dhaug_repack_desc* weights_device, int nweights, void* stream) {
    DHAUG_CHECK(ndesc >= 0 && nitems >= 0 && nweights >= 0 && nweights <= 65535, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(state);
    if (ndesc == 0 || nitems == 0) return dhaug_counter_add(state, 1, stream);      // nothing to update: the count still advances
    DHAUG_CHECK_PTR(param); DHAUG_CHECK_PTR(grad); DHAUG_CHECK_PTR(exp_avg); DHAUG_CHECK_PTR(exp_avg_sq); DHAUG_CHECK_PTR(descs_device);
    const unsigned grid = (unsigned)(nitems < 8192 ? nitems : 8192);
    hipLaunchKernelGGL(adam_nt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, lr, beta1,
                       beta2, eps, state, grad_scale, descs_device, ndesc, (long long)nitems);
    if (nweights > 0) {
        DHAUG_CHECK_PTR(weights_device);
        hipLaunchKernelGGL(nn_from_nt_kernel, dim3(64, (unsigned)nweights), dim3(256), 0, (hipStream_t)stream, weights_device, state);
        return dhaug_launch_status();
    }
    return dhaug_counter_add(state, 1, stream);
}

int dhaug_repack_weights(const dhaug_repack_desc* descs_device, int nparams, void* stream) {
    DHAUG_CHECK(nparams >= 0 && nparams <= 65535, DHAUG_EINVAL);
    if (nparams == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(descs_device);
    hipLaunchKernelGGL(repack_nt_kernel, dim3(32, (unsigned)nparams), dim3(256), 0, (hipStream_t)stream, descs_device);
    hipLaunchKernelGGL(repack_nn_kernel, dim3(32, (unsigned)nparams), dim3(256), 0, (hipStream_t)stream, descs_device);
    return dhaug_launch_status();
}

int dhaug_frame_diff(const float* x, float* out, int64_t rows, int R, int in_w, int w, int adjoint, void* stream) {
    DHAUG_CHECK(rows >= 0 && R >= 2 && w >= 1 && in_w >= w, DHAUG_EINVAL);
    if (rows == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(x); DHAUG_CHECK_PTR(out);
    const long long total = adjoint ? rows * (long long)R * in_w : rows * (long long)(R - 1) * w;
    hipLaunchKernelGGL(frame_diff_kernel, dim3(grid1d(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)rows, R,
                       in_w, w, adjoint);
    return dhaug_launch_status();
}

int dhaug_frame_reverse(const float* x, float* out, int64_t rows, int R, int w, void* stream) {
    DHAUG_CHECK(rows >= 0 && R >= 1 && w >= 1, DHAUG_EINVAL);
    if (rows == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(x); DHAUG_CHECK_PTR(out);
    DHAUG_CHECK(x != out, DHAUG_EINVAL);
    hipLaunchKernelGGL(frame_reverse_kernel, dim3(grid1d(rows * (long long)R * w, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                       (long long)rows, R, w);
    return dhaug_launch_status();
}

int dhaug_weighted_means(const float* const* arrays, const int64_t* counts, const float* weights, int n, float* out, void* stream) {
    DHAUG_CHECK(n >= 1 && n <= DHAUG_WEIGHTED_MEANS_MAX, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(arrays); DHAUG_CHECK_PTR(counts); DHAUG_CHECK_PTR(weights); DHAUG_CHECK_PTR(out);
    WMeans a;
    a.n = n;
    for (int i = 0; i < n; ++i) {
        DHAUG_CHECK(arrays[i] != nullptr && counts[i] >= 1, DHAUG_EINVAL);
        a.x[i] = arrays[i]; a.count[i] = counts[i]; a.weight[i] = weights[i];
    }
    hipLaunchKernelGGL(weighted_means_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, out);
    return dhaug_launch_status();
}

int dhaug_counter_add(int* counter, int value, void* stream) {
    DHAUG_CHECK_PTR(counter);
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, value);
    return dhaug_launch_status();
}

int dhaug_gp_assemble(const float* real, const float* fake, const float* alpha, float* out, int64_t B, int64_t W, void* stream) {
    DHAUG_CHECK(B >= 0 && W >= 1, DHAUG_EINVAL);
    if (B == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(real); DHAUG_CHECK_PTR(fake); DHAUG_CHECK_PTR(alpha); DHAUG_CHECK_PTR(out);
    hipLaunchKernelGGL(gp_assemble_kernel, dim3(grid1d(B * W, 256)), dim3(256), 0, (hipStream_t)stream, real, fake, alpha, out,
                       (long long)B, (long long)W, static_cast<uint16_t*>(nullptr), 0LL);
    return dhaug_launch_status();
}

int dhaug_gp_assemble_bf16(const float* real, const float* fake, const float* alpha, float* out, uint16_t* rows_bf16, int64_t ld_bf16,
                           int64_t B, int64_t W, void* stream) {
    DHAUG_CHECK(B >= 0 && W >= 1 && ld_bf16 >= W, DHAUG_EINVAL);
    if (B == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(real); DHAUG_CHECK_PTR(fake); DHAUG_CHECK_PTR(alpha); DHAUG_CHECK_PTR(out); DHAUG_CHECK_PTR(rows_bf16);
    hipLaunchKernelGGL(gp_assemble_kernel, dim3(grid1d(B * W, 256)), dim3(256), 0, (hipStream_t)stream, real, fake, alpha, out,
                       (long long)B, (long long)W, rows_bf16, (long long)ld_bf16);
    return dhaug_launch_status();
}

int dhaug_gp_penalty(const float* grad, float* v, float* pen, int64_t B, int64_t W, float coef, void* stream) {
    DHAUG_CHECK(B >= 0 && W >= 1 && W <= (1 << 20), DHAUG_EINVAL);
    if (B == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(grad); DHAUG_CHECK_PTR(v); DHAUG_CHECK_PTR(pen);
    hipLaunchKernelGGL(gp_penalty_kernel, dim3(grid1d((B + 3) / 4, 1)), dim3(256), 0, (hipStream_t)stream, grad, v, pen,
                       (long long)B, (int)W, coef, static_cast<uint16_t*>(nullptr), 0LL);
    return dhaug_launch_status();
}

int dhaug_gp_penalty_bf16(const float* grad, float* v, uint16_t* v_bf16, int64_t ld_bf16, float* pen, int64_t B, int64_t W, float coef,
                          void* stream) {
    DHAUG_CHECK(B >= 0 && W >= 1 && W <= (1 << 20) && ld_bf16 >= W, DHAUG_EINVAL);
    if (B == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(grad); DHAUG_CHECK_PTR(v); DHAUG_CHECK_PTR(v_bf16); DHAUG_CHECK_PTR(pen);
    hipLaunchKernelGGL(gp_penalty_kernel, dim3(grid1d((B + 3) / 4, 1)), dim3(256), 0, (hipStream_t)stream, grad, v, pen,
                       (long long)B, (int)W, coef, v_bf16, (long long)ld_bf16);
    return dhaug_launch_status();
}

int dhaug_critic_scalars(const float* logits, int64_t ld, const float* pen, int64_t B, int64_t P, float lambda, float* out5,
                         float* scratch, void* stream) {
    DHAUG_CHECK(B >= 1 && P >= 1 && ld >= 1, DHAUG_EINVAL);
    DHAUG_CHECK_PTR(logits); DHAUG_CHECK_PTR(pen); DHAUG_CHECK_PTR(out5); DHAUG_CHECK_PTR(scratch);
    hipLaunchKernelGGL(critic_scalars_partial_kernel, dim3(CS_BLOCKS), dim3(256), 0, (hipStream_t)stream, logits, (long long)ld, pen,
                       (long long)B, (long long)P, scratch);
    hipLaunchKernelGGL(critic_scalars_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, (long long)B, (long long)P,
                       lambda, out5);
    return dhaug_launch_status();
}

int dhaug_rank1_mask_bf16(const uint16_t* seed, int64_t ld_seed, const uint16_t* w, int64_t ld_w, const uint16_t* mask, int64_t ld_mask,
                          uint16_t* out, int64_t ld_out, int64_t M, int64_t N, int64_t pad_cols, int mask_act, float mask_slope,
                          void* stream) {
    DHAUG_CHECK(M >= 0 && N >= 1 && pad_cols >= N && pad_cols % 8 == 0, DHAUG_EINVAL);
    DHAUG_CHECK(pad_cols <= DHAUG_RANK1_MAX_N, DHAUG_EUNSUPPORTED);               /* (the weight row is staged in LDS) */
    DHAUG_CHECK(mask_act == DHAUG_ACT_RELU || mask_act == DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(seed); DHAUG_CHECK_PTR(w); DHAUG_CHECK_PTR(mask); DHAUG_CHECK_PTR(out);
    DHAUG_CHECK(ld_seed >= 1 && ld_w >= 1 && ld_mask >= pad_cols && ld_out >= pad_cols && ld_mask % 8 == 0 && ld_out % 8 == 0, DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(mask) && dhaug_aligned16(out), DHAUG_EALIGN);
    hipLaunchKernelGGL(rank1_mask_kernel, dim3(grid1d(M * (pad_cols >> 3), 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       (long long)ld_seed, w, (long long)ld_w, mask, (long long)ld_mask, out, (long long)ld_out, (long long)M, (int)N,
                       (int)pad_cols, mask_act == DHAUG_ACT_RELU ? 0.0f : mask_slope, (const uint32_t*)nullptr);
    return dhaug_launch_status();
}

/* see include/dhaug.h */
int dhaug_rank1_bits_bf16(const uint16_t* seed, int64_t ld_seed, const uint16_t* w, int64_t ld_w, const uint32_t* bits,
                          uint16_t* out, int64_t ld_out, int64_t M, int mask_act, float mask_slope, void* stream) {
    DHAUG_CHECK(M >= 0, DHAUG_EINVAL);
    DHAUG_CHECK(mask_act == DHAUG_ACT_RELU || mask_act == DHAUG_ACT_LRELU, DHAUG_EINVAL);
    if (M == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(seed); DHAUG_CHECK_PTR(w); DHAUG_CHECK_PTR(bits); DHAUG_CHECK_PTR(out);
    DHAUG_CHECK(ld_seed >= 1 && ld_w >= 1 && ld_out >= 256 && ld_out % 8 == 0, DHAUG_EALIGN);
    DHAUG_CHECK(dhaug_aligned16(out) && dhaug_aligned16(bits), DHAUG_EALIGN);
    hipLaunchKernelGGL(rank1_mask_kernel, dim3(grid1d(M * 32, 256)), dim3(256), 0, (hipStream_t)stream, seed, (long long)ld_seed, w,
                       (long long)ld_w, (const uint16_t*)nullptr, 0LL, out, (long long)ld_out, (long long)M, 256, 256,
                       mask_act == DHAUG_ACT_RELU ? 0.0f : mask_slope, bits);
    return dhaug_launch_status();
}

int dhaug_workgroup_cap_ = 0;

/* see include/dhaug.h */
int dhaug_set_workgroup_cap(int n) {
    DHAUG_CHECK(n >= 0 && n <= 256, DHAUG_EINVAL);
    const int old = dhaug_workgroup_cap_;
    dhaug_workgroup_cap_ = n;
    return old;
}

int dhaug_nan_propagation_ = 0;

/* see include/dhaug.h */
int dhaug_set_nan_propagation(int on) {
    const int old = dhaug_nan_propagation_;
    dhaug_nan_propagation_ = on != 0;
    return old;
}

int dhaug_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    DHAUG_CHECK(n >= 0, DHAUG_EINVAL);
    if (n == 0) return DHAUG_OK;
    DHAUG_CHECK_PTR(a); DHAUG_CHECK_PTR(b); DHAUG_CHECK_PTR(out);
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid1d(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)n);
    return dhaug_launch_status();
}

}  // extern "C"
