// TEST-ONLY: runs the per-pose FK arithmetic of csrc/dhaug_fk_math.h on the HOST so that the container
// without a GPU can compare it with the oracle (tests/test_hostcheck_cpu.py).  Never loaded by the product.
#include "dhaug_fk_math.h"
using namespace dhaug_fk;

extern "C" void hostcheck_fk_forward(const float* ang, const float* bl, const float* root, float* out16, long N) {
    for (long n = 0; n < N; ++n) {
        V3 p[16];
        fk_pose(ang + n * 37, bl + n * 15, p);
        for (int j = 0; j < 16; ++j) {
            out16[n * 48 + 3 * j + 0] = p[j].x + root[n * 3 + 0];
            out16[n * 48 + 3 * j + 1] = p[j].y + root[n * 3 + 1];
            out16[n * 48 + 3 * j + 2] = p[j].z + root[n * 3 + 2];
        }
    }
}

extern "C" void hostcheck_fk_backward(const float* ang, const float* bl, const float* g, float* gang, float* gbl,
                                      float* groot, long N) {
    for (long n = 0; n < N; ++n) {
        const float* mine = g + n * 48;
        V3 gr;
        fk_pose_backward(ang + n * 37, bl + n * 15,
                         [&](int j) { return mk(mine[3 * j], mine[3 * j + 1], mine[3 * j + 2]); },
                         gang + n * 37, gbl + n * 15, gr);
        groot[n * 3 + 0] = gr.x; groot[n * 3 + 1] = gr.y; groot[n * 3 + 2] = gr.z;
    }
}

extern "C" void hostcheck_tail_angles(const float* head, float* ang, float* root, long N, int preangle) {
    for (long n = 0; n < N; ++n) {
        float th[35];
        for (int c = 0; c < 35; ++c) th[c] = tanhf(head[n * 35 + c]);
        if (preangle) tail_angles<true>(th, ang + n * 37); else tail_angles<false>(th, ang + n * 37);
        for (int c = 0; c < 3; ++c) root[n * 3 + c] = th[32 + c] * 10.0f;
    }
}

extern "C" void hostcheck_sincos(const float* x, float* s, float* c, long N) {
    for (long n = 0; n < N; ++n) sincos_rad(x[n], s[n], c[n]);
}
