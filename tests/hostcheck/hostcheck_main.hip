// TEST-ONLY driver: the host build of the FK arithmetic (hostcheck.hip) as an executable, so that it can be built with
// -fsanitize=address,undefined on the CPU box (GPU sanitizers are not available on the pool).  Reads
//   N, then angles (N,37), bone_len (N,15), root (N,3), grad (N,48)  as raw float32 from argv[1]
// writes out16 (N,48), grad_angles (N,37), grad_bone_len (N,15), grad_root (N,3) to argv[2].
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" void hostcheck_fk_forward(const float*, const float*, const float*, float*, long);
extern "C" void hostcheck_fk_backward(const float*, const float*, const float*, float*, float*, float*, long);
extern "C" void hostcheck_sincos(const float*, float*, float*, long);

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    long N = 0;
    if (fread(&N, sizeof(long), 1, f) != 1 || N < 0 || N > (1 << 20)) return 4;
    std::vector<float> ang(N * 37), bl(N * 15), rt(N * 3), g(N * 48);
    if (fread(ang.data(), 4, ang.size(), f) != ang.size() || fread(bl.data(), 4, bl.size(), f) != bl.size() ||
        fread(rt.data(), 4, rt.size(), f) != rt.size() || fread(g.data(), 4, g.size(), f) != g.size())
        return 5;
    fclose(f);
    std::vector<float> out(N * 48), ga(N * 37), gb(N * 15), gr(N * 3), s(N * 37), c(N * 37);
    hostcheck_fk_forward(ang.data(), bl.data(), rt.data(), out.data(), N);
    hostcheck_fk_backward(ang.data(), bl.data(), g.data(), ga.data(), gb.data(), gr.data(), N);
    hostcheck_sincos(ang.data(), s.data(), c.data(), N * 37);
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 6;
    fwrite(out.data(), 4, out.size(), o);
    fwrite(ga.data(), 4, ga.size(), o);
    fwrite(gb.data(), 4, gb.size(), o);
    fwrite(gr.data(), 4, gr.size(), o);
    fclose(o);
    return 0;
}
