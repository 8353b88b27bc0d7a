// Development microbenchmark: dependent-load latency (pointer chase) at several working-set sizes, in shader clocks.
// Build: hipcc --offload-arch=gfx950 -O3 memlat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
__global__ void chase(const unsigned* p, int steps, unsigned* out, long long* cyc) {
    unsigned i = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) i = p[i];
    long long t1 = __builtin_readcyclecounter();
    out[0] = i; cyc[0] = t1 - t0;
}
int main() {
    for (size_t mb : {1, 16, 128, 1024, 4096}) {
        size_t n = mb * 1024 * 1024 / 256;                       // one entry per 256 bytes
        std::vector<unsigned> perm(n); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g);
        std::vector<unsigned> buf(n * 64, 0);
        for (size_t k = 0; k < n; ++k) buf[(size_t)perm[k] * 64] = perm[(k + 1) % n] * 64;
        unsigned *d, *o; long long* c;
        hipMalloc(&d, buf.size() * 4); hipMalloc(&o, 4); hipMalloc(&c, 8);
        hipMemcpy(d, buf.data(), buf.size() * 4, hipMemcpyHostToDevice);
        int steps = 20000;
        chase<<<1, 1>>>(d, 64, o, c); hipDeviceSynchronize();           // (code + a few entries warm; the chase below is cold)
        chase<<<1, 1>>>(d + 0, steps, o, c); hipDeviceSynchronize();
        long long cc; hipMemcpy(&cc, c, 8, hipMemcpyDeviceToHost);
        printf("working set %5zu MB: %.0f clocks per dependent load\n", mb, (double)cc / steps);
        hipFree(d); hipFree(o); hipFree(c);
    }
    return 0;
}
