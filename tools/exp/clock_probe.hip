// Experiment: the shader clock WHILE the fp32 conv kernel runs.  A one-wave probe kernel on a second stream reads
// s_memtime (shader clock ticks) and s_memrealtime (100 MHz) over ~40 ms while conv3_2-shaped launches of
// ptmi_conv3x3_fwd (through the C ABI of libptmi355.so) occupy the chip on the first stream.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/clock_probe.hip -o tools/exp/_bin/clock_probe -ldl
//   tools/exp/_bin/clock_probe probabilisticteacher_amd/libptmi355.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__global__ void probe(unsigned long long* out, unsigned long long ref_ticks)
{
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    unsigned long long r = r0;
    while (r - r0 < ref_ticks) { __builtin_amdgcn_s_sleep(32); r = wall_clock64(); }
    out[0] = __builtin_readcyclecounter() - c0;
    out[1] = r - r0;
}

typedef int (*fwd_t)(const float*, const float*, const float*, const float*, float*, int, int, int, int, int, int, void*);
typedef int (*pack_t)(const float*, float*, int, int, int, void*);
typedef int64_t (*pf_t)(int, int);

int main(int argc, char** argv)
{
    void* lib = dlopen(argc > 1 ? argv[1] : "probabilisticteacher_amd/libptmi355.so", RTLD_NOW);
    if (!lib) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    fwd_t fwd = (fwd_t)dlsym(lib, "ptmi_conv3x3_fwd");
    pack_t pack = (pack_t)dlsym(lib, "ptmi_conv3x3_pack_weights");
    pf_t pfl = (pf_t)dlsym(lib, "ptmi_conv3x3_packed_floats");
    const bool stem = argc > 2;                         // any second argument: the 3-channel stem layer instead of conv3_2
    const int n = 16, c = 256, h = stem ? 800 : 200, w = stem ? 1333 : 333;
    const int cin = stem ? 3 : c, cout = stem ? 64 : c;
    const size_t act = (size_t)n * (stem ? 64 : c) * h * w;
    float *x, *y, *wt, *wp, *b;
    hipMalloc(&x, act * 4); hipMalloc(&y, act * 4); hipMalloc(&wt, (size_t)c * c * 9 * 4); hipMalloc(&b, c * 4);
    hipMalloc(&wp, pfl(cin, cout) * 4);
    hipMemset(x, 0, act * 4); hipMemset(wt, 0, (size_t)c * c * 9 * 4); hipMemset(b, 0, c * 4);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    pack(wt, wp, cout, cin, 0, sa);
    unsigned long long* out; hipMalloc(&out, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {              // 0: idle chip, 1: conv kernel running
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            if (mode == 1) {
                hipEventRecord(e0, sa);
                for (int i = 0; i < (stem ? 30 : 6); ++i) fwd(x, wp, b, nullptr, y, n, cin, cout, h, w, 1, sa);      // ~9 ms (1.3 ms) each
                hipEventRecord(e1, sa);
            }
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, sb, out, 3000000ull);                 // 30 ms of 100 MHz ticks
            hipDeviceSynchronize();
            unsigned long long hst[2]; hipMemcpy(hst, out, 16, hipMemcpyDeviceToHost);
            float ms = 0.f; if (mode == 1) hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %llu shader ticks / %llu ref ticks -> %.0f MHz%s", mode ? "conv running" : "idle        ", hst[0], hst[1],
                   (double)hst[0] / (double)hst[1] * 100.0, mode ? "" : "\n");
            const int nl = stem ? 30 : 6;
            if (mode == 1) printf("   (%d %s launches n=16: %.2f ms each, %.1f TFLOP/s)\n", nl, stem ? "stem" : "conv3_2", ms / nl,
                                  2.0 * 9 * cin * cout * (double)h * w * n / (ms / nl) / 1e9);
        }
    }
    return 0;
}
