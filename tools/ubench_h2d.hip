// tools/ubench_h2d.hip -- what the host link gives the plugin mode: pinned host -> device copies of one 1080p I420 picture
// (3 110 400 B, the unit every EncodeOneFrame uploads) on 1 .. 16 HIP streams at once; GB/s and pictures/s.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_h2d.bin tools/ubench_h2d.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); exit(1); } } while (0)
int main()
{
    const size_t pic = 1920 * 1080 * 3 / 2;
    const int maxs = 16, reps = 200;
    std::vector<void*> h(maxs), d(maxs);
    std::vector<hipStream_t> st(maxs);
    for (int i = 0; i < maxs; i++) { CK(hipHostMalloc(&h[i], pic, hipHostMallocDefault)); CK(hipMalloc(&d[i], pic)); CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); }
    for (int pieces : {1, 4})
        for (int ns : {1, 2, 4, 8, 16}) {
            for (int w = 0; w < 2; w++) {   // second pass is the timed one
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; r++)
                    for (int i = 0; i < ns; i++)
                        for (int p = 0; p < pieces; p++) {
                            const size_t o = pic / pieces * p, len = p == pieces - 1 ? pic - o : pic / pieces;
                            CK(hipMemcpyAsync((char*)d[i] + o, (char*)h[i] + o, len, hipMemcpyHostToDevice, st[i]));
                        }
                for (int i = 0; i < ns; i++) CK(hipStreamSynchronize(st[i]));
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (w) printf("{\"streams\": %d, \"pieces_per_picture\": %d, \"GBps\": %.1f, \"pictures_per_s\": %.0f}\n", ns, pieces, (double)reps * ns * pic / dt / 1e9, reps * ns / dt);
            }
        }
    return 0;
}
