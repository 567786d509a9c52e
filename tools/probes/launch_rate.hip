// Host cost and device throughput of kernel launches on this stack (round 6): how many dependent / independent launches per second
// one host thread can enqueue, with small and large argument blocks, on 1 / 2 / 4 streams, and from two host threads.
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/launch_rate.hip -o tools/probes/launch_rate -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct big_args { int v[100]; };
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(big_args a, int* p) { if (p && threadIdx.x == 9999) *p = a.v[3]; }
__global__ void k_busy(int* p, int n) { int x = threadIdx.x; for (int i = 0; i < n; ++i) x = x * 1664525 + 1013904223; if (x == 42) *p = x; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void bench(const char* name, int n, std::vector<hipStream_t>& st, F launch)
{
    for (auto s : st) hipStreamSynchronize(s);
    const double t0 = now();
    for (int i = 0; i < n; ++i) launch(i);
    const double t1 = now();
    for (auto s : st) hipStreamSynchronize(s);
    const double t2 = now();
    printf("%-58s host %.2f us/launch   total %.2f us/launch\n", name, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n);
}
int main()
{
    int* d; hipMalloc(&d, 64);
    std::vector<hipStream_t> st(4);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 20000;
    big_args a{};
    std::vector<hipStream_t> one{st[0]};
    bench("empty kernel, 1 stream", N, one, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[0], d); });
    bench("empty kernel, 1 stream (again)", N, one, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[0], d); });
    bench("400-byte arguments, 1 stream", N, one, [&](int) { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st[0], a, d); });
    bench("216 x 256 threads, 1 stream", N, one, [&](int) { hipLaunchKernelGGL(k_empty, dim3(216), dim3(256), 0, st[0], d); });
    bench("empty kernel, round robin over 2 streams", N, st, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[i & 1], d); });
    bench("empty kernel, round robin over 4 streams", N, st, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[i & 3], d); });
    bench("10 us kernel, 1 stream", N / 4, one, [&](int) { hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, st[0], d, 4000); });
    bench("10 us kernel, round robin over 4 streams", N / 4, st, [&](int i) { hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, st[i & 3], d, 4000); });
    hipEvent_t ev[4]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    bench("record + cross-stream wait + kernel (2 streams)", N / 4, st, [&](int i) {
        hipEventRecord(ev[i & 1], st[i & 1]); hipStreamWaitEvent(st[(i + 1) & 1], ev[i & 1], 0);
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[(i + 1) & 1], d); });
    {
        for (auto s : st) hipStreamSynchronize(s);
        const double t0 = now();
        auto work = [&](int q) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[q], d); };
        std::thread ta(work, 0), tb(work, 1);
        ta.join(); tb.join();
        const double t1 = now();
        for (auto s : st) hipStreamSynchronize(s);
        const double t2 = now();
        printf("%-58s host %.2f us/launch   total %.2f us/launch\n", "two host threads, one stream each", 1e6 * (t1 - t0) / (2 * N), 1e6 * (t2 - t0) / (2 * N));
    }
    // a hipGraph of 20 kernel nodes on two streams' worth of dependencies (two chains of 10), replayed
    {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st[0], hipStreamCaptureModeRelaxed);
        hipEventRecord(ev[0], st[0]); hipStreamWaitEvent(st[1], ev[0], 0);
        for (int i = 0; i < 10; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[0], d); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[1], d); }
        hipEventRecord(ev[1], st[1]); hipStreamWaitEvent(st[0], ev[1], 0);
        hipStreamEndCapture(st[0], &g);
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
            const int reps = 2000;
            hipGraphLaunch(ge, st[0]); hipStreamSynchronize(st[0]);
            const double t0 = now();
            for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st[0]);
            const double t1 = now();
            hipStreamSynchronize(st[0]);
            const double t2 = now();
            printf("%-58s host %.2f us/node     total %.2f us/node\n", "hipGraph: 2 chains x 10 empty kernels, replayed", 1e6 * (t1 - t0) / (20 * reps), 1e6 * (t2 - t0) / (20 * reps));
        } else printf("graph instantiate failed\n");
    }
    return 0;
}
