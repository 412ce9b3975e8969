// What does a DEPENDENT kernel launch cost on this box, and which ingredient of the engine's launches sets it?
// Stand-alone microbenchmark (no engine, no torch): chains of N launches on one stream, wall time per launch between two
// stream synchronisations, for a grid of launch shapes:
//   threads per workgroup, dynamic LDS, kernarg size, barrier bit (hipExtAnyOrderLaunch), eager vs hipGraph replay,
//   an empty body vs a body that writes / reads one 8 KB row (what a decode projection hands to the next one).
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/launch_floor tools/launch_floor.hip ; run on the MI355X.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Small { float* p; int n; };
struct Big { float* p; int n; int pad[58]; };     // 240-byte kernarg, like GemmParams

template <class A>
__global__ void k_empty(const A a) {
    extern __shared__ unsigned char smem[];
    if (a.n < 0) a.p[threadIdx.x] = smem[threadIdx.x];       // never taken: keeps the arguments and LDS alive
}

// every workgroup reads the row the previous launch wrote and writes its own 32 B of the next one (a dependent chain with data)
template <class A>
__global__ void k_row(const A a) {
    extern __shared__ unsigned char smem[];
    const float* src = a.p + (a.n & 1) * 4096;
    float* dst = a.p + ((a.n + 1) & 1) * 4096;
    float v = src[threadIdx.x & 2047] + src[2048 + (threadIdx.x & 2047)];
    if (threadIdx.x < 8) dst[(blockIdx.x * 8 + threadIdx.x) & 4095] = v + 1.0f;
}

template <class K, class A>
static double chain(K kern, A arg, int grid, int threads, size_t lds, int n, unsigned flags, hipStream_t st, bool graph) {
    auto launch_all = [&]() {
        for (int i = 0; i < n; ++i) {
            A a = arg; a.n = (arg.n < 0) ? arg.n : i;
            if (flags) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, nullptr, nullptr, flags, a);
            else hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, a);
        }
    };
    hipGraphExec_t exec = nullptr;
    if (graph) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        launch_all();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        if (graph) CK(hipGraphLaunch(exec, st)); else launch_all();
        CK(hipStreamSynchronize(st));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        if (us < best) best = us;
    }
    if (exec) CK(hipGraphExecDestroy(exec));
    CK(hipGetLastError());
    return best;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* buf;
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMemset(buf, 0, 1 << 20));
    CK(hipFuncSetAttribute((const void*)k_empty<Small>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_empty<Big>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_row<Big>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int N = 2000;
    Small s{buf, -1};
    Big b{}; b.p = buf; b.n = -1;
    Big br{}; br.p = buf; br.n = 0;
    printf("us per dependent launch (best of 5 chains of %d), one stream\n", N);
    printf("%-64s %8s %8s\n", "launch", "eager", "graph");
    struct Row { const char* name; int grid, threads; size_t lds; int kind; unsigned flags; };
    const Row rows[] = {
        {"empty, 256 WG x 256 thr, small kernarg", 256, 256, 0, 0, 0},
        {"empty, 256 WG x 512 thr, small kernarg", 256, 512, 0, 0, 0},
        {"empty, 256 WG x 512 thr, 240 B kernarg", 256, 512, 0, 1, 0},
        {"empty, 256 WG x 512 thr, 240 B kernarg, 28 KB dyn LDS", 256, 512, 28 * 1024, 1, 0},
        {"empty, 256 WG x 512 thr, 240 B kernarg, 150 KB dyn LDS", 256, 512, 150 * 1024, 1, 0},
        {"empty, 1024 WG x 256 thr, 240 B kernarg", 1024, 256, 0, 1, 0},
        {"empty, 32 WG x 64 thr, small kernarg", 32, 64, 0, 0, 0},
        {"empty, 1 WG x 64 thr, small kernarg", 1, 64, 0, 0, 0},
        {"row chain (read 16 KB, write 32 B / WG), 256 WG x 512 thr, 28 KB LDS", 256, 512, 28 * 1024, 2, 0},
        {"ANY ORDER: empty, 256 WG x 512 thr, 240 B kernarg, 28 KB dyn LDS", 256, 512, 28 * 1024, 1, hipExtAnyOrderLaunch},
        {"ANY ORDER: empty, 256 WG x 256 thr, small kernarg", 256, 256, 0, 0, hipExtAnyOrderLaunch},
    };
    for (const Row& r : rows) {
        double e, g;
        if (r.kind == 0) { e = chain(k_empty<Small>, s, r.grid, r.threads, r.lds, N, r.flags, st, false); g = chain(k_empty<Small>, s, r.grid, r.threads, r.lds, N, r.flags, st, true); }
        else if (r.kind == 1) { e = chain(k_empty<Big>, b, r.grid, r.threads, r.lds, N, r.flags, st, false); g = chain(k_empty<Big>, b, r.grid, r.threads, r.lds, N, r.flags, st, true); }
        else { e = chain(k_row<Big>, br, r.grid, r.threads, r.lds, N, r.flags, st, false); g = chain(k_row<Big>, br, r.grid, r.threads, r.lds, N, r.flags, st, true); }
        printf("%-64s %8.2f %8.2f\n", r.name, e, g);
        fflush(stdout);
    }
    // the null stream (torch's current stream): the engine launches there unless graph replay is on
    printf("%-64s %8.2f %8s\n", "NULL STREAM: empty, 256 WG x 512 thr, 240 B kernarg, 28 KB LDS",
           chain(k_empty<Big>, b, 256, 512, 28 * 1024, N, 0, (hipStream_t)0, false), "-");
    return 0;
}
