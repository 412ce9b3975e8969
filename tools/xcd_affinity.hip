// Is there an exploitable XCD <-> HBM-stack affinity on this chip?  (round 4 experiment, stand-alone)
//
// Finding of round 3 (profiles/r03_kernel_timeline.md): next to a full-chip weight stream the odd-numbered XCDs stream ~10 % slower
// than the even ones, and the slowest workgroup ends every launch.  The packed weight layout and the tile -> workgroup map are ours,
// and workgroup b runs on XCD b % 8: IF the memory system interleaved addresses over the 8 HBM stacks in some simple way and an XCD
// reached "its" stacks faster, the packer could place every workgroup's tiles in memory near the XCD that streams them.
// Test: all 256 workgroups stream at once (8 waves, 16 x 1 KiB non-temporal loads in flight per wave, like the projection kernel),
// XCD x reading ONLY the blocks of residue class r = map(x) modulo 8 at block size G:
//     address(stream byte t of class r) = ((t / G) * 8 + r) * G + t % G
// for G = 256 B .. 1 MiB and map = (x + shift) % 8, (x ^ mask).  Reported: aggregate GB/s per (G, map) next to the plain linear split.
// If every cell equals the linear figure there is nothing to exploit at these granularities (the interleave is hashed / finer / the
// skew is not an affinity effect).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_affinity tools/xcd_affinity.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const unsigned char* buf;
    unsigned long long bytes_per_class;     // bytes every residue class holds (= total / 8)
    unsigned long long g;                   // block size in bytes; 0 = linear (workgroup b reads its own contiguous 1/256)
    int shift, mask;
    unsigned* sink;
    unsigned* xcc_seen;                      // [256] XCC id every workgroup really ran on
};

__global__ __launch_bounds__(512) void k_read(const Args a) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (tid == 0) a.xcc_seen[b] = xcc & 0xf;
    const int x = b & 7, idx = b >> 3;                                   // the observed placement (b % 8), 32 workgroups per XCD
    const int r = ((x + a.shift) & 7) ^ a.mask;
    const unsigned long long per_wg = a.bytes_per_class / 32;            // stream bytes of this workgroup inside its class
    const unsigned long long t0 = (unsigned long long)idx * per_wg;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long iters = per_wg / (512 * 16 * 16);           // 16 loads of 16 B per thread per iteration = 128 KiB per workgroup
    for (unsigned long long it = 0; it < iters; ++it) {
        u32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned long long t = t0 + ((it * 16 + j) * 512 + tid) * 16ull;
            unsigned long long addr;
            if (a.g == 0) addr = ((unsigned long long)b * per_wg) + ((it * 16 + j) * 512 + tid) * 16ull;       // linear: 256 contiguous slabs
            else addr = ((t / a.g) * 8 + (unsigned)r) * a.g + t % a.g;
            v[j] = __builtin_nontemporal_load((const u32x4*)(a.buf + addr));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) a.sink[tid] = 1;
}

int main() {
    const unsigned long long total = 4ull << 30;                           // 4 GiB: every cell streams all of it once
    unsigned char* buf;
    unsigned *sink, *xcc;
    CK(hipMalloc(&buf, total));
    CK(hipMemset(buf, 1, total));
    CK(hipMalloc(&sink, 4096));
    CK(hipMalloc(&xcc, 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](unsigned long long g, int shift, int mask) {
        Args a{buf, total / 8, g, shift, mask, sink, xcc};
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_read, dim3(256), dim3(512), 0, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return (double)total / (best * 1e-3) / 1e9;
    };
    printf("aggregate GB/s, 256 workgroups x 8 waves streaming 4 GiB (best of 3)\n");
    printf("linear split (workgroup b reads its own contiguous 16 MiB): %.0f\n", run(0, 0, 0));
    {
        std::vector<unsigned> h(256);
        CK(hipMemcpy(h.data(), xcc, 1024, hipMemcpyDeviceToHost));
        int ok = 0;
        for (int b = 0; b < 256; ++b) ok += (h[b] == (unsigned)(b & 7));
        printf("workgroups that ran on XCD b %% 8: %d / 256\n", ok);
    }
    const unsigned long long gs[] = {256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1048576, 2097152, 16777216};
    printf("%-10s |", "block G");
    for (int s = 0; s < 8; ++s) printf(" shift %d", s);
    printf(" |");
    for (int m = 1; m < 8; ++m) printf("  xor %d", m);
    printf("\n");
    for (unsigned long long g : gs) {
        printf("%-10llu |", g);
        for (int s = 0; s < 8; ++s) printf(" %7.0f", run(g, s, 0));
        printf(" |");
        for (int m = 1; m < 8; ++m) printf(" %6.0f", run(g, 0, m));
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
