// Can a chain of DEPENDENT weight-streaming launches overlap its boundaries?  (round 4 experiment, stand-alone: no engine, no torch)
//
// The decode path is ~415 dependent launches per speculation step, each of which pays a dispatch gap (1.25 us), a pipe fill (~1 us)
// and a tail (~1.5 us) around its weight stream (profiles/r03_kernel_timeline.md).  The weight stream of launch i+1 does not depend
// on launch i -- only its activation rows do.  So: launch every kernel WITHOUT the barrier bit (hipExtAnyOrderLaunch: the packet
// processor starts placing launch i+1's workgroups as soon as launch i's are placed, i.e. as launch i's workgroups retire), let a
// workgroup request its weight ring at once, and make it wait on a device-side completion counter of its predecessor before it
// reads the rows (write-through stores + drained + relaxed agent-scope counter on the producer side, sc1 loads on the consumer side:
// the hand-off form lsk_attn.h already uses inside one launch).  The ring is then in flight across the predecessor's tail, the
// dispatch gap and the hand-off.
//
// This file measures exactly that on a GEMV-shaped stand-in of the projection kernel (256 workgroups x 8 waves, 16-deep ring of
// 1 KiB non-temporal buffer loads per wave, x rows staged through LDS, every workgroup needs the WHOLE output row of its predecessor):
//   mode A  in-order launches, plain loads / stores                       (what the engine does today)
//   mode B  any-order launches + counter hand-off, ring requested FIRST   (the proposal)
//   mode C  any-order launches + counter hand-off, ring requested after the wait (the hand-off alone, no run-ahead)
//   mode D  in-order launches + the hand-off protocol                     (what the protocol itself costs when nothing overlaps)
//   mode E  any-order launches + GRANULE hand-off (8-byte {payload, tag} write-through stores, the consumer sweeps the data itself
//           until every tag matches: no store drain, no counter, no flag), ring requested first
//   mode F  in-order launches + the granule protocol
// for three launch sizes (33.5 / 100 / 168 MB of weights = o_proj / q,k,v / gate,up at llama2-7B) and two residency regimes (dynamic
// LDS small enough for two workgroups per CU, or one).  The final rows of all modes must be bit-identical (a stale read shows).
// Every spin is bounded: a broken assumption gives an error count, never a hang.
//
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/chain_overlap tools/chain_overlap.hip ; run on the MI355X.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVES 8
#define THREADS 512
#define RING 16
#define KSTEPS 128              // 1 KiB wave-loads per tile: 128 KiB of weights per 16 outputs
#define XN 4096                 // row length (floats)
#define SPIN_LIMIT 20000

struct Args {
    const void* w; unsigned w_bytes;
    const float* xin; float* xout;
    unsigned* done; unsigned wait_target;
    unsigned* err;
    int tiles_per_wg;
    int mode;                    // bit 0: counter hand-off (wait, sc1 rows, publish); bit 1: ring requested before the wait;
                                 // bit 2: GRANULE hand-off instead: every row element travels as one 8-byte {payload, tag} store, the
                                 // consumer sweeps the row until every tag is this launch's -- no drain, no counter, no separate flag
    unsigned tag;                // granules: the tag of the row this launch READS (it writes tag + 1)
    const unsigned long long* gin; unsigned long long* gout;
};

__device__ __forceinline__ float wave_rowgroup_sum(float v) {    // sum over lanes with the same (lane & 15)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__global__ __launch_bounds__(THREADS) void k_stream(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xs = (float*)smem;                       // [XN]
    float* slab = (float*)(smem + XN * 4);          // [2][WAVES][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool chained = a.mode & 1, ahead = a.mode & 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
    const int tile0 = blockIdx.x * a.tiles_per_wg;
    const int units = a.tiles_per_wg;
    auto unit_off = [&](int u) -> unsigned { return ((unsigned)(tile0 + u) * KSTEPS + (unsigned)(w * RING)) * 1024u + (unsigned)lane * 16u; };

    u32x4 ring[RING];
    f32x4 x0, x1;
    if (!chained && !(a.mode & 4)) {                // today's order: rows first (they must not queue behind the ring), then the ring
        x0 = *(const f32x4*)(a.xin + tid * 8);
        x1 = *(const f32x4*)(a.xin + tid * 8 + 4);
    }
    if (!chained || ahead || (a.mode & 4)) {
#pragma unroll
        for (int s = 0; s < RING; ++s) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, unit_off(0) + s * 1024u, 0, 2);
    }
    const bool gran = a.mode & 4;
    if (gran) {
        // sweep this thread's 8 granules until all carry the tag (bounded); the loads are sc1 (L1-bypassing)
        const unsigned long long* src = a.gin + tid * 8;
        unsigned long long g[8];
        int spins = 0;
        bool ok;
        do {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) ok &= (unsigned)(g[j] >> 32) == a.tag;
            if (!__all(ok)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT / 8) { if (lane == 0) atomicAdd(a.err, 1u); break; }
            }
        } while (!__all(ok));
        x0 = f32x4{__builtin_bit_cast(float, (unsigned)g[0]), __builtin_bit_cast(float, (unsigned)g[1]), __builtin_bit_cast(float, (unsigned)g[2]), __builtin_bit_cast(float, (unsigned)g[3])};
        x1 = f32x4{__builtin_bit_cast(float, (unsigned)g[4]), __builtin_bit_cast(float, (unsigned)g[5]), __builtin_bit_cast(float, (unsigned)g[6]), __builtin_bit_cast(float, (unsigned)g[7])};
    }
    if (chained) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(a.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.wait_target) {
                __builtin_amdgcn_s_sleep(2);
                // give up (and let every later waiter give up at once) instead of hanging the box on a broken assumption
                if (++spins > SPIN_LIMIT || ((spins & 255) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    atomicAdd(a.err, 1u);
                    break;
                }
            }
        }
        __syncthreads();
    }
    // rows: 8 floats per thread
    if (chained) {
        const unsigned long long* src = (const unsigned long long*)(a.xin + tid * 8);
        unsigned long long q0 = __hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long q1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long q2 = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long q3 = __hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x0 = __builtin_bit_cast(f32x4, (unsigned long long __attribute__((ext_vector_type(2)))){q0, q1});
        x1 = __builtin_bit_cast(f32x4, (unsigned long long __attribute__((ext_vector_type(2)))){q2, q3});
    }
    if (chained && !ahead) {
#pragma unroll
        for (int s = 0; s < RING; ++s) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, unit_off(0) + s * 1024u, 0, 2);
    }
    *(f32x4*)(xs + tid * 8) = x0;
    *(f32x4*)(xs + tid * 8 + 4) = x1;
    __syncthreads();
    // this wave's x fragments: k-step s of the wave covers floats [(16 w + s) * 16 + (lane >> 4) * 4, +4) of both half rows
    f32x4 xf[RING];
#pragma unroll
    for (int s = 0; s < RING; ++s) {
        const int k = (w * RING + s) * 16 + (lane >> 4) * 4;
        xf[s] = *(const f32x4*)(xs + k) + *(const f32x4*)(xs + 2048 + k);
    }
    for (int u = 0; u < units; ++u) {
        float acc = 0.f;
        const bool more = u + 1 < units;
        const unsigned noff = more ? unit_off(u + 1) : 0xF0000000u;
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            const f32x4 wv = __builtin_bit_cast(f32x4, ring[s]);
            acc += wv[0] * xf[s][0] + wv[1] * xf[s][1] + wv[2] * xf[s][2] + wv[3] * xf[s][3];
            ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, more ? noff + s * 1024u : 0xF0000000u, 0, 2);
        }
        acc = wave_rowgroup_sum(acc);
        float* sl = slab + ((u & 1) * WAVES + w) * 64;
        sl[lane] = acc;
        __syncthreads();
        if (w == (u & 7) && lane < 16) {
            float t = 0.f;
            const float* sb = slab + (u & 1) * WAVES * 64 + lane;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) t += sb[ww * 64];
            const float y = __sinf(t * 8.0f) + 0.001f * (float)(lane);
            const int n = (tile0 + u) * 16 + lane;
            if (gran) __hip_atomic_store(a.gout + n, ((unsigned long long)(a.tag + 1) << 32) | __builtin_bit_cast(unsigned, y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (chained) __hip_atomic_store(a.xout + n, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // sc1: write-through
            else a.xout[n] = y;
        }
    }
    if (chained) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct Result { double us; unsigned checksum; unsigned errs; };

static unsigned long long *g_g0, *g_g1, *g_ginit;

static Result run_chain(int mode, bool any_order, int tiles_per_wg, size_t lds, int n, const std::vector<void*>& wbufs, size_t w_bytes,
                        float* x0, float* x1, float* xinit, unsigned* done, unsigned* err, hipStream_t st, int grid) {
    const size_t out_floats = (size_t)grid * tiles_per_wg * 16;
    double best = 1e30;
    unsigned checksum = 0, errs = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpyAsync(x0, xinit, XN * 4, hipMemcpyDeviceToDevice, st));
        CK(hipMemcpyAsync(g_g0, g_ginit, XN * 8, hipMemcpyDeviceToDevice, st));      // tag 0
        CK(hipMemsetAsync(g_g1, 0xff, 1 << 20, st));                                 // tag 0xffffffff: never expected
        CK(hipMemsetAsync(done, 0, 4, st));
        CK(hipMemsetAsync(err, 0, 4, st));
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
            Args a;
            a.w = wbufs[i % wbufs.size()]; a.w_bytes = (unsigned)w_bytes;
            a.xin = (i & 1) ? x1 : x0; a.xout = (i & 1) ? x0 : x1;
            a.done = done; a.wait_target = (unsigned)(i * grid); a.err = err; a.tiles_per_wg = tiles_per_wg; a.mode = mode;
            a.tag = (unsigned)i; a.gin = (i & 1) ? g_g1 : g_g0; a.gout = (i & 1) ? g_g0 : g_g1;
            if (any_order) hipExtLaunchKernelGGL(k_stream, dim3(grid), dim3(THREADS), lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, a);
            else hipLaunchKernelGGL(k_stream, dim3(grid), dim3(THREADS), lds, st, a);
        }
        CK(hipStreamSynchronize(st));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        if (us < best) best = us;
        std::vector<float> h(XN);
        unsigned e = 0;
        if (mode & 4) {
            std::vector<unsigned long long> hg(XN);
            CK(hipMemcpy(hg.data(), (n & 1) ? g_g1 : g_g0, XN * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < XN; ++i) { unsigned b = (unsigned)hg[i]; memcpy(&h[i], &b, 4); }
        } else
        CK(hipMemcpy(h.data(), (n & 1) ? x1 : x0, XN * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        unsigned c = 0;
        for (int i = 0; i < XN; ++i) { unsigned b; memcpy(&b, &h[i], 4); c = c * 1000003u + b; }
        if (rep == 0) checksum = c; else if (c != checksum) checksum = 0xdeadbeef;
        errs += e;
        (void)out_floats;
    }
    CK(hipGetLastError());
    return {best, checksum, errs};
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 240;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = 256;
    float *x0, *x1, *xinit;
    unsigned *done, *err;
    CK(hipMalloc(&x0, 1 << 20)); CK(hipMalloc(&x1, 1 << 20)); CK(hipMalloc(&xinit, XN * 4));
    CK(hipMalloc(&done, 256)); CK(hipMalloc(&err, 256));
    CK(hipMalloc(&g_g0, 1 << 20)); CK(hipMalloc(&g_g1, 1 << 20)); CK(hipMalloc(&g_ginit, XN * 8));
    {
        std::vector<float> h(XN);
        unsigned s = 12345;
        for (int i = 0; i < XN; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        CK(hipMemcpy(xinit, h.data(), XN * 4, hipMemcpyHostToDevice));
        std::vector<unsigned long long> hg(XN);
        for (int i = 0; i < XN; ++i) { unsigned b; memcpy(&b, &h[i], 4); hg[i] = b; }
        CK(hipMemcpy(g_ginit, hg.data(), XN * 8, hipMemcpyHostToDevice));
    }
    printf("chains of %d dependent GEMV-shaped launches, 256 WG x 512 threads, us per launch (best of 4)\n", n);
    printf("%-28s %8s | %9s %9s %9s %9s %9s %9s | %s\n", "weights / launch", "LDS", "A inorder", "B ahead", "C behind", "D proto", "E granule", "F gran-io", "identical rows / spin errors");
    const int tpws[] = {1, 3, 5};
    const size_t ldss[] = {40 * 1024, 100 * 1024};
    for (int tpw : tpws) {
        const size_t w_bytes = (size_t)grid * tpw * KSTEPS * 1024;
        const int nbuf = (int)((700ull << 20) / w_bytes) + 1;           // > 256 MiB of distinct weights: nothing comes from the Infinity Cache
        std::vector<void*> wbufs(nbuf);
        {
            std::vector<float> h(w_bytes / 4);
            unsigned s = 777 + tpw;
            for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.05f; }
            for (int b = 0; b < nbuf; ++b) {
                CK(hipMalloc(&wbufs[b], w_bytes));
                h[b] += 0.01f * b;
                CK(hipMemcpy(wbufs[b], h.data(), w_bytes, hipMemcpyHostToDevice));
            }
        }
        for (size_t lds : ldss) {
            Result A = run_chain(0, false, tpw, lds, n, wbufs, w_bytes, x0, x1, xinit, done, err, st, grid);
            Result B = run_chain(3, true, tpw, lds, n, wbufs, w_bytes, x0, x1, xinit, done, err, st, grid);
            Result C = run_chain(1, true, tpw, lds, n, wbufs, w_bytes, x0, x1, xinit, done, err, st, grid);
            Result D = run_chain(3, false, tpw, lds, n, wbufs, w_bytes, x0, x1, xinit, done, err, st, grid);
            Result E = run_chain(4, true, tpw, lds, n, wbufs, w_bytes, x0, x1, xinit, done, err, st, grid);
            Result F = run_chain(4, false, tpw, lds, n, wbufs, w_bytes, x0, x1, xinit, done, err, st, grid);
            const bool same = A.checksum == B.checksum && A.checksum == C.checksum && A.checksum == D.checksum && A.checksum != 0xdeadbeef
                              && A.checksum == E.checksum && A.checksum == F.checksum;
            char name[64];
            snprintf(name, sizeof name, "%.1f MB (%d tiles/WG)", w_bytes / 1e6, tpw);
            printf("%-28s %6zuKB | %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f | %s / %u %u %u %u %u  (stream alone at 6.8 TB/s: %.2f us)\n", name, lds >> 10, A.us, B.us, C.us, D.us,
                   E.us, F.us, same ? "yes" : "NO", B.errs, C.errs, D.errs, E.errs, F.errs, w_bytes / 6.8e6);
            fflush(stdout);
        }
        for (void* p : wbufs) CK(hipFree(p));
    }
    return 0;
}
