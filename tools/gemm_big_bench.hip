// GPU micro-benchmark (not a test, not product code): the prompt-prefill projection kernel (layerskip_amd/csrc/lsk_gemm_big.h) at the
// four GEMM shapes of a llama2-7B prompt pass, one tile configuration after the other, HIP-event timed, with the number of output
// elements that differ from the first configuration of each shape (every configuration without a K-split walks K in the same order:
// bit-identical; round 6's transposed-product form was checked against the row-per-register form this way before the latter's residual
// and q/k/v epilogues were removed, profiles/r06_gemm_big_bench.txt).
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o build/gemm_big_bench tools/gemm_big_bench.hip && ./build/gemm_big_bench [rows ...]
//
// The weights of NL = 4 "layers" are cycled through (404 MB each: a launch never finds its panel in the 256 MiB Infinity Cache, as in
// the engine's layer loop).  Epilogues: the engine's own (RoPE + paged K / V^T append, SwiGLU, residual add).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../layerskip_amd/csrc/lsk_gemm_big.h"
#include "../layerskip_amd/csrc/lsk_small.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(elem_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (elem_t)(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}

__global__ void checksum_kernel(const unsigned short* p, size_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (unsigned long long)p[i] * (2 * (i % 1000003) + 1);
    atomicAdd(out, acc);
}

static const int NL = 4;
static const int H = 4096, I = 11008, NH = 32, NKV = 32, HD = 128;
static int g_iters = 20;

struct Bufs {
    elem_t *x, *act_in, *h, *act_out, *q, *kpool, *vpool, *cos, *sin;
    elem_t *wqkv[NL], *wo[NL], *wgu[NL], *wdown[NL];
    int *table, *kv_len;
    unsigned long long* sum;
};

static elem_t* dalloc(size_t n) { elem_t* p; CK(hipMalloc(&p, n * sizeof(elem_t))); return p; }

static elem_t* packed_weight(int n_rows, int k, unsigned seed, int tile_stride, int rope_hd) {
    elem_t* src = dalloc((size_t)n_rows * k);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, src, (size_t)n_rows * k, seed, 0.02f);
    elem_t* dst = dalloc((size_t)n_rows * k);
    const long long total = (long long)(n_rows / 16) * (k / 32) * 64;
    hipLaunchKernelGGL(lsk_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, src, n_rows, k, k, dst, 0, tile_stride, rope_hd);
    CK(hipDeviceSynchronize());
    CK(hipFree(src));
    return dst;
}

template <int EPI, int NTW, int MT, int PB, int NW, bool PIN, int KS, bool TR = true>
static float run_cfg(const char* name, BigGemmParams p, const elem_t* const* weights, elem_t* out, size_t out_elems, const elem_t* restore, Bufs& b,
                     std::vector<unsigned short>* first, double flops) {
    const int rb = (p.M + MT * 16 - 1) / (MT * 16);
    const int panels = (p.n_tiles + NW * NTW - 1) / (NW * NTW);
    const dim3 grid(rb * 8 * ((panels + 7) / 8));
    if ((p.K / LSK_BIG_BK / KS) % PB) { printf("  %-44s skipped (K-tiles per group not a multiple of the ring depth)\n", name); return 0.f; }
    hipEvent_t a, e;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&e));
    for (int it = 0; it < 3; ++it) {
        p.wp = weights[it % NL];
        hipLaunchKernelGGL((lsk_gemm_big_kernel<EPI, NTW, MT, PB, NW, PIN, KS, TR>), grid, dim3(NW * KS * 64), 0, 0, p);
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int it = 0; it < g_iters; ++it) {
        p.wp = weights[it % NL];
        hipLaunchKernelGGL((lsk_gemm_big_kernel<EPI, NTW, MT, PB, NW, PIN, KS, TR>), grid, dim3(NW * KS * 64), 0, 0, p);
    }
    CK(hipEventRecord(e, 0));
    CK(hipEventSynchronize(e));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, e));
    const float us = ms * 1e3f / g_iters;
    // one clean run for the comparison: the residual epilogue accumulates into h, so h is restored first
    if (restore) CK(hipMemcpy(out, restore, out_elems * sizeof(elem_t), hipMemcpyDeviceToDevice));
    p.wp = weights[0];
    hipLaunchKernelGGL((lsk_gemm_big_kernel<EPI, NTW, MT, PB, NW, PIN, KS, TR>), grid, dim3(NW * KS * 64), 0, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> host(out_elems);
    CK(hipMemcpy(host.data(), out, out_elems * sizeof(elem_t), hipMemcpyDeviceToHost));
    double maxd = 0.0, ref_rms = 0.0;
    size_t differ = 0;
    if (first->empty()) *first = host;
    for (size_t i = 0; i < out_elems; ++i) {
        unsigned ua = (unsigned)host[i] << 16, ub = (unsigned)(*first)[i] << 16;
        float fa, fb;
        memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
        const double d = fabs((double)fa - (double)fb);
        if (d > maxd) maxd = d;
        differ += host[i] != (*first)[i];
        ref_rms += (double)fb * fb;
    }
    printf("  %-44s %4u wgs %8.2f us %7.1f TFLOP/s   vs first: %zu of %zu differ, max |d| %.4g (rms %.3g)\n", name, grid.x, us, flops / us * 1e-6, differ, out_elems, maxd,
           sqrt(ref_rms / out_elems));
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(e));
    return us;
}

int main(int argc, char** argv) {
    std::vector<int> rows;
    for (int i = 1; i < argc; ++i) rows.push_back(atoi(argv[i]));
    if (rows.empty()) { rows.push_back(511); rows.push_back(2047); }
    const int max_rows = 4096;
    Bufs b;
    b.x = dalloc((size_t)max_rows * H); b.act_in = dalloc((size_t)max_rows * I); b.h = dalloc((size_t)max_rows * H);
    b.act_out = dalloc((size_t)max_rows * I); b.q = dalloc((size_t)max_rows * NH * HD);
    const int n_pages = max_rows / 128 + 1;
    b.kpool = dalloc((size_t)n_pages * 128 * NKV * HD); b.vpool = dalloc((size_t)n_pages * 128 * NKV * HD);
    b.cos = dalloc((size_t)(max_rows + 16) * HD / 2); b.sin = dalloc((size_t)(max_rows + 16) * HD / 2);
    elem_t* h0 = dalloc((size_t)max_rows * H);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, b.x, (size_t)max_rows * H, 11u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, b.act_in, (size_t)max_rows * I, 12u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, h0, (size_t)max_rows * H, 13u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, b.cos, (size_t)(max_rows + 16) * HD / 2, 14u, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, b.sin, (size_t)(max_rows + 16) * HD / 2, 15u, 1.0f);
    CK(hipMalloc(&b.table, sizeof(int) * n_pages)); CK(hipMalloc(&b.kv_len, sizeof(int))); CK(hipMalloc(&b.sum, 8));
    std::vector<int> tab(n_pages);
    for (int i = 0; i < n_pages; ++i) tab[i] = i;
    CK(hipMemcpy(b.table, tab.data(), sizeof(int) * n_pages, hipMemcpyHostToDevice));
    CK(hipMemset(b.kv_len, 0, sizeof(int)));
    for (int l = 0; l < NL; ++l) {
        b.wqkv[l] = packed_weight((NH + 2 * NKV) * HD, H, 100u + l, 1, 0);     // (the RoPE row permutation does not change the timing)
        b.wo[l] = packed_weight(H, NH * HD, 200u + l, 1, 0);
        b.wgu[l] = packed_weight(2 * I, H, 300u + l, 1, 0);
        b.wdown[l] = packed_weight(H, I, 400u + l, 1, 0);
    }
    for (int M : rows) {
        printf("== rows %d\n", M);
        {   // ---- q/k/v ----
            BigGemmParams p{};
            p.x = b.x; p.ldx = H; p.M = M; p.K = H; p.N = (NH + 2 * NKV) * HD; p.n_tiles = p.N / 16;
            p.q_out = b.q; p.ldq = NH * HD; p.kpool = b.kpool; p.vpool = b.vpool; p.block_table = b.table; p.page_size = 128;
            p.n_heads = NH; p.n_kv = NKV; p.head_dim = HD; p.rope_cos = b.cos; p.rope_sin = b.sin; p.kv_len = b.kv_len; p.pos_off = 0;
            const double fl = 2.0 * M * (double)p.N * p.K;
            std::vector<unsigned short> first;
            printf(" q/k/v  (N = %d, K = %d)\n", p.N, p.K);
            run_cfg<EPI_QKV, 2, 4, 2, 4, false, 1>("64 x 128, ring 2 (engine <= 1024 rows)", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
            run_cfg<EPI_QKV, 2, 8, 2, 4, false, 1>("128 x 128, ring 2 (engine > 1024 rows)", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
            run_cfg<EPI_QKV, 2, 4, 4, 4, false, 1>("64 x 128, ring 4", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
            run_cfg<EPI_QKV, 2, 4, 2, 8, false, 1>("64 x 256 (8 waves), ring 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
            run_cfg<EPI_QKV, 2, 4, 2, 4, false, 2>("64 x 128, ring 2, K-split 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
            if (getenv("LSK_BENCH_EXPLORE")) {      // tile shapes that are not in the engine (three 16-column tiles per wave: 192-column workgroup tiles)
                run_cfg<EPI_QKV, 3, 8, 2, 4, false, 1>("128 x 192, ring 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 3, 8, 2, 4, false, 2>("128 x 192, ring 2, K-split 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 3, 4, 2, 4, false, 1>("64 x 192, ring 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 3, 4, 2, 4, false, 2>("64 x 192, ring 2, K-split 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 4, 4, 2, 4, false, 1>("64 x 256 (4 waves), ring 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 4, 4, 2, 4, false, 2>("64 x 256 (4 waves), ring 2, K-split 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 2, 8, 2, 8, false, 1>("128 x 256 (8 waves), ring 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 2, 8, 4, 4, false, 1>("128 x 128, ring 4", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 2, 8, 2, 8, true, 1>("128 x 256 (8 waves), ring 2, pinned", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 2, 8, 4, 8, false, 1>("128 x 256 (8 waves), ring 4", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 2, 8, 4, 8, true, 1>("128 x 256 (8 waves), ring 4, pinned", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 3, 8, 2, 4, true, 1>("128 x 192, ring 2, pinned", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 3, 8, 2, 8, false, 1>("128 x 384 (8 waves), ring 2", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
                run_cfg<EPI_QKV, 3, 8, 2, 8, true, 1>("128 x 384 (8 waves), ring 2, pinned", p, b.wqkv, b.q, (size_t)M * NH * HD, nullptr, b, &first, fl);
            }
        }
        {   // ---- gate/up ----
            BigGemmParams p{};
            p.x = b.x; p.ldx = H; p.M = M; p.K = H; p.N = 2 * I; p.n_tiles = p.N / 16; p.act = b.act_out; p.ldact = I;
            const double fl = 2.0 * M * (double)p.N * p.K;
            std::vector<unsigned short> first;
            printf(" gate/up (N = %d, K = %d)\n", p.N, p.K);
            run_cfg<EPI_SWIGLU, 2, 8, 2, 4, false, 1, false>("128 x 128, ring 2, row-per-register (engine)", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
            run_cfg<EPI_SWIGLU, 2, 8, 2, 4, false, 1, true>("128 x 128, ring 2, transposed", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
            run_cfg<EPI_SWIGLU, 2, 8, 2, 4, false, 2, false>("128 x 128, ring 2, K-split 2", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
            if (getenv("LSK_BENCH_EXPLORE")) {
                run_cfg<EPI_SWIGLU, 2, 8, 2, 8, false, 1, false>("128 x 256 (8 waves), ring 2", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
                run_cfg<EPI_SWIGLU, 2, 8, 2, 8, false, 1, true>("128 x 256 (8 waves), ring 2, transposed", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
                run_cfg<EPI_SWIGLU, 4, 4, 2, 4, false, 1, false>("64 x 256 (4 waves), ring 2", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
                run_cfg<EPI_SWIGLU, 2, 8, 4, 4, false, 1, false>("128 x 128, ring 4", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
                run_cfg<EPI_SWIGLU, 2, 8, 2, 8, true, 1, false>("128 x 256 (8 waves), ring 2, pinned", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
                run_cfg<EPI_SWIGLU, 2, 8, 4, 8, true, 1, false>("128 x 256 (8 waves), ring 4, pinned", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
                run_cfg<EPI_SWIGLU, 2, 8, 2, 4, true, 1, false>("128 x 128, ring 2, pinned", p, b.wgu, b.act_out, (size_t)M * I, nullptr, b, &first, fl);
            }
        }
        for (int which = 0; which < 2; ++which) {   // ---- o_proj, down ----
            BigGemmParams p{};
            const int K = which ? I : NH * HD;
            p.x = which ? b.act_in : b.x; p.ldx = K; p.M = M; p.K = K; p.N = H; p.n_tiles = p.N / 16; p.h = b.h; p.ldh = H;
            const double fl = 2.0 * M * (double)p.N * p.K;
            std::vector<unsigned short> first;
            printf(" %s (N = %d, K = %d)\n", which ? "down" : "o_proj", p.N, p.K);
            run_cfg<EPI_RESID, 2, 4, 4, 4, true, 1>("64 x 128, ring 4, pinned (engine, > 384 wgs)", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
            run_cfg<EPI_RESID, 2, 4, 2, 4, true, 2>("64 x 128, ring 2, pinned, K-split 2 (engine, <= 384 wgs)", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
            run_cfg<EPI_RESID, 2, 8, 2, 4, false, 1>("128 x 128, ring 2 (engine, >= 512 wgs of 128 rows)", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
            run_cfg<EPI_RESID, 2, 4, 4, 4, true, 2>("64 x 128, ring 4, pinned, K-split 2", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
            run_cfg<EPI_RESID, 2, 8, 2, 4, false, 2>("128 x 128, ring 2, K-split 2", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
            run_cfg<EPI_RESID, 2, 2, 2, 4, false, 2>("32 x 128, ring 2, K-split 2", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
            if (getenv("LSK_BENCH_EXPLORE")) {
                run_cfg<EPI_RESID, 2, 8, 2, 8, false, 1>("128 x 256 (8 waves), ring 2", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
                run_cfg<EPI_RESID, 2, 8, 2, 8, true, 1>("128 x 256 (8 waves), ring 2, pinned", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
                run_cfg<EPI_RESID, 2, 4, 2, 8, true, 1>("64 x 256 (8 waves), ring 2, pinned", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
                run_cfg<EPI_RESID, 2, 8, 4, 4, false, 1>("128 x 128, ring 4", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
                run_cfg<EPI_RESID, 2, 8, 2, 4, true, 1>("128 x 128, ring 2, pinned", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
                run_cfg<EPI_RESID, 4, 4, 2, 4, true, 1>("64 x 256 (4 waves), ring 2, pinned", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
                run_cfg<EPI_RESID, 4, 4, 2, 4, true, 2>("64 x 256 (4 waves), ring 2, pinned, K-split 2", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
                run_cfg<EPI_RESID, 2, 8, 4, 8, true, 1>("128 x 256 (8 waves), ring 4, pinned", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
                run_cfg<EPI_RESID, 3, 8, 2, 8, true, 1>("128 x 384 (8 waves), ring 2, pinned", p, which ? b.wdown : b.wo, b.h, (size_t)M * H, h0, b, &first, fl);
            }
        }
    }
    return 0;
}
