// Is a dependent launch WITHOUT the barrier bit (hipExtAnyOrderLaunch) placed on the CUs while its predecessor still runs?
// Round 4 found it was not (profiles/r04_attn_oproj.md: o_proj's first instruction 0.69 us after attention's last acknowledgement) -- with an
// 8-wave, 166-register successor that did not FIT beside the predecessor's waves.  Round 6's four-wave projections would fit.  Probe:
// kernel A = `ga` workgroups of 256 threads that each hold the CU for `hold_us` (s_memrealtime loop) with REGS_A registers pinned;
// kernel B = 256 workgroups of `tb` threads with REGS_B registers, launched right behind A with or without the barrier bit, every
// workgroup stamping its first instruction.  Printed: B's first / median / last start relative to A's first start and A's end.
// build: hipcc --offload-arch=gfx950 -O3 -o build/anyorder_probe tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }     // 100 MHz

template <int REGS>
__device__ __forceinline__ float pin_registers(float seed) {
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = seed * (float)(i + 1);
#pragma unroll
    for (int i = 0; i < REGS; ++i) asm volatile("" : "+v"(r[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += r[i];
    return s;
}

template <int REGS>
__global__ __launch_bounds__(512) void k_hold(unsigned long long* stamps, float* sink, int hold_ticks) {
    const unsigned long long t0 = now();
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = (float)(threadIdx.x + i);
    while ((long long)(now() - t0) < hold_ticks) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) asm volatile("" : "+v"(r[i]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += r[i];
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = now(); }
    if (s == -1.f) sink[0] = s;
}

template <int REGS>
__global__ __launch_bounds__(512) void k_stamp(unsigned long long* stamps, float* sink) {
    const unsigned long long t0 = now();
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = (float)(threadIdx.x * i);
#pragma unroll
    for (int i = 0; i < REGS; ++i) asm volatile("" : "+v"(r[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += r[i];
    if (threadIdx.x == 0) stamps[blockIdx.x] = t0;
    if (s == -1.f) sink[0] = s;
}

template <int RA, int RB>
static void probe(const char* name, int ga, int tb, int hold_us, hipStream_t st, unsigned long long* sa, unsigned long long* sb, float* sink) {
    for (int any = 0; any < 2; ++any) {
        std::vector<double> first, med, last, aend;
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL((k_hold<RA>), dim3(ga), dim3(256), 0, st, sa, sink, hold_us * 100);
            if (any) hipExtLaunchKernelGGL((k_stamp<RB>), dim3(256), dim3(tb), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, sb, sink);
            else hipLaunchKernelGGL((k_stamp<RB>), dim3(256), dim3(tb), 0, st, sb, sink);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> ha(2 * ga), hb(256);
            CK(hipMemcpy(ha.data(), sa, sizeof(unsigned long long) * 2 * ga, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb.data(), sb, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost));
            unsigned long long a0 = ~0ull, a1 = 0;
            for (int i = 0; i < ga; ++i) { a0 = std::min(a0, ha[2 * i]); a1 = std::max(a1, ha[2 * i + 1]); }
            std::sort(hb.begin(), hb.end());
            if (rep < 4) continue;
            first.push_back((double)(long long)(hb[0] - a0) / 100.0);
            med.push_back((double)(long long)(hb[128] - a0) / 100.0);
            last.push_back((double)(long long)(hb[255] - a0) / 100.0);
            aend.push_back((double)(long long)(a1 - a0) / 100.0);
        }
        auto m = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("  %-58s %-9s A ends %6.2f us | B's workgroups start: first %6.2f  median %6.2f  last %6.2f us after A's first\n", name,
               any ? "any-order" : "in-order", m(aend), m(first), m(med), m(last));
    }
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *sa, *sb;
    float* sink;
    CK(hipMalloc(&sa, 8 * 2 * 4096)); CK(hipMalloc(&sb, 8 * 256)); CK(hipMalloc(&sink, 64));
    // A like the attention launch (640 four-wave workgroups, ~106 registers) or a projection (256 eight-wave ones are 512 threads: here 512
    // four-wave workgroups stand in), B like a four-wave projection (~176 registers) or an eight-wave one
    probe<96, 160>("A 640 x 256 thr x ~100 regs, 6 us | B 256 x 256 thr x ~170 regs", 640, 256, 6, st, sa, sb, sink);
    probe<96, 160>("A 640 x 256 thr x ~100 regs, 6 us | B 256 x 512 thr x ~170 regs", 640, 512, 6, st, sa, sb, sink);
    probe<96, 160>("A 192 x 256 thr x ~100 regs, 6 us | B 256 x 256 thr x ~170 regs", 192, 256, 6, st, sa, sb, sink);
    probe<160, 160>("A 512 x 256 thr x ~170 regs, 20 us | B 256 x 256 thr x ~170 regs", 512, 256, 20, st, sa, sb, sink);
    probe<160, 160>("A 256 x 256 thr x ~170 regs, 20 us | B 256 x 256 thr x ~170 regs", 256, 256, 20, st, sa, sb, sink);
    probe<32, 32>("A 256 x 256 thr x ~40 regs, 6 us | B 256 x 256 thr x ~40 regs", 256, 256, 6, st, sa, sb, sink);
    return 0;
}
