// Shared device helpers and kernel parameter blocks of the MI355X self-speculative decoding engine.
// gfx950 only: 64-wide wavefronts, v_mfma_f32_16x16x32_bf16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The model dtype is a BUILD-time choice: the same sources give liblayerskip_hip.so (bf16, the BASELINE configs) and,
// with -DLSK_ELEM_F16, liblayerskip_hip_f16.so (fp16, the dtype the reference's generate.py hard-codes, generate.py:63).
// Everything below is written against elem_t: storage, the rounding points (f2e / rnd_e) and the MFMA instruction.
#ifdef LSK_ELEM_F16
typedef _Float16 elem_t;
typedef _Float16 elem8 __attribute__((ext_vector_type(8)));
#define LSK_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define LSK_ELEM_DTYPE 1
#else
typedef __bf16 elem_t;
typedef __bf16 elem8 __attribute__((ext_vector_type(8)));
#define LSK_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define LSK_ELEM_DTYPE 0
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define LSK_WAVES 8            // waves per projection workgroup at most (the split-K factor inside a workgroup is 8 or 4: lsk_gemm_waves, lsk_gemm.h)
#define LSK_SPW 16             // k-steps per wave per (tile, chunk) unit = depth of the weight ring
#ifndef LSK_FORCE_WAVES
#define LSK_FORCE_WAVES 0      // measurement builds: -DLSK_FORCE_WAVES=8 (rounds 1-5's shape everywhere) or 4
#endif
#define LSK_ROWS 16
#define LSK_HDR_WORDS 40           // int32 words of the layer pipeline's message header (layout: lsk_accept.h); must fit ONE hidden row
#define LSK_ATTN_PAGE 128          // KV page size == keys per decode-attention workgroup
#define LSK_PAGE_SHIFT 7           // log2(LSK_ATTN_PAGE): lsk_check_cfg refuses any other page size, so position -> (page, slot) is a shift and a mask in
                                   // the prefill kernel (a division by the runtime page_size field is a ~35-instruction VALU sequence per row and lane:
                                   // 32 of them per lane in its q/k/v epilogue; the decode kernel's 8-row template has no register to spare for the change)
static_assert((1 << LSK_PAGE_SHIFT) == LSK_ATTN_PAGE, "page shift");
// 16-column tiles per head (head_dim 64 or 128 -> 4 or 8): tile index -> (head, tile in head) without a division
__device__ __forceinline__ int lsk_tph_shift(int head_dim) { return head_dim == 128 ? 3 : 2; }
#define LSK_SAMPLE_REG_VOCAB 32768  // sample=True: rows of up to this many logits live in one workgroup's registers (lsk_sample.h)

__device__ __forceinline__ float e2f(elem_t v) { return (float)v; }
__device__ __forceinline__ elem_t f2e(float v) { return (elem_t)v; }   // round-to-nearest-even
// One rounding to the model dtype, result widened again (the reference rounds wherever HF's
// bf16 modules materialise a tensor).
__device__ __forceinline__ float rnd_e(float v) { return (float)((elem_t)v); }

// Cross-lane traffic goes through DPP (VALU, a few cycles) instead of __shfl_xor, which hipcc lowers to
// ds_bpermute_b32: an LDS-pipe round trip of ~100 cycles per step that sat 6-deep on every reduction of the
// RMSNorm prologue and 8-deep on every softmax row of the (latency-bound) attention kernels.
template <int CTRL>
__device__ __forceinline__ float lsk_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int lsk_dpp(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
#define LSK_ROW_ROR(n) (0x120 + (n))      // rotate right inside each row of 16 lanes

// Lane i <-> lane i^8 of the same 16-lane row (a rotation by 8 of a row of 16 is exactly that exchange).
__device__ __forceinline__ float row_xor8(float v) { return lsk_dpp<LSK_ROW_ROR(8)>(v); }

// All-reduce over each 16-lane row.  Rotations by 8, 4, 2, 1 pair the same partial sums as the xor butterfly
// (after the step of stride s the value is periodic with period s), so the result is bit-identical to it.
__device__ __forceinline__ float row16_sum(float v) {
    v += lsk_dpp<LSK_ROW_ROR(8)>(v);
    v += lsk_dpp<LSK_ROW_ROR(4)>(v);
    v += lsk_dpp<LSK_ROW_ROR(2)>(v);
    v += lsk_dpp<LSK_ROW_ROR(1)>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, lsk_dpp<LSK_ROW_ROR(8)>(v));
    v = fmaxf(v, lsk_dpp<LSK_ROW_ROR(4)>(v));
    v = fmaxf(v, lsk_dpp<LSK_ROW_ROR(2)>(v));
    v = fmaxf(v, lsk_dpp<LSK_ROW_ROR(1)>(v));
    return v;
}

// All-reduce over the 4 lanes {c, c + 16, c + 32, c + 48} of a wave (one column of an MFMA C tile lives in these 4 lanes' registers):
// gfx950's row swaps.  v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second; fed two copies of a value, each leaves (even
// partner, odd partner) of every lane in the two registers, so all 4 lanes combine the same operands in the same order and end
// with bit-identical results.  (Inline assembly: the builtin of this toolchain loses the second result; the s_nop covers the
// VALU-write -> permlane-read hazard the compiler can no longer see.)
__device__ __forceinline__ void lsk_row_swap16(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lsk_row_swap32(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float col4_max(float v) {
    float a = v, b = v;
    lsk_row_swap16(a, b);
    a = fmaxf(a, b); b = a;
    lsk_row_swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float col4_sum(float v) {
    float a = v, b = v;
    lsk_row_swap16(a, b);
    a = a + b; b = a;
    lsk_row_swap32(a, b);
    return a + b;
}

// One step of the (value, index) maximum over a 16-lane row, lowest index wins ties (torch.argmax): the lane takes the pair of
// the lane ROT places to its right if that is better.
template <int ROT>
__device__ __forceinline__ void lsk_row16_argmax_step(float& v, int& idx) {
    const float ov = lsk_dpp<LSK_ROW_ROR(ROT)>(v);
    const int oi = lsk_dpp<LSK_ROW_ROR(ROT)>(idx);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
}

// Sum over the 64 lanes of a wave, same value (wave-uniform) in every lane: four row sums, then the four row
// totals through v_readlane in a fixed order.
__device__ __forceinline__ float wave_sum(float v) {
    const int r = __builtin_bit_cast(int, row16_sum(v));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(r, 48));
    return (r0 + r1) + (r2 + r3);
}

// ---- in-kernel timeline (measurement builds only: tools/kernel_timeline.py compiles the engine with -DLSK_TRACE) ----------
// Every workgroup of the decode kernels stamps the 100 MHz realtime counter at a few points of its life and writes the stamps,
// with the XCD / CU it ran on, to a host-provided buffer at its very end.  Without -DLSK_TRACE the macros expand to nothing and
// the parameter blocks have no extra field: the default build's device code is unchanged.
#ifdef LSK_TRACE
#define LSK_TRACE_WORDS 12
#ifndef LSK_TRACE_TID
#define LSK_TRACE_TID 0           // the thread (i.e. the wave) whose stamps are kept
#endif
#define LSK_TRACE_MAX_WGS 2048
struct LskTrace { unsigned long long* buf; int seq; };
#define LSK_TRACE_FIELD LskTrace trace;
#define LSK_TRACE_DECL unsigned long long lsk_tr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define LSK_TRACE_POINT(k) do { __builtin_amdgcn_sched_barrier(0); lsk_tr[k] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LSK_TRACE_FLUSH(p, wg) do { \
        if (threadIdx.x == LSK_TRACE_TID && (p).trace.buf != nullptr && (wg) < LSK_TRACE_MAX_WGS) { \
            unsigned hw_id, xcc_id; \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id)); \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id)); \
            unsigned long long* lsk_td = (p).trace.buf + ((size_t)(p).trace.seq * LSK_TRACE_MAX_WGS + (wg)) * LSK_TRACE_WORDS; \
            for (int lsk_k = 0; lsk_k < 10; ++lsk_k) lsk_td[lsk_k] = lsk_tr[lsk_k]; \
            lsk_td[10] = hw_id; lsk_td[11] = xcc_id; \
        } } while (0)
#else
#define LSK_TRACE_FIELD
#define LSK_TRACE_DECL
#define LSK_TRACE_POINT(k)
#define LSK_TRACE_FLUSH(p, wg)
#endif

// ---- projection (skinny GEMM) -----------------------------------------------------------------
enum { PRO_PLAIN = 0, PRO_RMS = 1 };
enum { EPI_F32 = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_QKV = 3, EPI_HEAD = 4 };

struct GemmParams {
    const elem_t* x;        // [M][ldx] input rows
    int ldx;
    int M;                  // 1..16
    int K;                  // multiple of 32
    const elem_t* wp;       // packed weight tiles [n_tiles][K/32][64][8]
    unsigned wp_bytes;
    int N;                  // logical output features (rows of the nn.Linear weight)
    int n_tiles;            // ceil(N / 16)
    int tiles_per_wg;       // <= 8 (<= 16 for SWIGLU: 8 gate/up pairs; any number for EPI_HEAD when K is one chunk): set by launch_gemm
    const elem_t* norm_w;   // PRO_RMS gain [K]
    float eps;
    // EPI_F32
    float* y;               // [M][N]
    // EPI_RESID: h[row][n] = bf16(h + bf16(acc))
    elem_t* h;
    int ldh;
    // EPI_SWIGLU: act[row][p*16+c]
    elem_t* act;
    int ldact;
    // EPI_QKV
    elem_t* q_out;          // [M][n_heads*head_dim]
    int ldq;
    elem_t* kpool;          // this layer's K pages [page][n_kv][page_size][head_dim]
    elem_t* vpool;
    const int* block_table;
    int page_size;
    int n_heads;
    int n_kv;
    int head_dim;
    const elem_t* rope_cos; // [rope_len][head_dim/2]
    const elem_t* rope_sin;
    const int* kv_len;      // device scalar: verified context length C
    int pos_off;            // row i sits at position *kv_len + pos_off + i
    // EPI_HEAD
    float* logits;          // optional [M][ld_logits] fp32 (bf16-rounded values)
    int ld_logits;
    float* part_val;        // [grid][16]
    int* part_idx;          // [grid][16]
    LSK_TRACE_FIELD
};

struct StepState {
    int kv_len;             // verified context length (all layers)
    int next_token;
    int pad[14];
};
