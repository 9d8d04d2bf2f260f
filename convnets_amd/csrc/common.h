// Shared device/host helpers for the gfx950 kernels behind include/cnk.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/cnk.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define CNK_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------------
void cnk_set_error(const char* fmt, ...);
#define CNK_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            cnk_set_error(__VA_ARGS__);   \
            return 1;                     \
        }                                 \
    } while (0)
#define CNK_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            cnk_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                       \
        }                                                                   \
    } while (0)

// ---- A/B switches (host; api.hip) ----------------------------------------------------------------
// Every CNK_* environment switch of the library is resolved ONCE, when libcnk.so is loaded, into this table; a
// slice / workspace query and the launch it sizes therefore always see the same choice, and no launch
// path calls getenv.  In-process A/B runs (tests/, tools/bench_*.py) change os.environ and then call the test hook
// cnk_debug_reload_switches() (exported, deliberately not part of include/cnk.h).
struct CnkSwitches {
    int pp3x3;                 // CNK_PP3X3 (1): ping-pong 3x3 core on / off
    int pp_cfg;                // CNK_PP_CFG (-1): force configuration 0..3 of conv_pp.hip where it fits
    int pp_full;               // CNK_PP_FULL (-1): 1 / 0 force the fused data gradients onto / off the ping-pong core
    int pp_ws;                 // CNK_PP_WS (1): the weight-stationary 64 -> 64 channel 3x3 kernel of conv_pp.hip on / off
    int gemm1x1;               // CNK_GEMM1X1 (1): 0 off, 1 by rule, 2 / 4 force that tile
    int igemm_halo;            // CNK_IGEMM_HALO (1)
    int igemm_halo_pf;         // CNK_IGEMM_HALO_PF (-1): 0 / 1 force the plain / weight-prefetching halo kernel
    int rowpatch_stats;        // CNK_ROWPATCH_STATS (1)
    int rowpatch_wide;         // CNK_ROWPATCH_WIDE (1): first-conv forward items of 32 pixels x 64 channels (0: x 32 channels)
    int wgrad_big;             // CNK_WGRAD_BIG (1)
    int wgrad3;                // CNK_WGRAD3 (1)
    int wgrad3_slots_pct;      // CNK_WGRAD3_SLOTS_PCT (-1: as wgrad_slots_pct): the same for the all-taps 3x3 kernel
    int wgrad_slots_pct;       // CNK_WGRAD_SLOTS_PCT (-1: all): A/B override of the share of the workgroup slots the weight gradients' split-K fills
    int bn_pool_elementwise;   // CNK_BN_POOL_ELEMENTWISE (0)
    int bn_fused_final;        // CNK_BN_FUSED_FINAL (1): finalisation of the BN partials inside the merge launch
    int fold3_nwg;             // CNK_FOLD3_NWG (512): workgroups of the one-pass folded backward (conv_fold3.hip; A/B)
};
const CnkSwitches& cnk_sw();

// Launch counters per kernel variant (test hook cnk_debug_counter(id), exported, not part of include/cnk.h): a parity
// test that forces a configuration asserts through these that the configuration actually ran.
enum {
    CNK_CNT_PP_A = 0, CNK_CNT_PP_A1 = 1, CNK_CNT_PP_B = 2, CNK_CNT_PP_C = 3,      // conv_pp.hip configurations
    CNK_CNT_HALO = 4, CNK_CNT_HALO_PF = 5, CNK_CNT_GATHER = 6, CNK_CNT_G1_2 = 7, CNK_CNT_G1_4 = 8,
    CNK_CNT_PP_FULL = 9,                                                          // ping-pong launches with the FULL epilogue
    CNK_CNT_BN_FUSED_FINAL = 10, CNK_CNT_BN_FINAL_LAUNCH = 11,
    CNK_CNT_PP_WS = 12,                                                           // weight-stationary 64 -> 64 launches
    CNK_CNT_BN_FINAL_NOBANK = 13,                                                 // merge + finalise fell back to two launches: no ticket bank
    CNK_CNT_MAX = 32
};
void cnk_count(int id);

// ---- opt-in for more than 64 KiB of dynamic LDS ------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one static object of this type per
// kernel instantiation remembers the devices already set (a bit per device, atomic: launches come from the Python main
// thread and from autograd's device thread), and the return code is checked.
struct LdsOptIn {
    std::atomic<uint64_t> done{0};
    hipError_t ensure(const void* fn, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
        return e;
    }
};
#define CNK_LDS_OPTIN(kernel, bytes, name)                                                              \
    do {                                                                                                \
        static LdsOptIn optin__;                                                                        \
        const hipError_t oe__ = optin__.ensure((const void*)(kernel), (bytes));                         \
        CNK_CHECK_ARG(oe__ == hipSuccess, "%s: cannot reserve %d bytes of LDS: %s", name, (int)(bytes), \
                      hipGetErrorString(oe__));                                                         \
    } while (0)

// ---- profiling hooks (host; api.hip) ------------------------------------------------------------
void cnk_prof_start(int cls, hipStream_t s, double flops, double bytes);
void cnk_prof_stop(int cls, hipStream_t s);
struct ProfScope {
    int cls;
    hipStream_t s;
    ProfScope(int c, hipStream_t st, double flops, double bytes) : cls(c), s(st) {
        cnk_prof_start(cls, s, flops, bytes);
    }
    ~ProfScope() { cnk_prof_stop(cls, s); }
};

// Kernel-level scopes: exactly one launch of the named kernel family between the two events (the class scopes
// above bracket a whole C-ABI call, which may issue several kernels).  Same record store, higher class bits.
#define CNK_KF_GATHER (1 << 16)        /* igemm_gather_kernel */
#define CNK_KF_HALO (1 << 17)          /* igemm_halo3x3_kernel, igemm_halo3x3_pf_kernel */
#define CNK_KF_WGRAD (1 << 18)         /* wgrad_mfma_kernel */
#define CNK_KF_BN_APPLY (1 << 19)      /* bn_apply_kernel */
#define CNK_KF_BN_BWD_APPLY (1 << 20)  /* bn_bwd_apply_kernel */

// ---- division by a run-time constant (valid for dividends < 2^31) -----------------------------
struct FastDiv {
    uint32_t mul, shr, d, pad;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    f.shr = l;
    f.d = d;
    f.pad = 0;
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t m, const FastDiv& f) {
    return (__umulhi(m, f.mul) + m) >> f.shr;
}

// ---- element helpers ------------------------------------------------------------------------------
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int PER16 = 4;   // elements per 16-byte chunk
};
template <> struct ElemTraits<bf16> {
    static constexpr int PER16 = 8;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 16-byte vector of T <-> floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = f; }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (bf16)f; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One stride-parity class of the data gradient (see conv_igemm.hip).
struct DgradClass {
    int ph, pw, OH, OW, ntaps;
    signed char kh[32], kw[32], dh[32], dw[32];
};

// Ping-pong 3x3 core (conv_pp.hip): arguments of one launch, filled by conv_igemm.hip from its GatherParams.
struct PPArgs {
    const void* src;
    const void* wpk;
    void* dst;
    const float* bias;
    const void* addend;
    const unsigned char* addend_mask;
    float* stats;
    const void* bnf_x;
    const float* bnf_mean;
    const float* bnf_invstd;
    const float* bnf_gamma;
    const float* bnf_beta;
    const unsigned char* bnf_mask;
    float* bnf_partial;
    int bnf_relu, bnf_tile0, relu;
    int N, H, W, C, Nout;
    int dgrad;                 // data-gradient launch (configuration choice, conv_pp.hip pp_pick)
    signed char dh[9], dw[9];
    double algo_bytes;
};
bool pp3x3_ok(int N, int H, int W, int C, int Nout, int dgrad, int fused);
int64_t pp3x3_slices(int N, int H, int W, int C, int Nout, int dgrad);
int pp3x3_launch(const PPArgs& a, hipStream_t s);
// conv_gemm1x1.hip: dense 1x1 / stride-1 convolutions as a GEMM with a four-stage LDS-DMA ring (bf16)
int gemm1x1_pick(int64_t M, int K, int Nout, int full);          // 0: no; 2 / 4: column blocks per wave
int64_t gemm1x1_slices(int64_t M);
struct G1Args {
    const void* a; const void* b; void* c;            // [M][K], [Nout][K + K2], [M][Nout]  (bf16)
    const void* a2; int K2;                           // optional second K segment [M][K2] (nullptr / 0: none)
    const float* bias; float* stats;
    const void* addend; const unsigned char* addend_mask;
    const void* bnf_x; const float* bnf_mean; const float* bnf_invstd; const float* bnf_gamma; const float* bnf_beta;
    const unsigned char* bnf_mask; float* bnf_partial; int bnf_relu, bnf_tile0;
    int64_t M; int K, Nout;
    double algo_flops, algo_bytes;
};
int gemm1x1_launch(int nj, const G1Args& a, hipStream_t s);

// cnk_conv2d_wgrad_bnfold: the BN layer behind the conv whose backward apply pass runs inside the weight-gradient kernel's staging
struct WgradFold {
    const void* y;            // conv output = BN input [M][Cout]
    const float* coef;        // [3][Cout] (cnk_bn_bwd_coef)
    const float* mean;
    const float* invstd;
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int conv_groups(const cnk_conv_desc* d) { return d->groups > 1 ? d->groups : 1; }
static inline int conv_dil(const cnk_conv_desc* d) { return d->dilation > 1 ? d->dilation : 1; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
