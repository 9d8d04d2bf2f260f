// Shared device/host helpers for the gfx950 kernels behind include/cnk.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
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
    const void* a; const void* b; void* c;            // [M][K], [Nout][K], [M][Nout]  (bf16)
    const float* bias; float* stats;
    const void* addend; const unsigned char* addend_mask;
    const void* bnf_x; const float* bnf_mean; const float* bnf_invstd; const float* bnf_gamma; const float* bnf_beta;
    const unsigned char* bnf_mask; float* bnf_partial; int bnf_relu, bnf_tile0;
    int64_t M; int K, Nout;
    double algo_flops, algo_bytes;
};
int gemm1x1_launch(int nj, const G1Args& a, hipStream_t s);

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int conv_groups(const cnk_conv_desc* d) { return d->groups > 1 ? d->groups : 1; }
static inline int conv_dil(const cnk_conv_desc* d) { return d->dilation > 1 ? d->dilation : 1; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
