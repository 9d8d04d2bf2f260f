// MFMA convolution for layers whose channel count does not fill a 128-byte K step: the first layer
// of ResNet (7x7/2, Cin=3), VGG (3x3, Cin=3), AlexNet (11x11/4, Cin=3) and the LeNet/ConvNet layers.
//
// The GEMM K axis is the flattened window k = (kh*KW + kw)*Cin + ci, zero-padded to a multiple
// of 32.  The im2col tile [pixels][k] is built in LDS by an element-wise gather (2- or 4-byte loads;
// the window rows are only element-aligned in NHWC), everything after that is the MFMA machinery of
// conv_igemm / conv_wgrad:
//   forward : A = im2col tile [128 px][<=128 k], B = packed weights [64 co][Kpad]; ds_read_b128 frags
//   wgrad   : reduction over pixels; LDS images dy[px][co] and im2col[px][k] are read with the
//             transposing ds_read_b64_tr_b16 (bf16) / ds_read_b32 (fp32); each wave takes a quarter
//             of every 64(32)-pixel stage and writes its own fp32 partial slab (summed by the
//             deterministic slab reducer of conv_wgrad.hip).
// Requires Cout <= 64.  The data gradient of such layers uses the direct kernel (the first layer of
// a network needs none).
#include "common.h"

namespace {

constexpr int KC = 128;      // k elements per forward chunk
constexpr int KG = 160;      // k elements per wgrad workgroup (5 MFMA column blocks)
constexpr int NTMAX = KG / 32;

struct SmallCParams {
    const void* x;
    const void* wpk;
    void* y;
    const float* bias;
    const void* dy;
    float* slab;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, sh, sw, ph, pw;
    int K, Kpad, M;
    int ksteps, ksteps_per_split, splits;
    FastDiv fd_howo, fd_wo, fd_kwc, fd_c;
};

template <typename T> struct Mma32s;
template <> struct Mma32s<bf16> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Mma32s<float> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
    }
};

// k -> window offset tables for k in [k0, k0+n)
__device__ __forceinline__ void build_ktab(const SmallCParams& p, int k0, int n, int* koff, int* kkh,
                                           int* kkw) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int k = k0 + i;
        if (k < p.K) {
            uint32_t kh = fd_div(k, p.fd_kwc), rem = k - kh * p.fd_kwc.d;
            uint32_t kw = fd_div(rem, p.fd_c), ci = rem - kw * p.fd_c.d;
            koff[i] = (kh * p.W + kw) * p.Cin + ci;
            kkh[i] = kh;
            kkw[i] = kw;
        } else {
            koff[i] = 0;
            kkh[i] = 1 << 20;        // fails every bounds check -> zero padding of K
            kkw[i] = 0;
        }
    }
}

// pixel rows [m0, m0+n) -> (base element offset of the window origin, ih0, iw0)
__device__ __forceinline__ void build_rtab(const SmallCParams& p, int m0, int n, int* rbase, int* rih,
                                           int* riw) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int m = m0 + i;
        if (m < p.M) {
            uint32_t img = fd_div(m, p.fd_howo), rem = m - img * p.fd_howo.d;
            uint32_t ho = fd_div(rem, p.fd_wo), wo = rem - ho * p.fd_wo.d;
            int ih0 = (int)ho * p.sh - p.ph, iw0 = (int)wo * p.sw - p.pw;
            rbase[i] = ((int)(img * p.H) + ih0) * p.W * p.Cin + iw0 * p.Cin;
            rih[i] = ih0;
            riw[i] = iw0;
        } else {
            rbase[i] = 0;
            rih[i] = -(1 << 20);
            riw[i] = 0;
        }
    }
}

// dst[r][kk] = x at (row r, window element kk), rows strided by `stride` bytes
template <typename T>
__device__ __forceinline__ void gather_tile(const SmallCParams& p, const T* x, unsigned char* dst,
                                            int stride, int rows, int nk, const int* koff,
                                            const int* kkh, const int* kkw, const int* rbase,
                                            const int* rih, const int* riw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int kk = lane; kk < nk; kk += 64) {
        const int ko = koff[kk], kh = kkh[kk], kw = kkw[kk];
        for (int r = wave; r < rows; r += 4) {
            int ih = rih[r] + kh, iw = riw[r] + kw;
            T v = from_f32<T>(0.f);
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) v = x[rbase[r] + ko];
            *(T*)(dst + r * stride + kk * (int)sizeof(T)) = v;
        }
    }
}

__host__ __device__ constexpr int odd16(int bytes) { return ((bytes / 16) & 1) ? bytes : bytes + 16; }

template <typename T>
__global__ __launch_bounds__(256) void smallc_fwd_kernel(const SmallCParams p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int STR = odd16(KC * ES);          // row stride with an odd count of 16-B slots
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                    // [128][STR]
    unsigned char* Bs = As + 128 * STR;          // [64][STR]
    int* koff = (int*)(Bs + 64 * STR);
    int* kkh = koff + KC;
    int* kkw = kkh + KC;
    int* rbase = kkw + KC;
    int* rih = rbase + 128;
    int* riw = rih + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 128;
    const T* x = (const T*)p.x;
    const T* wpk = (const T*)p.wpk;
    build_rtab(p, m0, 128, rbase, rih, riw);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int k0 = 0; k0 < p.Kpad; k0 += KC) {
        const int kc = min(KC, p.Kpad - k0);
        build_ktab(p, k0, kc, koff, kkh, kkw);
        __syncthreads();
        gather_tile<T>(p, x, As, STR, 128, kc, koff, kkh, kkw, rbase, rih, riw);
        const int cpr = kc * ES / 16;
        for (int i = tid; i < 64 * cpr; i += 256) {
            int r = i / cpr, c = i - r * cpr;
            *(u32x4*)(Bs + r * STR + c * 16) = *(const u32x4*)(wpk + (size_t)r * p.Kpad + k0 + c * (16 / ES));
        }
        __syncthreads();
        for (int s = 0; s < kc * ES / 32; ++s) {
            u32x4 fa = *(const u32x4*)(As + (wave * 32 + l31) * STR + (2 * s + lh) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4 fb = *(const u32x4*)(Bs + (j * 32 + l31) * STR + (2 * s + lh) * 16);
                Mma32s<T>::run(acc[j], fa, fb);
            }
        }
        __syncthreads();
    }
    T* y = (T*)p.y;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = j * 32 + l31;
        if (col >= p.Cout) continue;
        float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < p.M) y[(size_t)m * p.Cout + col] = from_f32<T>(acc[j][r] + bv);
        }
    }
}

template <typename T> struct SwCfg;
template <> struct SwCfg<bf16> { static constexpr int KP = 64; };
template <> struct SwCfg<float> { static constexpr int KP = 32; };

template <typename T>
__global__ __launch_bounds__(256) void smallc_wgrad_kernel(const SmallCParams p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int KP = SwCfg<T>::KP;                         // pixels per stage
    constexpr int STRD = 64 * ES + (ES == 2 ? 64 : 0);       // dy image row: 64 co (+pad: == 64 mod 256)
    constexpr int STRX = KG * ES;                            // 320 B (== 64 mod 256) / 640 B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ds = smem;                                // [KP][STRD]
    unsigned char* Xs = Ds + KP * STRD;                      // [KP][STRX]
    int* koff = (int*)(Xs + KP * STRX);
    int* kkh = koff + KG;
    int* kkw = kkh + KG;
    int* rbase = kkw + KG;
    int* rih = rbase + 64;
    int* riw = rih + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, kg = blockIdx.y;
    const int k0 = kg * KG, nk = min(KG, p.Kpad - k0), nt = nk / 32;
    const T* x = (const T*)p.x;
    const T* dy = (const T*)p.dy;
    build_ktab(p, k0, nk, koff, kkh, kkw);
    f32x16 acc[2][NTMAX];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTMAX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int ks0 = split * p.ksteps_per_split, ks1 = min(p.ksteps, ks0 + p.ksteps_per_split);
    for (int ks = ks0; ks < ks1; ++ks) {
        const int m0 = ks * KP;
        __syncthreads();                                     // previous stage's reads are done
        build_rtab(p, m0, KP, rbase, rih, riw);
        // dy tile [KP][64 co], zero beyond Cout / M
        if (p.Cout == 64) {
            constexpr int CPR = 64 * ES / 16;
            for (int i = tid; i < KP * CPR; i += 256) {
                int r = i / CPR, c = i - r * CPR;
                u32x4 v = {0, 0, 0, 0};
                if (m0 + r < p.M) v = *(const u32x4*)(dy + (size_t)(m0 + r) * 64 + c * (16 / ES));
                *(u32x4*)(Ds + r * STRD + c * 16) = v;
            }
        } else {
            for (int i = tid; i < KP * 64; i += 256) {
                int r = i >> 6, c = i & 63;
                T v = from_f32<T>(0.f);
                if (m0 + r < p.M && c < p.Cout) v = dy[(size_t)(m0 + r) * p.Cout + c];
                *(T*)(Ds + r * STRD + c * ES) = v;
            }
        }
        __syncthreads();                                     // tables visible
        gather_tile<T>(p, x, Xs, STRX, KP, nk, koff, kkh, kkw, rbase, rih, riw);
        __syncthreads();
        if constexpr (ES == 2) {
            // wave w reduces pixels 16w .. 16w+15 of the stage: one 32x32x16 step per block pair
            const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g >> 1;
            const int prow = wave * 16 + 8 * h + q;
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            bf16x8 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned char* a0 = Ds + prow * STRD + (i * 32 + 16 * (g & 1) + 4 * pp) * 2;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * STRD));
                s16x8 w8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fa[i] = __builtin_bit_cast(bf16x8, w8);
            }
#pragma unroll
            for (int j = 0; j < NTMAX; ++j) {
                if (j < nt) {
                    const unsigned char* b0 = Xs + prow * STRX + (j * 32 + 16 * (g & 1) + 4 * pp) * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0 + 4 * STRX));
                    s16x8 w8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bf16x8 fb = __builtin_bit_cast(bf16x8, w8);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
                }
            }
        } else {
            // wave w reduces pixels 8w .. 8w+7: four 32x32x2 steps
            const int l31 = lane & 31, lk = lane >> 5;
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const int prow = wave * 8 + ss * 2 + lk;
                float fa[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *(const float*)(Ds + prow * STRD + (i * 32 + l31) * 4);
#pragma unroll
                for (int j = 0; j < NTMAX; ++j) {
                    if (j < nt) {
                        float fb = *(const float*)(Xs + prow * STRX + (j * 32 + l31) * 4);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    }
    // slab[(split*4 + wave)][co 64][Kpad]
    const int l31 = lane & 31, lh = lane >> 5;
    float* slab = p.slab + ((size_t)(split * 4 + wave) * 64) * p.Kpad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTMAX; ++j) {
            if (j < nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    slab[(size_t)co * p.Kpad + k0 + j * 32 + l31] = acc[i][j][r];
                }
            }
        }
}

// wpk[co][k] (T, zero padded to [64][Kpad]) = w[co][ci][kh][kw], k = (kh*KW + kw)*Cin + ci
// transpose: image of the TRANSPOSED, tap-flipped conv (the data gradient of a stride-1 conv as a
// forward conv over dy): row = ci of the master weight, k = (r*KW + q)*Cout_master + co, value
// w[co][ci][KH-1-r][KW-1-q]; Cout / Cin are then those of the transposed problem.
template <typename T>
__global__ void pack_smallc_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin,
                                   int KH, int KW, int K, int Kpad, int transpose) {
    int total = 64 * Kpad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int co = i / Kpad, k = i - co * Kpad;
        float v = 0.f;
        if (co < Cout && k < K) {
            int t = k / Cin, ci = k - t * Cin;
            if (transpose) v = w[((size_t)ci * Cout + co) * KH * KW + (KH * KW - 1 - t)];
            else v = w[((size_t)co * Cin + ci) * KH * KW + t];
        }
        out[i] = from_f32<T>(v);
    }
}

SmallCParams make_sc(const cnk_conv_desc* d) {
    SmallCParams p = {};
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout;
    p.KH = d->kh; p.KW = d->kw; p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
    p.K = d->kh * d->kw * d->Cin;
    p.Kpad = (p.K + 31) / 32 * 32;
    p.M = d->N * Ho * Wo;
    p.fd_howo = make_fastdiv(Ho * Wo);
    p.fd_wo = make_fastdiv(Wo);
    p.fd_kwc = make_fastdiv(d->kw * d->Cin);
    p.fd_c = make_fastdiv(d->Cin);
    return p;
}

void sc_wgrad_plan(const cnk_conv_desc* d, int* splits, int* ksteps, int* per_split, int* kgroups,
                   int* kpad) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    int M = d->N * Ho * Wo;
    int kp = d->dtype == CNK_BF16 ? 64 : 32;
    int K = d->kh * d->kw * d->Cin;
    *kpad = (K + 31) / 32 * 32;
    *kgroups = ceil_div(*kpad, KG);
    *ksteps = ceil_div(M, kp);
    int want = ceil_div(1024, *kgroups);
    int max_by_k = *ksteps / 8 > 0 ? *ksteps / 8 : 1;
    int sp = want < max_by_k ? want : max_by_k;
    *per_split = ceil_div(*ksteps, sp);
    *splits = ceil_div(*ksteps, *per_split);
}

}  // namespace

int launch_wgrad_reduce(const float* slab, float* dw, int splits, size_t slab_stride, int Cout,
                        int Cin, int ntaps, size_t s_co, size_t s_t, size_t s_ci, int inner_is_ci,
                        int accumulate, hipStream_t s);

// The element offsets of the gather are 32-bit: the input tensor must stay below 2^31 elements.
bool smallc_supported(const cnk_conv_desc* d) {
    if (d->Cout > 64) return false;
    if ((int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 31)) return false;
    int kstep = d->dtype == CNK_BF16 ? 64 : 32;
    if (d->Cin % kstep == 0 && d->Cout % 64 == 0) return false;     // the implicit-GEMM tiles fit: use them
    return true;
}

size_t smallc_packed_bytes(const cnk_conv_desc* d) {
    int K = d->kh * d->kw * d->Cin, Kpad = (K + 31) / 32 * 32;
    return (size_t)64 * Kpad * (d->dtype == CNK_BF16 ? 2 : 4);
}

static int smallc_pack_t(const cnk_conv_desc* d, const float* w, void* out, int transpose, hipStream_t s) {
    int K = d->kh * d->kw * d->Cin, Kpad = (K + 31) / 32 * 32;
    int grid = ceil_div(64 * Kpad, 256);
    if (d->dtype == CNK_BF16)
        hipLaunchKernelGGL(pack_smallc_kernel<bf16>, dim3(grid), dim3(256), 0, s, w, (bf16*)out, d->Cout, d->Cin, d->kh, d->kw, K, Kpad, transpose);
    else
        hipLaunchKernelGGL(pack_smallc_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, d->Cout, d->Cin, d->kh, d->kw, K, Kpad, transpose);
    CNK_CHECK_LAUNCH("pack_smallc");
    return 0;
}

int smallc_pack(const cnk_conv_desc* d, const float* w, void* out, hipStream_t s) { return smallc_pack_t(d, w, out, 0, s); }

// Data gradient of a STRIDE-1 conv = forward conv of dy with the transposed, tap-flipped weights and
// padding k-1-p.  When that transposed problem is a small-channel one (Cin <= 64 of the original, e.g.
// MobileNet's 32 -> 64 pointwise conv at 112x112) it runs on the MFMA forward kernel above instead of
// the direct kernel.  Returns false when the trick does not apply.
bool smallc_dgrad_desc(const cnk_conv_desc* d, cnk_conv_desc* t) {
    if (d->stride_h != 1 || d->stride_w != 1) return false;
    if (d->kh - 1 - d->pad_h < 0 || d->kw - 1 - d->pad_w < 0) return false;
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    *t = *d;
    t->H = Ho; t->W = Wo; t->Cin = d->Cout; t->Cout = d->Cin;
    t->pad_h = d->kh - 1 - d->pad_h; t->pad_w = d->kw - 1 - d->pad_w;
    return smallc_supported(t);
}

size_t smallc_dgrad_packed_bytes(const cnk_conv_desc* d) {
    cnk_conv_desc t;
    return smallc_dgrad_desc(d, &t) ? smallc_packed_bytes(&t) : 0;
}

int smallc_dgrad_pack(const cnk_conv_desc* d, const float* w, void* out, hipStream_t s) {
    cnk_conv_desc t;
    smallc_dgrad_desc(d, &t);
    return smallc_pack_t(&t, w, out, 1, s);
}

int smallc_dgrad(const cnk_conv_desc* d, const void* dy, const void* wpk, void* dx, hipStream_t s);

int smallc_fwd(const cnk_conv_desc* d, const void* x, const void* wpk, const float* bias, void* y,
               hipStream_t s) {
    SmallCParams p = make_sc(d);
    p.x = x; p.wpk = wpk; p.y = y; p.bias = bias;
    int grid = ceil_div(p.M, 128);
    if (d->dtype == CNK_BF16) {
        size_t lds = (size_t)192 * odd16(KC * 2) + (3 * KC + 3 * 128) * sizeof(int);
        hipLaunchKernelGGL(smallc_fwd_kernel<bf16>, dim3(grid), dim3(256), lds, s, p);
    } else {
        size_t lds = (size_t)192 * odd16(KC * 4) + (3 * KC + 3 * 128) * sizeof(int);
        hipLaunchKernelGGL(smallc_fwd_kernel<float>, dim3(grid), dim3(256), lds, s, p);
    }
    CNK_CHECK_LAUNCH("smallc_fwd");
    return 0;
}

int smallc_dgrad(const cnk_conv_desc* d, const void* dy, const void* wpk, void* dx, hipStream_t s) {
    cnk_conv_desc t;
    smallc_dgrad_desc(d, &t);
    return smallc_fwd(&t, dy, wpk, nullptr, dx, s);
}

size_t smallc_wgrad_workspace(const cnk_conv_desc* d) {
    int splits, ksteps, per, kgroups, kpad;
    sc_wgrad_plan(d, &splits, &ksteps, &per, &kgroups, &kpad);
    return (size_t)splits * 4 * 64 * kpad * sizeof(float);
}

int smallc_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t s) {
    SmallCParams p = make_sc(d);
    int kgroups, kpad;
    sc_wgrad_plan(d, &p.splits, &p.ksteps, &p.ksteps_per_split, &kgroups, &kpad);
    size_t need = (size_t)p.splits * 4 * 64 * kpad * sizeof(float);
    CNK_CHECK_ARG(ws && ws_bytes >= need, "conv2d_wgrad(smallc): workspace too small (%zu < %zu)", ws_bytes, need);
    p.x = x; p.dy = dy; p.slab = (float*)ws;
    dim3 grid(p.splits, kgroups);
    if (d->dtype == CNK_BF16) {
        size_t lds = (size_t)64 * (64 * 2 + 64) + (size_t)64 * KG * 2 + (3 * KG + 3 * 64) * sizeof(int);
        hipLaunchKernelGGL(smallc_wgrad_kernel<bf16>, grid, dim3(256), lds, s, p);
    } else {
        size_t lds = (size_t)32 * (64 * 4) + (size_t)32 * KG * 4 + (3 * KG + 3 * 64) * sizeof(int);
        hipLaunchKernelGGL(smallc_wgrad_kernel<float>, grid, dim3(256), lds, s, p);
    }
    CNK_CHECK_LAUNCH("smallc_wgrad");
    // slab order [split*4+wave][co(64)][Kpad], k = t*Cin + ci
    return launch_wgrad_reduce(p.slab, dw, p.splits * 4, (size_t)64 * kpad, d->Cout, d->Cin,
                               d->kh * d->kw, (size_t)kpad, (size_t)d->Cin, (size_t)1, 1, accumulate, s);
}
