// Backward of conv(1x1, stride 1, no bias) -> BatchNorm2d with the BN apply pass folded into the conv's own backward GEMMs
// (reference sites: the last conv block of a bottleneck, /root/reference/src/resnet.py:172-176, built by conv2d_block,
// /root/reference/src/basemodel.py:999-1015; autograd runs both backward passes at basemodel.py:278).
//
// With y = W a (a: conv input [M][Cin], W: [Cout][Cin]) and the BN backward
//     dy_conv[p][k] = A_k * (g[p][k] - B1_k - (y[p][k] - mu_k) * invstd_k * B2_k)          (g: gradient w.r.t. the BN output, masked)
// the data gradient is LINEAR in g and a:
//     da[p][j] = sum_k dy_conv[p][k] W[k][j]
//              = sum_k g[p][k] * (A_k W[k][j])  -  sum_i a[p][i] * M[i][j]  +  c_j
//     E_k = A_k invstd_k B2_k,   M = W^T diag(E) W  [Cin][Cin],   c_j = sum_k W[k][j] * (E_k mu_k - A_k B1_k)
// i.e. ONE GEMM over the concatenated reduction [g | a] (K = Cout + Cin) with a per-step weight image and a bias.  The tensor
// dy_conv ([M][Cout], the widest of the block) is never written or read: per bottleneck that removes one 3 * M * Cout * 2 byte
// streaming pass (bn_bwd_apply_kernel) and turns the conv data gradient's M * Cout * 2 bytes of reads into M * (Cout + Cin) * 2.
// The weight gradient forms dy_conv on the fly while it stages its tile (wgrad_mfma_kernel<.., FOLD>, conv_wgrad.hip).
//
// (A second form for convs that NARROW -- the BN input y itself as second K segment, K = 2 Cout, no M matrix -- and the fold of
// strided 1x1 convs were built and measured in round 4: slower / neutral in the ResNet-50 step; removed in round 5.)
//
// This file: the per-step weight image.  One launch, three block roles:
//   * W' tiles:  wf[j][k] = bf16(A_k * W[k][j])                      from the fp32 master weights (transposed through LDS)
//   * M tiles :  wf[j][Cout + i] = bf16(-M[i][j])                     on the bf16 MFMA from the packed dgrad image wd[j][k]
//                (M is a correction of relative size |B2|; it is rounded to bf16 for the GEMM anyway)
//   * bias    :  c_j in fp32 from the master weights (fixed summation order)
#include "common.h"

namespace {

struct FoldPrepParams {
    const float* w;        // [Cout][Cin] fp32 master weights
    const bf16* wd;        // [Cin][Cout] bf16 data-gradient image (K = Cout contiguous)
    const float* coef;     // [3][Cout]: A, B1, B2 (bn_bwd_coef)
    const float* mean;
    const float* invstd;
    bf16* wf;              // [Cin][Cout + Cin]
    float* biasf;          // [Cin]
    int Cout, Cin;
    int nb_w, nb_m;        // blocks of the first two roles (the rest: bias)
};

constexpr int FP_MAXC = 4096;          // Cout bound: one fp32 vector of it lives in LDS

__global__ __launch_bounds__(256) void bnfold_prep_kernel(const FoldPrepParams p) {
    __shared__ float vec[FP_MAXC];                 // E_k (M role) or the bias role's per-k factor
    __shared__ float tile[4][32][33];              // W' transpose tile / the four waves' M partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KF = p.Cout + p.Cin;
    const float* cA = p.coef;
    const float* cB1 = p.coef + p.Cout;
    const float* cB2 = p.coef + 2 * (size_t)p.Cout;
    int b = blockIdx.x;
    if (b < p.nb_w) {
        // ---- W' tile: 32 output channels k x 32 input channels j
        const int tj = p.Cin / 32;
        const int k0 = (b / tj) * 32, j0 = (b % tj) * 32;
        const int c = tid & 31, r0 = tid >> 5;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 8 * q, k = k0 + r;
            tile[0][r][c] = cA[k] * p.w[(size_t)k * p.Cin + j0 + c];
        }
        __syncthreads();
        const int j = tid >> 3, k4 = (tid & 7) * 4;
        bf16x4 o = {(bf16)tile[0][k4][j], (bf16)tile[0][k4 + 1][j], (bf16)tile[0][k4 + 2][j], (bf16)tile[0][k4 + 3][j]};
        *(bf16x4*)(p.wf + (size_t)(j0 + j) * KF + k0 + k4) = o;
        return;
    }
    b -= p.nb_w;
    if (b < p.nb_m) {
        // ---- M tile: rows i0.., columns j0..; wave w takes the 16-channel k steps w, w + 4, ...
        for (int k = tid; k < p.Cout; k += 256) vec[k] = cA[k] * cB2[k] * p.invstd[k];
        __syncthreads();
        const int ti = p.Cin / 32;
        const int i0 = (b / ti) * 32, j0 = (b % ti) * 32;
        const int l31 = lane & 31, lh = lane >> 5;
        const bf16* arow = p.wd + (size_t)(i0 + l31) * p.Cout + 8 * lh;
        const bf16* brow = p.wd + (size_t)(j0 + l31) * p.Cout + 8 * lh;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // Cout / 64 steps per wave (Cout is a multiple of 64): loads of up to eight steps are requested together -- the loop is
        // latency-bound (a 32 x 32 tile: 0.8 us of MFMAs behind 32 dependent L2 round trips when it ran one step at a time)
        const int nsteps = p.Cout / 64;
        for (int s0 = 0; s0 < nsteps; s0 += 8) {
            bf16x8 av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k0 = ((s0 + u < nsteps ? s0 + u : nsteps - 1) * 4 + wave) * 16;
                av[u] = *(const bf16x8*)(arow + k0);
                bv[u] = *(const bf16x8*)(brow + k0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (s0 + u < nsteps) {
                    const int k0 = ((s0 + u) * 4 + wave) * 16;
                    bf16x8 bs;
#pragma unroll
                    for (int e = 0; e < 8; ++e) bs[e] = (bf16)((float)bv[u][e] * vec[k0 + 8 * lh + e]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[u], bs, acc, 0, 0, 0);
                }
            }
        }
        // lane (l31, lh), register r: D[i = (r & 3) + 8 (r >> 2) + 4 lh][j = l31]
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][l31] = acc[r];
        __syncthreads();
        const int j = tid >> 3, i4 = (tid & 7) * 4;
        float m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = (tile[0][i4 + e][j] + tile[1][i4 + e][j]) + (tile[2][i4 + e][j] + tile[3][i4 + e][j]);
        bf16x4 o = {(bf16)(-m[0]), (bf16)(-m[1]), (bf16)(-m[2]), (bf16)(-m[3])};
        *(bf16x4*)(p.wf + (size_t)(j0 + j) * KF + p.Cout + i0 + i4) = o;
        return;
    }
    b -= p.nb_m;
    {
        // ---- bias: 8 input channels j per block, 32 k phases, eight loads in flight per thread
        for (int k = tid; k < p.Cout; k += 256) {
            const float a = cA[k];
            vec[k] = a * cB2[k] * p.invstd[k] * p.mean[k] - a * cB1[k];
        }
        __syncthreads();
        const int j0 = b * 8, j = tid & 7, ph = tid >> 3;
        const float* wp = p.w + j0 + j;
        float a0 = 0.f, a1 = 0.f;
        const int nk = p.Cout / 32;                        // k = ph + 32 q, q < nk
        for (int q0 = 0; q0 < nk; q0 += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = wp[(size_t)(ph + 32 * (q0 + u < nk ? q0 + u : nk - 1)) * p.Cin];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                if (q0 + u < nk) a0 += wv[u] * vec[ph + 32 * (q0 + u)];
                if (q0 + u + 1 < nk) a1 += wv[u + 1] * vec[ph + 32 * (q0 + u + 1)];
            }
        }
        tile[0][ph][j] = a0 + a1;
        __syncthreads();
        if (tid < 8) {
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q) sacc += tile[0][q][tid];
            p.biasf[j0 + tid] = sacc;
        }
    }
}

}  // namespace

bool bnfold_supported(const cnk_conv_desc* d) {
    return d->dtype == CNK_BF16 && conv_groups(d) == 1 && conv_dil(d) == 1 && d->kh == 1 && d->kw == 1 && d->stride_h == 1 &&
           d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 && d->Cin % 64 == 0 &&
           d->Cout % 64 == 0 && d->Cout <= FP_MAXC;
}

size_t bnfold_image_bytes(const cnk_conv_desc* d) {
    return (size_t)d->Cin * (d->Cout + d->Cin) * sizeof(bf16);
}

int bnfold_prepare(const cnk_conv_desc* d, const float* w, const void* w_dgrad, const float* mean, const float* invstd,
                   const float* coef, void* w_fold, float* bias_fold, hipStream_t s) {
    FoldPrepParams p = {};
    p.w = w; p.wd = (const bf16*)w_dgrad; p.coef = coef; p.mean = mean; p.invstd = invstd;
    p.wf = (bf16*)w_fold; p.biasf = bias_fold; p.Cout = d->Cout; p.Cin = d->Cin;
    p.nb_w = (d->Cout / 32) * (d->Cin / 32);
    p.nb_m = (d->Cin / 32) * (d->Cin / 32);
    const int grid = p.nb_w + p.nb_m + d->Cin / 8;
    hipLaunchKernelGGL(bnfold_prep_kernel, dim3(grid), dim3(256), 0, s, p);
    CNK_CHECK_LAUNCH("bnfold_prep");
    return 0;
}
