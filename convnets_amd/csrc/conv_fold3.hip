// Backward of conv(1x1, stride 1) -> BatchNorm2d with the BN apply pass folded in (bnfold.hip), DATA GRADIENT AND WEIGHT GRADIENT IN
// ONE PASS over the operands -- the 64 -> 256-channel convs of the first ResNet stage (the last conv of a bottleneck and the
// projection shortcut at 56x56: /root/reference/src/resnet.py:172-176, 160-166; autograd's backward of both layers at
// /root/reference/src/basemodel.py:278).  Round 5.
//
// The two folded launches of round 4 each stream the block's widest tensor: the data gradient reads g ([M][256], the gradient
// w.r.t. the BN output) and a ([M][64], the conv input), the FOLD weight gradient reads g, the BN input y ([M][256]) and a --
// 925 MB at batch 256 for 26 GFLOP, on a part of the step where both streams together are HBM-bound.  Both are sums over the
// same rows:
//     da[p][j]  = sum_K [g | a][p][K] * wf[j][K] + c_j                                   (bnfold.hip: K = Cout + Cin)
//     dW[k][j]  = sum_p dy_conv[p][k] a[p][j],   dy_conv = A_k (g - B1_k - (y - mu_k) invstd_k B2_k),   y = W a
//               = A_k (G[k][j] - B1_k S[j])  -  E_k ((W X)[k][j] - mu_k S[j])
//     G = g^T a  [Cout][Cin],   X = a^T a  [Cin][Cin],   S[j] = sum_p a[p][j],   E_k = A_k invstd_k B2_k
// so y is not needed at all (it is W a, up to the bf16 rounding of the stored tensor), and one kernel that holds a tile of g and a in
// LDS can feed it to the matrix pipe three times: as the row operand of the data gradient, and transposed (ds_read_b64_tr_b16)
// as both operands of the two Gram products.  Traffic per launch: g + a + the BN-2 input of the fused reduction in, da out --
// what the folded data gradient alone moves today; the weight gradient's 925 MB (and its 13 MB of slabs) disappear.
//
// A gradient addend (the projection shortcut's data gradient joins the first conv's: no fused reduction there) takes the BN
// input's place in the stage.
//
// Structure (HBM-bound by a factor of 6 over its MFMA time: everything is arranged for bytes in flight, not for the matrix pipe):
//   * 256 threads, two workgroups per CU; a workgroup owns a contiguous run of 32-pixel steps.
//   * three LDS stages of (g 16 KiB | a 4 KiB | BN-2 input 4 KiB), filled by LDS-DMA two steps ahead, counted vmcnt (loads only:
//     DESIGN.md 7a), one raw s_barrier per step; every vector-memory operation inside the loop is inline asm.
//   * images are lane-linear (DMA) with a source-side XOR swizzle that serves BOTH read patterns without conflicts: plain
//     ds_read_b128 rows (data gradient: pixel = row) and transposing reads whose four rows are 4 or 8 apart (the k order of a
//     Gram step is a fixed permutation of its 16 pixels, the same for both operands).
//   * data gradient: wave w owns input channels 16w..16w+15 for all 32 pixels, v_mfma_f32_16x16x32_bf16 on the TRANSPOSED
//     product (weights as the row operand, stationary in 40 registers), so a lane ends up with four consecutive channels of one
//     pixel; per-wave staging, 16-byte row stores; bias, ReLU re-mask and the two sums of the producing BN layer's backward as in
//     igemm_epilogue (conv_igemm.hip).
//   * Gram products: wave w owns output channels 64w..64w+63 (2 x 2 blocks of 32x32x16) and one 32 x 32 block of X.
//   * per workgroup one fp32 slab [Cout + Cin + 1][Cin] (G | X | S), summed by launch_wgrad_reduce in fixed order; a last tiny
//     kernel forms dW from the sums.
#include "common.h"

int launch_wgrad_reduce(const float* slab, float* dw, int splits, size_t slab_stride, int Cout, int Cin, int ntaps, size_t s_co, size_t s_t,
                        size_t s_ci, int inner_is_ci, int accumulate, hipStream_t s);

namespace {

constexpr int F3_CIN = 64, F3_COUT = 256, F3_KF = F3_COUT + F3_CIN, F3_KP = 32, F3_NST = 3;
constexpr int F3_G = 0, F3_X = 16384, F3_Y = 20480, F3_STAGE = 24576;
constexpr int F3_CS = F3_NST * F3_STAGE;               // per-wave staging blocks: 32 rows x 48 B
constexpr int F3_CSW = 1536;
constexpr int F3_SA = F3_CS + 4 * F3_CSW;              // [4][64] floats: the waves' column-sum partials
constexpr int F3_LDS = F3_SA + 1024;
constexpr int F3_ROWS = F3_COUT + F3_CIN + 1;          // slab rows: G, X, S
constexpr int F3_PIECES = 6;                           // LDS-DMA pieces per wave and step

struct F3Params {
    const unsigned char* g;        // [M][256] bf16
    const unsigned char* x;        // [M][64] bf16 (the conv input a)
    const unsigned char* y2;       // [M][64] bf16: input of the BN layer that produced a (fused reduction), or nullptr
    const unsigned char* addend;   // [M][64] bf16: the gradient arriving at a through another branch, or nullptr (not with y2)
    const bf16* wf;                // [64][320]: bnfold_prepare's image
    const float* bias;             // [64]
    unsigned char* dx;             // [M][64] bf16
    const float* bnf_mean;
    const float* bnf_invstd;
    const float* bnf_gamma;
    const float* bnf_beta;
    float* bnf_partial;            // [nwg][64][2]
    int bnf_relu;
    float* slab;                   // [nwg][321][64]
    int M, ksteps, nwg;
};

__device__ __attribute__((aligned(16))) unsigned int f3_zero16[4] = {0, 0, 0, 0};
typedef __attribute__((address_space(3))) unsigned char lds_u8;

#define F3_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

__device__ __forceinline__ void f3_dma16(const void* g, uint32_t lds_addr) {
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

__device__ __forceinline__ bf16x8 f3_tr(const unsigned char* a0, const unsigned char* a1) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
    const s16x8 w8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, w8);
}

// byte offset of 16-byte chunk c of row r inside the g image (512-byte rows) / the a image (128-byte rows)
__device__ __forceinline__ int f3_goff(int r, int c) { return r * 512 + ((((c ^ r) & 15) | (c & 16)) << 4); }
__device__ __forceinline__ int f3_xoff(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256, 2) void fold3_kernel(const F3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks0 = (int)((long long)blockIdx.x * p.ksteps / p.nwg), ks1 = (int)((long long)(blockIdx.x + 1) * p.ksteps / p.nwg);
    const bool bnf = p.y2 != nullptr, has_add = p.addend != nullptr;
    const unsigned char* side_src = bnf ? p.y2 : p.addend;          // what the third image of a stage holds

    // ---- per-lane constants
    // data gradient: this wave's 16 x 320 slice of the image, the row operand of every step
    u32x4 wreg[10];
    {
        const bf16* wp = p.wf + (size_t)(16 * wave + (lane & 15)) * F3_KF + 8 * (lane >> 4);
#pragma unroll
        for (int s = 0; s < 10; ++s) wreg[s] = *(const u32x4*)(wp + 32 * s);
    }
    float bias4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = p.bias[16 * wave + 4 * (lane >> 4) + r];
    // epilogue: this lane's 16-byte chunk of a row = channels 8 * chunk .. + 7
    const int erow = lane >> 1, echunk = 2 * wave + (lane & 1);
    float bmu[8], bsc[8], bsf[8], s8[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s8[e] = q8[e] = 0.f;
        bmu[e] = bsc[e] = bsf[e] = 0.f;
    }
    if (bnf) {
        float mu[8], gm[8], is[8], bt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * echunk + e;
            mu[e] = p.bnf_mean[c]; gm[e] = p.bnf_gamma[c]; is[e] = p.bnf_invstd[c]; bt[e] = p.bnf_beta[c];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = gm[e] * is[e];
            bmu[e] = mu[e];
            bsc[e] = sc;
            bsf[e] = bt[e] - mu[e] * sc;
        }
    }
    // LDS-DMA sources: wave w fills g pieces 4w..4w+3 (two rows each), piece w of a and of the BN input (eight rows each)
    int goff[4], grow[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = 2 * (4 * wave + u) + (lane >> 5), cl = lane & 31;
        const int c = (cl & 16) | ((cl ^ r) & 15);
        grow[u] = r;
        goff[u] = r * 512 + c * 16;
    }
    const int xrow = 8 * wave + (lane >> 3);
    const int xoff = xrow * 128 + (((lane & 7) ^ ((xrow >> 1) & 7)) << 4);
    const int yoff = xrow * 128 + ((lane & 7) << 4);
    const unsigned char* zsrc = (const unsigned char*)f3_zero16;
    auto issue = [&](int ks, int buf) {
        const bool live = ks < ks1;
        const int m0 = ks * F3_KP;
        const uint32_t st = lds0 + (uint32_t)buf * F3_STAGE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = live && m0 + grow[u] < p.M;
            f3_dma16(ok ? p.g + (size_t)m0 * 512 + goff[u] : zsrc, st + F3_G + (4 * wave + u) * 1024);
        }
        const bool okx = live && m0 + xrow < p.M;
        f3_dma16(okx ? p.x + (size_t)m0 * 128 + xoff : zsrc, st + F3_X + wave * 1024);
        f3_dma16(okx && side_src ? side_src + (size_t)m0 * 128 + yoff : zsrc, st + F3_Y + wave * 1024);
    };

    // Gram fragments: lane l supplies the address of row q of a 4 x 16 block (ds_read_b64_tr_b16); k slot (h, read, q) of a
    // 16-pixel step is pixel ((q >> 1) << 3) ^ ((q & 1) * 5) ^ (h << 1) ^ read -- four rows that differ in bits (0, 2, 3), which
    // is what keeps both swizzled images conflict-free
    const int g4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g4 >> 1, ch = g4 & 1;
    const int rq = (((q >> 1) << 3) ^ ((q & 1) * 5) ^ (h << 1));
    int ga[2][2][2], xa[2][2][2];              // [block][k16 step][read]
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int r = 16 * ss + (rq ^ rd);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int byte = (64 * wave + 32 * i + 16 * ch + 4 * pp) * 2;
                ga[i][ss][rd] = F3_G + f3_goff(r, byte >> 4) + (byte & 15);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int byte = (32 * j + 16 * ch + 4 * pp) * 2;
                xa[j][ss][rd] = F3_X + f3_xoff(r, byte >> 4) + (byte & 15);
            }
        }
    // data-gradient fragments: lane (pixel n = l & 15 of pixel block nt, k group l >> 4)
    int da[2][10];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int r = 16 * nt + (lane & 15);
#pragma unroll
        for (int s = 0; s < 8; ++s) da[nt][s] = F3_G + f3_goff(r, 4 * s + (lane >> 4));
#pragma unroll
        for (int s = 8; s < 10; ++s) da[nt][s] = F3_X + f3_xoff(r, 4 * (s - 8) + (lane >> 4));
    }

    f32x16 acc[2][2], xx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xx[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][r] = 0.f;
    }
    float sa = 0.f;
    unsigned char* cs = smem + F3_CS + wave * F3_CSW;

    // everything the compiler loads is in registers before the first DMA: no compiler-counted vmcnt inside the loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 10; ++s) asm volatile("" : "+v"(wreg[s]));
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bias4[r]));
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bmu[e]), "+v"(bsc[e]), "+v"(bsf[e]));
    issue(ks0, 0);
    issue(ks0 + 1, 1);

    int buf = 0;
    for (int ks = ks0; ks < ks1; ++ks) {
        F3_WAIT_VM(F3_PIECES);                  // this wave's pieces of step ks (the six of step ks + 1 stay in flight)
        __builtin_amdgcn_s_barrier();           // ... and everyone's; every wave has finished reading the stage of step ks - 1
        issue(ks + 2, buf == 0 ? 2 : buf - 1);
        const unsigned char* st = smem + buf * F3_STAGE;

        // ---- Gram products
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = f3_tr(st + ga[i][ss][0], st + ga[i][ss][1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = f3_tr(st + xa[j][ss][0], st + xa[j][ss][1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            // (wave-uniform selects, not indexed registers)
            const bf16x8 fxa = (wave >> 1) ? fb[1] : fb[0], fxb = (wave & 1) ? fb[1] : fb[0];
            xx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fxa, fxb, xx, 0, 0, 0);
            if (ss == (wave >> 1)) {
                const bf16x8 f = fxb;
                sa += (((float)f[0] + (float)f[1]) + ((float)f[2] + (float)f[3])) + (((float)f[4] + (float)f[5]) + ((float)f[6] + (float)f[7]));
            }
        }

        // ---- data gradient of the step's 32 pixels (transposed product: rows = this wave's 16 input channels)
        f32x4 d[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            d[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                const u32x4 b = *(const u32x4*)(st + da[nt][s]);
                d[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wreg[s]), __builtin_bit_cast(bf16x8, b), d[nt], 0, 0, 0);
            }
        }
        // lane (pixel l & 15 of block nt) holds channels 16 w + 4 (l >> 4) + {0..3}: stage as bf16 rows, re-read as 16-byte chunks
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const bf16x4 o = {(bf16)(d[nt][0] + bias4[0]), (bf16)(d[nt][1] + bias4[1]), (bf16)(d[nt][2] + bias4[2]), (bf16)(d[nt][3] + bias4[3])};
            *(bf16x4*)(cs + (16 * nt + (lane & 15)) * 48 + (lane >> 4) * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8 v = *(const bf16x8*)(cs + erow * 48 + (lane & 1) * 16);
        const int m = ks * F3_KP + erow;
        if (bnf) {
            const bf16x8 yv = *(const bf16x8*)(st + F3_Y + erow * 128 + echunk * 16);
            const bool rowok = m < p.M;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float yf = (float)yv[e];
                const bool keep = rowok && (!p.bnf_relu || yf * bsc[e] + bsf[e] > 0.f);
                const float dv = keep ? (float)v[e] : 0.f;
                v[e] = (bf16)dv;
                s8[e] += dv;
                q8[e] += dv * (yf - bmu[e]);
            }
        }
        if (has_add) {
            // same rounding as a separate add kernel: round(round(conv) + addend), as igemm_epilogue
            const bf16x8 av = *(const bf16x8*)(st + F3_Y + erow * 128 + echunk * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)av[e]);
        }
        if (m < p.M) {
            const unsigned char* dst = p.dx + (size_t)m * 128 + echunk * 16;
            const u32x4 vv = __builtin_bit_cast(u32x4, v);
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(vv) : "memory");
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
    F3_WAIT_VM(0);                              // trailing dummy pieces and the last stores
    __builtin_amdgcn_s_barrier();

    // ---- slab: G rows (this wave's 64 output channels), one block of X, the column sums
    float* slab = p.slab + (size_t)blockIdx.x * (F3_ROWS * F3_CIN);
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 64 * wave + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[co * F3_CIN + 32 * j + l31] = acc[i][j][r];
            }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 32 * (wave >> 1) + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[(F3_COUT + i) * F3_CIN + 32 * (wave & 1) + l31] = xx[r];
    }
    float* sap = (float*)(smem + F3_SA);
    sap[wave * 64 + lane] = sa;
    __syncthreads();
    if (tid < F3_CIN) {
        // channel tid = block tid >> 5: waves (tid >> 5) and 2 + (tid >> 5) (the two k16 steps), both lane halves
        const int jb = tid >> 5, c = tid & 31;
        slab[(F3_COUT + F3_CIN) * F3_CIN + tid] = (sap[jb * 64 + c] + sap[jb * 64 + 32 + c]) + (sap[(2 + jb) * 64 + c] + sap[(2 + jb) * 64 + 32 + c]);
    }
    // ---- the two sums of the fused BN backward, one record per workgroup
    if (bnf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int mk = 2; mk < 64; mk <<= 1) {
                s8[e] += __shfl_xor(s8[e], mk, 64);
                q8[e] += __shfl_xor(q8[e], mk, 64);
            }
        }
        if (lane < 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 8 * echunk + e;
                float* o = p.bnf_partial + ((size_t)blockIdx.x * F3_CIN + c) * 2;
                o[0] = s8[e];
                o[1] = q8[e] * p.bnf_invstd[c];
            }
        }
    }
}

// dW[k][j] (+)= A_k (G[k][j] - B1_k S[j]) - E_k (sum_i bf16(W[k][i]) X[i][j] - mu_k S[j])
__global__ __launch_bounds__(256) void fold3_combine_kernel(const float* __restrict__ sums, const float* __restrict__ w, const float* __restrict__ coef,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dw,
                                                            int accumulate) {
    __shared__ float xs[F3_CIN][F3_CIN + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < F3_CIN * F3_CIN; e += 256) xs[e / F3_CIN][e % F3_CIN] = sums[F3_COUT * F3_CIN + e];
    __syncthreads();
    const int k = blockIdx.x * 4 + (tid >> 6), j = tid & 63;
    const float S = sums[(F3_COUT + F3_CIN) * F3_CIN + j];
    float wx = 0.f;
#pragma unroll 8
    for (int i = 0; i < F3_CIN; ++i) wx += (float)(bf16)w[k * F3_CIN + i] * xs[i][j];
    const float A = coef[k], B1 = coef[F3_COUT + k], B2 = coef[2 * F3_COUT + k];
    const float E = A * invstd[k] * B2;
    const float v = A * (sums[k * F3_CIN + j] - B1 * S) - E * (wx - mean[k] * S);
    float* o = dw + k * F3_CIN + j;
    *o = accumulate ? *o + v : v;
}

}  // namespace

bool fold3_supported(const cnk_conv_desc* d) {
    return d->dtype == CNK_BF16 && d->kh == 1 && d->kw == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 &&
           d->Cin == F3_CIN && d->Cout == F3_COUT && d->groups <= 1 && d->dilation <= 1 && (int64_t)d->N * d->H * d->W >= 4096 &&
           (int64_t)d->N * d->H * d->W * F3_COUT * 2 < ((int64_t)1 << 31);
}

int fold3_workgroups(const cnk_conv_desc* d) {
    const int ksteps = (int)(((int64_t)d->N * d->H * d->W + F3_KP - 1) / F3_KP);
    int n = cnk_sw().fold3_nwg;
    if (n < 1) n = 1;
    if (n > 2048) n = 2048;
    return ksteps < n ? ksteps : n;
}

size_t fold3_workspace(const cnk_conv_desc* d) { return ((size_t)fold3_workgroups(d) + 1) * F3_ROWS * F3_CIN * sizeof(float); }

int fold3_bwd(const cnk_conv_desc* d, const void* g, const void* x, const void* w_fold, const float* bias_fold, void* dx, const void* addend,
              const cnk_bn_bwd_fuse* bn, const float* w, const float* mean, const float* invstd, const float* coef, float* dw, int accumulate,
              void* ws, size_t ws_bytes, hipStream_t s) {
    CNK_CHECK_ARG(fold3_supported(d), "fold3_bwd: a dense bf16 1x1 / stride-1 conv of 64 -> 256 channels");
    CNK_CHECK_ARG(!bn || !bn->relu_mask, "fold3_bwd: the fused BN backward re-derives its ReLU mask (no packed mask)");
    CNK_CHECK_ARG(!bn || !addend, "fold3_bwd: a gradient addend OR the fused BN backward, not both");
    CNK_CHECK_ARG(ws && ws_bytes >= fold3_workspace(d), "fold3_bwd: workspace too small (%zu < %zu)", ws_bytes, fold3_workspace(d));
    F3Params p = {};
    p.g = (const unsigned char*)g; p.x = (const unsigned char*)x; p.wf = (const bf16*)w_fold; p.bias = bias_fold;
    p.dx = (unsigned char*)dx;
    p.addend = (const unsigned char*)addend;
    if (bn) {
        p.y2 = (const unsigned char*)bn->x; p.bnf_mean = bn->mean; p.bnf_invstd = bn->invstd; p.bnf_gamma = bn->gamma; p.bnf_beta = bn->beta;
        p.bnf_partial = bn->partial; p.bnf_relu = bn->relu;
    }
    p.M = d->N * d->H * d->W;
    p.ksteps = (p.M + F3_KP - 1) / F3_KP;
    p.nwg = fold3_workgroups(d);
    p.slab = (float*)ws;
    float* sums = p.slab + (size_t)p.nwg * F3_ROWS * F3_CIN;
    {
        ProfScope kscope(CNK_KF_GATHER, s, 2.0 * p.M * (double)F3_KF * F3_CIN * 2.0,
                         (double)p.M * (F3_COUT + 2 * F3_CIN + (bn || addend ? F3_CIN : 0)) * 2.0);
        CNK_LDS_OPTIN(fold3_kernel, F3_LDS, "fold3");
        hipLaunchKernelGGL(fold3_kernel, dim3(p.nwg), dim3(256), F3_LDS, s, p);
        CNK_CHECK_LAUNCH("fold3");
    }
    if (int rc = launch_wgrad_reduce(p.slab, sums, p.nwg, (size_t)F3_ROWS * F3_CIN, F3_ROWS, F3_CIN, 1, F3_CIN, F3_CIN, 1, 1, 0, s)) return rc;
    hipLaunchKernelGGL(fold3_combine_kernel, dim3(F3_COUT / 4), dim3(256), 0, s, sums, w, coef, mean, invstd, dw, accumulate);
    CNK_CHECK_LAUNCH("fold3_combine");
    return 0;
}
