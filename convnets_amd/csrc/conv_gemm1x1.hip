// Dense 1x1 / stride-1 convolution as a plain GEMM with an LDS-DMA ring, bf16:  C[m][n] = sum_k A[m][k] * B[n][k]
// (A = NHWC activations flattened to [M][K], B = the packed weight image [Nout][K]).  Sites served: the 1x1 convs of the
// bottleneck blocks (/root/reference/src/resnet.py:165-178) on the 28x28 / 14x14 / 7x7 maps: forward (bias, BN statistics
// partials) and data gradient (gradient addend + mask, fused BN-backward reduction), wherever gemm1x1_pick measured it ahead.
//
// Why a second kernel next to igemm_gather_kernel (conv_igemm.hip): ablated builds of that kernel on these layers (DESIGN.md
// section 9: full / loads only / MFMAs only) show that its single-buffer loop adds load time and MFMA time up, and that the
// loads already run at the rate the L2 -> LDS path delivers (22 TB/s) because every 128 x 128 tile re-reads its A rows once
// per column tile and its B rows once per row tile.  Here a 512-thread workgroup owns a 256 x 128 or 256 x 256 tile (a quarter
// / half fewer L2 bytes per flop) and walks K in steps of 32 channels through a ring of LDS stages filled by LDS-DMA NST - 1
// steps ahead, with one raw barrier per step and completions counted by s_waitcnt vmcnt(N):
//
//   step k:  wait for this wave's DMAs of stage k (those of the NST - 2 younger stages stay in flight) | s_barrier |
//            issue stage k + NST - 1 into the slot read in step k-1 | the stage's ds_read_b128s | its MFMAs
//
//   256 x 128: three stages = 72 KiB, 101 VGPRs  -> TWO workgroups per CU, whose barrier stalls and MFMA bursts interleave
//   256 x 256: four stages = 128 KiB             -> one workgroup per CU, for long K with few tiles
//
// LDS rows hold 32 channels (64 B) of one pixel / output channel; the 16-byte piece index is XOR-ed with (row >> 2) & 3 on
// the DMA SOURCE address (the LDS image of a DMA is lane-linear), which makes the fragment reads conflict-free.
// Every vector-memory operation of the loop is inline asm so that hipcc tracks none of them (cdna_hip_programming.md 5.7).
// Epilogue: a wave's 64 x (64 | 128) block is staged 32 rows at a time as bf16 in the (now idle) ring and written with
// 16-byte row stores; the per-channel sums (BN statistics of the ROUNDED outputs, one (n, mean, M2) record per 256-row tile;
// or the two sums of the fused BN backward) ride on that copy-out, as in igemm_epilogue.  Outputs are bit-identical to
// igemm_gather_kernel's (same MFMA order along K).
#include <cstdlib>

#include "common.h"

#ifndef G1_ABL
#define G1_ABL 0      // debug builds (tools/ablate_g1.sh): 1 no MFMAs, 2 no DMAs, 4 no fragment reads
#endif

namespace {

constexpr int G1_BM = 256, G1_KC = 32;
constexpr int G1_ROWB = G1_KC * 2;                                  // LDS row bytes
constexpr int G1_ASTAGE = G1_BM * G1_ROWB;
// NJ = 32-wide column blocks per wave: workgroup tile 256 x (64 NJ), wave tile 64 x (32 NJ)
template <int NJ> struct G1Cfg {
    static constexpr int BN = 64 * NJ;
    static constexpr int BSTAGE = BN * G1_ROWB, STAGE = G1_ASTAGE + BSTAGE;
    static constexpr int NST = NJ == 2 ? 3 : 4;                     // ring stages
    static constexpr int LDS = NST * STAGE;                         // 72 KiB (NJ = 2: two workgroups per CU) / 128 KiB (NJ = 4)
    static constexpr int SS = 32 * NJ * 2 + 16;                     // staged row: the wave's 32 NJ channels + pad
    static constexpr int BPW = BN / 16 / 8;                         // B pieces (1 KiB = 16 rows) per wave and stage
    static constexpr int DMA_PER_WAVE = 2 + BPW;
    static constexpr int CPR = 32 * NJ * 2 / 16;                    // 16-byte chunks per staged row
};

__device__ __attribute__((aligned(16))) unsigned int g1_zero16[4] = {0, 0, 0, 0};
typedef __attribute__((address_space(3))) unsigned char g1_lds_u8;

struct G1Params {
    const bf16* a;          // [M][K]
    const bf16* a2;         // optional second K segment [M][K2] (the loop then walks K + K2; nullptr: none)
    const bf16* b;          // [Nout][K + K2]
    bf16* c;                // [M][Nout]
    const float* bias;      // [Nout] or nullptr
    float* stats;           // [tiles_m][Nout][3] or nullptr
    // data-gradient epilogue (MODE 2), as igemm_epilogue: gradient addend (+ packed ReLU mask) and the first pass of the
    // backward of the BN(+ReLU) layer that produced the tensor this gradient belongs to
    const bf16* addend;
    const unsigned char* addend_mask;
    const bf16* bnf_x;
    const float* bnf_mean;
    const float* bnf_invstd;
    const float* bnf_gamma;
    const float* bnf_beta;
    const unsigned char* bnf_mask;
    float* bnf_partial;     // [bnf_tile0 + tiles_m][Nout][2]
    int bnf_relu, bnf_tile0;
    int M, K, K2, Nout, tiles_m, tiles_n, nk;
};

__device__ __forceinline__ void g1_dma16(const void* g, uint32_t lds_addr) {        // M0 saved / restored inside
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
#define G1_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

// MODE: 0 plain, 1 BN statistics partials of the output, 2 data-gradient side operands.  (The second launch-bounds argument is
// waves per SIMD: 4 = two 256 x 128 workgroups per CU; MODE 2 needs 151 registers -- squeezed into 128 it spilled and its
// fused data gradients got 25-50 % slower, so it keeps one workgroup per CU.)
template <int NJ, int MODE>
__global__ __launch_bounds__(512, (NJ == 2 && MODE < 2) ? 4 : 2) void gemm1x1_kernel(const G1Params p) {
    constexpr bool STATS = MODE == 1, FULL = MODE == 2;
    using CFG = G1Cfg<NJ>;
    constexpr int BN = CFG::BN, STAGE = CFG::STAGE, SS = CFG::SS, BPW = CFG::BPW, CPR = CFG::CPR, NST = CFG::NST, DPW = CFG::DMA_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(g1_lds_u8*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                        // 4 x 2 waves, 64 x (32 NJ) each
    // XCD-aware order: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of tiles (n fastest), so the column
    // tiles of one row tile -- same A rows -- meet in one L2
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int m0 = tile_m * G1_BM, n0 = tile_n * BN;

    // ---- DMA roles: a stage is pieces of 1 KiB (16 rows x 64 B): 16 A pieces, BN / 16 B pieces; wave w moves A pieces w and
    // w + 8 and B pieces w (+ 8).  Lane -> (row in piece, physical 16-byte slot); the logical slot is XOR-swizzled.
    const int prow = lane >> 2, pslot = lane & 3;
    const unsigned char* asrc[2];
    bool alive[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = (wave + 8 * h) * 16 + prow;
        const int logical = pslot ^ ((row >> 2) & 3);
        alive[h] = m0 + row < p.M;
        asrc[h] = alive[h] ? (const unsigned char*)(p.a + (size_t)(m0 + row) * p.K + logical * 8) : (const unsigned char*)g1_zero16;
    }
    // second K segment (p.a2): the A rows continue in another matrix after nk1 steps; the B rows are K + K2 long
    const int nk1 = p.K / G1_KC;
    const unsigned char* asrc2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = (wave + 8 * h) * 16 + prow;
        const int logical = pslot ^ ((row >> 2) & 3);
        asrc2[h] = (alive[h] && p.a2) ? (const unsigned char*)(p.a2 + (size_t)(m0 + row) * p.K2 + logical * 8) - (size_t)nk1 * (G1_KC * 2)
                                      : (const unsigned char*)g1_zero16;
    }
    const unsigned char* bsrc[BPW];
#pragma unroll
    for (int h = 0; h < BPW; ++h) {
        const int row = (wave + 8 * h) * 16 + prow;
        const int logical = pslot ^ ((row >> 2) & 3);
        bsrc[h] = (const unsigned char*)(p.b + (size_t)(n0 + row) * (p.K + p.K2) + logical * 8);
    }
    auto issue = [&](int k) {
        const uint32_t sb = lds0 + (uint32_t)(k % NST) * STAGE;
        const size_t ko = (size_t)k * (G1_KC * 2);
        if (G1_ABL & 2) return;
        const bool seg2 = k >= nk1;                              // (uniform)
        g1_dma16(alive[0] ? (seg2 ? asrc2[0] : asrc[0]) + ko : asrc[0], sb + (uint32_t)wave * 1024);
        g1_dma16(alive[1] ? (seg2 ? asrc2[1] : asrc[1]) + ko : asrc[1], sb + (uint32_t)(wave + 8) * 1024);
#pragma unroll
        for (int h = 0; h < BPW; ++h) g1_dma16(bsrc[h] + ko, sb + G1_ASTAGE + (uint32_t)(wave + 8 * h) * 1024);
    };

    // ---- fragment byte offsets inside a stage (k-step s of the stage: logical slot 2s + lh)
    int aoff[2][2], boff[NJ][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ra = wm * 64 + i * 32 + l31;
            aoff[i][s] = ra * G1_ROWB + (((2 * s + lh) ^ ((ra >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int rb = wn * (32 * NJ) + j * 32 + l31;
            boff[j][s] = G1_ASTAGE + rb * G1_ROWB + (((2 * s + lh) ^ ((rb >> 2) & 3)) << 4);
        }
    }

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // launches with a second K segment (the folded BN backward) take their bias as the accumulators' start value: the loads are
    // issued here, under the ring fill, not as dependent round trips in the epilogue
    const bool bias_first = p.a2 != nullptr && p.bias != nullptr;
    if (bias_first) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 b4 = *(const f32x4*)(p.bias + n0 + wn * (32 * NJ) + j * 32 + 8 * gq + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][j][4 * gq + e] = acc[1][j][4 * gq + e] = b4[e];
            }
    }

    // ---- main loop: one barrier per step.  (A ping-pong variant -- two wave groups half a step apart, two barriers per step,
    // fragments read in one phase and multiplied in the other, as conv_pp.hip does for the 3x3 -- measured 5-10 % slower
    // here: with one tap per K chunk there is no register-only phase long enough to pay for the second barrier.)
    const int nk = p.nk;
#pragma unroll
    for (int k = 0; k < NST - 1; ++k)
        if (k < nk) issue(k);
    for (int k = 0; k < nk; ++k) {
        const int rem = nk - 1 - k;                                 // stages issued after stage k: min(rem, NST - 2) stay in flight
        if (rem >= NST - 2) G1_WAIT_VM((NST - 2) * DPW);
        else if (rem == 2) G1_WAIT_VM(2 * DPW);
        else if (rem == 1) G1_WAIT_VM(DPW);
        else G1_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();                               // stage k is complete; everyone has left stage k-1
        asm volatile("" ::: "memory");
        if (k + NST - 1 < nk) issue(k + NST - 1);
        const unsigned char* sb = smem + (k % NST) * STAGE;
        u32x4 fa[2][2], fb[NJ][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i][s] = (G1_ABL & 4) ? u32x4{(unsigned)aoff[i][s], (unsigned)k, 0x3f803f80u, 0x3f803f80u} : *(const u32x4*)(sb + aoff[i][s]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j][s] = (G1_ABL & 4) ? u32x4{(unsigned)boff[j][s], (unsigned)k, 0x3f803f80u, 0x3f803f80u} : *(const u32x4*)(sb + boff[j][s]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (G1_ABL & 1) acc[i][j][s] += __builtin_bit_cast(float, fb[j][s][0] ^ fa[i][s][1]);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[j][s]),
                                                                        __builtin_bit_cast(bf16x8, fa[i][s]), acc[i][j], 0, 0, 0);
    }

    // ---- epilogue.  Lane (pixel l31, half lh) of block (i, j) holds channels j*32 + 8g + 4lh + {0..3} in regs 4g..4g+3.
    __syncthreads();                                                // the ring is idle: reuse it as staging
    unsigned char* st = smem + wave * 32 * SS;                      // 32 rows x (32 NJ) channels per wave
    float* red = (float*)(smem + 8 * 32 * SS);                      // [4 wm][BN][2]
    const int chunk = lane % CPR, crow = lane / CPR;                // copy-out: CPR lanes per staged row
    constexpr int RPI = 64 / CPR;                                   // rows per copy-out instruction
    float s8[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
    // fused BN backward: coefficients of this lane's 8 channels
    const bool bnf = FULL && p.bnf_x != nullptr;
    const bool bnf_remask = bnf && p.bnf_relu && !p.bnf_mask;
    float bmu[8], bsc[8], bsf[8], is_fin = 0.f;
    if constexpr (FULL) {
        if (bnf) {
            if (tid < BN) is_fin = p.bnf_invstd[n0 + tid];           // for the partial store at the end (not behind the row stores)
            const int c0 = n0 + wn * (32 * NJ) + chunk * 8;
            // unconditional, back-to-back loads (absent gamma / beta read the mean array): per-element "load or zero" selects
            // made hipcc branch around each load and wait for it alone (see igemm_epilogue)
            const float* gp = bnf_remask ? p.bnf_gamma : p.bnf_mean;
            const float* bp = bnf_remask ? p.bnf_beta : p.bnf_mean;
            float mu[8], gm[8], is[8], bt[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { mu[e] = p.bnf_mean[c0 + e]; gm[e] = gp[c0 + e]; is[e] = p.bnf_invstd[c0 + e]; bt[e] = bp[c0 + e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = gm[e] * is[e];
                bmu[e] = mu[e];
                bsc[e] = bnf_remask ? sc : 0.f;
                bsf[e] = bnf_remask ? bt[e] - mu[e] * sc : 0.f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int col = j * 32 + 8 * gq + 4 * lh;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * gq + e];
                if (p.bias && !bias_first) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += p.bias[n0 + wn * (32 * NJ) + col + e];
                }
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *(bf16x4*)(st + l31 * SS + col * 2) = o;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0): wave-private staging
        // side operands of this block's rows, requested together ahead of the stores (which may alias them as far as the
        // compiler knows: loads left inside the store loop would cost one latency per row group)
        constexpr int NT = 32 / RPI;
        Vec16<bf16> av[FULL ? NT : 1], xv[FULL ? NT : 1];
        unsigned am[FULL ? NT : 1], km[FULL ? NT : 1];
        if constexpr (FULL) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int r = crow + RPI * t, m = m0 + wm * 64 + i * 32 + r;
                am[t] = km[t] = 0xFFu;
                if (m < p.M) {
                    const size_t go = (size_t)m * p.Nout + n0 + wn * (32 * NJ) + chunk * 8;      // element offset of this 16-byte vector
                    if (p.addend) {
                        av[t].v = *(const decltype(av[t].v)*)(p.addend + go);
                        if (p.addend_mask) am[t] = p.addend_mask[go >> 3];
                    }
                    if (bnf) {
                        xv[t].v = *(const decltype(xv[t].v)*)(p.bnf_x + go);
                        if (p.bnf_relu && p.bnf_mask) km[t] = p.bnf_mask[go >> 3];
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int r = crow + RPI * t, m = m0 + wm * 64 + i * 32 + r;
            if (m < p.M) {
                Vec16<bf16> v;
                v.v = *(const decltype(v.v)*)(st + r * SS + chunk * 16);
                if constexpr (FULL) {
                    if (p.addend) {          // same rounding as a separate add kernel: round(round(conv) + addend)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v.set(e, v.get(e) + (((am[t] >> e) & 1u) ? av[t].get(e) : 0.f));
                    }
                    if (bnf) {
                        // dx' = dx * ReLU-mask of the producing BN layer; s8 += dx', q8 += dx' * (x - mean)
                        unsigned keep = km[t];
                        if (bnf_remask) {
                            keep = 0;
#pragma unroll
                            for (int e = 0; e < 8; ++e) keep |= (xv[t].get(e) * bsc[e] + bsf[e] > 0.f ? 1u : 0u) << e;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float dv = ((keep >> e) & 1u) ? v.get(e) : 0.f;
                            v.set(e, dv);
                            s8[e] += dv;
                            q8[e] += dv * (xv[t].get(e) - bmu[e]);
                        }
                    }
                }
                *(decltype(v.v)*)(p.c + (size_t)m * p.Nout + n0 + wn * (32 * NJ) + chunk * 8) = v.v;
                if constexpr (STATS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = v.get(e);
                        s8[e] += f;
                        q8[e] += f * f;
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                         // staged rows read before the next block overwrites them
    }
    if (STATS || bnf) {
        // lanes with the same chunk hold the same 8 channels: xor tree over the higher lane bits, then the four row waves of
        // a column half through LDS, in fixed order
#pragma unroll
        for (int off = CPR; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s8[e] += __shfl_xor(s8[e], off, 64);
                q8[e] += __shfl_xor(q8[e], off, 64);
            }
        __syncthreads();                                            // every wave has finished with its staging block
        if (lane < CPR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float* o = red + ((size_t)wm * BN + wn * (32 * NJ) + lane * 8 + e) * 2;
                o[0] = s8[e]; o[1] = q8[e];
            }
        }
        __syncthreads();
        if (tid < BN) {
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                S += red[((size_t)w * BN + tid) * 2];
                Q += red[((size_t)w * BN + tid) * 2 + 1];
            }
            if constexpr (FULL) {
                const int c = n0 + tid;
                float* o = p.bnf_partial + ((size_t)(p.bnf_tile0 + tile_m) * p.Nout + c) * 2;
                o[0] = S;
                o[1] = Q * is_fin;
            } else {
                const int nvalid = min(G1_BM, p.M - m0);
                const float mean = S / (float)nvalid;
                const float m2 = Q - S * mean;
                float* o = p.stats + ((size_t)tile_m * p.Nout + n0 + tid) * 3;
                o[0] = (float)nvalid;
                o[1] = mean;
                o[2] = m2 > 0.f ? m2 : 0.f;
            }
        }
    }
}

}  // namespace

// shapes this kernel takes (dense 1x1 stride-1 GEMM, bf16): K a multiple of the 32-channel step, Nout of the 128-wide tile
bool gemm1x1_shape_ok(int64_t M, int K, int Nout) {
    return M > 0 && K >= 64 && K % G1_KC == 0 && Nout % 128 == 0 && M * (int64_t)(K > Nout ? K : Nout) < (1ll << 31);
}

// 0: not taken; 2 / 4: column blocks per wave (256 x 128 / 256 x 256 tiles).  full: the launch carries data-gradient side
// operands (MODE 2: its registers allow one workgroup per CU).  CNK_GEMM1X1: 0 off, 1 by rule, 2 / 4 force that tile
// wherever the shape fits.
//
// The rule is stated in WORK PER CU (256 CUs, one or two resident workgroups each), not in batch-specific tile counts:
//   * the ring pays off when its fill / drain (3 steps) is amortised over a long K loop: >= 32 steps of 32 channels;
//   * a 256 x 256 tile needs >= 0.59 tiles per CU (else most CUs idle and the 128 x 128 kernel's finer grain wins);
//   * 256 x 128 tiles with a long K loop are taken for 0.375 .. 1.17 tiles per CU (one, partly filled round of the
//     chip: the old kernel's 128-row tiles would make several rounds), with or without the fused epilogue;
//   * with a medium K loop (>= 8 steps) and >= 256 output channels, 256 x 128 tiles are taken once there is at least one
//     per CU: two workgroups then share a CU and cover each other's barrier stalls.
// Measured on MI355X (tools/bench_g1.py, A/B in one process) at batch 256 and 128; tests/test_gpu_round3.py sweeps the
// batch 32 .. 256 for equal results whatever this picks.
constexpr int G1_CUS = 256;
int gemm1x1_pick(int64_t M, int K, int Nout, int full) {
    const int mode = cnk_sw().gemm1x1;
    if (mode == 0 || !gemm1x1_shape_ok(M, K, Nout)) return 0;
    if (mode == 4) return Nout % 256 == 0 ? 4 : 2;
    if (mode == 2) return 2;
    const int64_t tm = (M + G1_BM - 1) / G1_BM, t2 = tm * (Nout / 128), t4 = Nout % 256 == 0 ? tm * (Nout / 256) : 0;
    const int ksteps = K / G1_KC;
    const bool long_k = ksteps >= 32, one_round2 = t2 * 8 >= 3 * G1_CUS && t2 * 64 <= 75 * G1_CUS;
    if (long_k && t4 * 128 >= 75 * G1_CUS) return 4;             // 1024->256 @14x14 50 -> 41 us, 1024->512 78 -> 74, fused dgrad of 256->1024 65 -> 56
    if (full) return (long_k && one_round2) ? 2 : 0;           // fused data gradients: 512->2048 @7x7 68 -> 51 us; batch 128: 256->1024 @14x14 44 -> 34
    if (long_k && one_round2) return 2;                        // 2048->512 @7x7 62 -> 44 us; batch 128: 1024->256 @14x14 37 -> 28
    // 512->256 @28x28 94 -> 88 us, 256->1024 @14x14 57 -> 54, 512->2048 @7x7 46 -> 45.7; narrower outputs (512->128) and
    // shorter K (128->512, 64->256) stay on the 128 x 128 kernel
    if (ksteps >= 8 && Nout >= 256 && t2 >= G1_CUS) return 2;
    return 0;
}

int64_t gemm1x1_slices(int64_t M) { return (M + G1_BM - 1) / G1_BM; }

template <int NJ>
static int gemm1x1_launch_t(const G1Params& p0, double algo_flops, double algo_bytes, hipStream_t s) {
    G1Params p = p0;
    p.tiles_n = p.Nout / G1Cfg<NJ>::BN;
    // more than 64 KiB of dynamic LDS needs the opt-in (per device, common.h)
    CNK_LDS_OPTIN((gemm1x1_kernel<NJ, 0>), G1Cfg<NJ>::LDS, "gemm1x1");
    CNK_LDS_OPTIN((gemm1x1_kernel<NJ, 1>), G1Cfg<NJ>::LDS, "gemm1x1");
    CNK_LDS_OPTIN((gemm1x1_kernel<NJ, 2>), G1Cfg<NJ>::LDS, "gemm1x1");
    ProfScope kscope(CNK_KF_GATHER, s, algo_flops, algo_bytes);
    const int grid = p.tiles_m * p.tiles_n;
    cnk_count(NJ == 4 ? CNK_CNT_G1_4 : CNK_CNT_G1_2);
    if (p.addend || p.bnf_x) hipLaunchKernelGGL((gemm1x1_kernel<NJ, 2>), dim3(grid), dim3(512), G1Cfg<NJ>::LDS, s, p);
    else if (p.stats) hipLaunchKernelGGL((gemm1x1_kernel<NJ, 1>), dim3(grid), dim3(512), G1Cfg<NJ>::LDS, s, p);
    else hipLaunchKernelGGL((gemm1x1_kernel<NJ, 0>), dim3(grid), dim3(512), G1Cfg<NJ>::LDS, s, p);
    CNK_CHECK_LAUNCH("gemm1x1");
    return 0;
}

int gemm1x1_launch(int nj, const G1Args& a, hipStream_t s) {
    G1Params p = {};
    p.a = (const bf16*)a.a; p.a2 = (const bf16*)a.a2; p.b = (const bf16*)a.b; p.c = (bf16*)a.c; p.bias = a.bias; p.stats = a.stats;
    p.addend = (const bf16*)a.addend; p.addend_mask = a.addend_mask;
    p.bnf_x = (const bf16*)a.bnf_x; p.bnf_mean = a.bnf_mean; p.bnf_invstd = a.bnf_invstd; p.bnf_gamma = a.bnf_gamma;
    p.bnf_beta = a.bnf_beta; p.bnf_mask = a.bnf_mask; p.bnf_partial = a.bnf_partial; p.bnf_relu = a.bnf_relu; p.bnf_tile0 = a.bnf_tile0;
    p.M = (int)a.M; p.K = a.K; p.K2 = a.a2 ? a.K2 : 0; p.Nout = a.Nout;
    CNK_CHECK_ARG(p.K2 % G1_KC == 0, "gemm1x1: second K segment off the 32-channel grid");
    p.tiles_m = (int)((a.M + G1_BM - 1) / G1_BM); p.tiles_n = 0; p.nk = (p.K + p.K2) / G1_KC;
    CNK_CHECK_ARG(!(p.stats && (p.addend || p.bnf_x)), "gemm1x1: statistics partials and data-gradient side operands are exclusive");
    return nj == 4 ? gemm1x1_launch_t<4>(p, a.algo_flops, a.algo_bytes, s) : gemm1x1_launch_t<2>(p, a.algo_flops, a.algo_bytes, s);
}
