// Ping-pong implicit-GEMM core for the 3x3 / stride-1 / "same" convolutions (forward and data gradient), bf16.
// Sites served: the bottleneck 3x3 (/root/reference/src/resnet.py:171) and the VGG 3x3 stack (src/vggnet.py:69).
//
//   C[m][n] = sum_{c,t} A[m + tap t][c] * B[n][t][c]        m = (img, oh, ow) flat, n = output channel
//
// One 512-thread workgroup per CU, PERSISTENT over a contiguous run of BM(m) x BN(n) tiles.  The eight waves form
// two groups of four (one wave of each group per SIMD) that run the SAME instruction stream one phase apart
// (group 1 executes one extra s_barrier up front):
//
//   phase L(j): every ds_read_b128 of K step j (= one tap of one channel chunk), the LDS-DMAs of step j+2's weight
//               tile, counted vmcnt, barrier
//   phase C(j): the step's 16 v_mfma_f32_32x32x16_bf16 from registers only; underneath them the LDS-DMA of one 1-KiB
//               piece of the NEXT chunk's pixel patch with its address arithmetic, and the fragment addresses of
//               step j+1; barrier
//
// so on every SIMD one wave always owns the matrix pipe while its partner moves data: loads, LDS reads and MFMAs
// overlap inside the workgroup by construction (the r01 kernels relied on 3-4 independent workgroups per CU).
// Everything that stays in flight across barriers is an LDS-DMA retired by counted `s_waitcnt vmcnt(N)`; barriers are
// raw `s_barrier`; EVERY vector-memory operation of the kernel is inline asm, so hipcc tracks none of them and never
// drains the pipeline on its own (cdna_hip_programming.md 5.7): all completions are counted by hand.
//
// A operand = pixel patch in PADDED coordinates: row pitch P = W+1 (one shared zero cell between image rows),
// image pitch (H+1)*P (one shared zero row between images), so a tap is the uniform row offset dh*P+dw and the
// image borders are real zero cells in LDS -- no per-lane validity masks in the loop.  The patch of a channel chunk
// (BM + 2P + 2 + wraps rows) is double-buffered and reused by the nine taps; the next chunk's patch (possibly the next
// tile's) streams in during taps 0..6.  B operand = packed weights [n][t][c], a ring of three tiles.  LDS rows hold
// one chunk of one pixel / output channel (128 B or 64 B) with the 16-byte piece index XOR-swizzled by the row (on the
// DMA SOURCE address, the LDS image is lane-linear) => conflict-free ds_read_b128.
//
// Configurations (PPCfg): wave tile 64x64 with 64-channel K steps (one LDS read per MFMA) or 128x64 with 32-channel
// K steps (0.75 reads per MFMA, half the weight bytes per flop), workgroup tile 256x128, 256x64, 512x128 or 256x256.
//
// The epilogue runs per wave on a private 2.5-KiB staging block (no workgroup barrier): 32x32 blocks are staged
// as bf16 rows and written with 16-byte row stores; bias, the gradient addend (+ReLU bit mask), the fused BN-backward
// reduction and the BN statistics partials (one (n, mean, M2) record per wave slice of 64 / 128 rows) as in
// igemm_epilogue of conv_igemm.hip.
#include <cstdlib>

#include "common.h"

#ifndef WS_SIDE_Q
#define WS_SIDE_Q 8   // K step of a tile's second job at which the full epilogue's side operands are requested
#endif
#ifndef WS_ABL
#define WS_ABL 0      // debug builds only (ABL_MACRO=WS_ABL tools/ablate_pp.sh): ws3x3_kernel without 1 patch DMA, 4 LDS reads, 8 MFMA, 16 epilogue
#endif
#ifndef PP_ABL
#define PP_ABL 0      // debug builds only (tools/ablate_pp.sh): 1 no patch DMA, 2 no weight DMA, 4 no LDS reads, 8 no MFMA
#endif

namespace {

constexpr int PP_EPI = 2560;                 // per-wave staging: 32 rows x 80 B

__device__ __attribute__((aligned(16))) unsigned int pp_zero16[4] = {0, 0, 0, 0};
typedef __attribute__((address_space(3))) unsigned char lds_u8;

// MI x NJ 32x32 blocks per wave, GM waves along m (GN = 8/GM along n), KC channels per K step
template <int MI_, int NJ_, int GM_, int KC_> struct PPCfg {
    static constexpr int MI = MI_, NJ = NJ_, GM = GM_, GN = 8 / GM_, KC = KC_;
    static constexpr int BM = GM * MI * 32, BN = GN * NJ * 32;
    static constexpr int RB = KC * 2;                 // LDS row bytes
    static constexpr int LRB = KC == 64 ? 7 : 6;      // log2(RB)
    static constexpr int S = KC / 16;                 // MFMAs along K per block and K step
    static constexpr int CPR = RB / 16;               // 16-byte pieces per row
    static constexpr int RPP = 1024 / RB;             // rows per 1-KiB DMA
    static constexpr int SWS = KC == 64 ? 3 : 2;      // swizzle: piece ^= (row >> (4 - SWS)) & (CPR-1), as (row << SWS) & SWM on the byte
    static constexpr int SWM = (CPR - 1) << 4;
    static constexpr int BSTAGE = BN * RB;            // bytes of one weight stage
    static constexpr int NB = BSTAGE / 1024 / 8;      // weight DMAs per wave per K step
    static constexpr int WROWS = MI * 32;             // rows of a wave = one statistics slice
    static constexpr int NM = MI * NJ * S;            // MFMAs per wave per K step
    static constexpr int EPI_STORES = MI * 2 * NJ;    // row stores of the epilogue per wave
};

struct PPParams {
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
    int bnf_relu, bnf_tile0;
    int relu;
    int N, H, W, C, Nout, M;
    int P, IP;
    int csteps;
    int tiles_n, tiles_total, nwg;
    int npg;
    int toff[9];
    FastDiv fd_hw, fd_w, fd_ip, fd_p;
};

#define PP_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")
#define PP_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int OFF> __device__ __forceinline__ void ds_read128(u32x4& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}

// LDS-DMA as inline asm (cdna_hip_programming.md 5.7): M0 (the LDS destination) is saved and restored inside the statement.
__device__ __forceinline__ void dma16(const void* g, uint32_t lds_addr) {
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

template <class CFG> struct PPState {
    f32x16 acc[CFG::MI][CFG::NJ];
    u32x4 fa[CFG::MI][CFG::S], fb[CFG::NJ][CFG::S];
    int prow[CFG::MI];                             // patch row (relative) of this lane's pixel, per row block
    uint32_t abase[CFG::MI];                       // LDS byte address of the next phase L's A fragments (k step 0)
    uint32_t bB[CFG::NJ];                          // LDS byte address of this lane's weight rows (stage 0, k step 0)
    int brow[CFG::NB];                             // weight image element offset of this lane's DMA rows
    // uniform
    const unsigned char* src;
    const bf16* wpk;
    uint32_t lds_patch, lds_b, lds_dump, npgb;
    int wave, lane, l31, lh, wm, wn;
    int k, c, pb;                                  // tile, chunk, patch buffer of the current chunk
    int kn, cn, has_next;                          // next chunk
    int tile_m, tile_n;
    int pbase_n;                                   // padded index of patch row 0 of the next chunk's tile
    int wuni, wuni_next;                           // weight element offset of (tile_n, chunk) for current / next chunk
    int post_epi;
};

__device__ __forceinline__ int pp_index(const PPParams& p, int m) {
    uint32_t img = fd_div(m, p.fd_hw), rem = m - img * p.fd_hw.d;
    uint32_t h = fd_div(rem, p.fd_w), w = rem - h * p.fd_w.d;
    return (int)(img * p.IP + (h + 1) * p.P + w + 1);
}

// LDS byte address (k step 0) of the fragment row `row` of an image starting at `base`: the swizzled piece of lane half lh
template <class CFG> __device__ __forceinline__ uint32_t pp_frag_addr(uint32_t base, uint32_t row, int lh) {
    return (base + ((row << CFG::LRB) | ((row << CFG::SWS) & CFG::SWM))) ^ ((uint32_t)lh << 4);
}

// Source pointer and LDS destination of one 1-KiB piece of chunk cn of the tile whose patch starts at padded index
// pbase.  Branch-free 32-bit arithmetic (the tensor is < 2 GiB, checked on the host) so that it can issue between MFMAs.
template <class CFG, class ST> __device__ __forceinline__ void pp_patch_piece_addr(const PPParams& p, const ST& st, int piece, int pbase, int cn, int pbuf, bool live,
                                                                        const unsigned char*& g, uint32_t& dst) {
    const int row = piece * CFG::RPP + st.lane / CFG::CPR;
    const uint32_t q = (uint32_t)(pbase + row);
    const uint32_t img = fd_div(q, p.fd_ip), rem = q - img * p.fd_ip.d;
    const uint32_t r = fd_div(rem, p.fd_p), cc = rem - r * p.fd_p.d;
    const int ch = (st.lane % CFG::CPR) ^ ((row >> (4 - CFG::SWS)) & (CFG::CPR - 1));
    const bool use = live & (piece < p.npg);
    const bool ok = use & (r >= 1) & (cc >= 1) & (img < (uint32_t)p.N);
    const uint32_t pix = (img * p.H + (r - 1)) * p.W + (cc - 1);
    const uint32_t off = (pix * p.C + cn * CFG::KC + ch * 8) * 2;
    const unsigned char* gz = (const unsigned char*)pp_zero16;
    const unsigned char* gp = st.src + off;
    g = ok ? gp : gz;
    dst = use ? st.lds_patch + pbuf * st.npgb + piece * 1024 : st.lds_dump;
}

template <class CFG> __device__ __forceinline__ void pp_issue_b(const PPParams& p, const PPState<CFG>& st, int wuni, int tap, int stage, bool live) {
#pragma unroll
    for (int u = 0; u < CFG::NB; ++u) {
        const bf16* g = live ? st.wpk + (size_t)(uint32_t)(st.brow[u] + wuni + tap * p.C) : (const bf16*)pp_zero16;
        dma16(g, st.lds_b + stage * CFG::BSTAGE + (st.wave * CFG::NB + u) * 1024);
    }
}

// Phase L + phase C of K step (chunk st.c, tap T).
template <class CFG, int T, bool FIRST = false> __device__ __forceinline__ void pp_tap(const PPParams& p, PPState<CFG>& st) {
    constexpr int MI = CFG::MI, NJ = CFG::NJ, S = CFG::S, NB = CFG::NB, NM = CFG::NM;
    constexpr int STAGE = T % 3;
    // ---- phase L ---------------------------------------------------------------------------------------
    {
#if !(PP_ABL & 4)
        if (T == 0) {
            // first tap of a chunk: patch buffer / tile may have changed since the previous phase C
            const uint32_t pbuf = st.lds_patch + st.pb * st.npgb;
#pragma unroll
            for (int i = 0; i < MI; ++i) st.abase[i] = pp_frag_addr<CFG>(pbuf, (uint32_t)(st.prow[i] + p.toff[0]), st.lh);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int s = 0; s < S; ++s) ds_read128<0>(st.fa[i][s], st.abase[i] ^ (uint32_t)(s << 5));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < S; ++s) ds_read128<STAGE * CFG::BSTAGE>(st.fb[j][s], st.bB[j] ^ (uint32_t)(s << 5));
#else
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int s = 0; s < S; ++s) asm volatile("" : "+v"(st.fa[i][s]));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < S; ++s) asm volatile("" : "+v"(st.fb[j][s]));
#endif
        // weights of step j+2 (tap T+2 of this chunk, or tap T-7 of the next chunk) into stage (T+2)%3
#if !(PP_ABL & 2)
        if (T + 2 < 9) pp_issue_b<CFG>(p, st, st.wuni, T + 2, (T + 2) % 3, true);
        else pp_issue_b<CFG>(p, st, st.wuni_next, T + 2 - 9, (T + 2) % 3, st.has_next != 0);
#endif
        // retire everything issued before this phase except the patch piece of the previous phase C (the counter
        // retires in order: that piece, which comes from HBM, gets three phases this way)
        constexpr int KEEP = ((PP_ABL & 2) ? 0 : NB) + ((PP_ABL & 1) ? 0 : ((T >= 1 && T <= 7) ? 1 : 0));
        // (the first phase L after an epilogue waits for nothing: the epilogue retired every DMA issued before it, this
        // phase's own weight DMAs are for step 2, and the epilogue's stores are nobody's operand)
        if (!(T == 0 && st.post_epi != 0)) PP_WAIT_VM(KEEP);
        PP_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase C ---------------------------------------------------------------------------------------
#if (PP_ABL & 8)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int s = 0; s < S; ++s) asm volatile("" ::"v"(st.fa[i][s]));
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int s = 0; s < S; ++s) asm volatile("" ::"v"(st.fb[j][s]));
    if (FIRST) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) st.acc[i][j][r] = 0.f;
    }
    if (T < 8) {
        const uint32_t pbuf = st.lds_patch + st.pb * st.npgb;
#pragma unroll
        for (int i = 0; i < MI; ++i) st.abase[i] = pp_frag_addr<CFG>(pbuf, (uint32_t)(st.prow[i] + p.toff[T < 8 ? T + 1 : 0]), st.lh);
    }
#else
    // One piece of the next chunk's patch (taps 0..6: 56 slots) and the A fragment addresses of the next phase L: their
    // address arithmetic is interleaved with the MFMAs (sched_group_barrier: one MFMA, three VALU/SALU, ...) and the
    // DMA itself goes out before the last two.  FIRST: first K step of a tile, the accumulators start from zero.
    const unsigned char* pg = nullptr;
    uint32_t pdst = 0;
    constexpr bool PATCH = T <= 6 && !(PP_ABL & 1);
    __builtin_amdgcn_s_setprio(1);
    if (PATCH) pp_patch_piece_addr<CFG, PPState<CFG>>(p, st, T * 8 + st.wave, st.pbase_n, st.cn, st.pb ^ 1, st.has_next != 0, pg, pdst);
    if (T < 8) {
        const uint32_t pbuf = st.lds_patch + st.pb * st.npgb;
#pragma unroll
        for (int i = 0; i < MI; ++i) st.abase[i] = pp_frag_addr<CFG>(pbuf, (uint32_t)(st.prow[i] + p.toff[T < 8 ? T + 1 : 0]), st.lh);
    }
    auto mma = [&](int idx) {                       // flat order: k step, row block, column block
        const int s = idx / (MI * NJ), i = (idx / NJ) % MI, j = idx % NJ;
        if (FIRST && s == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            st.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, st.fb[j][s]), __builtin_bit_cast(bf16x8, st.fa[i][s]), z, 0, 0, 0);
        } else {
            st.acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, st.fb[j][s]), __builtin_bit_cast(bf16x8, st.fa[i][s]), st.acc[i][j], 0, 0, 0);
        }
    };
#pragma unroll
    for (int idx = 0; idx < NM - 2; ++idx) mma(idx);
#pragma unroll
    for (int g = 0; g < NM - 2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x6, T < 8 ? 3 : 0, 0);
    }
    if (PATCH) dma16(pg, pdst);
    mma(NM - 2);
    mma(NM - 1);
    __builtin_amdgcn_s_setprio(0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Stores as inline asm (hipcc tracks no vector-memory operation of this kernel).  The trailing
// s_nop keeps the data registers intact until the store has read them (cdna_hip_programming.md 5.7 item 1).
__device__ __forceinline__ void pp_store16(void* g, const bf16x8& v) {
    const u32x4 d = __builtin_bit_cast(u32x4, v);
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(d) : "memory");
}
__device__ __forceinline__ void pp_store12(void* g, float a, float b, float c) {
    typedef __attribute__((ext_vector_type(3))) float f32x3;
    f32x3 v = {a, b, c};
    asm volatile("global_store_dwordx3 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(v) : "memory");
}
__device__ __forceinline__ void pp_store8(void* g, float a, float b) {
    f32x2 v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(v) : "memory");
}

// Side-operand loads of the full epilogue, inline asm like every other vector-memory operation of this kernel
// (cdna_hip_programming.md 5.7 item 1 form (ii): the destination is named "+v" by the wait that retires it).
__device__ __forceinline__ void pp_load16(u32x4& d, const void* g) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(g) : "memory");
}
__device__ __forceinline__ void pp_load1(unsigned& d, const void* g) {
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(d) : "v"(g) : "memory");
}
template <int N> __device__ __forceinline__ void pp_wait_side(u32x4& a, u32x4& x, unsigned& am, unsigned& km) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(x), "+v"(am), "+v"(km) : "i"(N) : "memory");
}

enum { PP_EPI_PLAIN = 0, PP_EPI_STATS = 1, PP_EPI_FULL = 2 };

// Side operands of one 64-row x 32-channel block pair of the full epilogue, and the per-lane BN coefficients of a column block.
// The ping-pong core fetches them inside the epilogue; the weight-stationary kernel, whose epilogue nothing covers, issues the
// loads a whole job early (pp_side_issue at the start of the tile's last job) and keeps the coefficients for the launch.
struct PPSide {
    u32x4 av[4], xv[4];
    unsigned am[4], km[4];
};
struct PPCoef {
    float bmu[8], bsc[8], bsf[8], is_fin;
};

// rows of the four passes of row-block pair ib of this wave (-1: past the end)
template <class CFG, class ST> __device__ __forceinline__ void pp_side_rows(const PPParams& p, const ST& st, int ib, int (&ms)[4]) {
    const int crow = st.lane >> 2;
    const int R0 = st.tile_m * CFG::BM + st.wm * CFG::WROWS;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = R0 + (ib * 2 + (k >> 1)) * 32 + (k & 1) * 16 + crow;
        ms[k] = m < p.M ? m : -1;
    }
}

// the 16 side-operand loads of column block j, row-block pair ib (an absent operand reads the zero block)
template <class CFG, class ST> __device__ __forceinline__ void pp_side_issue(const PPParams& p, const ST& st, int j, int ib, PPSide& sd) {
    const int cc = st.lane & 3;
    const int cb = st.tile_n * CFG::BN + st.wn * (CFG::NJ * 32) + j * 32;
    const bool bnf = p.bnf_x != nullptr, has_add = p.addend != nullptr;
    int ms[4];
    pp_side_rows<CFG, ST>(p, st, ib, ms);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t goff = ((size_t)(ms[k] < 0 ? 0 : ms[k]) * p.Nout + cb) * 2 + cc * 16;
        const unsigned char* z = (const unsigned char*)pp_zero16;
        const bool on = ms[k] >= 0;
        pp_load16(sd.av[k], on && has_add ? (const unsigned char*)p.addend + goff : z);
        pp_load1(sd.am[k], on && has_add && p.addend_mask ? p.addend_mask + (goff >> 4) : z);
        pp_load16(sd.xv[k], on && bnf ? (const unsigned char*)p.bnf_x + goff : z);
        pp_load1(sd.km[k], on && bnf && p.bnf_relu && p.bnf_mask ? p.bnf_mask + (goff >> 4) : z);
    }
}

// the same coefficients as a per-channel table [mean | scale | shift | invstd][Nout] (thread ch < Nout writes channel ch)
__device__ __forceinline__ void pp_coef_table(const PPParams& p, int ch, float* tab) {
    const bool bnf = p.bnf_x != nullptr;
    const bool bnf_remask = bnf && p.bnf_relu && !p.bnf_mask;
    float mu = 0.f, sc = 0.f, sf = 0.f, is = 0.f;
    if (bnf) {
        mu = p.bnf_mean[ch];
        is = p.bnf_invstd[ch];
        if (bnf_remask) {
            sc = p.bnf_gamma[ch] * is;
            sf = p.bnf_beta[ch] - mu * sc;
        }
    }
    tab[ch] = mu; tab[p.Nout + ch] = sc; tab[2 * p.Nout + ch] = sf; tab[3 * p.Nout + ch] = is;
}

// BN coefficients of the 8 channels cb + (lane & 3) * 8.. this lane masks / sums, and invstd of the channel it finalises
__device__ __forceinline__ void pp_coef_load(const PPParams& p, int cb, int lane, PPCoef& cf) {
    const bool bnf = p.bnf_x != nullptr;
    const bool bnf_remask = bnf && p.bnf_relu && !p.bnf_mask;
#pragma unroll
    for (int e = 0; e < 8; ++e) cf.bmu[e] = cf.bsc[e] = cf.bsf[e] = 0.f;
    cf.is_fin = 0.f;
    if (bnf) {
        // unconditional, back-to-back loads (absent gamma / beta read the mean array): per-element "load or zero"
        // selects made hipcc branch around each load and wait for it alone (see igemm_epilogue)
        const float* gp = bnf_remask ? p.bnf_gamma : p.bnf_mean;
        const float* bp = bnf_remask ? p.bnf_beta : p.bnf_mean;
        const int ch0 = cb + (lane & 3) * 8;
        float mu[8], gm[8], is[8], bt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = p.bnf_mean[ch0 + e]; gm[e] = gp[ch0 + e]; is[e] = p.bnf_invstd[ch0 + e]; bt[e] = bp[ch0 + e]; }
        // invstd of the channel this lane finalises: fetched HERE -- a load in front of the partial store would
        // be waited for with vmcnt(0), i.e. behind every row store of the column block (the counter retires in order)
        cf.is_fin = p.bnf_invstd[cb + (lane & 31)];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = gm[e] * is[e];
            cf.bmu[e] = mu[e];
            cf.bsc[e] = bnf_remask ? sc : 0.f;
            cf.bsf[e] = bnf_remask ? bt[e] - mu[e] * sc : 0.f;
        }
    }
    // the coefficient loads above are hipcc's own (waited for by hipcc, ahead of every asm operation below)
    asm volatile("" : "+v"(cf.bmu[0]), "+v"(cf.bsc[0]), "+v"(cf.bsf[0]), "+v"(cf.is_fin));
}

// Per-wave epilogue of tile (tile_m, tile_n): see the header comment.  MODE: PLAIN = rounded rows only; STATS = + BN
// statistics partials; FULL = bias / ReLU / gradient addend (+mask) / fused BN backward / statistics, all run-time.
// The number of vector-memory operations per wave is static: CFG::EPI_STORES row stores, + NJ partial stores in STATS
// and FULL (when sums are written).
// FULL issues the 16 side-operand loads of two row blocks of a column block (4 row passes x {addend, its mask, BN
// input, its mask}; an absent operand reads the zero block) BEFORE their first store: vmcnt retires in order, so a
// load waited for behind a store would wait for that store's round trip too.
template <class CFG, int MODE, class ST = PPState<CFG>, bool WAIT0 = true, bool PRE = false>
__device__ __forceinline__ void pp_epilogue(const PPParams& p, ST& st, unsigned char* smem_epi, PPSide* pre_side = nullptr, const float* coef_tab = nullptr) {
    static_assert(!PRE || (CFG::NJ == 1 && CFG::MI == 2), "preloaded side operands: one column block, one row-block pair");
    constexpr int MI = CFG::MI, NJ = CFG::NJ;
    unsigned char* stg = smem_epi + st.wave * PP_EPI;
    const int lane = st.lane, l31 = st.l31, lh = st.lh;
    const int crow = lane >> 2, cc = lane & 3;
    const int R0 = st.tile_m * CFG::BM + st.wm * CFG::WROWS;
    const int C0 = st.tile_n * CFG::BN + st.wn * (NJ * 32);
    unsigned char* dst = (unsigned char*)p.dst;
    const bool bnf = MODE == PP_EPI_FULL && p.bnf_x != nullptr;
    const bool bnf_remask = bnf && p.bnf_relu && !p.bnf_mask;
    const bool has_add = MODE == PP_EPI_FULL && p.addend != nullptr;
    const bool want_sums = MODE == PP_EPI_STATS || (MODE == PP_EPI_FULL && (p.stats || bnf));
    // Every LDS-DMA of the K loop (the next tile's tap-1 weights, issued one phase ago, are the youngest) is retired HERE,
    // before the first store of the epilogue: the K loop's counted waits must never have to tell loads from stores, because
    // vmcnt does not retire them in one order (see the side-operand waits below).  (WAIT0 = false: the streaming kernel
    // below, whose one counted wait per job names loads issued after it only.)
    if (WAIT0) PP_WAIT_VM(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cb = C0 + j * 32;                                  // first channel of this column block
        __builtin_amdgcn_sched_barrier(0);                           // keep the column blocks apart (register pressure)
        float s8[8], q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
        PPCoef cf;
        if (MODE == PP_EPI_FULL && !PRE) pp_coef_load(p, cb, lane, cf);
        if (MODE == PP_EPI_FULL && PRE) {
            // per-channel table in LDS, written once per launch (pp_coef_table): [mean | scale | shift | invstd][Nout]
            const int ch0 = cb + cc * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { cf.bmu[e] = coef_tab[ch0 + e]; cf.bsc[e] = coef_tab[p.Nout + ch0 + e]; cf.bsf[e] = coef_tab[2 * p.Nout + ch0 + e]; }
            cf.is_fin = coef_tab[3 * p.Nout + cb + (lane & 31)];
        }
        const float is_fin = MODE == PP_EPI_FULL ? cf.is_fin : 0.f;
#pragma unroll
        for (int ib = 0; ib < MI / 2; ++ib) {
            __builtin_amdgcn_sched_barrier(0);
            PPSide sd_here;
            PPSide& sd = (MODE == PP_EPI_FULL && PRE) ? *pre_side : sd_here;
            int ms[4];
            pp_side_rows<CFG, ST>(p, st, ib, ms);
            if (MODE == PP_EPI_FULL && !PRE) pp_side_issue<CFG, ST>(p, st, j, ib, sd);
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const int i = ib * 2 + i2;
                // stage block (i, j): lane (l31, lh) owns pixel row l31 and channels 8g + 4lh + {0..3}
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = st.acc[i][j][4 * g + e];
                    if (MODE == PP_EPI_FULL) {
                        if (p.bias) {
                            // uniform (scalar) loads of the group's eight biases, the lane half picks its four
                            const float* bq = p.bias + __builtin_amdgcn_readfirstlane(cb + 8 * g);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += lh ? bq[4 + e] : bq[e];
                        }
                        if (p.relu && !has_add) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        }
                    }
                    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *(bf16x4*)(stg + l31 * 80 + (8 * g + 4 * lh) * 2) = o;
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int k = i2 * 2 + h2;
                    bf16x8 v = *(const bf16x8*)(stg + (h2 * 16 + crow) * 80 + cc * 16);
                    const int m = ms[k];
                    const size_t goff = ((size_t)(m < 0 ? 0 : m) * p.Nout + cb) * 2 + cc * 16;
                    if (MODE == PP_EPI_FULL) {
                        // retire this pass's four loads: the count is the 4*(3-k) LOADS of the later passes only.  The k
                        // stores of the earlier passes are NOT credited: loads retire in order among loads, but a store can
                        // be acknowledged before an older load has returned (round 3: crediting them -- vmcnt(9 / 6 / 3) --
                        // let a wait pass with one of its own loads still in flight as soon as the chip was busy enough
                        // for loads to be slower than store acks; 7 of 196 tiles wrong at batch 128, none at batch 16)
                        if (k == 0) pp_wait_side<12>(sd.av[0], sd.xv[0], sd.am[0], sd.km[0]);
                        else if (k == 1) pp_wait_side<8>(sd.av[1], sd.xv[1], sd.am[1], sd.km[1]);
                        else if (k == 2) pp_wait_side<4>(sd.av[2], sd.xv[2], sd.am[2], sd.km[2]);
                        else pp_wait_side<0>(sd.av[3], sd.xv[3], sd.am[3], sd.km[3]);
                    }
                    if (m >= 0) {
                        if (MODE == PP_EPI_FULL) {
                            const bf16x8 a8 = __builtin_bit_cast(bf16x8, sd.av[k]);
                            const bf16x8 x8 = __builtin_bit_cast(bf16x8, sd.xv[k]);
                            if (has_add) {
                                const unsigned amk = p.addend_mask ? sd.am[k] : 0xFFu;
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (((amk >> e) & 1u) ? (float)a8[e] : 0.f));
                                if (p.relu) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) v[e] = (float)v[e] > 0.f ? v[e] : (bf16)0.f;
                                }
                            }
                            if (bnf) {
                                unsigned keep = (p.bnf_relu && p.bnf_mask) ? sd.km[k] : 0xFFu;
                                if (bnf_remask) {
                                    keep = 0;
#pragma unroll
                                    for (int e = 0; e < 8; ++e) keep |= ((float)x8[e] * cf.bsc[e] + cf.bsf[e] > 0.f ? 1u : 0u) << e;
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const float dv = ((keep >> e) & 1u) ? (float)v[e] : 0.f;
                                    v[e] = (bf16)dv;
                                    s8[e] += dv;
                                    q8[e] += dv * ((float)x8[e] - cf.bmu[e]);
                                }
                            }
                        }
                        if (MODE == PP_EPI_STATS || (MODE == PP_EPI_FULL && p.stats)) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float f = (float)v[e];
                                s8[e] += f;
                                q8[e] += f * f;
                            }
                        }
                        // one store instruction per pass as long as ANY lane has a row (a partial last tile may skip
                        // whole passes)
                        pp_store16(dst + goff, v);
                    }
                }
            }
        }
        if (want_sums) {
            // per-channel sums over this wave's rows: lanes with equal (lane & 3) hold the same 8 channels
            float* red = (float*)stg;                                   // [16][32]
            float S = 0.f, Q = 0.f;
            const int chn = lane & 31, half = lane >> 5;
            // column ^ (row & 7): the 8 rows of a 32-lane half land on 8 different banks per (cc, e) instead of one
            const int wbase = crow * 32 + ((cc * 8) | (crow & 7));
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wbase ^ e] = s8[e];
#pragma unroll
            for (int r = 0; r < 8; ++r) S += red[(half * 8 + r) * 32 + (chn ^ r)];
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wbase ^ e] = q8[e];
#pragma unroll
            for (int r = 0; r < 8; ++r) Q += red[(half * 8 + r) * 32 + (chn ^ r)];
            S += __shfl_xor(S, 32, 64);
            Q += __shfl_xor(Q, 32, 64);
            const int ch = cb + chn;
            const int nvalid = min(CFG::WROWS, p.M - R0);
            const int slice = st.tile_m * CFG::GM + st.wm;
            if (lane < 32 && nvalid > 0) {
                if (bnf) {
                    pp_store8(p.bnf_partial + ((size_t)(p.bnf_tile0 + slice) * p.Nout + ch) * 2, S, Q * is_fin);
                } else {
                    const float mean = S / (float)nvalid;
                    float m2 = Q - S * mean;
                    pp_store12(p.stats + ((size_t)slice * p.Nout + ch) * 3, (float)nvalid, mean, m2 > 0.f ? m2 : 0.f);
                }
            }
        }
    }
}

template <class CFG, int MODE>
__global__ __launch_bounds__(512, 2) void pp3x3_kernel(const PPParams p) {
    constexpr int MI = CFG::MI, NJ = CFG::NJ, NB = CFG::NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* smem = smem_raw;
    PPState<CFG> st;
    const int tid = threadIdx.x;
    st.lane = tid & 63;
    st.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    st.l31 = st.lane & 31;
    st.lh = st.lane >> 5;
    const int grp = st.wave >> 2, q4 = st.wave & 3;
    st.wm = grp * (CFG::GM / 2) + q4 / CFG::GN;
    st.wn = q4 % CFG::GN;
    st.src = (const unsigned char*)p.src;
    st.wpk = (const bf16*)p.wpk;
    st.npgb = (uint32_t)p.npg * 1024u;
    // 128-byte aligned rows: the swizzle XORs are applied to whole LDS addresses
    const uint32_t lds_raw = (uint32_t)(uintptr_t)(lds_u8*)smem;
    st.lds_patch = (lds_raw + 127u) & ~127u;
    smem += st.lds_patch - lds_raw;
    st.lds_b = st.lds_patch + 2 * st.npgb;
    const uint32_t lds_epi = st.lds_b + 3 * CFG::BSTAGE;
    st.lds_dump = lds_epi + 8 * PP_EPI;
    unsigned char* smem_epi = smem + 2 * st.npgb + 3 * CFG::BSTAGE;

    // this workgroup's run of tiles (XCD-aware: blocks b, b+8, ... share an XCD and get adjacent runs)
    int L;
    {
        const int q = p.nwg >> 3, r = p.nwg & 7, xcd = blockIdx.x & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int k0 = (int)((long long)L * p.tiles_total / p.nwg), k1 = (int)((long long)(L + 1) * p.tiles_total / p.nwg);
    if (k0 >= k1) return;

    // static per-lane addresses of the weight tile: fragment rows and DMA rows
#pragma unroll
    for (int j = 0; j < NJ; ++j) st.bB[j] = pp_frag_addr<CFG>(st.lds_b, (uint32_t)(st.wn * (NJ * 32) + j * 32 + st.l31), st.lh);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int row = (st.wave * NB + u) * CFG::RPP + st.lane / CFG::CPR;
        const int ch = (st.lane % CFG::CPR) ^ ((row >> (4 - CFG::SWS)) & (CFG::CPR - 1));
        st.brow[u] = row * 9 * p.C + ch * 8;
    }

    auto set_tile = [&](int k) {
        st.tile_m = k / p.tiles_n;
        st.tile_n = k - st.tile_m * p.tiles_n;
        const int m0 = st.tile_m * CFG::BM;
        const int pbase = pp_index(p, m0) - p.P - 1;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int m = m0 + st.wm * CFG::WROWS + i * 32 + st.l31;
            m = m < p.M ? m : p.M - 1;
            st.prow[i] = pp_index(p, m) - pbase;
        }
        return pbase;
    };
    auto set_next = [&]() {
        if (st.c + 1 < p.csteps) { st.kn = st.k; st.cn = st.c + 1; st.has_next = 1; }
        else { st.kn = st.k + 1; st.cn = 0; st.has_next = st.kn < k1; }
        const int kk = st.has_next ? st.kn : st.k;
        const int tm = kk / p.tiles_n, tn = kk - tm * p.tiles_n;
        st.pbase_n = pp_index(p, tm * CFG::BM) - p.P - 1;
        st.wuni_next = tn * CFG::BN * 9 * p.C + st.cn * CFG::KC;
    };

#if (PP_ABL & 4)
#pragma unroll
    for (int s = 0; s < CFG::S; ++s) {
#pragma unroll
        for (int i = 0; i < MI; ++i) st.fa[i][s] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
        for (int j = 0; j < NJ; ++j) st.fb[j][s] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
#endif
    st.k = k0; st.c = 0; st.pb = 0; st.post_epi = 0;
    const int pbase0 = set_tile(k0);
    st.wuni = st.tile_n * CFG::BN * 9 * p.C;
    set_next();
    // prologue: the first chunk's whole patch, weights of steps 0 and 1
    for (int piece = st.wave; piece < p.npg; piece += 8) {
        const unsigned char* g;
        uint32_t d;
        pp_patch_piece_addr<CFG, PPState<CFG>>(p, st, piece, pbase0, 0, 0, true, g, d);
        dma16(g, d);
    }
    pp_issue_b<CFG>(p, st, st.wuni, 0, 0, true);
    pp_issue_b<CFG>(p, st, st.wuni, 1, 1, true);
    PP_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one phase behind group 0

    while (true) {
        if (st.c == 0) pp_tap<CFG, 0, true>(p, st);
        else pp_tap<CFG, 0, false>(p, st);
        st.post_epi = 0;
        pp_tap<CFG, 1>(p, st);
        pp_tap<CFG, 2>(p, st);
        pp_tap<CFG, 3>(p, st);
        pp_tap<CFG, 4>(p, st);
        pp_tap<CFG, 5>(p, st);
        pp_tap<CFG, 6>(p, st);
        pp_tap<CFG, 7>(p, st);
        pp_tap<CFG, 8>(p, st);
        if (st.c == p.csteps - 1) {
            pp_epilogue<CFG, MODE>(p, st, smem_epi);
            st.post_epi = 1;
        }
        if (!st.has_next) break;
        st.k = st.kn; st.c = st.cn; st.pb ^= 1;
        st.wuni = st.wuni_next;
        if (st.c == 0) set_tile(st.k);
        set_next();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

// ---- 64 -> 64 channels: weight-stationary streaming kernel (round 3) -----------------------------------------------------
// The 3x3 layers of the first ResNet stage (src/resnet.py:171 at 56x56: 64 input and 64 output channels) have K = 576: the whole
// filter bank is 72 KiB.  ONE 512-thread workgroup per CU (two waves per SIMD, 256 registers each); a wave owns 64 pixels x 32
// output channels and keeps ITS half of the bank -- 36 fragments, 144 registers -- in registers for the whole launch: no weight
// traffic through LDS at all (the halo kernel moves 72 KiB of weights through LDS per 128-pixel tile, more than the pixels).
// The layer is HBM-bound on paper (floor ~41 us forward at 5 TB/s, 24 us of MFMA time at the dense peak).
// A job = (256-pixel tile, 32-channel chunk): its patch (<= 32 KiB in padded coordinates, as above) sits in one of THREE
// buffers, filled by LDS-DMA two jobs ahead; a job is 36 MFMAs per wave with no barrier inside, one s_barrier between jobs.
// Every wave issues exactly PPW patch pieces per job (dummies into the dump block), so the one counted wait -- vmcnt(PPW) at
// the END of a job's MFMAs, BEFORE any store of an epilogue: it counts loads only, see pp_epilogue -- retires the next job's
// patch and leaves the one after it in flight.  The source offsets of a tile's pieces are computed once per tile (two
// fast divisions per piece), between the MFMAs of the previous tile.  Per-wave epilogue shared with the ping-pong core
// (pp_epilogue: plain / BN statistics / full data-gradient epilogue); the second wave of a SIMD covers it, nothing else does.
struct WSCfg {
    static constexpr int MI = 2, NJ = 1, GM = 4, GN = 2, KC = 32;
    static constexpr int BM = GM * MI * 32, BN = 64;
    static constexpr int LRB = 6, CPR = 4, RPP = 16, SWS = 2, SWM = (CPR - 1) << 4;      // 64-byte patch rows
    static constexpr int WROWS = MI * 32;
    static constexpr int PPW = 4;                          // patch pieces per wave and job (8 waves: <= 32 KiB)
    static constexpr int NBUF = 3;
    // of the 18 (chunk, tap) weight groups the LAST WL live in LDS instead of registers: the epilogues need the registers
    // (the full one ~100 of them; with everything in registers hipcc spilled fragments to scratch and re-read them inside
    // the loop, behind vmcnt waits that also retired the patch pieces in flight)
    static constexpr int WL_FULL = 13, WL_STATS = 6, WL_MAX = 13;
};

struct WSState {
    f32x16 acc[2][1];
    int prow[2];
    const unsigned char* src;
    uint32_t lds_patch, lds_dump, npgb;
    int wave, lane, l31, lh, wm, wn;
    int tile_m, tile_n;
};

template <int OFF> __device__ __forceinline__ void ws_no_read128(u32x4& d, uint32_t addr) { asm volatile("" : "+v"(d) : "v"(addr)); }
// s_waitcnt lgkmcnt(n) naming the fragments it retires (n: the reads issued after them; a constant once the K loop is unrolled)
#define WS_WAIT_CASES(OPS)                                                                                                 \
    switch (n) {                                                                                                           \
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : OPS : : "memory"); break;                                            \
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : OPS : : "memory"); break;                                            \
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : OPS : : "memory"); break;                                            \
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : OPS : : "memory"); break;                                            \
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : OPS : : "memory"); break;                                            \
        default: asm volatile("s_waitcnt lgkmcnt(6)" : OPS : : "memory"); break;                                           \
    }
#define WS_OPS2 "+v"(a), "+v"(b)
#define WS_OPS3 "+v"(a), "+v"(b), "+v"(c)
__device__ __forceinline__ void ws_wait_frag(int n, u32x4& a, u32x4& b) { WS_WAIT_CASES(WS_OPS2) }
__device__ __forceinline__ void ws_wait_frag3(int n, u32x4& a, u32x4& b, u32x4& c) { WS_WAIT_CASES(WS_OPS3) }
#undef WS_OPS2
#undef WS_OPS3
#undef WS_WAIT_CASES

template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ws3x3_kernel(const PPParams p) {
    typedef WSCfg CFG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* smem = smem_raw;
    WSState st;
    const int tid = threadIdx.x;
    st.lane = tid & 63;
    st.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    st.l31 = st.lane & 31;
    st.lh = st.lane >> 5;
    st.wm = st.wave & 3;
    st.wn = st.wave >> 2;
    st.tile_n = 0;
    st.src = (const unsigned char*)p.src;
    st.npgb = (uint32_t)p.npg * 1024u;
    const uint32_t lds_raw = (uint32_t)(uintptr_t)(lds_u8*)smem;
    st.lds_patch = (lds_raw + 127u) & ~127u;
    smem += st.lds_patch - lds_raw;
    const uint32_t lds_epi = st.lds_patch + CFG::NBUF * st.npgb;
    st.lds_dump = lds_epi + 8 * PP_EPI;
    unsigned char* smem_epi = smem + CFG::NBUF * st.npgb;
    // weight groups kept in LDS (fragment-major, lane * 16 B: conflict-free), read beside the pixel fragments of their K steps
    constexpr int WL = MODE == PP_EPI_FULL ? CFG::WL_FULL : CFG::WL_STATS;
    constexpr int DEPTH = MODE == PP_EPI_FULL ? 1 : 2;                       // K steps the LDS reads run ahead of the MFMAs
    const uint32_t lds_wl = st.lds_dump + 2048;                              // 1 KiB dump block, 1 KiB coefficient table
    float* coef_tab = (float*)(smem_epi + 8 * PP_EPI + 1024);
    if (MODE == PP_EPI_FULL && tid < 64) pp_coef_table(p, tid, coef_tab);

    int L;
    {
        const int q = p.nwg >> 3, r = p.nwg & 7, xcd = blockIdx.x & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int k0 = (int)((long long)L * p.tiles_total / p.nwg), k1 = (int)((long long)(L + 1) * p.tiles_total / p.nwg);
    if (k0 >= k1) return;

    auto patch_base = [&](int tile) { return pp_index(p, tile * CFG::BM) - p.P - 1; };
    // Source of this lane's 16 bytes of piece u * 8 + wave of tile `tile` (chunk 0; chunk 1 is 64 bytes further): byte offset
    // into the tensor (< 2 GiB, checked on the host), or -1 for a padding row / a piece past the patch / a tile past the run
    int poff[CFG::PPW];
    auto piece_offset = [&](int tile, int u) {
        const int piece = u * 8 + st.wave;
        const int row = piece * CFG::RPP + st.lane / CFG::CPR;
        const uint32_t q = (uint32_t)(patch_base(tile < k1 ? tile : k0) + row);
        const uint32_t img = fd_div(q, p.fd_ip), rem = q - img * p.fd_ip.d;
        const uint32_t r = fd_div(rem, p.fd_p), cc = rem - r * p.fd_p.d;
        const int ch = (st.lane % CFG::CPR) ^ ((row >> (4 - CFG::SWS)) & (CFG::CPR - 1));
        const bool ok = (tile < k1) & (piece < p.npg) & (r >= 1) & (cc >= 1) & (img < (uint32_t)p.N);
        const uint32_t pix = (img * p.H + (r - 1)) * p.W + (cc - 1);
        return ok ? (int)((pix * p.C + ch * 8) * 2) : -1;
    };
    // PPW pieces of chunk c of the tile whose offsets poff[] holds, into patch buffer `buf`
    auto issue_patch = [&](int c, int buf) {
#pragma unroll
        for (int u = 0; u < CFG::PPW; ++u) {
            const int piece = u * 8 + st.wave;
            const unsigned char* g = poff[u] >= 0 ? st.src + (uint32_t)(poff[u] + c * (CFG::KC * 2)) : (const unsigned char*)pp_zero16;
            dma16(g, piece < p.npg ? st.lds_patch + buf * st.npgb + piece * 1024 : st.lds_dump);
        }
    };
#pragma unroll
    for (int u = 0; u < CFG::PPW; ++u) poff[u] = piece_offset(k0, u);
    issue_patch(0, 0);
    issue_patch(1, 1);
#pragma unroll
    for (int u = 0; u < CFG::PPW; ++u) poff[u] = piece_offset(k0 + 1, u);

    // this wave's half of the filter bank, straight into registers: fragment (chunk c, tap t, k step s) of lane (l31, lh) is the
    // 8 channels c*32 + s*16 + lh*8.. of tap t of output channel wn*32 + l31 (image: [n][tap][C], as the core above reads it)
    u32x4 wf[2][9][2];
    {
        const bf16* wpk = (const bf16*)p.wpk;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (c * 9 + t >= 18 - WL) {
                        // fragment (c, t, s) of BOTH channel halves, placed by the eight waves in turn
                        const int f = (c * 9 + t - (18 - WL)) * 2 + s;
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            if (((f * 2 + h) & 7) == st.wave)
                                *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(lds_wl + (f * 2 + h) * 1024 + st.lane * 16) =
                                    *(const u32x4*)(wpk + (size_t)(((h * 32 + st.l31) * 9 + t) * 64 + c * 32 + s * 16 + st.lh * 8));
                    } else {
                        wf[c][t][s] = *(const u32x4*)(wpk + (size_t)(((st.wn * 32 + st.l31) * 9 + t) * 64 + c * 32 + s * 16 + st.lh * 8));
                    }
                }
    }
    __syncthreads();

    auto set_tile = [&](int tile) {
        st.tile_m = tile;
        const int m0 = tile * CFG::BM;
        const int pbase = patch_base(tile);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + st.wm * CFG::WROWS + i * 32 + st.l31;
            m = m < p.M ? m : p.M - 1;
            st.prow[i] = pp_index(p, m) - pbase;
        }
    };

    // the filter bank is complete HERE: hipcc waits for a load at its first use, which would be inside the loop, where its
    // count knows nothing of the patch pieces in flight (it would retire them with the weights, every job)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (c * 9 + t < 18 - WL) asm volatile("" : "+v"(wf[c][t][0]), "+v"(wf[c][t][1]));
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = p.toff[t];
    // full epilogue: the BN coefficients as a table in LDS, written once per launch (above); the side operands of a tile (gradient addend,
    // BN input, their masks: 16 loads per lane) are issued at the START of the tile's second job, a whole job ahead of the
    // epilogue that consumes them -- nothing but the other wave of the SIMD would cover their round trip there
    PPSide side;
    PP_WAIT_VM(0);                                 // (prologue only) the first tile's two patches
    int buf = 0;                                   // buffer of job j; job j+2 goes into the one job j-1 read
    for (int tile = k0; tile < k1; ++tile) {
        set_tile(tile);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) st.acc[i][0][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // every wave has waited for its own pieces of this job's patch and has finished reading the previous job's buffer
            __builtin_amdgcn_s_barrier();
            const int nb = buf == 0 ? CFG::NBUF - 1 : buf - 1;
#if !(WS_ABL & 1)
            issue_patch(c, nb);                    // chunk c of tile + 1: two jobs ahead
#else
            (void)nb;
#endif
            const uint32_t pbuf = st.lds_patch + (uint32_t)buf * st.npgb;
            // 18 K steps (tap t = q / 2, k step s = q & 1); the fragments of step q + DEPTH are read from LDS while step q's
            // MFMAs run (two steps ahead where the registers allow: 56 vs 59 us for the K loops alone)
            constexpr int NF = DEPTH + 1;
            u32x4 fa[NF][2], fw[NF];
#if (WS_ABL & 4)
#pragma unroll
            for (int i = 0; i < NF; ++i) fa[i][0] = fa[i][1] = fw[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, pbuf};
#define ds_read128 ws_no_read128
#endif
            auto in_lds = [&](int q) { return c * 9 + (q >> 1) >= 18 - WL; };          // step q's weights come from LDS
            auto nreads = [&](int q) { return q < 18 ? (in_lds(q) ? 3 : 2) : 0; };
            auto read_step = [&](int q, u32x4 (&a)[2], u32x4& w) {
                const int t1 = q >> 1, s1 = q & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) ds_read128<0>(a[i], pp_frag_addr<CFG>(pbuf, (uint32_t)(st.prow[i] + toff[t1]), st.lh) ^ (uint32_t)(s1 << 5));
                if (in_lds(q)) ds_read128<0>(w, lds_wl + (uint32_t)((((c * 9 + t1 - (18 - WL)) * 2 + s1) * 2 + st.wn) * 1024 + st.lane * 16));
            };
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) read_step(q, fa[q], fw[q]);
#pragma unroll
            for (int q = 0; q < 18; ++q) {
                if (q + DEPTH < 18) read_step(q + DEPTH, fa[(q + DEPTH) % NF], fw[(q + DEPTH) % NF]);
                const int left = nreads(q + 1) + (DEPTH > 1 ? nreads(q + 2) : 0);     // reads issued after step q's
                if (in_lds(q)) ws_wait_frag3(left, fa[q % NF][0], fa[q % NF][1], fw[q % NF]);
                else ws_wait_frag(left, fa[q % NF][0], fa[q % NF][1]);
                if (MODE == PP_EPI_FULL && c == 1 && q == WS_SIDE_Q) pp_side_issue<CFG, WSState>(p, st, 0, 0, side);
                // next tile's piece offsets, one piece per early K step of the second chunk: VALU work in the MFMAs' shadow
                if (c == 1 && q >= 1 && q <= CFG::PPW) poff[q - 1] = piece_offset(tile + 2, q - 1);
#if (WS_ABL & 8)
                asm volatile("" ::"v"(in_lds(q) ? fw[q % NF] : wf[c][q >> 1][q & 1]), "v"(fa[q % NF][0]), "v"(fa[q % NF][1]));
#else
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    st.acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, in_lds(q) ? fw[q % NF] : wf[c][q >> 1][q & 1]),
                                                                            __builtin_bit_cast(bf16x8, fa[q % NF][i]), st.acc[i][0], 0, 0, 0);
#endif
            }
#if (WS_ABL & 4)
#undef ds_read128
#endif
            // the NEXT job's patch (issued a whole job ago) has landed; the one just issued (and the side operands behind it)
            // stay in flight.  Before any store.
            if (MODE == PP_EPI_FULL && c == 1) PP_WAIT_VM(CFG::PPW + 16);
            else PP_WAIT_VM(CFG::PPW);
            buf = buf == CFG::NBUF - 1 ? 0 : buf + 1;
        }
#if !(WS_ABL & 16)
        pp_epilogue<CFG, MODE, WSState, false, MODE == PP_EPI_FULL>(p, st, smem_epi, &side, coef_tab);
#else
        if (st.acc[0][0][0] == 123.456f) pp_epilogue<CFG, PP_EPI_PLAIN, WSState, false>(p, st, smem_epi);
#endif
    }
    PP_WAIT_VM(0);                                 // trailing dummy pieces must not outlive the workgroup's LDS
}

typedef PPCfg<2, 2, 4, 64> CfgA;      // 256 x 128, wave 64 x 64, 64-channel steps
typedef PPCfg<2, 1, 4, 64> CfgA1;     // 256 x 64 (64 output channels)
typedef PPCfg<4, 2, 4, 32> CfgB;      // 512 x 128, wave 128 x 64, 32-channel steps
typedef PPCfg<4, 2, 2, 32> CfgC;      // 256 x 256, wave 128 x 64, 32-channel steps

struct PPGeom {
    int cfg;          // 0 A, 1 A1, 2 B, 3 C, 4 weight-stationary 64 -> 64
    int bm, bn, kc, wrows, bstage, rpp;
};
const PPGeom PP_GEOM[5] = {{0, CfgA::BM, CfgA::BN, CfgA::KC, CfgA::WROWS, CfgA::BSTAGE, CfgA::RPP},
                           {1, CfgA1::BM, CfgA1::BN, CfgA1::KC, CfgA1::WROWS, CfgA1::BSTAGE, CfgA1::RPP},
                           {2, CfgB::BM, CfgB::BN, CfgB::KC, CfgB::WROWS, CfgB::BSTAGE, CfgB::RPP},
                           {3, CfgC::BM, CfgC::BN, CfgC::KC, CfgC::WROWS, CfgC::BSTAGE, CfgC::RPP},
                           {4, WSCfg::BM, WSCfg::BN, WSCfg::KC, WSCfg::WROWS, 0, WSCfg::RPP}};

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------
static int pp_patch_rows(int H, int W, int bm) {
    // worst-case padded-index span of bm consecutive pixels, plus the halo on both sides
    const int P = W + 1;
    const int row_wraps = (bm - 1) / W + 1;
    const int img_wraps = (bm - 1) / (H * W) + 1;
    return bm + row_wraps + img_wraps * P + 2 * P + 3;
}

static int pp_npg(int H, int W, const PPGeom& g) { return (pp_patch_rows(H, W, g.bm) + g.rpp - 1) / g.rpp; }

static size_t pp_lds_bytes(int H, int W, const PPGeom& g) {
    if (g.cfg == 4) return (size_t)WSCfg::NBUF * pp_npg(H, W, g) * 1024 + 8 * PP_EPI + 2048 + WSCfg::WL_MAX * 4096 + 128;
    return (size_t)2 * pp_npg(H, W, g) * 1024 + (size_t)3 * g.bstage + 8 * PP_EPI + 1024 + 128;
}

// Configuration for a shape (-1: not taken): bf16 3x3 taps within +-1 on an output grid equal to the source grid,
// channel counts on the 64 grid, a patch that fits the 56 prefetch slots and LDS, tensors below 2 GiB, >= 8 tiles.
// CNK_PP3X3=0 turns the core off; CNK_PP_CFG=0..3 forces a configuration where it fits (tools/bench_pp.py A/B runs).
static int pp_pick(int N, int H, int W, int C, int Nout, int dgrad, int fused) {
    const int enabled = cnk_sw().pp3x3, forced = cnk_sw().pp_cfg;
    if (!enabled) return -1;
    if (C % 64 || Nout % 64) return -1;
    const long long M = (long long)N * H * W;
    if (M >= (1ll << 30) || M * (C > Nout ? C : Nout) * 2 >= (1ll << 31)) return -1;
    auto fits = [&](int c) {
        const PPGeom& g = PP_GEOM[c];
        if (Nout % g.bn) return false;
        if (pp_npg(H, W, g) > 56 || pp_lds_bytes(H, W, g) > 160 * 1024) return false;
        return (M + g.bm - 1) / g.bm * (Nout / g.bn) >= 8;
    };
    if (forced >= 0 && forced < 4 && fits(forced) && !(dgrad && forced >= 2)) return forced;      // (bench / test runs only)
    // 64 -> 64 channels: the weight-stationary kernel (filter bank in registers), forward and data gradient alike; its patch
    // must fit the WSCfg::PPW pieces a wave issues per job
    // The fused data gradients too (CNK_PP_FULL=0 sends them back to the halo kernel): solo the full-epilogue variant wins
    // (113-120 vs 126 us).  In round 3's step it did not (153 vs 147 us on the same launches of a traced replay, step time
    // equal); in round 4's step -- the folded BN backward moved traffic from the compute stream to the weight-gradient stream,
    // the halo kernel's three launches now take 167-170 us there -- it does by a little: 19.32 -> 19.23 ms over four
    // alternating pairs.
    if (cnk_sw().pp_ws && !(dgrad && fused && cnk_sw().pp_full == 0) && forced < 0 && C == 64 && Nout == 64 &&
        pp_npg(H, W, PP_GEOM[4]) <= 8 * WSCfg::PPW && fits(4))
        return 4;
    // Rule in work per CU (one persistent workgroup per CU, 256 CUs): the biggest tile -- 256 x 256, then 512 x 128, both
    // with the cheaper 128 x 64 wave tile -- as long as its tiles cover at least 5/8 of the CUs; below that the 256 x 128
    // tile's finer grain fills the chip better (7x7 layers at batch 256: 98 big tiles, 141 vs 75 us).
    constexpr int PP_CUS = 256;
    const int order[4] = {3, 2, 0, 1};
    for (int c : order) {
        if (!fits(c)) continue;
        const PPGeom& g = PP_GEOM[c];
        const long long tiles = (M + g.bm - 1) / g.bm * (Nout / g.bn);
        if ((c == 3 || c == 2) && tiles * 8 < 5 * PP_CUS) continue;
        // the data gradient's full epilogue (addend, fused BN backward) does not fit the registers beside 128
        // accumulators: 64 x 64 wave tiles there
        if ((c == 3 || c == 2) && dgrad) continue;
        // Measured (tools/bench_pp.py, interleaved with the r01 halo kernels in one process, bs 256): 64 output
        // channels (9 K steps per tile, epilogue-dominated) 111 vs 104 us forward: stays on the halo kernel.  With
        // the BN backward fused into the epilogue (every 3x3 data gradient of a training step) this core's
        // per-wave epilogue stalls both groups, the halo kernel hides its epilogue behind two other workgroups:
        // 136 vs 114 us at 28x28, 94 vs 92 at 14x14, 76 vs 83 at 7x7 -- taken only when every workgroup has ONE tile.
        if (c == 1) continue;
        if (dgrad && fused && cnk_sw().pp_full == 0) continue;
        if (dgrad && fused && tiles > 256 && cnk_sw().pp_full != 1) continue;
        return c;
    }
    return -1;
}

bool pp3x3_ok(int N, int H, int W, int C, int Nout, int dgrad, int fused) { return pp_pick(N, H, W, C, Nout, dgrad, fused) >= 0; }

// slices of the partials this core writes for the shape (forward: BN statistics; data gradient: fused BN backward), 0: not taken
int64_t pp3x3_slices(int N, int H, int W, int C, int Nout, int dgrad) {
    const int c = pp_pick(N, H, W, C, Nout, dgrad, 1);
    if (c < 0) return 0;
    return ((int64_t)N * H * W + PP_GEOM[c].wrows - 1) / PP_GEOM[c].wrows;
}

template <class CFG> static int pp_launch_cfg(const PPParams& p, int mode, size_t lds, hipStream_t s) {
#define PP_LAUNCH(MODEV)                                                                                                   \
    do {                                                                                                                   \
        CNK_LDS_OPTIN((pp3x3_kernel<CFG, MODEV>), 160 * 1024, "pp3x3");                                                    \
        hipLaunchKernelGGL((pp3x3_kernel<CFG, MODEV>), dim3(p.nwg), dim3(512), lds, s, p);                                 \
    } while (0)
    if (mode == PP_EPI_FULL) PP_LAUNCH(PP_EPI_FULL);
    else if (mode == PP_EPI_STATS) PP_LAUNCH(PP_EPI_STATS);
    else PP_LAUNCH(PP_EPI_PLAIN);
#undef PP_LAUNCH
    CNK_CHECK_LAUNCH("pp3x3");
    return 0;
}

int pp3x3_launch(const PPArgs& a, hipStream_t s) {
    const int c = pp_pick(a.N, a.H, a.W, a.C, a.Nout, a.dgrad, a.bnf_x != nullptr);
    CNK_CHECK_ARG(c >= 0, "pp3x3: shape not supported");
    const PPGeom& g = PP_GEOM[c];
    PPParams p = {};
    p.src = a.src; p.wpk = a.wpk; p.dst = a.dst; p.bias = a.bias; p.addend = a.addend; p.addend_mask = a.addend_mask;
    p.stats = a.stats; p.bnf_x = a.bnf_x; p.bnf_mean = a.bnf_mean; p.bnf_invstd = a.bnf_invstd; p.bnf_gamma = a.bnf_gamma;
    p.bnf_beta = a.bnf_beta; p.bnf_mask = a.bnf_mask; p.bnf_partial = a.bnf_partial; p.bnf_relu = a.bnf_relu;
    p.bnf_tile0 = a.bnf_tile0; p.relu = a.relu;
    p.N = a.N; p.H = a.H; p.W = a.W; p.C = a.C; p.Nout = a.Nout;
    p.M = a.N * a.H * a.W;
    p.P = a.W + 1; p.IP = (a.H + 1) * p.P;
    p.csteps = a.C / g.kc;
    p.tiles_n = a.Nout / g.bn;
    p.tiles_total = (p.M + g.bm - 1) / g.bm * p.tiles_n;
    p.nwg = p.tiles_total < 256 ? p.tiles_total : 256;
    p.npg = pp_npg(a.H, a.W, g);
    for (int t = 0; t < 9; ++t) p.toff[t] = a.dh[t] * p.P + a.dw[t];
    p.fd_hw = make_fastdiv(a.H * a.W);
    p.fd_w = make_fastdiv(a.W);
    p.fd_ip = make_fastdiv(p.IP);
    p.fd_p = make_fastdiv(p.P);
    const size_t lds = pp_lds_bytes(a.H, a.W, g);
    const int mode = (a.bias || a.addend || a.bnf_x || a.relu) ? PP_EPI_FULL : (a.stats ? PP_EPI_STATS : PP_EPI_PLAIN);
    ProfScope kscope(CNK_KF_HALO, s, 2.0 * p.M * (double)a.Nout * a.C * 9, a.algo_bytes);
    if (c < 4) cnk_count(CNK_CNT_PP_A + c);
    if (mode == PP_EPI_FULL) cnk_count(CNK_CNT_PP_FULL);
    if (c == 4) {
        cnk_count(CNK_CNT_PP_WS);
#define WS_LAUNCH(MODEV)                                                                                                   \
    do {                                                                                                                   \
        CNK_LDS_OPTIN((ws3x3_kernel<MODEV>), 160 * 1024, "ws3x3");                                                         \
        hipLaunchKernelGGL((ws3x3_kernel<MODEV>), dim3(p.nwg), dim3(512), lds, s, p);                                      \
    } while (0)
        if (mode == PP_EPI_FULL) WS_LAUNCH(PP_EPI_FULL);
        else if (mode == PP_EPI_STATS) WS_LAUNCH(PP_EPI_STATS);
        else WS_LAUNCH(PP_EPI_PLAIN);
#undef WS_LAUNCH
        CNK_CHECK_LAUNCH("ws3x3");
        return 0;
    }
    switch (c) {
        case 0: return pp_launch_cfg<CfgA>(p, mode, lds, s);
        case 1: return pp_launch_cfg<CfgA1>(p, mode, lds, s);
        case 2: return pp_launch_cfg<CfgB>(p, mode, lds, s);
        default: return pp_launch_cfg<CfgC>(p, mode, lds, s);
    }
}
