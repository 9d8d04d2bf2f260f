// "Row-patch" MFMA convolution for the first layer of a network (Cin = 1..4, Cout <= 64): ResNet's
// 7x7/2 stem, AlexNet's 11x11/4, VGG's first 3x3.  Supersedes the element-gather kernels of
// conv_smallc.hip whenever the input row fits the patch (they remain the fallback).
//
// In NHWC the kw*Cin values a window needs from one input row are CONTIGUOUS: for output pixel ow
// and window row kh they are patch[kh][(ow*s - p)*Cin ... + kw*Cin).  So a workgroup stages the
// KH input rows of one output row in LDS with coalesced 16-byte loads (zero rows/columns for the
// padding) and the MFMA fragments are read straight out of that patch -- no im2col, no gather:
//   K' = kh*Rp + j,  j in [0,Rp),  Rp = kw*Cin (+lead) rounded up to the 16-byte chunk;
//   the weight image [64 co][K'pad] has zeros at the padded j, so whatever finite patch values
//   the over-read touches are multiplied by zero.
// `lead` (0/1) shifts the run start one element left when that makes every run start 4-byte
// aligned (bf16, even s*Cin): the fragment is then four ds_read_b32; otherwise eight ds_read_u16.
//   forward : acc^T[co][px]  += W[co][k'] * patch(px,k')      (px on the MFMA lane)
//   wgrad   : dW[co][k']     += dy^T[co][px] * patch(px,k')   (reduction over px; dy through the
//             transposing LDS read, patch column fragments by strided element reads); every
//             workgroup owns a run of output rows and writes one fp32 slab, summed by
//             rowpatch_reduce in fixed order.
#include "common.h"

namespace {

struct RowPatchParams {
    const void* x;
    const void* wpk;       // [64][Kp] weights (T)
    void* y;
    const void* dy;
    float* slab;           // wgrad: [grid][64][Kp]
    const float* bias;
    float* stats;          // forward: optional BN partials [grid][Cout][3]
    int ksteps;            // forward: k-steps of two 16-B chunks that cover the KH*Rp real columns
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, s, p;
    int R, Rp, lead, Kp;   // run length, padded run, lead shift, padded K'
    int A0;                // element offset of iw = -p inside a patch row is (A0 - p*Cin)
    int PWE;               // patch row length in elements (incl. zero tail)
    int rows_total;        // N*Ho
    int rows_per_wg;
};

constexpr int PATCH_A0 = 16;     // data of an input row starts 16 elements into the patch row (16-B aligned)

template <typename T> struct RpCfg;
template <> struct RpCfg<bf16> { static constexpr int CH = 8; };   // elements per 16-B chunk
template <> struct RpCfg<float> { static constexpr int CH = 4; };

// Stage input rows ih0 .. ih0+nrows-1 of image n (zero outside) into patch[r][...]
template <typename T>
__device__ __forceinline__ void load_patch(const RowPatchParams& p, const T* x, T* patch, int n,
                                           int ih0, int nrows) {
    constexpr int CH = RpCfg<T>::CH;
    const int row_elems = p.W * p.Cin;                 // multiple of CH checked on the host
    const int cpr = p.PWE / CH;                        // chunks per patch row
    const int data0 = PATCH_A0 / CH, data1 = data0 + row_elems / CH;
    for (int i = threadIdx.x; i < nrows * cpr; i += blockDim.x) {
        int r = i / cpr, c = i - r * cpr;
        int ih = ih0 + r;
        u32x4 v = {0, 0, 0, 0};
        if ((unsigned)ih < (unsigned)p.H && c >= data0 && c < data1)
            v = *(const u32x4*)(x + ((size_t)(n * p.H + ih) * p.W) * p.Cin + (size_t)(c - data0) * CH);
        *(u32x4*)(patch + (size_t)r * p.PWE + c * CH) = v;
    }
}

// 16 bytes (one MFMA k-chunk) starting at element e of a patch row
template <typename T, bool ALIGN4>
__device__ __forceinline__ u32x4 read_chunk(const T* row, int e) {
    if constexpr (sizeof(T) == 4) {
        const float* q = (const float*)row + e;
        f32x4 v = {q[0], q[1], q[2], q[3]};
        return __builtin_bit_cast(u32x4, v);
    } else if constexpr (ALIGN4) {
        const unsigned int* q = (const unsigned int*)((const unsigned short*)row + e);
        u32x4 v = {q[0], q[1], q[2], q[3]};
        return v;
    } else {
        const unsigned short* q = (const unsigned short*)row + e;
        u32x4 v = {(unsigned)q[0] | ((unsigned)q[1] << 16), (unsigned)q[2] | ((unsigned)q[3] << 16),
                   (unsigned)q[4] | ((unsigned)q[5] << 16), (unsigned)q[6] | ((unsigned)q[7] << 16)};
        return v;
    }
}

template <typename T> struct RpMma;
template <> struct RpMma<bf16> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct RpMma<float> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& a, const u32x4& b) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------------------------------
// forward: one workgroup = rows_per_wg consecutive output rows (n, oh); wave w takes pixel chunks
// w, w+4, ... of 32 pixels.  LDS: weights [64][Kp+pad] | input-row ring [KH+s+1][PWE] | per-wave out stage.
//
// Consecutive output rows of one image share KH - s input rows: the patch is a RING of KH + s rows (row ih lives in
// slot (ih + p) mod (KH + s); one more all-zero row serves the K' padding).  While row oh is computed the s input rows
// that row oh + 1 adds are already in flight global -> registers (issued before the MFMA loop, written to their slots
// -- which row oh does not read -- after it): one barrier per output row, each input row fetched once per workgroup,
// no exposed load latency.  A full refill happens at the first row of a workgroup and at image boundaries.
// STATS: per-channel BN partial (n, mean, M2) of the ROUNDED outputs of this workgroup's rows, one slice per workgroup
// (each lane sums its <= rows_per_wg * ceil(Wo/128) pixels, lanes / waves are merged with Chan's update).
// The steady-state loop's vector memory operations are inline asm: hipcc then tracks none of them and inserts no
// s_waitcnt vmcnt(0) at the loop head (it cannot count across the back edge, and loads and stores share the counter, so
// its conservative wait drained every output store once per row).  The prefetch goes global -> LDS directly (LDS-DMA:
// a wave fills 64 consecutive 16-byte chunks of a ring row, no registers for the compiler to shuffle while the data is
// in flight); the one wait the loop needs -- DMA done before the row-end barrier -- is rp_wait_vm0(), placed BEFORE the
// row's output stores are issued.
#ifndef RP_ABL
#define RP_ABL 0      // debug builds (tools/ablate_rp.sh): 1 no prefetch DMA, 2 one k-step, 4 no output stores, 8 no epilogue (k-loop
                      // dies with it), 16 no statistics sums, 64 token epilogue (k-loop stays),
                      // 32 no A-fragment reads, 128 no MFMA (a VALU op consumes the fragments); weight gradient: 256 no patch element
                      // reads, 512 no MFMA, 1024 operands staged for the first row only
#endif
typedef __attribute__((address_space(3))) unsigned char rp_lds_u8;
__device__ __forceinline__ void rp_dma16(const void* g, uint32_t lds_addr) {          // M0 saved / restored inside
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ __forceinline__ void rp_store16(void* g, const u32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(v) : "memory");
}
__device__ __forceinline__ void rp_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename T>
__device__ __forceinline__ void rp_load_rows(const RowPatchParams& p, const T* x, T* ring, int n, int ih0, int nrows,
                                             int ring_rows) {
    constexpr int CH = RpCfg<T>::CH;
    const int row_elems = p.W * p.Cin;
    const int cpr = p.PWE / CH;
    const int data0 = PATCH_A0 / CH, data1 = data0 + row_elems / CH;
    for (int i = threadIdx.x; i < nrows * cpr; i += blockDim.x) {
        int r = i / cpr, c = i - r * cpr;
        int ih = ih0 + r;
        u32x4 v = {0, 0, 0, 0};
        if ((unsigned)ih < (unsigned)p.H && c >= data0 && c < data1)
            v = *(const u32x4*)(x + ((size_t)(n * p.H + ih) * p.W) * p.Cin + (size_t)(c - data0) * CH);
        *(u32x4*)(ring + (size_t)((ih + p.p) % ring_rows) * p.PWE + c * CH) = v;
    }
}

// 8 waves per workgroup; a work item is (32-pixel chunk, 32-channel half of Cout) and item it goes to wave it % 8, so a wave
// always works on the same channel half (it & 1 == wave & 1).  Per item: ksteps MFMAs 32x32, the 32 x 32 tile staged
// wave-privately and copied out as 64-byte segments.
constexpr int RP_FWD_WAVES = 4;

// NS > 0: the K' loop is fully unrolled over NS = Kp / (2 CH) k-steps (the ResNet stem: 12, VGG's first conv: 4).  Where
// the two k-chunks of a step start inside the ring depends on the output row only, so it is computed once per row with
// scalar instructions (offA / offB below) and a lane spends one select and one add per step; the generic loop (NS == 0)
// re-derives (kh, chunk) per lane and step -- ~25 vector instructions per MFMA, which made the kernel VALU-bound at a
// quarter of its store-bandwidth floor.
// WIDE (round 3; NS > 0, 64 output channels): a work item is a 32-pixel chunk with BOTH channel halves -- every patch fragment
// read (four ds_read_b32 on 12-byte lane strides, the K loop's cost) feeds two MFMAs instead of one, and a wave writes whole
// 128-byte pixel rows instead of 64-byte halves.  The wave keeps both halves' weight fragments (2 NS x 4 VGPRs).
template <typename T, bool ALIGN4, bool STATS, int NS, bool WIDE = false>
__global__ __launch_bounds__(64 * RP_FWD_WAVES) void rowpatch_fwd_kernel(const RowPatchParams p) {
    constexpr int CH = RpCfg<T>::CH, ES = (int)sizeof(T), NW = RP_FWD_WAVES;
    constexpr int NH = WIDE ? 2 : 1;                             // channel halves per wave
    constexpr int SS = 32 * NH * ES + 16;                        // staged pixel row: this wave's 32 NH channels (+ pad)
    constexpr int CPRO = 32 * NH * ES / 16;                      // 16-B chunks per staged pixel
    static_assert(!WIDE || NS > 0, "the wide item needs the unrolled K loop");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int WS = p.Kp * ES + 16;                               // weight row stride (odd # of 16-B slots)
    const int RING = p.KH + p.s;
    unsigned char* Wl = smem;                                    // [Kp / (2 CH) k-steps][2 halves][2][32] x 16 B
    T* ring = (T*)(Wl + 64 * WS);                                // [RING + 1][PWE], last row all zero
    unsigned char* stage = (unsigned char*)(ring + (size_t)(RING + 1) * p.PWE);   // [NW waves][32][SS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int half = WIDE ? 0 : (wave & 1);                      // channels half*32 .. half*32 + 32 NH - 1
    const T* x = (const T*)p.x;
    const T* wpk = (const T*)p.wpk;
    {
        const int cpr = p.Kp / CH;
        // weights in MFMA-fragment order: the 16 bytes lane (l31, lh) of channel half h reads at k-step s sit at
        // s*2048 + h*1024 + lh*512 + l31*16, so a wave's ds_read_b128 covers 1 KiB linearly (row-major rows of Kp
        // elements put 4 lanes on every bank: the weight reads alone then cost more LDS cycles than the MFMAs)
        for (int i = tid; i < 64 * cpr; i += 64 * NW) {
            const int co = i / cpr, q = i - co * cpr;
            *(u32x4*)(Wl + (q >> 1) * 2048 + (co >> 5) * 1024 + (q & 1) * 512 + (co & 31) * 16) =
                *(const u32x4*)(wpk + (size_t)co * p.Kp + q * CH);
        }
        for (int i = tid; i < p.PWE; i += 64 * NW) ring[(size_t)RING * p.PWE + i] = from_f32<T>(0.f);
    }
    const int nsteps = p.ksteps;                                 // MFMA k-steps that hold real taps (two chunks each)
    const int cps = p.Rp / CH;                                   // chunks per kh run
    // NS > 0: this wave's weight fragments (its channel half, all k-steps) stay in registers for the whole workgroup --
    // with them in LDS every MFMA needed 2 KiB of LDS reads and the kernel ran at the LDS port's rate, not the MFMA's
    u32x4 wreg[NH][NS > 0 ? NS : 1];
    if constexpr (NS > 0) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) wreg[h][sidx] = *(const u32x4*)(Wl + sidx * 2048 + (half + h) * 1024 + lh * 512 + l31 * 16);
    }
    // bias of this lane's 16 output channels per half (4 groups of 4)
    // (the wide item has no registers to spare for them: it fetches the bias in its epilogue, when there is one)
    f32x4 breg[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (half + h) * 32 + 8 * gq + 4 * lh + e;
                breg[h][gq][e] = (!WIDE && p.bias && c < p.Cout) ? p.bias[c] : 0.f;
            }
    unsigned char* st = stage + wave * 32 * SS;
    const int row0 = blockIdx.x * p.rows_per_wg;
    const int nrows = min(p.rows_per_wg, p.rows_total - row0);
    const int cpr_in = p.PWE / CH, data0 = PATCH_A0 / CH, data1 = data0 + (p.W * p.Cin) / CH;
    const int segs = (cpr_in + 63) / 64;                         // 64-chunk segments per input row
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(rp_lds_u8*)ring;
    const int nitems = (WIDE ? 1 : 2) * ((p.Wo + 31) / 32);
    // STATS (Cout == 64): in the copy-out every lane always moves the same 16-byte channel group (lane % CPRO) of pixels
    // lane / CPRO + k * (64 / CPRO): its sums cover CH channels
    float sS[CH], sQ[CH], cnt = 0.f;
    if constexpr (STATS) {
#pragma unroll
        for (int e = 0; e < CH; ++e) sS[e] = sQ[e] = 0.f;
    }
    int n = row0 / p.Ho, oh = row0 - n * p.Ho;
    bool need_full = true;
    for (int rr = 0; rr < nrows; ++rr) {
        const int row = row0 + rr;
        if (need_full) {
            __syncthreads();                                     // previous row's ring reads done
            rp_load_rows<T>(p, x, ring, n, oh * p.s - p.p, p.KH, RING);
            __syncthreads();
        }
        const bool next_ok = rr + 1 < nrows && oh + 1 < p.Ho;
        if (next_ok) {
            // the s input rows that output row oh + 1 adds: wave-sized segments of 64 chunks, dealt to the waves
            const int ihn = oh * p.s - p.p + p.KH;
            for (int it = wave; it < p.s * segs; it += NW) {
                const int r = it / segs, c = (it - r * segs) * 64 + lane, ih = ihn + r;
                const int slot = (ihn + r + p.p) % RING;
                unsigned char* rowl = (unsigned char*)(ring + (size_t)slot * p.PWE);
                if (c < cpr_in) {
                    if (!(RP_ABL & 1) && (unsigned)ih < (unsigned)p.H && c >= data0 && c < data1)
                        rp_dma16(x + ((size_t)(n * p.H + ih) * p.W) * p.Cin + (size_t)(c - data0) * CH,
                                 ring_lds + (uint32_t)((size_t)slot * p.PWE * ES) + (uint32_t)(c - lane) * 16);
                    else
                        *(u32x4*)(rowl + c * 16) = u32x4{0, 0, 0, 0};      // padding columns / rows outside the image
                }
            }
        }
        const int base = (oh * p.s) % RING;                      // slot of this row's kh = 0
        bool dma_done = false;
        int offA[NS > 0 ? NS : 1], offB[NS > 0 ? NS : 1];        // byte offset in the ring of k-chunk 2s / 2s+1 (uniform)
        if constexpr (NS > 0) {
            int kh = 0, jc = 0, slot = base;
#pragma unroll
            for (int q = 0; q < 2 * NS; ++q) {
                const int off = (kh < p.KH ? slot : RING) * (p.PWE * ES) + jc * 16;
                if (q & 1) offB[q >> 1] = off; else offA[q >> 1] = off;
                if (++jc == cps) {
                    jc = 0;
                    ++kh;
                    if (++slot == RING) slot = 0;
                }
            }
        }
        for (int it = wave; it < nitems; it += NW) {
            const int c0 = (WIDE ? it : (it >> 1)) * 32;
            const int ow = c0 + l31;
            const int owc = ow < p.Wo ? ow : p.Wo - 1;           // clamp: reads stay inside the ring row
            const int e0 = PATCH_A0 - p.p * p.Cin - p.lead + owc * p.s * p.Cin;
            f32x16 accs[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[h][r] = 0.f;
            f32x16& acc = accs[0];
            if constexpr (NS > 0) {
                const unsigned char* arow = (const unsigned char*)ring;
#pragma unroll
                for (int sidx = 0; sidx < ((RP_ABL & 2) ? 1 : NS); ++sidx) {
                    u32x4 fa;
                    if (RP_ABL & 32) fa = u32x4{(unsigned)e0, (unsigned)sidx, 0x3f803f80u, 0x3f803f80u};
                    else fa = read_chunk<T, ALIGN4>((const T*)(arow + (lh ? offB[sidx] : offA[sidx])), e0);
                    if (RP_ABL & 128) acc[sidx & 15] += __builtin_bit_cast(float, fa[0] ^ fa[1] ^ fa[2] ^ fa[3] ^ wreg[0][sidx][0]);
                    else {
#pragma unroll
                        for (int h = 0; h < NH; ++h) RpMma<T>::run(accs[h], wreg[h][sidx], fa);     // transposed: rows = co, cols = pixels
                    }
                }
            } else {
            // this lane's k-chunk q = 2*step + lh walks (kh, chunk-in-run); fragments of step+1 are read before the
            // MFMA of step
            int kh = lh / cps, jc = lh - kh * cps;
            auto frag_a = [&]() {
                int slot = kh < p.KH ? base + kh : RING;
                slot = (slot >= RING && kh < p.KH) ? slot - RING : slot;
                return read_chunk<T, ALIGN4>(ring + (size_t)slot * p.PWE, e0 + jc * CH);
            };
            const unsigned char* wrow = Wl + half * 1024 + lh * 512 + l31 * 16;
            u32x4 fa = frag_a();
            u32x4 fb = *(const u32x4*)(wrow);
            for (int sidx = 0; sidx < ((RP_ABL & 2) ? 1 : nsteps); ++sidx) {
                const u32x4 ca = fa, cb = fb;
                if (sidx + 1 < nsteps) {
                    jc += 2;
                    while (jc >= cps) { jc -= cps; ++kh; }
                    wrow += 2048;
                    fa = frag_a();
                    fb = *(const u32x4*)(wrow);
                }
                RpMma<T>::run(acc, cb, ca);                      // transposed: rows = co, cols = pixels
            }
            }
            if (next_ok && !dma_done) {
                dma_done = true;
                rp_wait_vm0();        // before this item's output stores: waiting later would also wait for their round trip
            }
            if (RP_ABL & 8) continue;
            if (RP_ABL & 64) {                                   // keep the MFMA loop alive with a token epilogue
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[r];
                if (t == 12345.678f) ((float*)p.y)[0] = t;
                continue;
            }
            // lane (pixel l31, lh) holds channels (half + h)*32 + 8g + 4lh + {0..3} in regs 4g..4g+3 of accumulator h
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int col = h * 32 + 8 * gq + 4 * lh;    // within this wave's channels
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = accs[h][4 * gq + e] + (WIDE ? 0.f : breg[h][gq][e]);
                    if (WIDE && p.bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += p.bias[col + e];           // (64 output channels: all live)
                    }
                    if constexpr (ES == 2) {
                        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *(bf16x4*)(st + l31 * SS + col * 2) = o;
                    } else {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *(f32x4*)(st + l31 * SS + col * 4) = o;
                    }
                }
            // wave-private staging: no workgroup barrier needed, only this wave's own LDS ordering
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
            T* y = (T*)p.y + ((size_t)row * p.Wo + c0) * p.Cout + half * 32;
            if (p.Cout == 64) {
#pragma unroll
                for (int i = lane; i < 32 * CPRO; i += 64) {
                    const int pr = i / CPRO, c = i - pr * CPRO;
                    if (c0 + pr < p.Wo) {
                        const u32x4 v = *(const u32x4*)(st + pr * SS + c * 16);
                        if (!(RP_ABL & 4)) rp_store16((unsigned char*)y + (size_t)pr * 64 * ES + c * 16, v);
                        if constexpr (STATS && !(RP_ABL & 16)) {
                            Vec16<T> vv;
                            vv.v = __builtin_bit_cast(decltype(vv.v), v);
                            cnt += 1.f;
#pragma unroll
                            for (int e = 0; e < CH; ++e) {
                                const float f = vv.get(e);
                                sS[e] += f;
                                sQ[e] += f * f;
                            }
                        }
                    }
                }
            } else {
                const int nc = min(32, p.Cout - half * 32);      // this half's live channels (<= 0: none)
                for (int i = lane; i < 32 * nc; i += 64) {
                    int pr = i / nc, c = i - pr * nc;
                    if (c0 + pr < p.Wo) y[(size_t)pr * p.Cout + c] = *(const T*)(st + pr * SS + c * ES);
                }
            }
        }
        if (next_ok) {
            if (!dma_done) rp_wait_vm0();                        // a wave without an item in this row
            // new rows visible, everyone done with row oh.  Only the LDS traffic is waited for: the output stores of this
            // row stay in flight across the barrier (__syncthreads would drain them)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        need_full = !next_ok;
        if (++oh == p.Ho) { oh = 0; ++n; }
    }
    if constexpr (STATS) {
        // lane -> (n, mean, M2) of its own values, then Chan merges: the 64 / CPRO lanes that share a channel group (xor
        // shuffles), the 4 waves of this channel half through LDS.  Fixed order: deterministic.
        float mean[CH], m2[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const float m = cnt > 0.f ? sS[e] / cnt : 0.f;
            const float q = sQ[e] - sS[e] * m;
            mean[e] = m;
            m2[e] = q > 0.f ? q : 0.f;
        }
        float nn = cnt;
#pragma unroll
        for (int off = CPRO; off < 64; off <<= 1) {
            const float nb = __shfl_xor(nn, off, 64);
            const float nt = nn + nb;
            const float wb = nt > 0.f ? nb / nt : 0.f;
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const float mb = __shfl_xor(mean[e], off, 64), qb = __shfl_xor(m2[e], off, 64);
                const float d = mb - mean[e];
                m2[e] = m2[e] + qb + d * d * nn * wb;
                mean[e] = mean[e] + d * wb;
            }
            nn = nt;
        }
        __syncthreads();                                         // the stage area is free now
        float* red = (float*)stage;                              // [NW waves][32 NH ch][3]
        if (lane < CPRO) {
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                float* o = red + ((size_t)wave * (32 * NH) + lane * CH + e) * 3;
                o[0] = nn; o[1] = mean[e]; o[2] = m2[e];
            }
        }
        __syncthreads();
        if (tid < 64) {
            float na = 0.f, ma = 0.f, qa = 0.f;
#pragma unroll
            for (int w = WIDE ? 0 : (tid >> 5); w < NW; w += WIDE ? 1 : 2) {      // the waves that hold channel tid
                const float* o = red + ((size_t)w * (32 * NH) + (WIDE ? tid : (tid & 31))) * 3;
                const float nb = o[0], nt = na + nb, wb = nt > 0.f ? nb / nt : 0.f, d = o[1] - ma;
                qa = qa + o[2] + d * d * na * wb;
                ma = ma + d * wb;
                na = nt;
            }
            float* o = p.stats + ((size_t)blockIdx.x * 64 + tid) * 3;
            o[0] = na; o[1] = ma; o[2] = qa;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// wgrad: dW[co][k'] += sum_px dy[px][co] * patch(px, k').  32x32 output blocks (2 co-blocks x
// Kp/32 k'-blocks) are dealt round-robin to the 4 waves; each wave sweeps every 32-pixel chunk.
// PAIR (bf16): the wave pair (2w, 2w + 1) owns the k' blocks nj = w, w + 2, ... for BOTH co halves, and the two waves of a pair
// take alternate 32-pixel chunks of the row; their accumulators meet in LDS once, at the end.  A wave then reads each dy^T
// fragment once per step for all its blocks and each patch fragment (eight 2-byte LDS reads, the expensive one) once for two
// MFMAs: 4.7 LDS reads per MFMA instead of 8.7 -- this kernel is bound by the LDS pipe, not by the matrix pipe.
template <typename T, int NBLK, bool PAIR = false>
__global__ __launch_bounds__(256) void rowpatch_wgrad_kernel(const RowPatchParams p) {
    static_assert(!PAIR || sizeof(T) == 2, "PAIR: bf16 only");
    constexpr int NACC = PAIR ? 2 * NBLK : NBLK;
    constexpr int CH = RpCfg<T>::CH, ES = (int)sizeof(T);
    constexpr int DS = 64 * ES + (ES == 2 ? 64 : 0);              // dy image row stride (tr-read friendly)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* patch = (T*)smem;                                          // [KH+1][PWE]
    unsigned char* Dl = (unsigned char*)(patch + (size_t)(p.KH + 1) * p.PWE);   // [WoPad][DS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* x = (const T*)p.x;
    const T* dy = (const T*)p.dy;
    const int WoPad = (p.Wo + 31) / 32 * 32;
    const int nblk = 2 * (p.Kp / 32);                            // 32x32 blocks of dW
    f32x16 acc[NACC];
#pragma unroll
    for (int b = 0; b < NACC; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    for (int i = tid; i < p.PWE; i += 256) patch[(size_t)p.KH * p.PWE + i] = from_f32<T>(0.f);
    // column k' = nj*32 + lane%32 of block b: where its kh run starts inside the patch (element offset), once per kernel
    // (the division by the run length used to sit in the innermost loop: ~40 VALU instructions per MFMA)
    int koff[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int kp = (PAIR ? (wave >> 1) + 2 * b : (wave + 4 * b) >> 1) * 32 + (lane & 31);
        int kh = kp / p.Rp;
        const int jj = kp - kh * p.Rp;
        kh = kh < p.KH ? kh : p.KH;
        koff[b] = kh * p.PWE + (PATCH_A0 - p.p * p.Cin - p.lead + jj);
    }
    const int row0 = blockIdx.x * p.rows_per_wg;
    // Operands of output row r + 1 are loaded into registers BEFORE row r is computed and written to LDS after it: the
    // load latency hides behind the MFMA loop (it used to sit between two barriers of every row).  PFX / PFD chunks of
    // 16 bytes per thread for the KH patch rows / the dy row; shapes that need more take the synchronous path.
    constexpr int PFX = 3, PFD = 4, CPRD = 64 * ES / 16;
    const int cpr = p.PWE / CH, xchunks = p.KH * cpr, dchunks = WoPad * CPRD;
    const bool pf = p.Cout == 64 && xchunks <= 256 * PFX && dchunks <= 256 * PFD && !(RP_ABL & 1024);
    int xlds[PFX], xrow[PFX], xcol[PFX];                         // this thread's patch chunks: LDS element offset, kh row, column
#pragma unroll
    for (int t = 0; t < PFX; ++t) {
        const int i = tid + t * 256, r = i / cpr, c = i - r * cpr;
        const int d0 = PATCH_A0 / CH, d1 = d0 + (p.W * p.Cin) / CH;
        xrow[t] = i < xchunks ? r : -1;
        xcol[t] = (c >= d0 && c < d1) ? (c - d0) * CH : -1;       // element offset inside the input row, -1: padding column
        xlds[t] = r * p.PWE + c * CH;
    }
    u32x4 xr[PFX], dr[PFD];
    auto prefetch = [&](int row) {
        const int n = row / p.Ho, oh = row - n * p.Ho, ih0 = oh * p.s - p.p;
#pragma unroll
        for (int t = 0; t < PFX; ++t) {
            xr[t] = u32x4{0, 0, 0, 0};
            const int ih = ih0 + xrow[t];
            if (xrow[t] >= 0 && xcol[t] >= 0 && (unsigned)ih < (unsigned)p.H)
                xr[t] = *(const u32x4*)(x + ((size_t)(n * p.H + ih) * p.W) * p.Cin + xcol[t]);
        }
#pragma unroll
        for (int t = 0; t < PFD; ++t) {
            const int i = tid + t * 256, pr = i / CPRD, c = i - pr * CPRD;
            dr[t] = u32x4{0, 0, 0, 0};
            if (i < dchunks && pr < p.Wo) dr[t] = *(const u32x4*)(dy + ((size_t)row * p.Wo + pr) * 64 + c * CH);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int t = 0; t < PFX; ++t)
            if (xrow[t] >= 0) *(u32x4*)(patch + xlds[t]) = xr[t];
#pragma unroll
        for (int t = 0; t < PFD; ++t) {
            const int i = tid + t * 256, pr = i / CPRD, c = i - pr * CPRD;
            if (i < dchunks) *(u32x4*)(Dl + pr * DS + c * 16) = dr[t];
        }
    };
    if (pf && row0 < p.rows_total) {
        prefetch(row0);
        commit();
    }
    for (int rr = 0; rr < p.rows_per_wg; ++rr) {
        const int row = row0 + rr;
        if (row >= p.rows_total) break;
        const int n = row / p.Ho, oh = row - n * p.Ho;
        const bool more = rr + 1 < p.rows_per_wg && row + 1 < p.rows_total;
        if (pf) {
            __syncthreads();                                     // this row's operands are in LDS
            if (more) prefetch(row + 1);
        } else {
        __syncthreads();
        if (!(RP_ABL & 1024) || rr == 0) load_patch<T>(p, x, patch, n, oh * p.s - p.p, p.KH);
        // dy row -> LDS [px][64 co], zero beyond Wo / Cout
        if ((RP_ABL & 1024) && rr > 0) {
        } else if (p.Cout == 64) {
            constexpr int CPR = 64 * ES / 16;
            for (int i = tid; i < WoPad * CPR; i += 256) {
                int pr = i / CPR, c = i - pr * CPR;
                u32x4 v = {0, 0, 0, 0};
                if (pr < p.Wo) v = *(const u32x4*)(dy + ((size_t)row * p.Wo + pr) * 64 + c * CH);
                *(u32x4*)(Dl + pr * DS + c * 16) = v;
            }
        } else {
            for (int i = tid; i < WoPad * 64; i += 256) {
                int pr = i >> 6, c = i & 63;
                T v = from_f32<T>(0.f);
                if (pr < p.Wo && c < p.Cout) v = dy[((size_t)row * p.Wo + pr) * p.Cout + c];
                *(T*)(Dl + pr * DS + c * ES) = v;
            }
        }
        __syncthreads();
        }
        for (int c0 = 0; c0 < WoPad; c0 += 32) {
            if constexpr (PAIR) {
                if (((c0 >> 5) & 1) != (wave & 1)) continue;      // the other wave of the pair takes this chunk
                const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g >> 1;
                typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const int prow = c0 + ss * 16 + 8 * h;
                    bf16x8 fa[2];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        const unsigned char* a0 = Dl + (prow + q) * DS + (mi * 32 + 16 * (g & 1) + 4 * pp) * 2;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * DS));
                        const s16x8 w8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        fa[mi] = __builtin_bit_cast(bf16x8, w8);
                    }
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) {
                        if ((wave >> 1) + 2 * b < (nblk >> 1)) {
                            const unsigned short* prw = (const unsigned short*)patch + koff[b];
                            const int stp = p.s * p.Cin;
                            unsigned short e[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                int px = prow + u;
                                px = px < p.Wo ? px : p.Wo - 1;   // dy is zero there; keep the read in range
                                e[u] = prw[px * stp];
                            }
                            const u32x4 fbv = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                               (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
                            const bf16x8 fb = __builtin_bit_cast(bf16x8, fbv);
                            acc[2 * b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb, acc[2 * b], 0, 0, 0);
                            acc[2 * b + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb, acc[2 * b + 1], 0, 0, 0);
                        }
                    }
                }
            } else if constexpr (ES == 2) {
                const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g >> 1;
                const int l31 = lane & 31;
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {                  // two 16-pixel reduction steps
                    const int prow = c0 + ss * 16 + 8 * h;       // first of this lane half's 8 pixels
                    // A: dy^T fragment (rows co) through the transposing read.  Its co half mi = blk & 1 = wave & 1 is the same for
                    // all of a wave's blocks (blk = wave + 4 b): read ONCE per step, not once per block
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const unsigned char* a0 = Dl + (prow + q) * DS + ((wave & 1) * 32 + 16 * (g & 1) + 4 * pp) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * DS));
                    const s16x8 w8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, w8);
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) {
                        const int blk = wave + 4 * b;
                        if (blk < nblk) {
                            const int nj = blk >> 1;
                            // B: column k' = nj*32 + l31 of the virtual im2col, pixels prow..prow+7
                            const unsigned short* prw = (const unsigned short*)patch + koff[b];
                            const int stp = p.s * p.Cin;
                            unsigned short e[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                int px = prow + u;
                                px = px < p.Wo ? px : p.Wo - 1;   // dy is zero there; keep the read in range
                                e[u] = (RP_ABL & 256) ? (unsigned short)(px + u) : prw[px * stp];
                            }
                            u32x4 fbv = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                         (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
                            if (RP_ABL & 512) acc[b][ss] += __builtin_bit_cast(float, fbv[0] ^ fbv[1] ^ fbv[2] ^ fbv[3] ^ (unsigned)w8[0] ^ (unsigned)w8[5]);
                            else acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, fbv), acc[b], 0, 0, 0);
                        }
                    }
                }
            } else {
                const int l31 = lane & 31, lk = lane >> 5;
                for (int ss = 0; ss < 16; ++ss) {                 // 32 pixels, two per 32x32x2 step
                    int px = c0 + ss * 2 + lk;
                    const int pxc = px < p.Wo ? px : p.Wo - 1;
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) {
                        const int blk = wave + 4 * b;
                        if (blk < nblk) {
                            const int mi = blk & 1, nj = blk >> 1;
                            float fa = *(const float*)(Dl + px * DS + (mi * 32 + l31) * 4);
                            float fb = ((const float*)patch)[koff[b] + pxc * p.s * p.Cin];
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[b], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (pf && more) {
            __syncthreads();                                     // every wave is done with this row's operands
            commit();
        }
    }
    const int l31 = lane & 31, lh = lane >> 5;
    float* slab = p.slab + (size_t)blockIdx.x * 64 * p.Kp;
    if constexpr (PAIR) {
        // the odd wave of each pair hands its partial sums over through LDS (one pair at a time: NACC * 16 floats per lane,
        // 24 KiB for the ResNet stem -- the operand images are no longer needed)
        float* hand = (float*)smem;
        for (int pr = 0; pr < 2; ++pr) {
            __syncthreads();
            if (wave == 2 * pr + 1) {
#pragma unroll
                for (int b = 0; b < NACC; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hand[(b * 16 + r) * 64 + lane] = acc[b][r];
            }
            __syncthreads();
            if (wave == 2 * pr) {
#pragma unroll
                for (int b = 0; b < NACC; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][r] += hand[(b * 16 + r) * 64 + lane];
            }
        }
        if (wave & 1) return;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const int nj = (wave >> 1) + 2 * b;
            if (nj < (nblk >> 1)) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        slab[(size_t)co * p.Kp + nj * 32 + l31] = acc[2 * b + mi][r];
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int blk = wave + 4 * b;
        if (blk < nblk) {
            const int mi = blk & 1, nj = blk >> 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[(size_t)co * p.Kp + nj * 32 + l31] = acc[b][r];
            }
        }
    }
}

// dw[co][ci][kh][kw] (+)= sum_wg slab[wg][co][kh*Rp + lead + kw*Cin + ci]; LPE lanes per element, each sums every LPE-th
// slab in slab order, then a fixed xor tree: deterministic.  (16 lanes: the ~1000 slabs of the stem are 50 MB and an
// element per 4 lanes left the chip mostly idle -- 80 us.)
template <int LPE>
__global__ __launch_bounds__(256) void rowpatch_reduce_kernel(const float* __restrict__ slab,
                                                              float* __restrict__ dw, int nslabs,
                                                              int Cout, int Cin, int KH, int KW, int Rp,
                                                              int lead, int Kp, int accumulate) {
    const int total = Cout * Cin * KH * KW, q = threadIdx.x & (LPE - 1);
    const int per_wave = 64 / LPE, total_r = (total + per_wave - 1) / per_wave * per_wave;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) / LPE; i < total_r; i += (gridDim.x * blockDim.x) / LPE) {
        const bool live = i < total;
        int ii = live ? i : 0;
        int kw = ii % KW, r1 = ii / KW;
        int kh = r1 % KH, r2 = r1 / KH;
        int ci = r2 % Cin, co = r2 / Cin;
        const float* sp = slab + (size_t)co * Kp + kh * Rp + lead + kw * Cin + ci;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = q;
        for (; s + 3 * LPE < nslabs; s += 4 * LPE) {              // four independent loads in flight
            a0 += sp[(size_t)s * 64 * Kp];
            a1 += sp[(size_t)(s + LPE) * 64 * Kp];
            a2 += sp[(size_t)(s + 2 * LPE) * 64 * Kp];
            a3 += sp[(size_t)(s + 3 * LPE) * 64 * Kp];
        }
        for (; s < nslabs; s += LPE) a0 += sp[(size_t)s * 64 * Kp];
        float a = (a0 + a1) + (a2 + a3);
#pragma unroll
        for (int o = 1; o < LPE; o <<= 1) a += __shfl_xor(a, o, 64);
        if (q == 0 && live) dw[ii] = accumulate ? dw[ii] + a : a;
    }
}

// weights: wpk[co][kh*Rp + lead + kw*Cin + ci] = w[co][ci][kh][kw], zero elsewhere ([64][Kp])
template <typename T>
__global__ void rowpatch_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout,
                                     int Cin, int KH, int KW, int Rp, int lead, int Kp) {
    const int total = 64 * Kp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int co = i / Kp, k = i - co * Kp;
        int kh = k / Rp, j = k - kh * Rp - lead;
        float v = 0.f;
        if (co < Cout && kh < KH && j >= 0 && j < KW * Cin) {
            int kw = j / Cin, ci = j - kw * Cin;
            v = w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw];
        }
        out[i] = from_f32<T>(v);
    }
}

// the staging area is sized for the wide item (32 pixels x 64 channels per wave)
size_t rp_fwd_lds(const RowPatchParams& p, int es) {
    return (size_t)64 * (p.Kp * es + 16) + (size_t)(p.KH + p.s + 1) * p.PWE * es + (size_t)RP_FWD_WAVES * 32 * (64 * es + 16);
}

bool make_rp(const cnk_conv_desc* d, RowPatchParams* out) {
    RowPatchParams p = {};
    const int es = d->dtype == CNK_BF16 ? 2 : 4, ch = 16 / es;
    if (d->Cout > 64 || d->stride_h != d->stride_w || d->pad_h != d->pad_w) return false;
    if ((d->W * d->Cin) % ch) return false;                       // 16-byte row loads
    if (d->pad_h * d->Cin + 1 > PATCH_A0) return false;
    if ((int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 31)) return false;
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout;
    p.KH = d->kh; p.KW = d->kw; p.s = d->stride_h; p.p = d->pad_h;
    p.R = d->kw * d->Cin;
    p.A0 = PATCH_A0;
    // lead: make (A0 - p*Cin - lead) even when every run start then is 4-byte aligned (bf16, even s*Cin)
    p.lead = (es == 2 && ((p.s * p.Cin) % 2 == 0) && ((PATCH_A0 - p.p * p.Cin) & 1)) ? 1 : 0;
    p.Rp = (p.R + p.lead + ch - 1) / ch * ch;
    int kraw = p.KH * p.Rp;
    p.Kp = (kraw + 31) / 32 * 32;
    // patch row: A0 lead-in, (W+p)*Cin data+right pad, then the largest over-read (Rp) and chunk rounding
    int need = PATCH_A0 + (p.W + p.p) * p.Cin + p.Rp + 2 * ch;
    p.PWE = (need + ch - 1) / ch * ch;
    p.rows_total = d->N * Ho;
    size_t lds_f = rp_fwd_lds(p, es);
    size_t lds_w = (size_t)(p.KH + 1) * p.PWE * es + (size_t)((Wo + 31) / 32 * 32) * (64 * es + (es == 2 ? 64 : 0));
    if (lds_f > 150 * 1024 || lds_w > 150 * 1024) return false;
    if (2 * (p.Kp / 32) > 4 * 6) return false;                   // wgrad accumulator budget (NBLK <= 6)
    *out = p;
    return true;
}

bool rp_align4(const cnk_conv_desc* d, const RowPatchParams& p) {
    return d->dtype == CNK_BF16 && ((p.s * p.Cin) % 2 == 0) && (((PATCH_A0 - p.p * p.Cin - p.lead) & 1) == 0);
}

int rp_rows_per_wg(const RowPatchParams& p, int target_wgs) {
    int r = ceil_div(p.rows_total, target_wgs);
    return r < 1 ? 1 : r;
}

}  // namespace

bool rowpatch_supported(const cnk_conv_desc* d) {
    RowPatchParams p;
    int kstep = d->dtype == CNK_BF16 ? 64 : 32;
    if (d->Cin % kstep == 0 && d->Cout % 64 == 0) return false;   // implicit-GEMM tiles fit
    if (d->Cin > 4) return false;
    return make_rp(d, &p);
}

size_t rowpatch_packed_bytes(const cnk_conv_desc* d) {
    RowPatchParams p;
    make_rp(d, &p);
    return (size_t)64 * p.Kp * (d->dtype == CNK_BF16 ? 2 : 4);
}

int rowpatch_pack(const cnk_conv_desc* d, const float* w, void* out, hipStream_t s) {
    RowPatchParams p;
    make_rp(d, &p);
    int grid = ceil_div(64 * p.Kp, 256);
    if (d->dtype == CNK_BF16)
        hipLaunchKernelGGL(rowpatch_pack_kernel<bf16>, dim3(grid), dim3(256), 0, s, w, (bf16*)out, d->Cout, d->Cin, d->kh, d->kw, p.Rp, p.lead, p.Kp);
    else
        hipLaunchKernelGGL(rowpatch_pack_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, d->Cout, d->Cin, d->kh, d->kw, p.Rp, p.lead, p.Kp);
    CNK_CHECK_LAUNCH("rowpatch_pack");
    return 0;
}

static int rp_fwd_grid(const cnk_conv_desc* d, const RowPatchParams& p, int* rows_per_wg) {
    // ~2048 workgroups, each a whole number of output rows; the weight image (64 x Kp) is staged once per workgroup.
    // The 7x7 stems' wide bf16 variant (12 unrolled k-steps, 234 registers: two workgroups per CU) gets ONE resident round of
    // 512 instead -- 243 -> 221 us on the ResNet stem at batch 256, and a quarter of the statistics partials to merge
    const bool stem_wide = cnk_sw().rowpatch_wide && d->Cout == 64 && d->dtype == CNK_BF16 && p.Kp / 16 == 12 && p.Rp / 8 >= 2;
    *rows_per_wg = rp_rows_per_wg(p, stem_wide ? 512 : 2048);
    return ceil_div(p.rows_total, *rows_per_wg);
}

int64_t rowpatch_fwd_stat_slices(const cnk_conv_desc* d) {
    RowPatchParams p;
    if (!make_rp(d, &p) || d->Cout != 64) return 0;      // the statistics ride on the 64-channel copy-out
    int rpw;
    return rp_fwd_grid(d, p, &rpw);
}

int rowpatch_fwd(const cnk_conv_desc* d, const void* x, const void* wpk, const float* bias, void* y, float* stats,
                 hipStream_t s) {
    RowPatchParams p;
    make_rp(d, &p);
    CNK_CHECK_ARG(!stats || d->Cout == 64, "cnk_conv2d_fwd: this conv shape cannot emit BN partials (cnk_conv2d_fwd_bn_partial_slices == 0)");
    p.x = x; p.wpk = wpk; p.y = y; p.bias = bias; p.stats = stats;
    int grid = rp_fwd_grid(d, p, &p.rows_per_wg);
    const int es = d->dtype == CNK_BF16 ? 2 : 4, ch = 16 / es;
    p.ksteps = ceil_div(p.KH * p.Rp, 2 * ch);
    size_t lds = rp_fwd_lds(p, es);
    const bool wide = cnk_sw().rowpatch_wide && d->Cout == 64 && d->dtype == CNK_BF16;
#define RP_FWD_NSW(TT, A4, NSV, WD)                                                                                 \
    do {                                                                                                           \
        if (stats) hipLaunchKernelGGL((rowpatch_fwd_kernel<TT, A4, true, NSV, WD>), dim3(grid), dim3(64 * RP_FWD_WAVES), lds, s, p); \
        else hipLaunchKernelGGL((rowpatch_fwd_kernel<TT, A4, false, NSV, WD>), dim3(grid), dim3(64 * RP_FWD_WAVES), lds, s, p);      \
    } while (0)
#define RP_FWD_NS(TT, A4, NSV)                                                                                      \
    do {                                                                                                           \
        if (NSV > 0 && wide) RP_FWD_NSW(TT, A4, NSV, (NSV > 0));                                                    \
        else RP_FWD_NSW(TT, A4, NSV, false);                                                                       \
    } while (0)
#define RP_FWD(TT, A4)                                                                                             \
    do {                                                                                                           \
        const int ns = p.Kp / (2 * ch);                                                                            \
        if (ns == 12 && p.Rp / ch >= 2) RP_FWD_NS(TT, A4, 12);                                                     \
        else if (ns == 4 && p.Rp / ch >= 2) RP_FWD_NS(TT, A4, 4);                                                  \
        else RP_FWD_NS(TT, A4, 0);                                                                                 \
    } while (0)
    if (d->dtype == CNK_BF16) {
        if (rp_align4(d, p)) RP_FWD(bf16, true);
        else RP_FWD(bf16, false);
    } else {
        RP_FWD(float, true);
    }
#undef RP_FWD_NS
#undef RP_FWD_NSW
#undef RP_FWD
    CNK_CHECK_LAUNCH("rowpatch_fwd");
    return 0;
}

// bf16, 64 output channels, four to eight k' blocks (the 7x7 stems: six): the pair layout of rowpatch_wgrad_kernel
static bool rp_wgrad_pair(const cnk_conv_desc* d, const RowPatchParams& p) {
    const int njw = ceil_div(p.Kp / 32, 2);
    return d->dtype == CNK_BF16 && njw >= 2 && njw <= 4 && d->Cout == 64 && !(RP_ABL & 2048);
}
static int rp_wgrad_grid(const cnk_conv_desc* d, const RowPatchParams& p, int* rows_per_wg) {
    // one resident round: the pair layout holds two workgroups per CU (232 registers), the others three -- and half the
    // slabs to sum behind it (512 instead of 1024 workgroups: 224 -> 201 us on the ResNet stem)
    *rows_per_wg = rp_rows_per_wg(p, rp_wgrad_pair(d, p) ? 512 : 1024);
    return ceil_div(p.rows_total, *rows_per_wg);
}

size_t rowpatch_wgrad_workspace(const cnk_conv_desc* d) {
    RowPatchParams p;
    make_rp(d, &p);
    int rpw;
    int grid = rp_wgrad_grid(d, p, &rpw);
    return (size_t)grid * 64 * p.Kp * sizeof(float);
}

int rowpatch_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate,
                   void* ws, size_t ws_bytes, hipStream_t s) {
    RowPatchParams p;
    make_rp(d, &p);
    int grid = rp_wgrad_grid(d, p, &p.rows_per_wg);
    size_t need = (size_t)grid * 64 * p.Kp * sizeof(float);
    CNK_CHECK_ARG(ws && ws_bytes >= need, "conv2d_wgrad(rowpatch): workspace too small (%zu < %zu)", ws_bytes, need);
    p.x = x; p.dy = dy; p.slab = (float*)ws;
    const int es = d->dtype == CNK_BF16 ? 2 : 4;
    size_t lds = (size_t)(p.KH + 1) * p.PWE * es + (size_t)((p.Wo + 31) / 32 * 32) * (64 * es + (es == 2 ? 64 : 0));
    int nblk = 2 * (p.Kp / 32), per_wave = ceil_div(nblk, 4);
#define RP_LAUNCH(TT, NB) hipLaunchKernelGGL((rowpatch_wgrad_kernel<TT, NB>), dim3(grid), dim3(256), lds, s, p)
    // the pair layout's hand-over needs NACC * 16 * 64 floats of LDS
    const int njw = ceil_div(p.Kp / 32, 2);
    if (rp_wgrad_pair(d, p)) {
        const size_t hand = (size_t)2 * njw * 16 * 64 * sizeof(float);
        if (hand > lds) lds = hand;
        if (njw == 2) hipLaunchKernelGGL((rowpatch_wgrad_kernel<bf16, 2, true>), dim3(grid), dim3(256), lds, s, p);
        else if (njw == 3) hipLaunchKernelGGL((rowpatch_wgrad_kernel<bf16, 3, true>), dim3(grid), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((rowpatch_wgrad_kernel<bf16, 4, true>), dim3(grid), dim3(256), lds, s, p);
    } else
    if (d->dtype == CNK_BF16) {
        if (per_wave <= 1) RP_LAUNCH(bf16, 1); else if (per_wave <= 2) RP_LAUNCH(bf16, 2);
        else if (per_wave <= 3) RP_LAUNCH(bf16, 3); else if (per_wave <= 4) RP_LAUNCH(bf16, 4);
        else RP_LAUNCH(bf16, 6);
    } else {
        if (per_wave <= 1) RP_LAUNCH(float, 1); else if (per_wave <= 2) RP_LAUNCH(float, 2);
        else if (per_wave <= 3) RP_LAUNCH(float, 3); else if (per_wave <= 4) RP_LAUNCH(float, 4);
        else RP_LAUNCH(float, 6);
    }
#undef RP_LAUNCH
    CNK_CHECK_LAUNCH("rowpatch_wgrad");
    int total = d->Cout * d->Cin * d->kh * d->kw;
    if (grid >= 256) {
        int rgrid = ceil_div(total * 16, 256);
        hipLaunchKernelGGL(rowpatch_reduce_kernel<16>, dim3(rgrid), dim3(256), 0, s, (const float*)ws, dw, grid, d->Cout,
                           d->Cin, d->kh, d->kw, p.Rp, p.lead, p.Kp, accumulate);
    } else {
        int rgrid = ceil_div(total * 4, 256);
        hipLaunchKernelGGL(rowpatch_reduce_kernel<4>, dim3(rgrid), dim3(256), 0, s, (const float*)ws, dw, grid, d->Cout,
                           d->Cin, d->kh, d->kw, p.Rp, p.lead, p.Kp, accumulate);
    }
    CNK_CHECK_LAUNCH("rowpatch_reduce");
    return 0;
}
