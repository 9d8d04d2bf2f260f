// extern "C" entry points of include/cnk.h: argument checks, kernel-family dispatch, error text,
// optional HIP-event timing.  No allocation, no synchronisation (except cnk_prof_collect).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"
#include "wgrad_chain.h"

// ---- kernel families (other translation units) ---------------------------------------------------
bool igemm_supported(const cnk_conv_desc* d, int pass);
bool igemm_grouped_ok(const cnk_conv_desc* d);
bool igemm_on_grid(const cnk_conv_desc* d, int pass);
size_t igemm_image_elems(const cnk_conv_desc* d, int pass, int ntaps);
int igemm_fwd(const cnk_conv_desc* d, const void* x, const void* w_fwd, const float* bias, void* y, float* stats, const void* residual, int relu, hipStream_t s);
int64_t igemm_fwd_stat_slices(const cnk_conv_desc* d);
int igemm_dgrad(const cnk_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* addend, const unsigned char* addend_mask, const cnk_bn_bwd_fuse* bn, hipStream_t s);
int igemm_dgrad_bnfold(const cnk_conv_desc* d, const void* g, const void* x2, const void* w_fold, const float* bias_fold, void* dx, const void* addend, const unsigned char* addend_mask, const cnk_bn_bwd_fuse* bn, hipStream_t s);
bool bnfold_supported(const cnk_conv_desc* d);
bool fold3_supported(const cnk_conv_desc* d);
int fold3_workgroups(const cnk_conv_desc* d);
size_t fold3_workspace(const cnk_conv_desc* d);
int fold3_bwd(const cnk_conv_desc* d, const void* g, const void* x, const void* w_fold, const float* bias_fold, void* dx, const void* addend, const cnk_bn_bwd_fuse* bn, const float* w, const float* mean, const float* invstd, const float* coef, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
size_t bnfold_image_bytes(const cnk_conv_desc* d);
int bnfold_prepare(const cnk_conv_desc* d, const float* w, const void* w_dgrad, const float* mean, const float* invstd, const float* coef, void* w_fold, float* bias_fold, hipStream_t s);
int bn_bwd_coef(const void* dy, const void* x, const float* ext_partial, int64_t ext_slices, const float* gamma, const float* sm, const float* si, float* dgamma, float* dbeta, int accumulate, float* coef, int64_t M, int C, int dtype, float* ws, hipStream_t s);
int64_t igemm_dgrad_bn_slices(const cnk_conv_desc* d);
bool dwconv_supported(const cnk_conv_desc* d);
int dwconv_fwd(const cnk_conv_desc* d, const void* x, const float* w, const float* bias, void* y, hipStream_t s);
int dwconv_dgrad(const cnk_conv_desc* d, const void* dy, const float* w, void* dx, hipStream_t s);
size_t dwconv_wgrad_workspace(const cnk_conv_desc* d);
int dwconv_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, float* ws, hipStream_t s);
size_t linear_workspace(int pass, int B, int K, int N);
int linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, float* ws, size_t ws_bytes, hipStream_t s);
int linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int N, float* ws, size_t ws_bytes, hipStream_t s);
int linear_wgrad(const float* dy, const float* x, float* dw, int B, int K, int N, int accumulate, float* ws, size_t ws_bytes, hipStream_t s);
int softmax_ce_sum(const float* logits, const int64_t* target, float* loss, float* dlogits, int32_t* argmax, float* row_loss_ws, int B, int N, hipStream_t s);
int dropout_fwd(const void* x, void* y, unsigned char* mask, int64_t n, float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int dtype, hipStream_t s);
int dropout_bwd(const void* dy, const unsigned char* mask, void* dx, int64_t n, float p, int dtype, hipStream_t s);
int avgpool_dispatch(bool bwd, const void* a, void* o, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s);
int sigmoid_fwd(const float* x, float* y, int64_t n, hipStream_t s);
int sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, hipStream_t s);
int channel_scale_fwd(const void* x, const float* g, void* y, int N, int HW, int C, int dtype, hipStream_t s);
int channel_scale_bwd(const void* dy, const void* x, const float* g, void* dx, float* dg, int N, int HW, int C, int dtype, hipStream_t s);
bool sk_supported(int C, int P, int dtype);
int sk_scale_fwd(const void* x, const float* gate, void* y, int N, int HW, int C, int dtype, hipStream_t s);
int sk_scale_bwd(const void* dy, const void* x, const float* gate, void* dx, float* dgate, int N, int HW, int C, int dtype, float* ws, hipStream_t s);
size_t sk_workspace(int N, int HW, int C, int P);
int sk_pool_fwd(const void* const* u, int P, float* pooled, int N, int HW, int C, int dtype, float* ws, hipStream_t s);
int sk_select_fwd(const void* const* u, const float* const* logits, int P, float* attn, void* out, int N, int HW, int C, int dtype, hipStream_t s);
int sk_select_bwd_logits(const void* const* u, const float* attn, const void* dout, float* const* dlogits, int P, int N, int HW, int C, int dtype, float* ws, hipStream_t s);
int sk_maps_bwd(const void* dout, const float* attn, const float* dpooled, void* const* du, int P, int N, int HW, int C, int dtype, hipStream_t s);
int input_pipeline(const unsigned char* x, void* y, const int* crop_yx, const unsigned char* flip, const int* affine_fix, const float* mean, const float* stdv, int N, int H, int W, int C, int Ho, int Wo, int pad, int dtype, hipStream_t s);
void affine_fixed_coeffs(double angle, double tx, double ty, double scale, double shear_x, double shear_y, int W, int H, int* out6);
int grouped_expand(const float* w, float* wd, int Cout, int Cin, int groups, int T, hipStream_t s);
int grouped_extract(const float* dwd, float* dw, int Cout, int Cin, int groups, int T, hipStream_t s);
int channel_shuffle(const void* x, void* y, int64_t pixels, int C, int groups, int dtype, hipStream_t s);
size_t grad_clip_workspace_bytes();
int grad_clip_norm(float* g, int64_t n, float max_norm, float* norm_out, float* ws, hipStream_t s);
int grad_clip_value(float* g, int64_t n, float c, hipStream_t s);
int dgrad_classes(const cnk_conv_desc* d, DgradClass* out);
bool wgrad_mfma_supported(const cnk_conv_desc* d);
bool wgrad3x3_supported(const cnk_conv_desc* d);
size_t wgrad3x3_workspace(const cnk_conv_desc* d);
int wgrad3x3(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s, const WgradPending* prev = nullptr, WgradPending* out = nullptr);
int wgrad_pending_reduce(const WgradPending& q, hipStream_t s);
size_t wgrad_mfma_workspace(const cnk_conv_desc* d);
int wgrad_mfma(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s, const WgradFold* fold = nullptr, const WgradPending* prev = nullptr, WgradPending* out = nullptr);
int direct_fwd(const cnk_conv_desc* d, const void* x, const void* w_t, const float* bias, void* y, hipStream_t s);
int direct_dgrad(const cnk_conv_desc* d, const void* dy, const void* w_t, void* dx, const void* addend, hipStream_t s);
size_t direct_wgrad_workspace(const cnk_conv_desc* d);
int direct_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
size_t colsum_workspace(int64_t M, int C);
int colsum(const void* a, float* out, int64_t M, int C, int dtype, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
int pack_direct(const cnk_conv_desc* d, const float* w, void* out, hipStream_t s);
int pack_taps(const cnk_conv_desc* d, const float* w, void* out, int ntaps, const signed char* kh, const signed char* kw, int transpose, hipStream_t s);
bool rowpatch_supported(const cnk_conv_desc* d);
size_t rowpatch_packed_bytes(const cnk_conv_desc* d);
int rowpatch_pack(const cnk_conv_desc* d, const float* w, void* out, hipStream_t s);
int rowpatch_fwd(const cnk_conv_desc* d, const void* x, const void* wpk, const float* bias, void* y, float* stats, hipStream_t s);
int64_t rowpatch_fwd_stat_slices(const cnk_conv_desc* d);
size_t rowpatch_wgrad_workspace(const cnk_conv_desc* d);
int rowpatch_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
bool smallc_supported(const cnk_conv_desc* d);
size_t smallc_packed_bytes(const cnk_conv_desc* d);
int smallc_pack(const cnk_conv_desc* d, const float* w, void* out, hipStream_t s);
int smallc_fwd(const cnk_conv_desc* d, const void* x, const void* wpk, const float* bias, void* y, hipStream_t s);
bool smallc_dgrad_desc(const cnk_conv_desc* d, cnk_conv_desc* t);
size_t smallc_dgrad_packed_bytes(const cnk_conv_desc* d);
int smallc_dgrad_pack(const cnk_conv_desc* d, const float* w, void* out, hipStream_t s);
int smallc_dgrad(const cnk_conv_desc* d, const void* dy, const void* wpk, void* dx, hipStream_t s);
size_t smallc_wgrad_workspace(const cnk_conv_desc* d);
int smallc_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
int pack_batch(const cnk_pack_job* jobs_dev, const void* block_map_dev, int nblocks, hipStream_t s);
int pack_job_blocks(const cnk_pack_job& j);
size_t bn_workspace_bytes(int64_t M, int C);
size_t bn_mask_bytes(int64_t M, int C, int dtype);
int bn_fwd_train(const void* x, const float* gamma, const float* beta, const void* res, void* y, unsigned char* mask, float* rm, float* rv, float* sm, float* si, const float* partial, int64_t partial_slices, int64_t M, int C, float eps, float mom, int relu, int dtype, float* ws, hipStream_t s, const float* raff = nullptr);
int bn_fwd_eval(const void* x, const float* gamma, const float* beta, const float* rm, const float* rv, const void* res, void* y, int64_t M, int C, float eps, int relu, int dtype, hipStream_t s);
int bn_bwd(const void* dy, const void* x, const void* y, const unsigned char* mask, const float* gamma, const float* beta, const float* sm, const float* si, void* dx, void* dres, float* dgamma, float* dbeta, int accumulate, const float* ext_partial, int64_t ext_slices, int64_t M, int C, int relu, int dtype, float* ws, hipStream_t s);
int bn_bwd_pool(const void* dy, const unsigned char* idx, const void* x, const void* xmax, const float* gamma, const float* beta, const float* sm, const float* si, void* dx, float* dgamma, float* dbeta, int accumulate, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int relu, int dtype, float* ws, hipStream_t s);
int maxpool_affine_dispatch(const void* x, const float* scale, const float* shift, int relu, void* y, unsigned char* idx, void* xmax, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s);
int maxpool_dispatch(bool bwd, const void* a, void* b, void* o, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int dtype, hipStream_t s);
template <int OP> int eltwise_dispatch(const void* a, const void* b, void* o, int64_t n, int dtype, hipStream_t s);
int gap_dispatch(bool bwd, const void* a, void* o, int N, int HW, int C, int dtype, hipStream_t s);
int nchw_to_nhwc_dispatch(const float* x, void* y, int N, int C, int H, int W, int dtype, hipStream_t s);
int adam_dispatch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step, const float* bias_dev, hipStream_t s);
void adam_bias_corrections(float b1, float b2, int step, float* out);

// ---- A/B switches: read once at load (common.h) ------------------------------------------------------
static CnkSwitches read_switches() {
    auto env = [](const char* name, int dflt) {
        const char* e = getenv(name);
        return (e && *e) ? atoi(e) : dflt;
    };
    CnkSwitches w;
    w.pp3x3 = env("CNK_PP3X3", 1);
    w.pp_cfg = env("CNK_PP_CFG", -1);
    w.pp_full = env("CNK_PP_FULL", -1);
    w.pp_ws = env("CNK_PP_WS", 1);
    w.gemm1x1 = env("CNK_GEMM1X1", 1);
    w.igemm_halo = env("CNK_IGEMM_HALO", 1);
    w.igemm_halo_pf = env("CNK_IGEMM_HALO_PF", -1);
    w.rowpatch_stats = env("CNK_ROWPATCH_STATS", 1);
    w.rowpatch_wide = env("CNK_ROWPATCH_WIDE", 1);
    w.wgrad_big = env("CNK_WGRAD_BIG", 1);
    w.wgrad3 = env("CNK_WGRAD3", 1);
    w.wgrad_slots_pct = env("CNK_WGRAD_SLOTS_PCT", -1);
    w.wgrad3_slots_pct = env("CNK_WGRAD3_SLOTS_PCT", -1);
    w.bn_pool_elementwise = getenv("CNK_BN_POOL_ELEMENTWISE") ? 1 : 0;
    w.bn_fused_final = env("CNK_BN_FUSED_FINAL", 1);
    w.fold3_nwg = env("CNK_FOLD3_NWG", 512);
    return w;
}
static CnkSwitches g_switches = read_switches();          // dynamic initialiser: runs when the library is loaded
const CnkSwitches& cnk_sw() { return g_switches; }
// test hook (not in include/cnk.h): re-resolve the table after os.environ changed.  Not thread-safe against
// concurrent launches -- callers are single-threaded A/B scripts and tests.
extern "C" void cnk_debug_reload_switches() { g_switches = read_switches(); }

static std::atomic<int64_t> g_counters[CNK_CNT_MAX];
void cnk_count(int id) { g_counters[id].fetch_add(1, std::memory_order_relaxed); }
// test / tool hook (not in include/cnk.h): one thread writes the device's 100 MHz wall clock to *slot -- a time stamp that also
// works inside a captured graph, where events cannot be read (tools/probe_graph_schedule.py)
__global__ void debug_stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
extern "C" int cnk_debug_stamp(void* slot, void* stream) {
    hipLaunchKernelGGL(debug_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" int64_t cnk_debug_counter(int id) { return id >= 0 && id < CNK_CNT_MAX ? g_counters[id].load(std::memory_order_relaxed) : -1; }

// ---- error text -----------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void cnk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- HIP-event timing -----------------------------------------------------------------------------
namespace {
struct ProfRec {
    int cls;
    hipEvent_t e0, e1;
    double flops, bytes;
};
std::mutex g_prof_mu;
int g_prof_mask = 0;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void cnk_prof_start(int cls, hipStream_t s, double flops, double bytes) {
    if (!(g_prof_mask & cls)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.cls = cls; r.flops = flops; r.bytes = bytes;
    r.e0 = get_event(); r.e1 = get_event();
    (void)hipEventRecord(r.e0, s);
    g_prof.push_back(r);
}
void cnk_prof_stop(int cls, hipStream_t s) {
    if (!(g_prof_mask & cls)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = g_prof.size(); i-- > 0;)
        if (g_prof[i].cls == cls) {
            (void)hipEventRecord(g_prof[i].e1, s);
            break;
        }
}

static inline hipStream_t S(void* s) { return (hipStream_t)s; }
static inline size_t esize(int dtype) { return dtype == CNK_BF16 ? 2 : 4; }

static int check_desc(const cnk_conv_desc* d, const char* fn) {
    CNK_CHECK_ARG(d, "%s: null descriptor", fn);
    CNK_CHECK_ARG(d->dtype == CNK_F32 || d->dtype == CNK_BF16, "%s: bad dtype %d", fn, d->dtype);
    CNK_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "%s: non-positive dims", fn);
    CNK_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_h >= 0 && d->pad_w >= 0, "%s: bad kernel/stride/pad", fn);
    const int dl = conv_dil(d);
    CNK_CHECK_ARG(d->H + 2 * d->pad_h >= dl * (d->kh - 1) + 1 && d->W + 2 * d->pad_w >= dl * (d->kw - 1) + 1, "%s: kernel larger than padded input", fn);
    CNK_CHECK_ARG(dl == 1 || (igemm_supported(d, CNK_PASS_FWD) && igemm_supported(d, CNK_PASS_DGRAD) && wgrad_mfma_supported(d) && (dl * (d->kh - 1) < 120 && dl * (d->kw - 1) < 120)),
                  "%s: dilation=%d is implemented on the implicit-GEMM kernels only (channel counts that are multiples of 8)", fn, dl);
    CNK_CHECK_ARG(conv_groups(d) == 1 || igemm_grouped_ok(d), "%s: groups=%d needs Cin == Cout, Cin %% 64 == 0 and a group width that divides 64 (depthwise: cnk_dwconv2d_*)", fn, d->groups);
    int64_t Ho = (d->H + 2 * d->pad_h - dl * (d->kh - 1) - 1) / d->stride_h + 1, Wo = (d->W + 2 * d->pad_w - dl * (d->kw - 1) - 1) / d->stride_w + 1;
    CNK_CHECK_ARG((int64_t)d->N * d->H * d->W < (1ll << 31) && (int64_t)d->N * Ho * Wo < (1ll << 31), "%s: pixel count exceeds 2^31", fn);
    return 0;
}

static int conv_cls(const cnk_conv_desc* d, int pass) {
    if (d->kh == 3 && d->kw == 3) return pass == 0 ? CNK_CLS_CONV3X3_FWD : pass == 1 ? CNK_CLS_CONV3X3_DGRAD : CNK_CLS_CONV3X3_WGRAD;
    if (d->kh == 1 && d->kw == 1) return pass == 0 ? CNK_CLS_CONV1X1_FWD : pass == 1 ? CNK_CLS_CONV1X1_DGRAD : CNK_CLS_CONV1X1_WGRAD;
    return CNK_CLS_OTHER;
}
static void conv_cost(const cnk_conv_desc* d, double* flops, double* bytes) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    double M = (double)d->N * Ho * Wo;
    *flops = 2.0 * M * d->Cout * d->Cin * d->kh * d->kw;
    *bytes = ((double)d->N * d->H * d->W * d->Cin + M * d->Cout) * esize(d->dtype) + (double)d->Cout * d->Cin * d->kh * d->kw * esize(d->dtype);
}

extern "C" {

int cnk_version(void) { return 100; }
const char* cnk_last_error(void) { return g_err; }

int cnk_stream_create(void** stream) {
    CNK_CHECK_ARG(stream, "cnk_stream_create: null pointer");
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    CNK_CHECK_ARG(e == hipSuccess, "cnk_stream_create: %s", hipGetErrorString(e));
    *stream = (void*)s;
    return 0;
}

// An empty launch: its only use is its place in a captured graph (include/cnk.h).
namespace { __global__ void graph_anchor_kernel() {} }
int cnk_graph_anchor(void* stream) {
    hipLaunchKernelGGL(graph_anchor_kernel, dim3(1), dim3(64), 0, S(stream));
    CNK_CHECK_LAUNCH("graph_anchor");
    return 0;
}

int cnk_stream_destroy(void* stream) {
    CNK_CHECK_ARG(stream, "cnk_stream_destroy: null pointer");
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    CNK_CHECK_ARG(e == hipSuccess, "cnk_stream_destroy: %s", hipGetErrorString(e));
    return 0;
}

int cnk_conv2d_out_hw(const cnk_conv_desc* d, int32_t* Ho, int32_t* Wo) {
    const int dl = conv_dil(d);
    *Ho = (d->H + 2 * d->pad_h - dl * (d->kh - 1) - 1) / d->stride_h + 1;
    *Wo = (d->W + 2 * d->pad_w - dl * (d->kw - 1) - 1) / d->stride_w + 1;
    return 0;
}

size_t cnk_conv2d_packed_bytes(const cnk_conv_desc* d, int pass) {
    size_t n = (size_t)d->Cout * d->Cin * d->kh * d->kw;
    if (pass != CNK_PASS_FWD && pass != CNK_PASS_DGRAD) return 0;
    if (igemm_supported(d, pass)) {
        if (pass == CNK_PASS_FWD) return igemm_image_elems(d, pass, d->kh * d->kw) * esize(d->dtype);
        DgradClass cls[16];
        if (d->stride_h * d->stride_w > 16) return 0;
        int nc = dgrad_classes(d, cls);
        size_t e = 0;
        for (int c = 0; c < nc; ++c) e += igemm_image_elems(d, pass, cls[c].ntaps);
        return e * esize(d->dtype);
    }
    if (pass == CNK_PASS_FWD && rowpatch_supported(d)) return rowpatch_packed_bytes(d);
    if (pass == CNK_PASS_FWD && smallc_supported(d)) return smallc_packed_bytes(d);
    if (pass == CNK_PASS_DGRAD && smallc_dgrad_packed_bytes(d)) {
        // room for both images: the transposed-forward one (used without an addend) and the direct one
        size_t a = (smallc_dgrad_packed_bytes(d) + 255) & ~(size_t)255;
        return a + n * sizeof(float);
    }
    return n * sizeof(float);
}

size_t cnk_conv2d_workspace_bytes(const cnk_conv_desc* d, int pass) {
    if (pass != CNK_PASS_WGRAD) return 0;
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    // (the 3x3 all-taps kernel and the one-tap kernel may both serve a shape -- CNK_WGRAD3 switches at run time for A/B
    // runs -- so the workspace covers whichever asks for more)
    size_t a3 = wgrad3x3_supported(d) ? wgrad3x3_workspace(d) : 0;
    size_t a = wgrad_mfma_supported(d) ? wgrad_mfma_workspace(d)
               : rowpatch_supported(d) ? rowpatch_wgrad_workspace(d)
               : smallc_supported(d) ? smallc_wgrad_workspace(d) : direct_wgrad_workspace(d);
    if (a3 > a) a = a3;
    a = (a + 255) & ~(size_t)255;
    return a + colsum_workspace((int64_t)d->N * Ho * Wo, d->Cout);
}

int cnk_conv2d_pack_weights(const cnk_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_pack_weights")) return rc;
    CNK_CHECK_ARG(w, "cnk_conv2d_pack_weights: null weight");
    bool direct_done = false;
    if (w_fwd) {
        if (igemm_supported(d, CNK_PASS_FWD)) {
            signed char kh[32], kw[32];
            for (int t = 0; t < d->kh * d->kw; ++t) { kh[t] = (signed char)(t / d->kw); kw[t] = (signed char)(t % d->kw); }
            if (int rc = pack_taps(d, w, w_fwd, d->kh * d->kw, kh, kw, 0, S(stream))) return rc;
        } else if (rowpatch_supported(d)) {
            if (int rc = rowpatch_pack(d, w, w_fwd, S(stream))) return rc;
        } else if (smallc_supported(d)) {
            if (int rc = smallc_pack(d, w, w_fwd, S(stream))) return rc;
        } else {
            if (int rc = pack_direct(d, w, w_fwd, S(stream))) return rc;
            direct_done = true;
        }
    }
    if (w_dgrad) {
        if (igemm_supported(d, CNK_PASS_DGRAD)) {
            DgradClass cls[16];
            CNK_CHECK_ARG(d->stride_h * d->stride_w <= 16, "cnk_conv2d_pack_weights: stride too large");
            int nc = dgrad_classes(d, cls);
            size_t off = 0;
            for (int c = 0; c < nc; ++c) {
                CNK_CHECK_ARG(cls[c].ntaps <= 32, "cnk_conv2d_pack_weights: too many taps");
                if (int rc = pack_taps(d, w, (char*)w_dgrad + off * esize(d->dtype), cls[c].ntaps, cls[c].kh, cls[c].kw, 1, S(stream))) return rc;
                off += igemm_image_elems(d, CNK_PASS_DGRAD, cls[c].ntaps);
            }
        } else if (smallc_dgrad_packed_bytes(d)) {
            size_t a = (smallc_dgrad_packed_bytes(d) + 255) & ~(size_t)255;
            if (int rc = smallc_dgrad_pack(d, w, w_dgrad, S(stream))) return rc;
            if (int rc = pack_direct(d, w, (char*)w_dgrad + a, S(stream))) return rc;
        } else if (!(direct_done && w_dgrad == w_fwd)) {
            if (int rc = pack_direct(d, w, w_dgrad, S(stream))) return rc;
        }
    }
    return 0;
}

int cnk_conv2d_pack_jobs(const cnk_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad, cnk_pack_job* jobs, int32_t max_jobs) {
    if (check_desc(d, "cnk_conv2d_pack_jobs")) return -1;
    int n = 0;
    auto add = [&](void* out, int ntaps, const signed char* kh, const signed char* kw, int transpose) {
        if (ntaps == 0) return true;
        if (n >= max_jobs || ntaps > 32) return false;
        if ((uint64_t)d->Cout * d->Cin * (uint64_t)ntaps >= (1ull << 31)) return false;   // pack_batch_kernel indexes in 32 bits
        cnk_pack_job& j = jobs[n++];
        memset(&j, 0, sizeof(j));
        j.w = w; j.out = out; j.Cout = d->Cout; j.Cin = d->Cin; j.KH = d->kh; j.KW = d->kw;
        j.ntaps = ntaps; j.transpose = transpose; j.dtype = d->dtype; j.groups = conv_groups(d);
        for (int t = 0; t < ntaps; ++t) { j.kh[t] = kh[t]; j.kw[t] = kw[t]; }
        j.nblocks = pack_job_blocks(j);
        return true;
    };
    if (w_fwd) {
        if (!igemm_supported(d, CNK_PASS_FWD) || !igemm_on_grid(d, CNK_PASS_FWD)) return -1;   // only unpadded implicit-GEMM images batch
        signed char kh[32], kw[32];
        for (int t = 0; t < d->kh * d->kw; ++t) { kh[t] = (signed char)(t / d->kw); kw[t] = (signed char)(t % d->kw); }
        if (!add(w_fwd, d->kh * d->kw, kh, kw, 0)) return -1;
    }
    if (w_dgrad) {
        if (!igemm_supported(d, CNK_PASS_DGRAD) || !igemm_on_grid(d, CNK_PASS_DGRAD) || d->stride_h * d->stride_w > 16) return -1;
        DgradClass cls[16];
        int nc = dgrad_classes(d, cls);
        size_t off = 0;
        for (int c = 0; c < nc; ++c) {
            if (!add((char*)w_dgrad + off * esize(d->dtype), cls[c].ntaps, cls[c].kh, cls[c].kw, 1)) return -1;
            off += (size_t)cls[c].ntaps * d->Cin * d->Cout;
        }
    }
    return n;
}

int cnk_pack_batch(const cnk_pack_job* jobs_dev, const void* block_map_dev, int32_t nblocks, void* stream) {
    CNK_CHECK_ARG(jobs_dev && block_map_dev && nblocks >= 0, "cnk_pack_batch: bad arguments");
    return pack_batch(jobs_dev, block_map_dev, nblocks, S(stream));
}

int64_t cnk_conv2d_fwd_bn_partial_slices(const cnk_conv_desc* d) {
    if (igemm_supported(d, CNK_PASS_FWD)) return igemm_fwd_stat_slices(d);
    return cnk_sw().rowpatch_stats && rowpatch_supported(d) ? rowpatch_fwd_stat_slices(d) : 0;
}

int cnk_conv2d_fwd(const cnk_conv_desc* d, const void* x, const void* w_fwd, const float* bias, void* y, float* bn_partial, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    if (int rc = check_desc(d, "cnk_conv2d_fwd")) return rc;
    CNK_CHECK_ARG(x && w_fwd && y, "cnk_conv2d_fwd: null pointer");
    double fl, by;
    conv_cost(d, &fl, &by);
    ProfScope ps(conv_cls(d, 0), S(stream), fl, by);
    if (igemm_supported(d, CNK_PASS_FWD)) return igemm_fwd(d, x, w_fwd, bias, y, bn_partial, nullptr, 0, S(stream));
    if (rowpatch_supported(d)) return rowpatch_fwd(d, x, w_fwd, bias, y, bn_partial, S(stream));
    CNK_CHECK_ARG(!bn_partial, "cnk_conv2d_fwd: this conv shape cannot emit BN partials (cnk_conv2d_fwd_bn_partial_slices == 0)");
    if (smallc_supported(d)) return smallc_fwd(d, x, w_fwd, bias, y, S(stream));
    return direct_fwd(d, x, w_fwd, bias, y, S(stream));
}

int cnk_conv2d_fwd_act_supported(const cnk_conv_desc* d) {
    return check_desc(d, "cnk_conv2d_fwd_act_supported") == 0 && igemm_supported(d, CNK_PASS_FWD) &&
           d->Cout % (d->dtype == CNK_BF16 ? 8 : 4) == 0;
}

int cnk_conv2d_fwd_act(const cnk_conv_desc* d, const void* x, const void* w_fwd, const float* bias, const void* residual, int relu, void* y, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_fwd_act")) return rc;
    CNK_CHECK_ARG(x && w_fwd && y, "cnk_conv2d_fwd_act: null pointer");
    CNK_CHECK_ARG(cnk_conv2d_fwd_act_supported(d), "cnk_conv2d_fwd_act: this conv shape has no fused-activation forward (cnk_conv2d_fwd_act_supported == 0)");
    double fl, by;
    conv_cost(d, &fl, &by);
    ProfScope ps(conv_cls(d, 0), S(stream), fl, by);
    return igemm_fwd(d, x, w_fwd, bias, y, nullptr, residual, relu, S(stream));
}

int cnk_conv2d_dgrad_accepts_mask(const cnk_conv_desc* d) {
    return igemm_supported(d, CNK_PASS_DGRAD) && d->Cin % (d->dtype == CNK_BF16 ? 8 : 4) == 0;
}

int64_t cnk_conv2d_dgrad_bn_slices(const cnk_conv_desc* d) {
    if (check_desc(d, "cnk_conv2d_dgrad_bn_slices")) return 0;
    return igemm_supported(d, CNK_PASS_DGRAD) ? igemm_dgrad_bn_slices(d) : 0;
}

int cnk_conv2d_dgrad(const cnk_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* addend, const uint8_t* addend_mask, const cnk_bn_bwd_fuse* bn, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    if (int rc = check_desc(d, "cnk_conv2d_dgrad")) return rc;
    CNK_CHECK_ARG(dy && w_dgrad && dx, "cnk_conv2d_dgrad: null pointer");
    double fl, by;
    conv_cost(d, &fl, &by);
    if (bn) by += (double)d->N * d->H * d->W * d->Cin * esize(d->dtype);      // the fused BN backward reads that layer's input
    if (addend) by += (double)d->N * d->H * d->W * d->Cin * esize(d->dtype);
    ProfScope ps(conv_cls(d, 1), S(stream), fl, by);
    CNK_CHECK_ARG(!addend_mask || (addend && igemm_supported(d, CNK_PASS_DGRAD) && d->Cin % (d->dtype == CNK_BF16 ? 8 : 4) == 0), "cnk_conv2d_dgrad: addend_mask needs an addend and the implicit-GEMM path");
    if (bn) {
        CNK_CHECK_ARG(cnk_conv2d_dgrad_bn_slices(d) > 0, "cnk_conv2d_dgrad: this conv shape cannot fuse the BN backward (cnk_conv2d_dgrad_bn_slices == 0)");
        CNK_CHECK_ARG(bn->x && bn->mean && bn->invstd && bn->partial, "cnk_conv2d_dgrad: bn fuse: null pointer");
        CNK_CHECK_ARG(!bn->relu || bn->relu_mask || (bn->gamma && bn->beta), "cnk_conv2d_dgrad: bn fuse: ReLU without a mask needs gamma and beta");
    }
    if (igemm_supported(d, CNK_PASS_DGRAD)) return igemm_dgrad(d, dy, w_dgrad, dx, addend, addend_mask, bn, S(stream));
    if (size_t sb = smallc_dgrad_packed_bytes(d)) {
        if (!addend) return smallc_dgrad(d, dy, w_dgrad, dx, S(stream));
        w_dgrad = (const char*)w_dgrad + ((sb + 255) & ~(size_t)255);       // the direct image sits behind it
    }
    return direct_dgrad(d, dy, w_dgrad, dx, addend, S(stream));
}

// The smallest workspace cnk_conv2d_wgrad accepts: split-K kernels take ANY workspace of at least one fp32 slab (+ the bias
// sum's scratch) -- less than cnk_conv2d_workspace_bytes() bounds the number of pixel-range splits (include/cnk.h; why a caller
// would: conv_wgrad.hip, wgrad_cap_splits); every other kernel needs its whole workspace.
size_t cnk_conv2d_wgrad_min_workspace_bytes(const cnk_conv_desc* d, int has_bias) {
    if (check_desc(d, "cnk_conv2d_wgrad_min_workspace_bytes")) return 0;
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    const size_t bias_ws = has_bias ? colsum_workspace((int64_t)d->N * Ho * Wo, d->Cout) : 0;
    if (!(wgrad3x3_supported(d) || wgrad_mfma_supported(d))) return cnk_conv2d_workspace_bytes(d, CNK_PASS_WGRAD);
    const size_t slab = (size_t)d->Cout * (conv_groups(d) > 1 ? 64 : d->Cin) * d->kh * d->kw * sizeof(float);
    return ((slab + 255) & ~(size_t)255) + bias_ws;
}

int cnk_conv2d_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_wgrad")) return rc;
    CNK_CHECK_ARG(dy && x && dw, "cnk_conv2d_wgrad: null pointer");
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    const size_t bias_ws = dbias ? colsum_workspace((int64_t)d->N * Ho * Wo, d->Cout) : 0;
    const bool splitk = wgrad3x3_supported(d) || wgrad_mfma_supported(d);
    const size_t least = cnk_conv2d_wgrad_min_workspace_bytes(d, dbias != nullptr);
    CNK_CHECK_ARG(workspace && workspace_bytes >= least, "cnk_conv2d_wgrad: workspace too small (%zu < %zu)", workspace_bytes, least);
    double fl, by;
    conv_cost(d, &fl, &by);
    ProfScope ps(conv_cls(d, 2), S(stream), fl, by);
    int rc;
    size_t a;
    if (splitk) {
        const bool w3 = wgrad3x3_supported(d);
        const size_t planned = w3 ? wgrad3x3_workspace(d) : wgrad_mfma_workspace(d);
        size_t avail = (workspace_bytes - bias_ws) & ~(size_t)255;
        a = avail < planned ? avail : planned;
        rc = w3 ? wgrad3x3(d, dy, x, dw, accumulate, workspace, a, S(stream)) : wgrad_mfma(d, dy, x, dw, accumulate, workspace, a, S(stream));
    } else if (rowpatch_supported(d)) {
        a = rowpatch_wgrad_workspace(d);
        rc = rowpatch_wgrad(d, dy, x, dw, accumulate, workspace, a, S(stream));
    } else if (smallc_supported(d)) {
        a = smallc_wgrad_workspace(d);
        rc = smallc_wgrad(d, dy, x, dw, accumulate, workspace, a, S(stream));
    } else {
        a = direct_wgrad_workspace(d);
        rc = direct_wgrad(d, dy, x, dw, accumulate, workspace, a, S(stream));
    }
    if (rc) return rc;
    if (dbias) {
        a = (a + 255) & ~(size_t)255;
        return colsum(dy, dbias, (int64_t)d->N * Ho * Wo, d->Cout, d->dtype, accumulate, (char*)workspace + a, workspace_bytes - a, S(stream));
    }
    return 0;
}

// ---- chained slab sums (csrc/wgrad_chain.h): cnk_wgrad_pending is WgradPending
int cnk_conv2d_wgrad_chain_supported(const cnk_conv_desc* d) {
    return check_desc(d, "cnk_conv2d_wgrad_chain_supported") == 0 && conv_groups(d) == 1 && d->Cin % 4 == 0 &&
           (wgrad3x3_supported(d) || wgrad_mfma_supported(d));
}

int cnk_conv2d_wgrad_chain(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, void* workspace, size_t workspace_bytes, const cnk_wgrad_pending* prev, cnk_wgrad_pending* out, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_wgrad_chain")) return rc;
    CNK_CHECK_ARG(dy && x && dw && out, "cnk_conv2d_wgrad_chain: null pointer");
    CNK_CHECK_ARG(cnk_conv2d_wgrad_chain_supported(d), "cnk_conv2d_wgrad_chain: this conv has no chained slab sum (cnk_conv2d_wgrad_chain_supported == 0)");
    const size_t least = cnk_conv2d_wgrad_min_workspace_bytes(d, 0);
    CNK_CHECK_ARG(workspace && workspace_bytes >= least, "cnk_conv2d_wgrad_chain: workspace too small (%zu < %zu)", workspace_bytes, least);
    if (prev && prev->slab) {
        const char* a0 = (const char*)prev->slab; const char* a1 = a0 + (size_t)prev->splits * prev->slab_stride * sizeof(float);
        const char* b0 = (const char*)workspace; const char* b1 = b0 + workspace_bytes;
        CNK_CHECK_ARG(a1 <= b0 || b1 <= a0, "cnk_conv2d_wgrad_chain: the pending slabs overlap this call's workspace");
    }
    double fl, by;
    conv_cost(d, &fl, &by);
    ProfScope ps(conv_cls(d, 2), S(stream), fl, by);
    const bool w3 = wgrad3x3_supported(d);
    const size_t planned = w3 ? wgrad3x3_workspace(d) : wgrad_mfma_workspace(d);
    const size_t avail = workspace_bytes & ~(size_t)255;
    const size_t a = avail < planned ? avail : planned;
    const WgradPending* pv = reinterpret_cast<const WgradPending*>(prev);
    WgradPending* po = reinterpret_cast<WgradPending*>(out);
    return w3 ? wgrad3x3(d, dy, x, dw, accumulate, workspace, a, S(stream), pv, po) : wgrad_mfma(d, dy, x, dw, accumulate, workspace, a, S(stream), nullptr, pv, po);
}

int cnk_wgrad_reduce_pending(const cnk_wgrad_pending* pending, void* stream) {
    CNK_CHECK_ARG(pending, "cnk_wgrad_reduce_pending: null pointer");
    return wgrad_pending_reduce(*reinterpret_cast<const WgradPending*>(pending), S(stream));
}

int cnk_conv2d_bnfold_supported(const cnk_conv_desc* d) {
    return check_desc(d, "cnk_conv2d_bnfold_supported") == 0 && bnfold_supported(d) && igemm_supported(d, CNK_PASS_DGRAD) &&
           wgrad_mfma_supported(d);
}

size_t cnk_conv2d_bnfold_image_bytes(const cnk_conv_desc* d) { return bnfold_image_bytes(d); }

int cnk_bn_bwd_coef(const void* dy, const void* x, const float* ext_partial, int64_t ext_slices, const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, int accumulate, float* coef, int64_t M, int32_t C, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(gamma && save_mean && save_invstd && dgamma && dbeta && coef, "cnk_bn_bwd_coef: null pointer");
    CNK_CHECK_ARG(ext_partial ? ext_slices > 0 : (dy && x && ext_slices == 0), "cnk_bn_bwd_coef: either the partial sums of a fused data gradient (ext_partial, ext_slices > 0) or (dy, x) for this call's own reduction");
    CNK_CHECK_ARG(M > 0 && C > 0, "cnk_bn_bwd_coef: empty tensor");
    CNK_CHECK_ARG(workspace && workspace_bytes >= bn_workspace_bytes(M, C), "cnk_bn_bwd_coef: workspace too small");
    ProfScope ps(CNK_CLS_BN_BWD, S(stream), 0.0, ext_partial ? 0.0 : 2.0 * (double)M * C * esize(dtype));
    return bn_bwd_coef(dy, x, ext_partial, ext_slices, gamma, save_mean, save_invstd, dgamma, dbeta, accumulate, coef, M, C, dtype, (float*)workspace, S(stream));
}

int cnk_conv2d_bnfold_prepare(const cnk_conv_desc* d, const float* w, const void* w_dgrad, const float* save_mean, const float* save_invstd, const float* coef, void* w_fold, float* bias_fold, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_bnfold_prepare")) return rc;
    CNK_CHECK_ARG(cnk_conv2d_bnfold_supported(d), "cnk_conv2d_bnfold_prepare: this conv shape has no folded BN backward (cnk_conv2d_bnfold_supported == 0)");
    CNK_CHECK_ARG(w && w_dgrad && save_mean && save_invstd && coef && w_fold && bias_fold, "cnk_conv2d_bnfold_prepare: null pointer");
    return bnfold_prepare(d, w, w_dgrad, save_mean, save_invstd, coef, w_fold, bias_fold, S(stream));
}

int cnk_conv2d_dgrad_bnfold(const cnk_conv_desc* d, const void* g, const void* x, const void* w_fold, const float* bias_fold, void* dx, const void* addend, const uint8_t* addend_mask, const cnk_bn_bwd_fuse* bn, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_dgrad_bnfold")) return rc;
    CNK_CHECK_ARG(!addend_mask || addend, "cnk_conv2d_dgrad_bnfold: addend_mask needs an addend");
    CNK_CHECK_ARG(cnk_conv2d_bnfold_supported(d), "cnk_conv2d_dgrad_bnfold: this conv shape has no folded BN backward (cnk_conv2d_bnfold_supported == 0)");
    CNK_CHECK_ARG(g && x && w_fold && bias_fold && dx, "cnk_conv2d_dgrad_bnfold: null pointer");
    double fl, by;
    conv_cost(d, &fl, &by);
    const double px = (double)d->N * d->H * d->W;
    {
        int Ho, Wo;
        cnk_conv2d_out_hw(d, &Ho, &Wo);
        const double pxo = (double)d->N * Ho * Wo;                      // the pixels the conv samples (stride)
        fl += 2.0 * pxo * d->Cin * d->Cin;                              // the second K segment
        by += pxo * d->Cin * esize(d->dtype);                           // ... reads the conv input there
    }
    if (bn) {
        by += px * d->Cin * esize(d->dtype);
        CNK_CHECK_ARG(cnk_conv2d_dgrad_bn_slices(d) > 0, "cnk_conv2d_dgrad_bnfold: this conv shape cannot fuse the BN backward (cnk_conv2d_dgrad_bn_slices == 0)");
        CNK_CHECK_ARG(bn->x && bn->mean && bn->invstd && bn->partial, "cnk_conv2d_dgrad_bnfold: bn fuse: null pointer");
        CNK_CHECK_ARG(!bn->relu || bn->relu_mask || (bn->gamma && bn->beta), "cnk_conv2d_dgrad_bnfold: bn fuse: ReLU without a mask needs gamma and beta");
    }
    if (addend) by += px * d->Cin * esize(d->dtype);
    ProfScope ps(conv_cls(d, 1), S(stream), fl, by);
    return igemm_dgrad_bnfold(d, g, x, w_fold, bias_fold, dx, addend, addend_mask, bn, S(stream));
}

int cnk_conv2d_wgrad_bnfold(const cnk_conv_desc* d, const void* g, const void* y, const void* x, const float* save_mean, const float* save_invstd, const float* coef, float* dw, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_wgrad_bnfold")) return rc;
    CNK_CHECK_ARG(cnk_conv2d_bnfold_supported(d), "cnk_conv2d_wgrad_bnfold: this conv shape has no folded BN backward (cnk_conv2d_bnfold_supported == 0)");
    CNK_CHECK_ARG(g && y && x && save_mean && save_invstd && coef && dw, "cnk_conv2d_wgrad_bnfold: null pointer");
    const size_t slab = ((size_t)d->Cout * d->Cin * sizeof(float) + 255) & ~(size_t)255;
    CNK_CHECK_ARG(workspace && workspace_bytes >= slab, "cnk_conv2d_wgrad_bnfold: workspace too small (%zu < %zu)", workspace_bytes, slab);
    double fl, by;
    conv_cost(d, &fl, &by);
    {
        int Ho, Wo;
        cnk_conv2d_out_hw(d, &Ho, &Wo);
        by += (double)d->N * Ho * Wo * d->Cout * esize(d->dtype);       // the BN input is read beside the gradient
    }
    ProfScope ps(conv_cls(d, 2), S(stream), fl, by);
    const size_t planned = wgrad_mfma_workspace(d);
    const size_t avail = workspace_bytes & ~(size_t)255;
    const WgradFold fold = {y, coef, save_mean, save_invstd};
    return wgrad_mfma(d, g, x, dw, accumulate, workspace, avail < planned ? avail : planned, S(stream), &fold);
}

int cnk_conv2d_wgrad_bnfold_chain(const cnk_conv_desc* d, const void* g, const void* y, const void* x, const float* save_mean, const float* save_invstd, const float* coef, float* dw, int accumulate, void* workspace, size_t workspace_bytes, const cnk_wgrad_pending* prev, cnk_wgrad_pending* out, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_wgrad_bnfold_chain")) return rc;
    CNK_CHECK_ARG(cnk_conv2d_bnfold_supported(d) && cnk_conv2d_wgrad_chain_supported(d), "cnk_conv2d_wgrad_bnfold_chain: this conv shape has no folded BN backward with a chained slab sum");
    CNK_CHECK_ARG(g && y && x && save_mean && save_invstd && coef && dw && out, "cnk_conv2d_wgrad_bnfold_chain: null pointer");
    const size_t slab = ((size_t)d->Cout * d->Cin * sizeof(float) + 255) & ~(size_t)255;
    CNK_CHECK_ARG(workspace && workspace_bytes >= slab, "cnk_conv2d_wgrad_bnfold_chain: workspace too small (%zu < %zu)", workspace_bytes, slab);
    if (prev && prev->slab) {
        const char* a0 = (const char*)prev->slab; const char* a1 = a0 + (size_t)prev->splits * prev->slab_stride * sizeof(float);
        const char* b0 = (const char*)workspace; const char* b1 = b0 + workspace_bytes;
        CNK_CHECK_ARG(a1 <= b0 || b1 <= a0, "cnk_conv2d_wgrad_bnfold_chain: the pending slabs overlap this call's workspace");
    }
    double fl, by;
    conv_cost(d, &fl, &by);
    {
        int Ho, Wo;
        cnk_conv2d_out_hw(d, &Ho, &Wo);
        by += (double)d->N * Ho * Wo * d->Cout * esize(d->dtype);
    }
    ProfScope ps(conv_cls(d, 2), S(stream), fl, by);
    const size_t planned = wgrad_mfma_workspace(d);
    const size_t avail = workspace_bytes & ~(size_t)255;
    const WgradFold fold = {y, coef, save_mean, save_invstd};
    return wgrad_mfma(d, g, x, dw, accumulate, workspace, avail < planned ? avail : planned, S(stream), &fold,
                      reinterpret_cast<const WgradPending*>(prev), reinterpret_cast<WgradPending*>(out));
}

// ---- folded data gradient + weight gradient in one pass (conv_fold3.hip)
int cnk_conv2d_bwd_bnfold_supported(const cnk_conv_desc* d) {
    return check_desc(d, "cnk_conv2d_bwd_bnfold_supported") == 0 && fold3_supported(d) && cnk_conv2d_bnfold_supported(d);
}

size_t cnk_conv2d_bwd_bnfold_workspace_bytes(const cnk_conv_desc* d) { return cnk_conv2d_bwd_bnfold_supported(d) ? fold3_workspace(d) : 0; }

int64_t cnk_conv2d_bwd_bnfold_bn_slices(const cnk_conv_desc* d) { return cnk_conv2d_bwd_bnfold_supported(d) ? fold3_workgroups(d) : 0; }

int cnk_conv2d_bwd_bnfold(const cnk_conv_desc* d, const void* g, const void* x, const void* w_fold, const float* bias_fold, void* dx, const void* addend, const cnk_bn_bwd_fuse* bn, const float* w, const float* save_mean, const float* save_invstd, const float* coef, float* dw, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d, "cnk_conv2d_bwd_bnfold")) return rc;
    CNK_CHECK_ARG(cnk_conv2d_bwd_bnfold_supported(d), "cnk_conv2d_bwd_bnfold: this conv shape has no one-pass folded backward (cnk_conv2d_bwd_bnfold_supported == 0)");
    CNK_CHECK_ARG(g && x && w_fold && bias_fold && dx && w && save_mean && save_invstd && coef && dw, "cnk_conv2d_bwd_bnfold: null pointer");
    CNK_CHECK_ARG(!bn || (bn->x && bn->mean && bn->invstd && bn->gamma && bn->beta && bn->partial), "cnk_conv2d_bwd_bnfold: incomplete cnk_bn_bwd_fuse");
    double fl, by;
    conv_cost(d, &fl, &by);
    ProfScope ps(conv_cls(d, 1), S(stream), 2.0 * fl, by);
    CNK_CHECK_ARG(!bn || !addend, "cnk_conv2d_bwd_bnfold: a gradient addend or the fused BN backward, not both");
    return fold3_bwd(d, g, x, w_fold, bias_fold, dx, addend, bn, w, save_mean, save_invstd, coef, dw, accumulate, workspace, workspace_bytes, S(stream));
}

size_t cnk_bn_workspace_bytes(int64_t M, int32_t C) { return bn_workspace_bytes(M, C); }

size_t cnk_bn_mask_bytes(int64_t M, int32_t C, int dtype) { return bn_mask_bytes(M, C, dtype); }

int cnk_bn_fwd_train(const void* x, const float* gamma, const float* beta, const void* residual, void* y, uint8_t* relu_mask, float* running_mean, float* running_var, float* save_mean, float* save_invstd, const float* partial, int64_t partial_slices, int64_t M, int32_t C, float eps, float momentum, int relu, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    // x may be NULL when it is never read: statistics from the caller's partials, no apply pass
    CNK_CHECK_ARG((x || (!y && partial)) && gamma && beta && save_mean && save_invstd, "cnk_bn_fwd_train: null pointer");
    CNK_CHECK_ARG(y || (!residual && !relu_mask), "cnk_bn_fwd_train: y == NULL (statistics only) takes no residual / mask");
    CNK_CHECK_ARG(M > 0 && C > 0, "cnk_bn_fwd_train: empty tensor");
    CNK_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "cnk_bn_fwd_train: running stats must both be given or both NULL");
    CNK_CHECK_ARG(workspace && workspace_bytes >= bn_workspace_bytes(M, C), "cnk_bn_fwd_train: workspace too small");
    CNK_CHECK_ARG((partial == nullptr) == (partial_slices == 0) && partial_slices >= 0, "cnk_bn_fwd_train: partial and partial_slices must come together");
    ProfScope ps(CNK_CLS_BN_FWD, S(stream), 0.0, (partial ? 2.0 : 3.0) * (double)M * C * esize(dtype));
    CNK_CHECK_ARG(!relu_mask || relu, "cnk_bn_fwd_train: relu_mask without relu");
    return bn_fwd_train(x, gamma, beta, residual, y, relu_mask, running_mean, running_var, save_mean, save_invstd, partial, partial_slices, M, C, eps, momentum, relu, dtype, (float*)workspace, S(stream));
}

int cnk_bn_fwd_train_res_affine(const void* x, const float* gamma, const float* beta, const void* residual, const float* residual_scale_shift, void* y, uint8_t* relu_mask, float* running_mean, float* running_var, float* save_mean, float* save_invstd, const float* partial, int64_t partial_slices, int64_t M, int32_t C, float eps, float momentum, int relu, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(x && gamma && beta && save_mean && save_invstd && y && residual && residual_scale_shift, "cnk_bn_fwd_train_res_affine: null pointer");
    CNK_CHECK_ARG(M > 0 && C > 0, "cnk_bn_fwd_train_res_affine: empty tensor");
    CNK_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "cnk_bn_fwd_train_res_affine: running stats must both be given or both NULL");
    CNK_CHECK_ARG(workspace && workspace_bytes >= bn_workspace_bytes(M, C), "cnk_bn_fwd_train_res_affine: workspace too small");
    CNK_CHECK_ARG((partial == nullptr) == (partial_slices == 0) && partial_slices >= 0, "cnk_bn_fwd_train_res_affine: partial and partial_slices must come together");
    CNK_CHECK_ARG(!relu_mask || relu, "cnk_bn_fwd_train_res_affine: relu_mask without relu");
    ProfScope ps(CNK_CLS_BN_FWD, S(stream), 0.0, (partial ? 3.0 : 4.0) * (double)M * C * esize(dtype));
    return bn_fwd_train(x, gamma, beta, residual, y, relu_mask, running_mean, running_var, save_mean, save_invstd, partial, partial_slices, M, C, eps, momentum, relu, dtype, (float*)workspace, S(stream), residual_scale_shift);
}

int cnk_bn_fwd_eval(const void* x, const float* gamma, const float* beta, const float* running_mean, const float* running_var, const void* residual, void* y, int64_t M, int32_t C, float eps, int relu, int dtype, void* stream) {
    CNK_CHECK_ARG(x && gamma && beta && y && running_mean && running_var, "cnk_bn_fwd_eval: null pointer");
    CNK_CHECK_ARG(M > 0 && C > 0, "cnk_bn_fwd_eval: empty tensor");
    return bn_fwd_eval(x, gamma, beta, running_mean, running_var, residual, y, M, C, eps, relu, dtype, S(stream));
}

int cnk_bn_bwd(const void* dy, const void* x, const void* y, const uint8_t* relu_mask, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dx, void* dresidual, float* dgamma, float* dbeta, int accumulate, const float* ext_partial, int64_t ext_slices, int64_t M, int32_t C, int relu, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta, "cnk_bn_bwd: null pointer");
    CNK_CHECK_ARG(!relu || ext_partial || y || relu_mask || (beta && !dresidual), "cnk_bn_bwd: relu backward needs the saved mask, the forward output, or (no residual) beta");
    CNK_CHECK_ARG(!ext_partial || (ext_slices > 0 && !dresidual), "cnk_bn_bwd: ext_partial needs ext_slices > 0 and dresidual == NULL (dy is already the masked gradient)");
    CNK_CHECK_ARG(M > 0 && C > 0, "cnk_bn_bwd: empty tensor");
    CNK_CHECK_ARG(workspace && workspace_bytes >= bn_workspace_bytes(M, C), "cnk_bn_bwd: workspace too small");
    // reduce pass (2 reads) + apply pass (2 reads, 1 write); a fused conv dgrad already did the first
    ProfScope ps(CNK_CLS_BN_BWD, S(stream), 0.0, (ext_partial ? 3.0 : 5.0) * (double)M * C * esize(dtype));
    return bn_bwd(dy, x, y, relu_mask, gamma, beta, save_mean, save_invstd, dx, dresidual, dgamma, dbeta, accumulate, ext_partial, ext_slices, M, C, relu, dtype, (float*)workspace, S(stream));
}

int cnk_avgpool2d_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream) {
    CNK_CHECK_ARG(x && y, "cnk_avgpool2d_fwd: null pointer");
    CNK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && 2 * ph <= kh && 2 * pw <= kw && H + 2 * ph >= kh && W + 2 * pw >= kw, "cnk_avgpool2d_fwd: bad geometry");
    return avgpool_dispatch(false, x, y, N, H, W, C, kh, kw, sh, sw, ph, pw, dtype, S(stream));
}
int cnk_avgpool2d_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream) {
    CNK_CHECK_ARG(dy && dx, "cnk_avgpool2d_bwd: null pointer");
    CNK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && 2 * ph <= kh && 2 * pw <= kw && H + 2 * ph >= kh && W + 2 * pw >= kw, "cnk_avgpool2d_bwd: bad geometry");
    return avgpool_dispatch(true, dy, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, dtype, S(stream));
}

int cnk_maxpool2d_fwd(const void* x, void* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream) {
    CNK_CHECK_ARG(x && y, "cnk_maxpool2d_fwd: null pointer");
    CNK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && 2 * ph <= kh && 2 * pw <= kw && H + 2 * ph >= kh && W + 2 * pw >= kw, "cnk_maxpool2d_fwd: bad geometry");
    CNK_CHECK_ARG(kh * kw <= 255, "cnk_maxpool2d_fwd: window too large for uint8 positions");
    return maxpool_dispatch(false, x, argmax, y, N, H, W, C, kh, kw, sh, sw, ph, pw, dtype, S(stream));
}
int cnk_maxpool2d_bwd(const void* dy, const uint8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream) {
    CNK_CHECK_ARG(dy && argmax && dx, "cnk_maxpool2d_bwd: null pointer");
    CNK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && 2 * ph <= kh && 2 * pw <= kw && H + 2 * ph >= kh && W + 2 * pw >= kw, "cnk_maxpool2d_bwd: bad geometry");
    return maxpool_dispatch(true, dy, (void*)argmax, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, dtype, S(stream));
}

static int bn_bwd_maxpool_common(const void* dy_pooled, const uint8_t* argmax, const void* x, const void* x_at_max, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta, int accumulate, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int relu, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(dy_pooled && argmax && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta, "cnk_bn_bwd_maxpool: null pointer");
    CNK_CHECK_ARG(!relu || beta, "cnk_bn_bwd_maxpool: relu backward re-derives the mask from x and needs beta");
    CNK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && kh * kw <= 255, "cnk_bn_bwd_maxpool: bad geometry");
    CNK_CHECK_ARG(H + 2 * ph >= kh && W + 2 * pw >= kw && 2 * ph <= kh && 2 * pw <= kw, "cnk_bn_bwd_maxpool: bad padding");
    CNK_CHECK_ARG(kh <= 2 * sh && kw <= 2 * sw, "cnk_bn_bwd_maxpool: more than 2 x 2 windows cover an element (kernel %dx%d, stride %dx%d): use cnk_maxpool2d_bwd + cnk_bn_bwd", kh, kw, sh, sw);
    CNK_CHECK_ARG(dtype == CNK_F32 || dtype == CNK_BF16, "cnk_bn_bwd_maxpool: bad dtype %d", dtype);
    const int64_t M = (int64_t)N * H * W;
    CNK_CHECK_ARG(M * C < (1ll << 31), "cnk_bn_bwd_maxpool: tensor too large");
    CNK_CHECK_ARG(workspace && workspace_bytes >= bn_workspace_bytes(M, C), "cnk_bn_bwd_maxpool: workspace too small");
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    // two passes over x, each with a gather of the pooled gradient and its arg-max bytes, one write of dx (with x_at_max the
    // first pass reads the pooled gradient and the kept inputs only)
    ProfScope ps(CNK_CLS_BN_BWD, S(stream), 0.0, (x_at_max ? 2.0 : 3.0) * (double)M * C * esize(dtype) + 2.0 * (double)N * Ho * Wo * C * (esize(dtype) + 1));
    return bn_bwd_pool(dy_pooled, argmax, x, x_at_max, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, accumulate, N, H, W, C, kh, kw, sh, sw, ph, pw, relu, dtype, (float*)workspace, S(stream));
}

int cnk_bn_bwd_maxpool(const void* dy_pooled, const uint8_t* argmax, const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta, int accumulate, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int relu, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return bn_bwd_maxpool_common(dy_pooled, argmax, x, nullptr, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, accumulate, N, H, W, C, kh, kw, sh, sw, ph, pw, relu, dtype, workspace, workspace_bytes, stream);
}

int cnk_bn_bwd_maxpool_kept(const void* dy_pooled, const uint8_t* argmax, const void* x, const void* x_at_max, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta, int accumulate, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int relu, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(x_at_max, "cnk_bn_bwd_maxpool_kept: null pointer");
    return bn_bwd_maxpool_common(dy_pooled, argmax, x, x_at_max, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, accumulate, N, H, W, C, kh, kw, sh, sw, ph, pw, relu, dtype, workspace, workspace_bytes, stream);
}

int cnk_maxpool2d_fwd_bn(const void* x, const float* scale, const float* shift, int relu, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream) {
    CNK_CHECK_ARG(x && scale && shift && y && idx, "cnk_maxpool2d_fwd_bn: null pointer");
    CNK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && kh * kw <= 255, "cnk_maxpool2d_fwd_bn: bad geometry");
    CNK_CHECK_ARG(H + 2 * ph >= kh && W + 2 * pw >= kw && 2 * ph <= kh && 2 * pw <= kw, "cnk_maxpool2d_fwd_bn: bad padding");
    CNK_CHECK_ARG(dtype == CNK_F32 || dtype == CNK_BF16, "cnk_maxpool2d_fwd_bn: bad dtype %d", dtype);
    return maxpool_affine_dispatch(x, scale, shift, relu, y, idx, nullptr, N, H, W, C, kh, kw, sh, sw, ph, pw, dtype, S(stream));
}

int cnk_maxpool2d_fwd_bn_keep(const void* x, const float* scale, const float* shift, int relu, void* y, uint8_t* idx, void* x_at_max, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream) {
    CNK_CHECK_ARG(x && scale && shift && y && idx && x_at_max, "cnk_maxpool2d_fwd_bn_keep: null pointer");
    CNK_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && kh * kw <= 255, "cnk_maxpool2d_fwd_bn_keep: bad geometry");
    CNK_CHECK_ARG(H + 2 * ph >= kh && W + 2 * pw >= kw && 2 * ph <= kh && 2 * pw <= kw, "cnk_maxpool2d_fwd_bn_keep: bad padding");
    CNK_CHECK_ARG(dtype == CNK_F32 || dtype == CNK_BF16, "cnk_maxpool2d_fwd_bn_keep: bad dtype %d", dtype);
    return maxpool_affine_dispatch(x, scale, shift, relu, y, idx, x_at_max, N, H, W, C, kh, kw, sh, sw, ph, pw, dtype, S(stream));
}

int cnk_relu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
    CNK_CHECK_ARG(x && y && n >= 0, "cnk_relu_fwd: bad arguments");
    return eltwise_dispatch<0>(x, nullptr, y, n, dtype, S(stream));
}
int cnk_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream) {
    CNK_CHECK_ARG(dy && y && dx && n >= 0, "cnk_relu_bwd: bad arguments");
    return eltwise_dispatch<1>(dy, y, dx, n, dtype, S(stream));
}
int cnk_add_relu_fwd(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream) {
    CNK_CHECK_ARG(a && b && y && n >= 0, "cnk_add_relu_fwd: bad arguments");
    return eltwise_dispatch<2>(a, b, y, n, dtype, S(stream));
}

int cnk_gap_fwd(const void* x, float* y, int32_t N, int32_t HW, int32_t C, int dtype, void* stream) {
    CNK_CHECK_ARG(x && y && N > 0 && HW > 0 && C > 0, "cnk_gap_fwd: bad arguments");
    return gap_dispatch(false, x, y, N, HW, C, dtype, S(stream));
}
int cnk_gap_bwd(const float* dy, void* dx, int32_t N, int32_t HW, int32_t C, int dtype, void* stream) {
    CNK_CHECK_ARG(dy && dx && N > 0 && HW > 0 && C > 0, "cnk_gap_bwd: bad arguments");
    return gap_dispatch(true, dy, dx, N, HW, C, dtype, S(stream));
}

int cnk_nchw_to_nhwc(const float* x_nchw, void* y_nhwc, int32_t N, int32_t C, int32_t H, int32_t W, int dtype, void* stream) {
    CNK_CHECK_ARG(x_nchw && y_nhwc && N > 0 && C > 0 && H > 0 && W > 0, "cnk_nchw_to_nhwc: bad arguments");
    return nchw_to_nhwc_dispatch(x_nchw, y_nhwc, N, C, H, W, dtype, S(stream));
}

int cnk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
    CNK_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "cnk_adam_step: bad arguments");
    CNK_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "cnk_adam_step: arenas must be 16-byte aligned");
    return adam_dispatch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, S(stream));
}
void cnk_adam_bias_corrections(float beta1, float beta2, int32_t step, float* out2) {
    if (out2) adam_bias_corrections(beta1, beta2, step < 1 ? 1 : step, out2);
}
int cnk_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, const float* bias_dev, void* stream) {
    CNK_CHECK_ARG(p && g && m && v && n > 0 && bias_dev, "cnk_adam_step_dev: bad arguments");
    CNK_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "cnk_adam_step_dev: arenas must be 16-byte aligned");
    return adam_dispatch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, 1, bias_dev, S(stream));
}

int cnk_dwconv2d_supported(const cnk_conv_desc* d) {
    return check_desc(d, "cnk_dwconv2d_supported") == 0 && dwconv_supported(d);
}

static size_t dw_ws(const cnk_conv_desc* d) {
    int Ho, Wo;
    cnk_conv2d_out_hw(d, &Ho, &Wo);
    size_t a = dwconv_wgrad_workspace(d), c = colsum_workspace((int64_t)d->N * Ho * Wo, d->Cout);
    return a > c ? a : c;
}

size_t cnk_dwconv2d_workspace_bytes(const cnk_conv_desc* d) {
    if (check_desc(d, "cnk_dwconv2d_workspace_bytes") || !dwconv_supported(d)) return 0;
    return dw_ws(d);
}

int cnk_dwconv2d_fwd(const cnk_conv_desc* d, const void* x, const float* w, const float* bias, void* y, void* stream) {
    if (int rc = check_desc(d, "cnk_dwconv2d_fwd")) return rc;
    CNK_CHECK_ARG(dwconv_supported(d), "cnk_dwconv2d_fwd: needs Cin == Cout and kh*kw <= 25");
    CNK_CHECK_ARG(x && w && y, "cnk_dwconv2d_fwd: null pointer");
    return dwconv_fwd(d, x, w, bias, y, S(stream));
}

int cnk_dwconv2d_dgrad(const cnk_conv_desc* d, const void* dy, const float* w, void* dx, void* stream) {
    if (int rc = check_desc(d, "cnk_dwconv2d_dgrad")) return rc;
    CNK_CHECK_ARG(dwconv_supported(d), "cnk_dwconv2d_dgrad: needs Cin == Cout and kh*kw <= 25");
    CNK_CHECK_ARG(dy && w && dx, "cnk_dwconv2d_dgrad: null pointer");
    return dwconv_dgrad(d, dy, w, dx, S(stream));
}

int cnk_dwconv2d_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d, "cnk_dwconv2d_wgrad")) return rc;
    CNK_CHECK_ARG(dwconv_supported(d), "cnk_dwconv2d_wgrad: needs Cin == Cout and kh*kw <= 25");
    CNK_CHECK_ARG(dy && x && dw, "cnk_dwconv2d_wgrad: null pointer");
    CNK_CHECK_ARG(workspace && workspace_bytes >= dw_ws(d), "cnk_dwconv2d_wgrad: workspace too small");
    if (int rc = dwconv_wgrad(d, dy, x, dw, accumulate, (float*)workspace, S(stream))) return rc;
    if (dbias) {
        int Ho, Wo;
        cnk_conv2d_out_hw(d, &Ho, &Wo);
        return colsum(dy, dbias, (int64_t)d->N * Ho * Wo, d->Cout, d->dtype, accumulate, workspace, workspace_bytes, S(stream));
    }
    return 0;
}

int cnk_grouped_expand(const float* w, float* w_dense, int32_t Cout, int32_t Cin, int32_t groups, int32_t T, void* stream) {
    CNK_CHECK_ARG(w && w_dense && Cout > 0 && Cin > 0 && T > 0 && groups > 0 && Cout % groups == 0 && Cin % groups == 0, "cnk_grouped_expand: bad arguments");
    return grouped_expand(w, w_dense, Cout, Cin, groups, T, S(stream));
}
int cnk_grouped_extract(const float* dw_dense, float* dw, int32_t Cout, int32_t Cin, int32_t groups, int32_t T, void* stream) {
    CNK_CHECK_ARG(dw_dense && dw && Cout > 0 && Cin > 0 && T > 0 && groups > 0 && Cout % groups == 0 && Cin % groups == 0, "cnk_grouped_extract: bad arguments");
    return grouped_extract(dw_dense, dw, Cout, Cin, groups, T, S(stream));
}
int cnk_channel_shuffle(const void* x, void* y, int64_t pixels, int32_t C, int32_t groups, int dtype, void* stream) {
    CNK_CHECK_ARG(x && y && pixels >= 0 && C > 0 && groups > 0 && C % groups == 0, "cnk_channel_shuffle: bad arguments");
    return channel_shuffle(x, y, pixels, C, groups, dtype, S(stream));
}

size_t cnk_linear_workspace_bytes(int pass, int32_t B, int32_t K, int32_t N) {
    if (B <= 0 || K <= 0 || N <= 0 || pass < 0 || pass > 2) return 0;
    size_t n = linear_workspace(pass, B, K, N);
    if (pass == CNK_PASS_WGRAD) {
        size_t c = colsum_workspace(B, N);
        if (c > n) n = c;
    }
    return n;
}

int cnk_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t K, int32_t N, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(x && w && y && B > 0 && K > 0 && N > 0, "cnk_linear_fwd: bad arguments");
    CNK_CHECK_ARG(workspace_bytes >= linear_workspace(0, B, K, N) && (workspace || !linear_workspace(0, B, K, N)), "cnk_linear_fwd: workspace too small");
    return linear_fwd(x, w, bias, y, B, K, N, (float*)workspace, workspace_bytes, S(stream));
}

int cnk_linear_dgrad(const float* dy, const float* w, float* dx, int32_t B, int32_t K, int32_t N, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(dy && w && dx && B > 0 && K > 0 && N > 0, "cnk_linear_dgrad: bad arguments");
    CNK_CHECK_ARG(workspace_bytes >= linear_workspace(1, B, K, N) && (workspace || !linear_workspace(1, B, K, N)), "cnk_linear_dgrad: workspace too small");
    return linear_dgrad(dy, w, dx, B, K, N, (float*)workspace, workspace_bytes, S(stream));
}

int cnk_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias, int accumulate, int32_t B, int32_t K, int32_t N, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(dy && x && dw && B > 0 && K > 0 && N > 0, "cnk_linear_wgrad: bad arguments");
    size_t need = cnk_linear_workspace_bytes(CNK_PASS_WGRAD, B, K, N);
    CNK_CHECK_ARG(workspace_bytes >= (dbias ? need : linear_workspace(2, B, K, N)) && (workspace || !need), "cnk_linear_wgrad: workspace too small");
    if (int rc = linear_wgrad(dy, x, dw, B, K, N, accumulate, (float*)workspace, workspace_bytes, S(stream))) return rc;
    if (dbias) return colsum(dy, dbias, B, N, CNK_F32, accumulate, workspace, workspace_bytes, S(stream));
    return 0;
}

int cnk_softmax_ce_sum(const float* logits, const int64_t* target, float* loss, float* dlogits, int32_t* argmax, int32_t B, int32_t N, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(logits && target && loss && B > 0 && N > 0, "cnk_softmax_ce_sum: bad arguments");
    CNK_CHECK_ARG(workspace && workspace_bytes >= (size_t)B * sizeof(float), "cnk_softmax_ce_sum: workspace too small");
    return softmax_ce_sum(logits, target, loss, dlogits, argmax, (float*)workspace, B, N, S(stream));
}

int cnk_dropout_fwd(const void* x, void* y, uint8_t* mask_bits, int64_t n, float p, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int dtype, void* stream) {
    CNK_CHECK_ARG(x && y && mask_bits && n >= 0 && p >= 0.f && p < 1.f, "cnk_dropout_fwd: bad arguments (0 <= p < 1)");
    CNK_CHECK_ARG(dtype == CNK_F32 || dtype == CNK_BF16, "cnk_dropout_fwd: bad dtype %d", dtype);
    CNK_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "cnk_dropout_fwd: tensors must be 16-byte aligned");
    if (n == 0) return 0;
    return dropout_fwd(x, y, mask_bits, n, p, seed, offset, offset_dev, dtype, S(stream));
}

int cnk_dropout_bwd(const void* dy, const uint8_t* mask_bits, void* dx, int64_t n, float p, int dtype, void* stream) {
    CNK_CHECK_ARG(dy && mask_bits && dx && n >= 0 && p >= 0.f && p < 1.f, "cnk_dropout_bwd: bad arguments (0 <= p < 1)");
    CNK_CHECK_ARG(dtype == CNK_F32 || dtype == CNK_BF16, "cnk_dropout_bwd: bad dtype %d", dtype);
    CNK_CHECK_ARG((uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0, "cnk_dropout_bwd: tensors must be 16-byte aligned");
    if (n == 0) return 0;
    return dropout_bwd(dy, mask_bits, dx, n, p, dtype, S(stream));
}

int cnk_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream) {
    CNK_CHECK_ARG(x && y && n >= 0, "cnk_sigmoid_fwd: bad arguments");
    return n ? sigmoid_fwd(x, y, n, S(stream)) : 0;
}
int cnk_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    CNK_CHECK_ARG(dy && y && dx && n >= 0, "cnk_sigmoid_bwd: bad arguments");
    return n ? sigmoid_bwd(dy, y, dx, n, S(stream)) : 0;
}
int cnk_channel_scale_fwd(const void* x, const float* gate, void* y, int32_t N, int32_t HW, int32_t C, int dtype, void* stream) {
    CNK_CHECK_ARG(x && gate && y && N > 0 && HW > 0 && C > 0, "cnk_channel_scale_fwd: bad arguments");
    if (sk_supported(C, 1, dtype)) return sk_scale_fwd(x, gate, y, N, HW, C, dtype, S(stream));      // 16-byte lanes along C
    return channel_scale_fwd(x, gate, y, N, HW, C, dtype, S(stream));
}
size_t cnk_channel_scale_workspace_bytes(int32_t N, int32_t HW, int32_t C, int dtype) {
    return N > 0 && HW > 0 && C > 0 && sk_supported(C, 1, dtype) ? sk_workspace(N, HW, C, 1) : 0;
}
int cnk_channel_scale_bwd(const void* dy, const void* x, const float* gate, void* dx, float* dgate, int32_t N, int32_t HW, int32_t C, int dtype, void* ws, size_t ws_bytes, void* stream) {
    CNK_CHECK_ARG(dy && x && gate && dgate && N > 0 && HW > 0 && C > 0, "cnk_channel_scale_bwd: bad arguments");
    CNK_CHECK_ARG(dx || sk_supported(C, 1, dtype), "cnk_channel_scale_bwd: dx may be NULL only on the vector path (C a multiple of one 16-byte vector)");
    if (sk_supported(C, 1, dtype)) {
        CNK_CHECK_ARG(ws && ws_bytes >= sk_workspace(N, HW, C, 1), "cnk_channel_scale_bwd: workspace too small (cnk_channel_scale_workspace_bytes)");
        return sk_scale_bwd(dy, x, gate, dx, dgate, N, HW, C, dtype, (float*)ws, S(stream));
    }
    return channel_scale_bwd(dy, x, gate, dx, dgate, N, HW, C, dtype, S(stream));
}

static int sk_check(const char* fn, const void* const* ptrs, int P, int N, int HW, int C, int dtype) {
    CNK_CHECK_ARG(dtype == CNK_F32 || dtype == CNK_BF16, "%s: bad dtype %d", fn, dtype);
    CNK_CHECK_ARG(N > 0 && HW > 0 && C > 0 && (int64_t)N * C < (1ll << 31), "%s: bad shape", fn);
    CNK_CHECK_ARG(sk_supported(C, P, dtype), "%s: needs 1..4 paths and C %% %d == 0 (got P=%d, C=%d)", fn, dtype == CNK_BF16 ? 8 : 4, P, C);
    CNK_CHECK_ARG(ptrs, "%s: null pointer table", fn);
    for (int i = 0; i < P; ++i) CNK_CHECK_ARG(ptrs[i], "%s: null pointer for path %d", fn, i);
    return 0;
}
size_t cnk_sk_workspace_bytes(int32_t N, int32_t HW, int32_t C, int32_t P) {
    return N > 0 && HW > 0 && C > 0 && P > 0 ? sk_workspace(N, HW, C, P) : 0;
}
int cnk_sk_pool_fwd(const void* const* u, int32_t P, float* pooled, int32_t N, int32_t HW, int32_t C, int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = sk_check("cnk_sk_pool_fwd", u, P, N, HW, C, dtype)) return rc;
    CNK_CHECK_ARG(pooled && ws && ws_bytes >= sk_workspace(N, HW, C, 1), "cnk_sk_pool_fwd: null output or workspace too small");
    return sk_pool_fwd(u, P, pooled, N, HW, C, dtype, (float*)ws, S(stream));
}
int cnk_sk_select_fwd(const void* const* u, const float* const* logits, int32_t P, float* attn, void* out, int32_t N, int32_t HW, int32_t C, int dtype, void* stream) {
    if (int rc = sk_check("cnk_sk_select_fwd", u, P, N, HW, C, dtype)) return rc;
    if (int rc = sk_check("cnk_sk_select_fwd", (const void* const*)logits, P, N, HW, C, dtype)) return rc;
    CNK_CHECK_ARG(attn && out, "cnk_sk_select_fwd: null output");
    return sk_select_fwd(u, logits, P, attn, out, N, HW, C, dtype, S(stream));
}
int cnk_sk_select_bwd_logits(const void* const* u, const float* attn, const void* dout, float* const* dlogits, int32_t P, int32_t N, int32_t HW, int32_t C, int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = sk_check("cnk_sk_select_bwd_logits", u, P, N, HW, C, dtype)) return rc;
    if (int rc = sk_check("cnk_sk_select_bwd_logits", (const void* const*)dlogits, P, N, HW, C, dtype)) return rc;
    CNK_CHECK_ARG(attn && dout && ws && ws_bytes >= sk_workspace(N, HW, C, P), "cnk_sk_select_bwd_logits: null operand or workspace too small");
    return sk_select_bwd_logits(u, attn, dout, dlogits, P, N, HW, C, dtype, (float*)ws, S(stream));
}
int cnk_sk_maps_bwd(const void* dout, const float* attn, const float* dpooled, void* const* du, int32_t P, int32_t N, int32_t HW, int32_t C, int dtype, void* stream) {
    if (int rc = sk_check("cnk_sk_maps_bwd", (const void* const*)du, P, N, HW, C, dtype)) return rc;
    CNK_CHECK_ARG((dout && attn) || dpooled, "cnk_sk_maps_bwd: needs dout + attn and/or dpooled");
    CNK_CHECK_ARG(!dout == !attn, "cnk_sk_maps_bwd: dout and attn go together");
    return sk_maps_bwd(dout, attn, dpooled, du, P, N, HW, C, dtype, S(stream));
}

int cnk_input_pipeline(const uint8_t* x, void* y, const int32_t* crop_yx, const uint8_t* flip, const float* mean, const float* std, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t pad, int dtype, void* stream) {
    CNK_CHECK_ARG(x && y && mean && std && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && pad >= 0, "cnk_input_pipeline: bad arguments");
    CNK_CHECK_ARG(Ho <= H + 2 * pad && Wo <= W + 2 * pad, "cnk_input_pipeline: crop larger than the padded image");
    return input_pipeline(x, y, crop_yx, flip, nullptr, mean, std, N, H, W, C, Ho, Wo, pad, dtype, S(stream));
}
int cnk_input_pipeline_affine(const uint8_t* x, void* y, const int32_t* crop_yx, const uint8_t* flip, const int32_t* affine_fix, const float* mean, const float* std, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t pad, int dtype, void* stream) {
    CNK_CHECK_ARG(x && y && mean && std && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && pad >= 0, "cnk_input_pipeline_affine: bad arguments");
    CNK_CHECK_ARG(Ho <= H + 2 * pad && Wo <= W + 2 * pad, "cnk_input_pipeline_affine: crop larger than the padded image");
    CNK_CHECK_ARG(Ho < 16384 && Wo < 16384, "cnk_input_pipeline_affine: image too large for 16.16 fixed-point coordinates");
    return input_pipeline(x, y, crop_yx, flip, affine_fix, mean, std, N, H, W, C, Ho, Wo, pad, dtype, S(stream));
}
int cnk_affine_fixed_coeffs(double angle_deg, double translate_x, double translate_y, double scale, double shear_x_deg, double shear_y_deg, int32_t W, int32_t H, int32_t* out6) {
    CNK_CHECK_ARG(out6 && W > 0 && H > 0 && scale > 0.0, "cnk_affine_fixed_coeffs: bad arguments");
    affine_fixed_coeffs(angle_deg, translate_x, translate_y, scale, shear_x_deg, shear_y_deg, W, H, out6);
    return 0;
}

size_t cnk_grad_clip_workspace_bytes(void) { return grad_clip_workspace_bytes(); }

int cnk_grad_clip_norm(float* grads, int64_t n, float max_norm, float* total_norm_out, void* workspace, size_t workspace_bytes, void* stream) {
    CNK_CHECK_ARG(grads && n > 0 && max_norm > 0.f, "cnk_grad_clip_norm: bad arguments");
    CNK_CHECK_ARG((uintptr_t)grads % 16 == 0, "cnk_grad_clip_norm: arena must be 16-byte aligned");
    CNK_CHECK_ARG(workspace && workspace_bytes >= grad_clip_workspace_bytes(), "cnk_grad_clip_norm: workspace too small");
    return grad_clip_norm(grads, n, max_norm, total_norm_out, (float*)workspace, S(stream));
}

int cnk_grad_clip_value(float* grads, int64_t n, float clip_value, void* stream) {
    CNK_CHECK_ARG(grads && n > 0 && clip_value >= 0.f, "cnk_grad_clip_value: bad arguments");
    CNK_CHECK_ARG((uintptr_t)grads % 16 == 0, "cnk_grad_clip_value: arena must be 16-byte aligned");
    return grad_clip_value(grads, n, clip_value, S(stream));
}

int cnk_prof_begin(int32_t class_mask) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
    g_prof.clear();
    g_prof_mask = class_mask;
    return 0;
}
int cnk_prof_collect(int32_t cls, int64_t* count, double* total_ms, double* total_flops, double* total_bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int64_t n = 0;
    double ms = 0, fl = 0, by = 0;
    for (auto& r : g_prof) {
        if (!(r.cls & cls)) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) { cnk_set_error("cnk_prof_collect: event sync failed"); return 3; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) { cnk_set_error("cnk_prof_collect: elapsed failed"); return 3; }
        ms += t; fl += r.flops; by += r.bytes; ++n;
    }
    if (count) *count = n;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (total_bytes) *total_bytes = by;
    return 0;
}
int cnk_prof_end(void) { return cnk_prof_begin(0); }

}  // extern "C"
