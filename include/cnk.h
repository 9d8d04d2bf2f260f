/* cnk.h -- C ABI of the MI355X-native conv / batch-norm / pool / ReLU training path.
 *
 * The reference (milan-pavlovic-ai/ConvNets) has no FFI or plugin interface of its own: its
 * narrowest seam is the layer factory of MultiClassBaseModel (src/basemodel.py:966-1042), whose
 * returned torch.nn modules do all arithmetic inside third-party torch.  Each entry point below
 * therefore replaces the torch operator that one reference call site reaches; the citation on
 * every declaration names that call site (file:line in /root/reference/src).  The Python host
 * side (convnets_amd/) binds these through ctypes and mirrors the reference factory on top.
 *
 * Conventions
 *   - every function returns 0 on success, nonzero on error; cnk_last_error() gives the message
 *     (thread-local).  The library never exits, aborts, allocates, frees, or synchronises.
 *   - all tensor arguments are raw DEVICE pointers owned by the caller (inputs, outputs, saved
 *     statistics, workspaces); nothing is retained across calls.
 *   - activations are NHWC ("channels_last"): x[n][h][w][c], element type given by `dtype`
 *     (CNK_F32 or CNK_BF16).  Master weights, weight/bias gradients, BN parameters and
 *     statistics are always fp32.  Master conv weights keep the reference layout
 *     [Cout][Cin][kh][kw] (nn.Conv2d.weight, basemodel.py:974).
 *   - `stream` is a hipStream_t passed as void*; kernels are launched only on it.
 *   - re-entrant: forward is called from the Python main thread, backward from the autograd
 *     engine thread (loss.backward(), basemodel.py:278).
 */
#ifndef CNK_H
#define CNK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CNK_F32 = 0, CNK_BF16 = 1 };
enum { CNK_PASS_FWD = 0, CNK_PASS_DGRAD = 1, CNK_PASS_WGRAD = 2 };

/* One nn.Conv2d as the reference factory builds it: dilation=1, zero padding (basemodel.py:966-979;
 * kwargs forwarded verbatim at :974).  groups: 0 or 1 = dense.  groups > 1 (ResNeXt's cardinality,
 * resnext.py:82-96) needs Cin == Cout, Cin % 64 == 0 and a group width Cin/groups that
 * divides 64: each 64-channel chunk of the output depends only on the same chunk of the input, so the
 * implicit-GEMM kernels run it as a block-diagonal 64 -> 64 conv per chunk (weights [Cout][Cin/groups][kh][kw]).
 * groups == Cin is the depthwise case and has its own entry points (cnk_dwconv2d_*). */
typedef struct cnk_conv_desc {
    int32_t N, H, W, Cin;      /* input  x[N][H][W][Cin]              */
    int32_t Cout, kh, kw;      /* weight w[Cout][Cin/groups][kh][kw] (fp32)  */
    int32_t stride_h, stride_w;
    int32_t pad_h, pad_w;
    int32_t dtype;             /* element type of x, y, dy, dx        */
    int32_t groups;
    int32_t dilation;          /* 0 or 1 = none; > 1 (SKNet's second path, sknet.py:198): implicit-GEMM shapes only */
} cnk_conv_desc;

int cnk_version(void);
const char* cnk_last_error(void);

/* A HIP stream of the caller's own (hipStreamNonBlocking) on the current device, for the second stream the backward pass runs
 * its weight gradients on (autograd issues both gradients of a conv from one node, basemodel.py:278; the host side overlaps
 * them).  Why the library hands it out: a host framework's stream objects may share handles -- torch.cuda.Stream() deals the
 * 32 streams of a per-device pool round-robin, so the 33rd object of a process IS the first again -- and a "second" stream
 * that silently equals the compute stream serialises the step and lets both sides' scratch buffers alias. */
int cnk_stream_create(void** stream);
/* An empty kernel launch on `stream`.  For hosts that capture the two-stream step into a hipGraph: the runtime's graph executor
 * derives its hardware queues from the order in which a node's children were created.  When a fork point (event record on the
 * compute stream + wait on the weight-gradient stream) is followed by a weight-gradient launch BEFORE the compute stream's next
 * launch, the replay of the ResNet-50 step degraded from 18.6 to 26-30 ms (two different changes); this launch
 * on the compute stream right behind the event record restored it.  An empirical remedy for that pattern -- measure the
 * replay after any change of launch order (behind the all-reduce forks of a data-parallel step it made things worse). */
int cnk_graph_anchor(void* stream);
int cnk_stream_destroy(void* stream);

/* Output spatial size, floor((H + 2p - dilation*(k-1) - 1)/s) + 1 -- the reference's shape tracker formula
 * (basemodel.py:770-771). */
int cnk_conv2d_out_hw(const cnk_conv_desc* d, int32_t* Ho, int32_t* Wo);

/* Bytes of packed weights for `pass` (FWD or DGRAD) and of workspace for `pass`. */
size_t cnk_conv2d_packed_bytes(const cnk_conv_desc* d, int pass);
size_t cnk_conv2d_workspace_bytes(const cnk_conv_desc* d, int pass);

/* Re-lays the fp32 master weight into the operand images the kernels stream:
 *   w_fwd   [Cout][kh][kw][Cin]            in `dtype` (forward B operand, K contiguous)
 *   w_dgrad [class][Cin][taps(class)][Cout] in `dtype` (dgrad B operand per stride-parity class)
 * Either output may be NULL.  Replaces nothing in the reference (torch keeps its own packed
 * copies inside oneDNN/cuDNN); called once per optimizer step. */
int cnk_conv2d_pack_weights(const cnk_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad,
                            void* stream);

/* Batched form of the re-layout for whole models: cnk_conv2d_pack_jobs appends to a HOST array the
 * jobs that produce the same images as cnk_conv2d_pack_weights for one conv (returns the number
 * written, or -1 if this conv's images cannot be batched: small-C / direct paths); the caller
 * uploads the jobs of every conv plus a block map {job index, first block of that job} -- one
 * block per 1024 output elements -- and then re-lays ALL weights with ONE launch per optimizer
 * step (cnk_pack_batch) instead of two to five launches per conv. */
typedef struct cnk_pack_job {
    const float* w;            /* master weight [Cout][Cin][KH][KW]            */
    void* out;                 /* image: [R][ntaps][Cred], R/Cred = Cout/Cin or swapped */
    int32_t Cout, Cin, KH, KW, ntaps, transpose, dtype, groups;   /* groups <= 1: dense */
    int8_t kh[32], kw[32];     /* tap list                                      */
    int32_t nblocks;           /* workgroups this job takes in cnk_pack_batch (filled by cnk_conv2d_pack_jobs) */
    int32_t pad_;
} cnk_pack_job;
int cnk_conv2d_pack_jobs(const cnk_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad,
                         cnk_pack_job* jobs, int32_t max_jobs);
int cnk_pack_batch(const cnk_pack_job* jobs_dev, const void* block_map_dev, int32_t nblocks,
                   void* stream);

/* y = conv2d(x, w) (+ bias).  nn.Conv2d.forward reached from basemodel.py:974 via :1004.
 * bias (fp32[Cout]) may be NULL: the reference has a bias only when batch_norm is off (:973).
 * Optional fusion with the BatchNorm2d that follows the conv in conv2d_block (:1008): when
 * bn_partial is non-NULL the epilogue also emits, per 128-row output tile and channel, the
 * (count, mean, M2) of the stored (rounded) outputs: bn_partial[slices][Cout][3] fp32 with
 * slices = cnk_conv2d_fwd_bn_partial_slices(d) (0 when this conv shape cannot emit them);
 * cnk_bn_fwd_train then skips its own statistics pass. */
int64_t cnk_conv2d_fwd_bn_partial_slices(const cnk_conv_desc* d);
int cnk_conv2d_fwd(const cnk_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                   void* y, float* bn_partial, void* workspace, size_t workspace_bytes, void* stream);

/* Inference forward with the layer's whole epilogue in the conv kernel:
 *     y = act( conv(x, w) + bias [+ residual] ),  act = ReLU when relu != 0.
 * With BatchNorm folded into w and bias (w' = w * gamma/sqrt(var+eps), bias' = beta - mean*scale) this is
 * conv2d_block in eval mode (valid_model / test(), basemodel.py:304-346, 601-722) and, with the
 * residual operand, the tail of a ResNet block (resnet.py:151-154,191-194) -- one kernel, no BN pass.
 * residual (may be NULL) is shaped like y.  Implicit-GEMM shapes only: query _supported first. */
int cnk_conv2d_fwd_act_supported(const cnk_conv_desc* d);
int cnk_conv2d_fwd_act(const cnk_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                       const void* residual, int relu, void* y, void* stream);

/* dx = conv2d_transpose(dy, w) [+ addend]: autograd's input gradient of the call above
 * (loss.backward(), basemodel.py:258/278).  `addend` (may be NULL; shaped like dx) is the
 * gradient arriving at the same tensor through another branch -- the identity / projection
 * shortcut of a residual block (resnet.py:141-151,181-191); summing it in the epilogue replaces
 * the separate accumulation pass autograd would run.  addend_mask (may be NULL): the packed ReLU
 * mask cnk_bn_fwd_train wrote for the block output; then `addend` is the gradient of the block
 * OUTPUT and addend*mask -- the identity-shortcut gradient -- is formed in the epilogue instead of
 * being materialised by the BN backward (implicit-GEMM path only). */
int cnk_conv2d_dgrad_accepts_mask(const cnk_conv_desc* d);      /* 1 if addend_mask is usable */

/* Optional fusion of the data gradient with the backward of the BatchNorm(+ReLU) layer whose OUTPUT
 * is this conv's input (conv2d_block chains, basemodel.py:999-1015: ... -> BN -> ReLU -> conv).
 * dx is then the gradient w.r.t. that BN output; the epilogue, which holds each dx tile, also
 *   - applies the layer's ReLU mask to it (relu != 0: from relu_mask, the packed mask
 *     cnk_bn_fwd_train wrote, else re-derived from x*scale+shift > 0 as cnk_bn_bwd does), and
 *   - emits per-tile per-channel partial sums  sum(dx')  and  sum(dx' * (x-mean)*invstd)
 * i.e. the whole first pass of cnk_bn_bwd (one read of dy and one of x saved).  Hand `partial`
 * to cnk_bn_bwd as ext_partial together with the (already masked) dx as its dy.
 * x: the BN layer's INPUT, [N*H*W][Cin] like dx.  partial: fp32 [slices][Cin][2] with
 * slices = cnk_conv2d_dgrad_bn_slices(d) (0: this conv shape cannot fuse; pass bn = NULL). */
typedef struct cnk_bn_bwd_fuse {
    const void* x;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    const uint8_t* relu_mask;
    float* partial;
    int32_t relu;
} cnk_bn_bwd_fuse;
int64_t cnk_conv2d_dgrad_bn_slices(const cnk_conv_desc* d);
int cnk_conv2d_dgrad(const cnk_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                     const void* addend, const uint8_t* addend_mask, const cnk_bn_bwd_fuse* bn,
                     void* workspace, size_t workspace_bytes, void* stream);

/* dw[Cout][Cin][kh][kw] (fp32) = sum over N,Ho,Wo of dy * x ; dbias[Cout] = sum dy (may be NULL).
 * accumulate != 0 adds into dw/dbias (torch's .grad accumulation), else overwrites.
 * workspace: cnk_conv2d_workspace_bytes(d, CNK_PASS_WGRAD) gives the deepest split of the pixel range (it fills the chip).
 * The MFMA kernels accept any workspace of at least ONE fp32 slab [Cout][kh][kw][Cin] (rounded up to 256 bytes, plus the
 * bias sum's scratch when dbias != NULL): a smaller workspace bounds the number of splits -- fewer, longer workgroups, less
 * slab traffic -- which is what a caller wants when the weight gradient runs on a side stream BESIDE other kernels and only
 * has to finish with them (convnets_amd/ops.py).  Results differ only in the fp32 summation order of the slabs. */
int cnk_conv2d_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw,
                     float* dbias, int accumulate, void* workspace, size_t workspace_bytes,
                     void* stream);
/* Chained slab sums (round 5; csrc/wgrad_chain.h).  A split-K weight gradient leaves fp32 slabs in its workspace that a second,
 * tiny launch sums into dw: 52 latency-bound launches per ResNet-50 step on the weight-gradient stream (basemodel.py:278 runs
 * all of them inside loss.backward()).  The _chain entry points run the kernel WITHOUT that sum and describe it in `*out`;
 * handed as `prev` to the NEXT weight-gradient call on the same stream, that launch's workgroups sum it in their prologue (no
 * launch of its own); the last one of a chain goes to cnk_wgrad_reduce_pending.  The caller keeps `workspace` untouched until
 * the call that received the descriptor as `prev` (or cnk_wgrad_reduce_pending) has been issued, i.e. consecutive calls
 * alternate between two workspaces.  Same results as cnk_conv2d_wgrad up to the fp32 summation order of the slabs;
 * bit-reproducible.  _supported: dense convs on the MFMA weight-gradient kernels, Cin % 4 == 0, no dbias. */
typedef struct cnk_wgrad_pending {
    const float* slab;         /* NULL: nothing pending */
    float* dw;
    uint64_t slab_stride;      /* floats between two slabs */
    int32_t splits, Cout, Cin, ntaps, accumulate;
    int32_t lanes;             /* lanes that share a 4-channel group of the sum (chosen by the producing call) */
} cnk_wgrad_pending;
int cnk_conv2d_wgrad_chain_supported(const cnk_conv_desc* d);
int cnk_conv2d_wgrad_chain(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, int accumulate, void* workspace,
                           size_t workspace_bytes, const cnk_wgrad_pending* prev, cnk_wgrad_pending* out, void* stream);
int cnk_wgrad_reduce_pending(const cnk_wgrad_pending* pending, void* stream);

/* The smallest workspace cnk_conv2d_wgrad accepts for this shape (has_bias: dbias != NULL): one slab (+ the bias sum's
 * scratch) for the split-K kernels, cnk_conv2d_workspace_bytes(d, CNK_PASS_WGRAD) for every other kernel. */
size_t cnk_conv2d_wgrad_min_workspace_bytes(const cnk_conv_desc* d, int has_bias);

/* nn.BatchNorm2d in training mode (basemodel.py:1008,1061), optionally fused with the ReLU that
 * follows it in conv2d_block (:1012) and with the residual add + F.relu of the ResNet blocks
 * (resnet.py:151-154,191-194):   y = act( (x-mean)*invstd*gamma + beta [+ residual] ).
 * x,y,residual: [M][C] NHWC-flattened, M = N*H*W.  Batch mean / biased variance are computed in
 * fp32; running_mean/var (may be NULL) get momentum updates with the UNBIASED variance;
 * save_mean/save_invstd (fp32[C]) are outputs kept for the backward.  partial/partial_slices:
 * optional statistics partials from cnk_conv2d_fwd (NULL/0: computed here).  y == NULL: statistics only (scale[C] | shift[C]
 * are left at the start of the workspace); x may then be NULL too when the partials are given (it is never read).
 * relu_mask (may be NULL; cnk_bn_mask_bytes(M, C, dtype) bytes): packed ReLU mask of the output,
 * one bit per element; handing it to cnk_bn_bwd replaces two full reads of y by 1/16 of the bytes.
 * workspace: cnk_bn_workspace_bytes(M, C). */
/* Concurrency: calls on ONE stream may follow each other freely (stream order is all the BN entry points rely on); calls on
 * different streams may run concurrently.  The merge + finalise launch behind long partial lists keeps last-block tickets in
 * device memory owned by the library, banked per stream (csrc/bn.hip); the one unsupported pattern is replaying two CAPTURED
 * graphs that contain BN calls of the same original stream concurrently on two streams. */
size_t cnk_bn_workspace_bytes(int64_t M, int32_t C);
size_t cnk_bn_mask_bytes(int64_t M, int32_t C, int dtype);
int cnk_bn_fwd_train(const void* x, const float* gamma, const float* beta, const void* residual,
                     void* y, uint8_t* relu_mask, float* running_mean, float* running_var, float* save_mean,
                     float* save_invstd, const float* partial, int64_t partial_slices, int64_t M,
                     int32_t C, float eps, float momentum, int relu, int dtype, void* workspace,
                     size_t workspace_bytes, void* stream);

/* The same with a residual operand whose own BatchNorm apply pass was DEFERRED into this call: a projection shortcut is
 * conv -> BN without activation (resnet.py:90-91) and its only consumer is this layer's `output += identity`
 * (resnet.py:191-194).  The shortcut's BN is then run with y == NULL (statistics only: its scale[C] | shift[C] are left at
 * the start of its workspace; copy them out) and `residual` here is the shortcut CONV's output;
 *   y = act( (x-mean)*invstd*gamma + beta + round_dtype(residual*scale + shift) )
 * -- bit for bit what the two-call sequence stores, without writing and re-reading the normalised shortcut tensor.
 * residual_scale_shift: fp32 [2][C]. */
int cnk_bn_fwd_train_res_affine(const void* x, const float* gamma, const float* beta, const void* residual,
                                const float* residual_scale_shift, void* y, uint8_t* relu_mask, float* running_mean,
                                float* running_var, float* save_mean, float* save_invstd, const float* partial,
                                int64_t partial_slices, int64_t M, int32_t C, float eps, float momentum, int relu,
                                int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Eval mode (valid_model, basemodel.py:311): running statistics, same fused epilogue. */
int cnk_bn_fwd_eval(const void* x, const float* gamma, const float* beta, const float* running_mean,
                    const float* running_var, const void* residual, void* y, int64_t M, int32_t C,
                    float eps, int relu, int dtype, void* stream);

/* Backward of cnk_bn_fwd_train.  The ReLU mask is taken from the forward OUTPUT like nn.ReLU's
 * backward: from relu_mask when given (the packed mask the forward wrote), else from y; when both
 * are NULL (allowed only without a residual operand) it is re-derived as x*scale+shift > 0 with
 * the same fp32 arithmetic as the forward.  dresidual (may be NULL) receives the gradient of the residual operand
 * (= masked dy).  dgamma/dbeta fp32[C]; accumulate as in wgrad.
 * ext_partial/ext_slices (NULL/0: computed here): the partial sums a fused cnk_conv2d_dgrad emitted;
 * dy is then ALREADY masked (it is also the residual gradient: dresidual must be NULL) and the
 * reduction pass is skipped. */
int cnk_bn_bwd(const void* dy, const void* x, const void* y, const uint8_t* relu_mask,
               const float* gamma, const float* beta,
               const float* save_mean, const float* save_invstd, void* dx, void* dresidual,
               float* dgamma, float* dbeta, int accumulate, const float* ext_partial,
               int64_t ext_slices, int64_t M, int32_t C, int relu,
               int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of conv(1x1, stride 1, no bias) -> BatchNorm2d with the BN layer's apply pass FOLDED into the conv's own backward
 * (conv2d_block, basemodel.py:999-1015, as the bottleneck's last block uses it, resnet.py:172-176; autograd runs the two
 * backward passes at basemodel.py:278).  The BN backward dy_conv = A (g - B1 - xhat B2) is linear in g (the masked gradient
 * w.r.t. the BN output) and, because y = W a, in the conv input a, so
 *     da = [g | a] x [A W ; -W^T diag(A invstd B2) W]  +  c
 * is ONE GEMM over K = Cout + Cin and the [M][Cout] tensor dy_conv is never written (csrc/bnfold.hip).  Sequence:
 *   1. a consumer's cnk_conv2d_dgrad with a cnk_bn_bwd_fuse record leaves g (masked) and the partial sums;
 *   2. cnk_bn_bwd_coef: dgamma / dbeta and coef[3][C] = A | B1 | B2 (fp32) from those partials (no apply pass);
 *      workspace: cnk_bn_workspace_bytes(M, C).  With ext_partial == NULL (and ext_slices == 0) the call runs the
 *      reduction pass itself over (dy, x): for a BN layer WITHOUT activation whose gradient arrives as a plain tensor (a
 *      projection shortcut, resnet.py:90-91);
 *   3. cnk_conv2d_bnfold_prepare: the per-step weight image w_fold (cnk_conv2d_bnfold_image_bytes) and bias_fold[Cin] from
 *      the fp32 master weights w [Cout][Cin], the packed data-gradient image w_dgrad and coef;
 *   4. cnk_conv2d_dgrad_bnfold(g, x = the conv input, ...) -> dx, with the same optional gradient addend (+ packed mask) and
 *      fused BN backward of the layer that produced x as cnk_conv2d_dgrad;  cnk_conv2d_wgrad_bnfold(g, y = the conv output, x, ...) -> dw, which forms dy_conv
 *      while staging its tiles, bit for bit the tensor cnk_bn_bwd would have written (workspace as cnk_conv2d_wgrad).
 * cnk_conv2d_bnfold_supported: bf16, 1x1, stride 1, no padding, dense, both channel counts multiples of 64.  (Round 4 also had a
 * second form for convs that narrow -- the BN input as second K segment -- and the fold of strided 1x1 convs; both measured
 * slower / neutral in the training step, DESIGN.md 7b, and were removed in round 5.) */
int cnk_conv2d_bnfold_supported(const cnk_conv_desc* d);
size_t cnk_conv2d_bnfold_image_bytes(const cnk_conv_desc* d);
int cnk_bn_bwd_coef(const void* dy, const void* x, const float* ext_partial, int64_t ext_slices, const float* gamma,
                    const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, int accumulate,
                    float* coef, int64_t M, int32_t C, int dtype, void* workspace, size_t workspace_bytes, void* stream);
int cnk_conv2d_bnfold_prepare(const cnk_conv_desc* d, const float* w, const void* w_dgrad, const float* save_mean,
                              const float* save_invstd, const float* coef, void* w_fold, float* bias_fold, void* stream);
int cnk_conv2d_dgrad_bnfold(const cnk_conv_desc* d, const void* g, const void* x, const void* w_fold,
                            const float* bias_fold, void* dx, const void* addend, const uint8_t* addend_mask,
                            const cnk_bn_bwd_fuse* bn, void* stream);
int cnk_conv2d_wgrad_bnfold(const cnk_conv_desc* d, const void* g, const void* y, const void* x, const float* save_mean,
                            const float* save_invstd, const float* coef, float* dw, int accumulate, void* workspace,
                            size_t workspace_bytes, void* stream);
/* ... with its slab sum chained like cnk_conv2d_wgrad_chain (declared above) */
int cnk_conv2d_wgrad_bnfold_chain(const cnk_conv_desc* d, const void* g, const void* y, const void* x, const float* save_mean,
                                  const float* save_invstd, const float* coef, float* dw, int accumulate, void* workspace,
                                  size_t workspace_bytes, const struct cnk_wgrad_pending* prev, struct cnk_wgrad_pending* out,
                                  void* stream);

/* Both gradients of the folded conv -> BN pair in ONE pass over the operands (csrc/conv_fold3.hip): the 64 -> 256-channel 1x1 convs
 * of ResNet's first stage (resnet.py:172-176 and the projection shortcut 160-166 at 56x56; autograd's backward, basemodel.py:278).
 * dx as cnk_conv2d_dgrad_bnfold (`addend` without a mask, or `bn` -- not both; `bn` re-derives its ReLU mask: relu_mask must be
 * NULL; its `partial` holds cnk_conv2d_bwd_bnfold_bn_slices(d) records).  dw as cnk_conv2d_wgrad_bnfold, from the Gram products g^T x and
 * x^T x instead of the BN input:  dw[k][j] (+)= A_k (G[k][j] - B1_k S_j) - A_k invstd_k B2_k ((bf16(w) X)[k][j] - mean_k S_j),
 * so the conv output is never read (it is bf16(w) x up to the rounding of the stored tensor) and neither operand is read twice.
 * w: the fp32 master weights [Cout][Cin].  save_mean / save_invstd / coef as for cnk_conv2d_wgrad_bnfold.  Everything runs on
 * `stream`, in order.  workspace: cnk_conv2d_bwd_bnfold_workspace_bytes(d). */
int cnk_conv2d_bwd_bnfold_supported(const cnk_conv_desc* d);
size_t cnk_conv2d_bwd_bnfold_workspace_bytes(const cnk_conv_desc* d);
int64_t cnk_conv2d_bwd_bnfold_bn_slices(const cnk_conv_desc* d);
int cnk_conv2d_bwd_bnfold(const cnk_conv_desc* d, const void* g, const void* x, const void* w_fold, const float* bias_fold, void* dx,
                          const void* addend, const cnk_bn_bwd_fuse* bn, const float* w, const float* save_mean, const float* save_invstd,
                          const float* coef, float* dw, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* nn.AvgPool2d (avgpool2d, basemodel.py:1026-1033; DenseNet's transition block, densenet.py:129): floor mode,
 * count_include_pad=True (every window divides by kh*kw), zero padding.  SURVEY.md 8f n4. */
int cnk_avgpool2d_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh,
                      int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream);
int cnk_avgpool2d_bwd(const void* dy, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh,
                      int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream);

/* nn.MaxPool2d (basemodel.py:1017-1024): floor mode, implicit -inf padding, dilation 1.
 * The forward optionally saves, per output element, the window-relative position r*kw+q (uint8)
 * of the arg-max -- the first maximum in row-major window order, as torch -- and the backward
 * routes the gradient from it (argmax[N][Ho][Wo][C]; kh*kw <= 255). */
int cnk_maxpool2d_fwd(const void* x, void* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W,
                      int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph,
                      int32_t pw, int dtype, void* stream);
int cnk_maxpool2d_bwd(const void* dy, const uint8_t* argmax, void* dx, int32_t N, int32_t H,
                      int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                      int32_t ph, int32_t pw, int dtype, void* stream);

/* Forward of a training-mode BatchNorm(+ReLU) and the MaxPool2d that follows it (the stems, resnet.py:88-89) without
 * the full-size normalised tensor: call cnk_bn_fwd_train with y == NULL (statistics only: save_mean / save_invstd / the
 * running statistics are produced and the per-channel scale[C] | shift[C], fp32, are left at the START of its
 * workspace), then this pooling pass, which applies x * scale + shift (+ReLU) to the conv output on the fly, each
 * value rounded to `dtype` as the BN pass would have stored it.  idx as cnk_maxpool2d_fwd (required).  The backward is
 * cnk_bn_bwd_maxpool below, or the regular pair: cnk_maxpool2d_bwd, then cnk_bn_bwd with y == NULL and no mask (it
 * re-derives the ReLU mask). */
int cnk_maxpool2d_fwd_bn(const void* x, const float* scale, const float* shift, int relu, void* y, uint8_t* idx,
                         int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw,
                         int32_t ph, int32_t pw, int dtype, void* stream);
/* The same pass, also keeping x_at_max [N][Ho][Wo][C] (dtype): the INPUT value x at each window's arg-max, for
 * cnk_bn_bwd_maxpool_kept. */
int cnk_maxpool2d_fwd_bn_keep(const void* x, const float* scale, const float* shift, int relu, void* y, uint8_t* idx,
                              void* x_at_max, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw,
                              int32_t sh, int32_t sw, int32_t ph, int32_t pw, int dtype, void* stream);

/* Backward of that pair in one call, for pools where at most 2 x 2 windows cover an element (kh <= 2*sh, kw <= 2*sw:
 * the 3x3/2 stem pool, VGG's 2x2/2): cnk_bn_bwd whose incoming gradient is max-pool-backward(dy_pooled, argmax), gathered
 * on the fly in both BN passes and summed in fp32 in cnk_maxpool2d_bwd's window order -- the full-size gradient of the
 * BN output is neither written nor read.  x: the BN input [N][H][W][C]; dy_pooled / argmax: [N][Ho][Wo][C]; relu != 0
 * re-derives the mask as x*scale+shift > 0 (needs beta).  fp32 results equal the two-call pair up to the order of the
 * per-slice sums (stride-2 pools walk 2 x 2 cells of elements); bf16 ones also differ by the skipped rounding of the
 * intermediate.  Deterministic.  workspace: cnk_bn_workspace_bytes(N*H*W, C). */
int cnk_bn_bwd_maxpool(const void* dy_pooled, const uint8_t* argmax, const void* x, const float* gamma,
                       const float* beta, const float* save_mean, const float* save_invstd, void* dx, float* dgamma,
                       float* dbeta, int accumulate, int32_t N, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw,
                       int32_t sh, int32_t sw, int32_t ph, int32_t pw, int relu, int dtype, void* workspace,
                       size_t workspace_bytes, void* stream);
/* The same backward with the inputs kept by cnk_maxpool2d_fwd_bn_keep: the pooled gradient only reaches arg-max elements,
 * so the reduction pass (dgamma, dbeta and the coefficients of dx) runs over dy_pooled and x_at_max -- a quarter of the
 * bytes of the gather over x for a stride-2 pool (stride-2 pools with <= 2 x 2 covering windows; other geometries ignore
 * x_at_max); the sums equal cnk_bn_bwd_maxpool's up to fp32 summation order.  dx as before. */
int cnk_bn_bwd_maxpool_kept(const void* dy_pooled, const uint8_t* argmax, const void* x, const void* x_at_max,
                            const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                            void* dx, float* dgamma, float* dbeta, int accumulate, int32_t N, int32_t H, int32_t W,
                            int32_t C, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw, int relu,
                            int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* nn.ReLU (basemodel.py:1012,1065) and the block-level  output += identity; F.relu(output)
 * (resnet.py:151-154,191-194) as stand-alone ops for unfused use. */
int cnk_relu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int cnk_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream);
int cnk_add_relu_fwd(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);

/* nn.AdaptiveAvgPool2d(1) (basemodel.py:1035-1042): y[N][C] = mean over H*W; output fp32. */
int cnk_gap_fwd(const void* x, float* y, int32_t N, int32_t HW, int32_t C, int dtype, void* stream);
int cnk_gap_bwd(const float* dy, void* dx, int32_t N, int32_t HW, int32_t C, int dtype,
                void* stream);

/* Layout/precision hand-off at the model boundary: the reference feeds NCHW fp32 batches
 * (basemodel.py:185-188); the path computes in NHWC `dtype`. */
int cnk_nchw_to_nhwc(const float* x_nchw, void* y_nhwc, int32_t N, int32_t C, int32_t H, int32_t W,
                     int dtype, void* stream);

/* On-device input pipeline (mngrdata.py:139-165): RandomCrop(Ho x Wo, padding=pad) +
 * RandomHorizontalFlip + ToTensor + Normalize(mean, std), from uint8 images x[N][H][W][C] (HWC, as decoded) straight
 * to the NHWC `dtype` tensor the convs consume.  The random draws are the caller's: crop_yx (int32 [N][2], origin of
 * the crop in the zero-padded image, 0 <= oy <= H + 2*pad - Ho; NULL: centred at `pad`) and flip (uint8 [N]; NULL: none).
 * mean / std: fp32 [C]. */
int cnk_input_pipeline(const uint8_t* x, void* y, const int32_t* crop_yx, const uint8_t* flip, const float* mean,
                       const float* std, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                       int32_t pad, int dtype, void* stream);

/* The same pass with RandomAffine(degrees, shear, scale) (mngrdata.py:148) as the last geometric transform, i.e. applied
 * to the cropped and flipped Ho x Wo image about its centre, nearest neighbour, fill 0 -- exactly what
 * torchvision hands to PIL: affine_fix (int32 [N][6], NULL: none) holds per image the six 16.16 fixed-point
 * coefficients of PIL's Image.transform(AFFINE, NEAREST) for the inverse matrix torchvision builds from the drawn
 * (angle, translate, scale, shear); cnk_affine_fixed_coeffs computes them on the host (angles in degrees; torchvision's
 * RandomAffine(shear=s) draws shear_x in (-s, s) and leaves shear_y = 0).  Bit-exact with PIL for uint8 images. */
int cnk_input_pipeline_affine(const uint8_t* x, void* y, const int32_t* crop_yx, const uint8_t* flip,
                              const int32_t* affine_fix, const float* mean, const float* std, int32_t N, int32_t H,
                              int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t pad, int dtype, void* stream);
int cnk_affine_fixed_coeffs(double angle_deg, double translate_x, double translate_y, double scale,
                            double shear_x_deg, double shear_y_deg, int32_t W, int32_t H, int32_t* out6);

/* optim.Adam with L2 weight decay as the reference configures it (basemodel.py:64-67; betas
 * (0.9,0.999), eps 1e-8), over one flat fp32 arena holding every parameter:
 *   g += wd*p; m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v/(1-b2^t))+eps)
 * `step` is the 1-based step count t. */
int cnk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, void* stream);
/* hipGraph-captured training step (BASELINE config 5): a captured launch replays with the kernel arguments of
 * capture time, so the step-varying scalars are read from device memory that the host refreshes before every
 * replay.  bias_dev = {1 - beta1^t, sqrt(1 - beta2^t)} of the replayed step t, computed by
 * cnk_adam_bias_corrections (host, the same arithmetic as cnk_adam_step => bitwise-identical updates). */
void cnk_adam_bias_corrections(float beta1, float beta2, int32_t step, float* out2);
int cnk_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, const float* bias_dev, void* stream);

/* ---- depthwise convolution ------------------------------------------------------
 * conv2d_depthwise of the layer factory (basemodel.py:981-997): nn.Conv2d(C, C, groups=C, ...), the
 * spatial half of MobileNet-v1's Conv2dBlockDW (mobilenet_v1.py:92-125).  Same descriptor with
 * Cin == Cout == C (C % 8 == 0 for bf16, % 4 for fp32; kh*kw <= 25); x/y/dy/dx NHWC in `dtype`;
 * w is the fp32 MASTER weight [C][1][kh][kw] (no packed image); dw fp32 in the same layout.
 * wgrad workspace: cnk_dwconv2d_workspace_bytes(d); dbias (may be NULL) = column sums of dy. */
int cnk_dwconv2d_supported(const cnk_conv_desc* d);
size_t cnk_dwconv2d_workspace_bytes(const cnk_conv_desc* d);
int cnk_dwconv2d_fwd(const cnk_conv_desc* d, const void* x, const float* w, const float* bias, void* y,
                     void* stream);
int cnk_dwconv2d_dgrad(const cnk_conv_desc* d, const void* dy, const float* w, void* dx, void* stream);
int cnk_dwconv2d_wgrad(const cnk_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias,
                       int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* General grouped conv (ShuffleNet's grouped 1x1 convs, shufflenet_v1.py:117-141: Cin != Cout, group widths off
 * the 64-channel chunks): run the DENSE conv of the block-diagonal weight.  cnk_grouped_expand writes
 * wd[Cout][Cin][T] (zeros off the diagonal) from w[Cout][Cin/groups][T] (T = kh*kw); cnk_grouped_extract reads the
 * diagonal blocks of the dense weight gradient back into dw[Cout][Cin/groups][T].
 * cnk_channel_shuffle: y[p][j*groups + i] = x[p][i*(C/groups) + j] on NHWC rows (shufflenet_v1.py:173-186); its
 * gradient is the same call with groups' = C/groups. */
int cnk_grouped_expand(const float* w, float* w_dense, int32_t Cout, int32_t Cin, int32_t groups, int32_t T,
                       void* stream);
int cnk_grouped_extract(const float* dw_dense, float* dw, int32_t Cout, int32_t Cin, int32_t groups, int32_t T,
                        void* stream);
int cnk_channel_shuffle(const void* x, void* y, int64_t pixels, int32_t C, int32_t groups, int dtype, void* stream);

/* ---- classifier head -- fp32, row-major, no workspace ---------------------------
 * nn.Linear (resnet.py:106, vggnet.py:83-92): y[B][N] = x[B][K] W[N][K]^T + bias (bias may be NULL);
 * dx[B][K] = dy[B][N] W[N][K]; dW[N][K] (+)= dy^T x, dbias[N] (+)= column sums of dy (dbias may be
 * NULL).  Skinny problems (batch rows x a large weight matrix) are split along the reduction into
 * fp32 slabs summed in fixed order: workspace = cnk_linear_workspace_bytes(pass, B, K, N), pass as
 * CNK_PASS_FWD / _DGRAD / _WGRAD (may be 0). */
size_t cnk_linear_workspace_bytes(int pass, int32_t B, int32_t K, int32_t N);
int cnk_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t K,
                   int32_t N, void* workspace, size_t workspace_bytes, void* stream);
int cnk_linear_dgrad(const float* dy, const float* w, float* dx, int32_t B, int32_t K, int32_t N,
                     void* workspace, size_t workspace_bytes, void* stream);
int cnk_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias, int accumulate, int32_t B,
                     int32_t K, int32_t N, void* workspace, size_t workspace_bytes, void* stream);

/* nn.CrossEntropyLoss(reduction='sum') (basemodel.py:46) on logits[B][N] with int64 class targets:
 * *loss = sum_b (logsumexp(logits_b) - logits_b[target_b]).  dlogits (may be NULL) receives
 * softmax - onehot, i.e. d(loss)/d(logits) for an upstream gradient of 1; argmax (may be NULL,
 * int32[B]) the predicted class per sample (score(), basemodel.py:119).  workspace: B floats. */
int cnk_softmax_ce_sum(const float* logits, const int64_t* target, float* loss, float* dlogits,
                       int32_t* argmax, int32_t B, int32_t N, void* workspace, size_t workspace_bytes,
                       void* stream);

/* nn.Dropout in training mode (resnet.py:105; feature-map dropout: densenet.py dense layers, squeezenet.py head):
 * y = keep ? x/(1-p) : 0 with a counter-based keep decision hash(seed, offset, index) -- reproducible for a given
 * (seed, offset), independent of the launch geometry and of the tensor layout (index = position in memory); x / y
 * are `dtype` (fp32 or bf16), 16-byte aligned.  mask_bits ((n+7)/8 bytes, bit k of byte g = element 8g+k) is the
 * keep pattern for the backward dx = keep ? dy/(1-p) : 0.  offset_dev (NULL: none): the call counter is
 * offset + *offset_dev, device memory the host refreshes before each replay of a hipGraph-captured step.
 * It does not reproduce torch's Philox stream: parity runs use dropout_rate = 0. */
int cnk_dropout_fwd(const void* x, void* y, uint8_t* mask_bits, int64_t n, float p, uint64_t seed,
                    uint64_t offset, const uint64_t* offset_dev, int dtype, void* stream);
int cnk_dropout_bwd(const void* dy, const uint8_t* mask_bits, void* dx, int64_t n, float p, int dtype,
                    void* stream);

/* Squeeze-and-excitation unit (SEUnit, senet.py:215-254 / se_resnet.py:216-255): nn.Sigmoid on the fp32 [N][C]
 * descriptor, and the scale step x * gate.expand_as(x) on NHWC activations (gate fp32 [N][C]):
 *   y[n][p][c] = x[n][p][c] * gate[n][c];   dx = dy * gate;   dgate[n][c] = sum_p dy * x.   SURVEY.md 8f n4. */
int cnk_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream);
int cnk_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
int cnk_channel_scale_fwd(const void* x, const float* gate, void* y, int32_t N, int32_t HW, int32_t C, int dtype,
                          void* stream);
/* ws: cnk_channel_scale_workspace_bytes(N, HW, C, dtype) bytes (0 when C is not a multiple of one 16-byte vector:
 * the scalar kernels then need none) -- fixed-order partial sums of dgate.  dx may be NULL on the vector path: only
 * dgate is produced, and the caller writes dx = dy * gate + dpooled / HW in one pass with cnk_sk_maps_bwd (P = 1,
 * attn = gate) once the squeeze branch's gradient dpooled is known (layers.SEUnit does). */
size_t cnk_channel_scale_workspace_bytes(int32_t N, int32_t HW, int32_t C, int dtype);
int cnk_channel_scale_bwd(const void* dy, const void* x, const float* gate, void* dx, float* dgate, int32_t N,
                          int32_t HW, int32_t C, int dtype, void* ws, size_t ws_bytes, void* stream);

/* ---- selective-kernel unit (reference src/sknet.py:172-240 == src/sk_resnet.py:171-239) -----------------
 * P (<= 4) NHWC feature maps u[i] of one shape [N][HW][C] (the grouped, dilated conv paths, sknet.py:186-198).
 * `u`, `logits`, `dlogits`, `du` are HOST arrays of P device pointers.  C must be a multiple of 8 (bf16) / 4 (f32).
 *   cnk_sk_pool_fwd    pooled[n][c] = mean_hw sum_i u_i        -- torch.sum(feature_maps, 1) + global_avgpool, :222-228
 *   cnk_sk_select_fwd  attn[i][n][c] = softmax_i(logits_i[n][c]); out = sum_i u_i * attn_i   -- :231-237
 *   cnk_sk_select_bwd_logits  dlogits_i from dout (reduction over HW + the softmax Jacobian)
 *   cnk_sk_maps_bwd    du_i = dout * attn_i + dpooled / HW  (the gradient of BOTH reads of u_i in one pass;
 *                      dout/attn or dpooled may be NULL, not both)
 * ws: cnk_sk_workspace_bytes(N, HW, C, P) bytes (fixed-order partial sums; deterministic). */
size_t cnk_sk_workspace_bytes(int32_t N, int32_t HW, int32_t C, int32_t P);
int cnk_sk_pool_fwd(const void* const* u, int32_t P, float* pooled, int32_t N, int32_t HW, int32_t C, int dtype,
                    void* ws, size_t ws_bytes, void* stream);
int cnk_sk_select_fwd(const void* const* u, const float* const* logits, int32_t P, float* attn, void* out, int32_t N,
                      int32_t HW, int32_t C, int dtype, void* stream);
int cnk_sk_select_bwd_logits(const void* const* u, const float* attn, const void* dout, float* const* dlogits, int32_t P,
                             int32_t N, int32_t HW, int32_t C, int dtype, void* ws, size_t ws_bytes, void* stream);
int cnk_sk_maps_bwd(const void* dout, const float* attn, const float* dpooled, void* const* du, int32_t P, int32_t N,
                    int32_t HW, int32_t C, int dtype, void* stream);

/* Gradient clipping over the same flat fp32 gradient arena (gradient_clipping, basemodel.py:131-178).
 * cnk_grad_clip_norm == torch.nn.utils.clip_grad_norm_(params, max_norm) (L2, basemodel.py:156): the
 * norm is taken over all gradients together and every gradient is scaled by
 * min(1, max_norm / (norm + 1e-6)); *total_norm_out (device float, may be NULL) receives the norm.
 * Arena padding must be zero.  workspace: cnk_grad_clip_workspace_bytes().
 * cnk_grad_clip_value == clip_grad_value_(params, clip_value) (basemodel.py:171): clamp to [-c, +c]. */
size_t cnk_grad_clip_workspace_bytes(void);
int cnk_grad_clip_norm(float* grads, int64_t n, float max_norm, float* total_norm_out, void* workspace,
                       size_t workspace_bytes, void* stream);
int cnk_grad_clip_value(float* grads, int64_t n, float clip_value, void* stream);

/* Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline
 * leg).  cnk_prof_begin(tag_mask) arms recording for kernels whose class bit is set;
 * cnk_prof_collect synchronises the recorded events (call after the timed region) and returns
 * count / total milliseconds / total algorithmic flops for `cls`. */
enum { CNK_CLS_CONV3X3_FWD = 1, CNK_CLS_CONV3X3_DGRAD = 2, CNK_CLS_CONV3X3_WGRAD = 4,
       CNK_CLS_CONV1X1_FWD = 8, CNK_CLS_CONV1X1_DGRAD = 16, CNK_CLS_CONV1X1_WGRAD = 32,
       CNK_CLS_BN_FWD = 64, CNK_CLS_BN_BWD = 128, CNK_CLS_OTHER = 256 };
int cnk_prof_begin(int32_t class_mask);
int cnk_prof_collect(int32_t cls, int64_t* count, double* total_ms, double* total_flops,
                     double* total_bytes);
int cnk_prof_end(void);

#ifdef __cplusplus
}
#endif
#endif /* CNK_H */
