/* ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement, in the reference's own NCHW fp32 layout, of the operator semantics the
 * reference's hot path exercises through torch.nn.  The reference has no
 * arithmetic of its own: every function below restates the published semantics of the torch
 * operator named at the reference call site it cites (torch pinned at 1.6.0+cu101,
 * /root/reference/requirements.txt:18).  Accumulation is in double so that the oracle's own
 * error is below the fp32 paths it checks.  Pinned by tests/test_oracle_golden.py against
 * tests/golden/ops_*.npz, which tests/golden/make_golden.py produced from layers built by the
 * reference's own factory (basemodel.py:966-1042) run on torch-CPU.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define IDX4(n, c, h, w, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (h)) * (W) + (w))

/* nn.Conv2d forward, cross-correlation, zero padding, dilation 1, groups 1
 * (reference: basemodel.py:974; bias iff not batch_norm, :973). */
void orc_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Ci,
                    int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw) {
    int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    double acc = bias ? bias[co] : 0.0;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int r = 0; r < kh; ++r) {
                            int h = ho * sh - ph + r;
                            if (h < 0 || h >= H) continue;
                            for (int s = 0; s < kw; ++s) {
                                int ww = wo * sw - pw + s;
                                if (ww < 0 || ww >= W) continue;
                                acc += (double)x[IDX4(n, ci, h, ww, Ci, H, W)] *
                                       (double)w[IDX4(co, ci, r, s, Ci, kh, kw)];
                            }
                        }
                    y[IDX4(n, co, ho, wo, Co, Ho, Wo)] = (float)acc;
                }
}

/* autograd of the above w.r.t. the input (loss.backward(), basemodel.py:278). */
void orc_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int Ci, int H, int W,
                      int Co, int kh, int kw, int sh, int sw, int ph, int pw) {
    int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Ci; ++ci)
            for (int h = 0; h < H; ++h)
                for (int ww = 0; ww < W; ++ww) {
                    double acc = 0.0;
                    for (int r = 0; r < kh; ++r) {
                        int t = h + ph - r;
                        if (t < 0 || t % sh) continue;
                        int ho = t / sh;
                        if (ho >= Ho) continue;
                        for (int s = 0; s < kw; ++s) {
                            int u = ww + pw - s;
                            if (u < 0 || u % sw) continue;
                            int wo = u / sw;
                            if (wo >= Wo) continue;
                            for (int co = 0; co < Co; ++co)
                                acc += (double)dy[IDX4(n, co, ho, wo, Co, Ho, Wo)] *
                                       (double)w[IDX4(co, ci, r, s, Ci, kh, kw)];
                        }
                    }
                    dx[IDX4(n, ci, h, ww, Ci, H, W)] = (float)acc;
                }
}

/* autograd w.r.t. the weight and (optional) bias. */
void orc_conv2d_wgrad(const float* dy, const float* x, float* dw, float* dbias, int N, int Ci,
                      int H, int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw) {
    int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    for (int co = 0; co < Co; ++co) {
        for (int ci = 0; ci < Ci; ++ci)
            for (int r = 0; r < kh; ++r)
                for (int s = 0; s < kw; ++s) {
                    double acc = 0.0;
                    for (int n = 0; n < N; ++n)
                        for (int ho = 0; ho < Ho; ++ho) {
                            int h = ho * sh - ph + r;
                            if (h < 0 || h >= H) continue;
                            for (int wo = 0; wo < Wo; ++wo) {
                                int ww = wo * sw - pw + s;
                                if (ww < 0 || ww >= W) continue;
                                acc += (double)dy[IDX4(n, co, ho, wo, Co, Ho, Wo)] *
                                       (double)x[IDX4(n, ci, h, ww, Ci, H, W)];
                            }
                        }
                    dw[IDX4(co, ci, r, s, Ci, kh, kw)] = (float)acc;
                }
        if (dbias) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n)
                for (int i = 0; i < Ho * Wo; ++i) acc += dy[((size_t)n * Co + co) * Ho * Wo + i];
            dbias[co] = (float)acc;
        }
    }
}

/* nn.BatchNorm2d training forward (basemodel.py:1008,1061): biased variance normalises, the
 * running variance takes the unbiased one, momentum 0.1, eps 1e-5. */
void orc_bn_fwd_train(const float* x, const float* gamma, const float* beta, float* y,
                      float* running_mean, float* running_var, float* save_mean,
                      float* save_invstd, int N, int C, int HW, float eps, float momentum) {
    size_t M = (size_t)N * HW;
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) s += x[((size_t)n * C + c) * HW + i];
        double mean = s / (double)M, v = 0.0;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                double d = x[((size_t)n * C + c) * HW + i] - mean;
                v += d * d;
            }
        double var = v / (double)M;
        double invstd = 1.0 / sqrt(var + (double)eps);
        save_mean[c] = (float)mean;
        save_invstd[c] = (float)invstd;
        if (running_mean) {
            double unb = M > 1 ? v / (double)(M - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                size_t k = ((size_t)n * C + c) * HW + i;
                y[k] = (float)((x[k] - mean) * invstd * gamma[c] + beta[c]);
            }
    }
}

/* eval-mode forward (valid_model, basemodel.py:311): running statistics. */
void orc_bn_fwd_eval(const float* x, const float* gamma, const float* beta, const float* rmean,
                     const float* rvar, float* y, int N, int C, int HW, float eps) {
    for (int c = 0; c < C; ++c) {
        double invstd = 1.0 / sqrt((double)rvar[c] + (double)eps);
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                size_t k = ((size_t)n * C + c) * HW + i;
                y[k] = (float)((x[k] - (double)rmean[c]) * invstd * gamma[c] + beta[c]);
            }
    }
}

/* training backward: dbeta = sum dy, dgamma = sum dy*xhat,
 * dx = gamma*invstd/M * (M*dy - dbeta - xhat*dgamma). */
void orc_bn_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean,
                const float* save_invstd, float* dx, float* dgamma, float* dbeta, int N, int C,
                int HW) {
    double M = (double)N * HW;
    for (int c = 0; c < C; ++c) {
        double mean = save_mean[c], invstd = save_invstd[c], sb = 0.0, sg = 0.0;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                size_t k = ((size_t)n * C + c) * HW + i;
                sb += dy[k];
                sg += dy[k] * (x[k] - mean) * invstd;
            }
        dbeta[c] = (float)sb;
        dgamma[c] = (float)sg;
        double a = gamma[c] * invstd / M;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < HW; ++i) {
                size_t k = ((size_t)n * C + c) * HW + i;
                double xh = (x[k] - mean) * invstd;
                dx[k] = (float)(a * (M * dy[k] - sb - xh * sg));
            }
    }
}

/* nn.ReLU (basemodel.py:1012) / F.relu (resnet.py:154,194): mask taken from the output. */
void orc_relu_fwd(const float* x, float* y, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : 0.f;
}
void orc_relu_bwd(const float* dy, const float* y, float* dx, size_t n) {
    for (size_t i = 0; i < n; ++i) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
/* residual add (resnet.py:151,191) followed by F.relu. */
void orc_add_relu_fwd(const float* a, const float* b, float* y, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        float s = a[i] + b[i];
        y[i] = s > 0.f ? s : 0.f;
    }
}

/* nn.MaxPool2d (basemodel.py:1021): floor mode, -inf padding, dilation 1; the backward sends the
 * gradient to the first maximum in row-major window order (idx = h*W + w, like torch). */
void orc_maxpool_fwd(const float* x, float* y, int64_t* idx, int N, int C, int H, int W, int kh,
                     int kw, int sh, int sw, int ph, int pw) {
    int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    float best = -INFINITY;
                    int64_t bi = -1;
                    for (int r = 0; r < kh; ++r) {
                        int h = ho * sh - ph + r;
                        if (h < 0 || h >= H) continue;
                        for (int s = 0; s < kw; ++s) {
                            int ww = wo * sw - pw + s;
                            if (ww < 0 || ww >= W) continue;
                            float v = x[IDX4(n, c, h, ww, C, H, W)];
                            if (bi < 0 || v > best || v != v) {
                                best = v;
                                bi = (int64_t)h * W + ww;
                            }
                        }
                    }
                    y[IDX4(n, c, ho, wo, C, Ho, Wo)] = best;
                    if (idx) idx[IDX4(n, c, ho, wo, C, Ho, Wo)] = bi;
                }
}
void orc_maxpool_bwd(const float* dy, const int64_t* idx, float* dx, int N, int C, int H, int W,
                     int Ho, int Wo) {
    memset(dx, 0, sizeof(float) * (size_t)N * C * H * W);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < Ho * Wo; ++i) {
                size_t o = ((size_t)n * C + c) * Ho * Wo + i;
                dx[((size_t)n * C + c) * H * W + idx[o]] += dy[o];
            }
}

/* nn.AdaptiveAvgPool2d(1) (basemodel.py:1039). */
void orc_gap_fwd(const float* x, float* y, int N, int C, int HW) {
    for (int i = 0; i < N * C; ++i) {
        double s = 0.0;
        for (int k = 0; k < HW; ++k) s += x[(size_t)i * HW + k];
        y[i] = (float)(s / HW);
    }
}
void orc_gap_bwd(const float* dy, float* dx, int N, int C, int HW) {
    for (int i = 0; i < N * C; ++i)
        for (int k = 0; k < HW; ++k) dx[(size_t)i * HW + k] = dy[i] / (float)HW;
}
