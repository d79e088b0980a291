/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * Plain-C CPU restatement of the convolution arithmetic that the reference
 * delegates to torch.nn.Conv2d / torch.nn.ConvTranspose2d (third-party
 * dependency `torch`, version unpinned by the reference).
 * Call sites restated: /root/reference/augmented_cyclegan/networks.py:158-189,
 * 210-244 (generators), 321-338 (Discriminator), 365-382 (Discriminator_edges),
 * 445-471 (LatentEncoder); modules.py:111-118,162,180,211,227.
 *
 * Published algorithm restated (cross-correlation, NCHW, weight OIHW):
 *   y[n,co,oh,ow] = b[co] + sum_{ci,kh,kw} xp[n,ci,oh*s+kh,ow*s+kw] * w[co,ci,kh,kw]
 * where xp is the input ALREADY padded by the caller (zero or reflection);
 * padding and its adjoint live in oracle/ops.py so the C stays trivially
 * checkable.  data-gradient and weight-gradient are the exact adjoints.
 *
 * Compiled twice: -DREAL=float (timed as the CPU baseline "port") and
 * -DREAL=double (used to bound fp32 rounding when judging the 1e-3 bar).
 * Accumulation order is a fixed row-saxpy; no -ffast-math.
 */
#include <stddef.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#define IDX4(a, b, c, d, B, C, D) ((((size_t)(a) * (B) + (b)) * (C) + (c)) * (D) + (d))

/* forward: xp[N,Ci,Hp,Wp] (pre-padded), w[Co,Ci,K,K], b[Co] or NULL -> y[N,Co,Ho,Wo] */
void conv_fwd(const REAL *xp, const REAL *w, const REAL *b, REAL *y,
              int N, int Ci, int Hp, int Wp, int Co, int K, int s, int Ho, int Wo)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co) {
            REAL *yp = y + IDX4(n, co, 0, 0, Co, Ho, Wo);
            const REAL bias = b ? b[co] : (REAL)0;
            for (int i = 0; i < Ho * Wo; ++i) yp[i] = bias;
            for (int ci = 0; ci < Ci; ++ci)
                for (int kh = 0; kh < K; ++kh)
                    for (int kw = 0; kw < K; ++kw) {
                        const REAL wv = w[IDX4(co, ci, kh, kw, Ci, K, K)];
                        for (int oh = 0; oh < Ho; ++oh) {
                            const REAL *xr = xp + IDX4(n, ci, oh * s + kh, kw, Ci, Hp, Wp);
                            REAL *yr = yp + (size_t)oh * Wo;
                            if (s == 1)
                                for (int ow = 0; ow < Wo; ++ow) yr[ow] += wv * xr[ow];
                            else
                                for (int ow = 0; ow < Wo; ++ow) yr[ow] += wv * xr[ow * s];
                        }
                    }
        }
}

/* data gradient: dy[N,Co,Ho,Wo], w[Co,Ci,K,K] -> dxp[N,Ci,Hp,Wp] (w.r.t. the padded input) */
void conv_dgrad(const REAL *dy, const REAL *w, REAL *dxp,
                int N, int Ci, int Hp, int Wp, int Co, int K, int s, int Ho, int Wo)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Ci; ++ci) {
            REAL *dxc = dxp + IDX4(n, ci, 0, 0, Ci, Hp, Wp);
            memset(dxc, 0, sizeof(REAL) * (size_t)Hp * Wp);
            for (int co = 0; co < Co; ++co)
                for (int kh = 0; kh < K; ++kh)
                    for (int kw = 0; kw < K; ++kw) {
                        const REAL wv = w[IDX4(co, ci, kh, kw, Ci, K, K)];
                        for (int oh = 0; oh < Ho; ++oh) {
                            const REAL *dyr = dy + IDX4(n, co, oh, 0, Co, Ho, Wo);
                            REAL *dxr = dxc + (size_t)(oh * s + kh) * Wp + kw;
                            if (s == 1)
                                for (int ow = 0; ow < Wo; ++ow) dxr[ow] += wv * dyr[ow];
                            else
                                for (int ow = 0; ow < Wo; ++ow) dxr[ow * s] += wv * dyr[ow];
                        }
                    }
        }
}

/* weight gradient: xp[N,Ci,Hp,Wp], dy[N,Co,Ho,Wo] -> dw[Co,Ci,K,K]  (overwrites dw) */
void conv_wgrad(const REAL *xp, const REAL *dy, REAL *dw,
                int N, int Ci, int Hp, int Wp, int Co, int K, int s, int Ho, int Wo)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int kh = 0; kh < K; ++kh)
                for (int kw = 0; kw < K; ++kw) {
                    double acc = 0.0; /* long reduction over N*Ho*Wo: keep it in double in both builds */
                    for (int n = 0; n < N; ++n)
                        for (int oh = 0; oh < Ho; ++oh) {
                            const REAL *xr = xp + IDX4(n, ci, oh * s + kh, kw, Ci, Hp, Wp);
                            const REAL *dyr = dy + IDX4(n, co, oh, 0, Co, Ho, Wo);
                            REAL row = (REAL)0;
                            if (s == 1) {
#pragma omp simd reduction(+ : row)
                                for (int ow = 0; ow < Wo; ++ow) row += xr[ow] * dyr[ow];
                            } else {
#pragma omp simd reduction(+ : row)
                                for (int ow = 0; ow < Wo; ++ow) row += xr[ow * s] * dyr[ow];
                            }
                            acc += (double)row;
                        }
                    dw[IDX4(co, ci, kh, kw, Ci, K, K)] = (REAL)acc;
                }
}

int conv_ref_real_bytes(void) { return (int)sizeof(REAL); }
