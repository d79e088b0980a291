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
 * Accumulation order per output element is fixed and documented at each function; no -ffast-math.
 * The loops are blocked for the cache (an output ROW of CB channels stays in L1 while every
 * (ci, kh, kw) contribution is added to it; a weight-gradient pair streams its two planes once for
 * all K*K taps) — blocking changes which elements are worked on together, never the order in which
 * one element's terms are added, so results are bit-identical to the naive loop nest.
 */
#include <stddef.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#define IDX4(a, b, c, d, B, C, D) ((((size_t)(a) * (B) + (b)) * (C) + (c)) * (D) + (d))

#define CB 4 /* output (fwd) / input (dgrad) channels that share one streamed row */

#define OWB 16 /* output pixels of a row kept in registers per channel (2 AVX2 vectors x CB channels = 8 accumulators) */

/* forward: xp[N,Ci,Hp,Wp] (pre-padded), w[Co,Ci,K,K], b[Co] or NULL -> y[N,Co,Ho,Wo]
 * y[n,co,oh,ow] = bias, then += terms in (ci, kh, kw) lexicographic order.  Stride 1: a strip of OWB output pixels x CB
 * channels lives in registers across the whole (ci, kh, kw) loop (same per-element order, fewer loads/stores). */
void conv_fwd(const REAL *xp, const REAL *w, const REAL *b, REAL *y,
              int N, int Ci, int Hp, int Wp, int Co, int K, int s, int Ho, int Wo)
{
    const int ncb = (Co + CB - 1) / CB;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int cb = 0; cb < ncb; ++cb)
            for (int oh = 0; oh < Ho; ++oh) {
                const int co0 = cb * CB, nco = (Co - co0 < CB) ? Co - co0 : CB;
                REAL *yr[CB];
                for (int c = 0; c < nco; ++c) yr[c] = y + IDX4(n, co0 + c, oh, 0, Co, Ho, Wo);
                int ow0 = 0;
                if (s == 1 && nco == CB) {
                    const size_t wstep = (size_t)Ci * K * K;     /* weights of consecutive output channels */
                    for (; ow0 + OWB <= Wo; ow0 += OWB) {
                        REAL acc[CB][OWB];
                        for (int c = 0; c < CB; ++c) {
                            const REAL bias = b ? b[co0 + c] : (REAL)0;
                            for (int v = 0; v < OWB; ++v) acc[c][v] = bias;
                        }
                        for (int ci = 0; ci < Ci; ++ci)
                            for (int kh = 0; kh < K; ++kh) {
                                const REAL *restrict xrow = xp + IDX4(n, ci, oh + kh, ow0, Ci, Hp, Wp);
                                const REAL *restrict wr = w + IDX4(co0, ci, kh, 0, Ci, K, K);
                                for (int kw = 0; kw < K; ++kw) {
                                    const REAL w0 = wr[kw], w1 = wr[wstep + kw], w2 = wr[2 * wstep + kw], w3 = wr[3 * wstep + kw];
                                    for (int v = 0; v < OWB; ++v) {
                                        const REAL xv = xrow[kw + v];
                                        acc[0][v] += w0 * xv;
                                        acc[1][v] += w1 * xv;
                                        acc[2][v] += w2 * xv;
                                        acc[3][v] += w3 * xv;
                                    }
                                }
                            }
                        for (int c = 0; c < CB; ++c)
                            for (int v = 0; v < OWB; ++v) yr[c][ow0 + v] = acc[c][v];
                    }
                }
                if (ow0 == Wo) continue;
                /* remainder of the row (and every stride-2 / ragged-channel case): the row itself is the accumulator */
                for (int c = 0; c < nco; ++c) {
                    const REAL bias = b ? b[co0 + c] : (REAL)0;
                    for (int ow = ow0; ow < Wo; ++ow) yr[c][ow] = bias;
                }
                for (int ci = 0; ci < Ci; ++ci)
                    for (int kh = 0; kh < K; ++kh) {
                        const REAL *xrow = xp + IDX4(n, ci, oh * s + kh, 0, Ci, Hp, Wp);
                        for (int kw = 0; kw < K; ++kw) {
                            const REAL *restrict xr = xrow + kw;
                            for (int c = 0; c < nco; ++c) {
                                const REAL wv = w[IDX4(co0 + c, ci, kh, kw, Ci, K, K)];
                                REAL *restrict yc = yr[c];
                                if (s == 1)
                                    for (int ow = ow0; ow < Wo; ++ow) yc[ow] += wv * xr[ow];
                                else
                                    for (int ow = ow0; ow < Wo; ++ow) yc[ow] += wv * xr[ow * s];
                            }
                        }
                    }
            }
}

/* data gradient: dy[N,Co,Ho,Wo], w[Co,Ci,K,K] -> dxp[N,Ci,Hp,Wp] (w.r.t. the padded input)
 * dxp[n,ci,ih,iw] = 0, then += terms in (co, kh, kw) lexicographic order (oh = (ih - kh) / s where that is an
 * output row). */
void conv_dgrad(const REAL *dy, const REAL *w, REAL *dxp,
                int N, int Ci, int Hp, int Wp, int Co, int K, int s, int Ho, int Wo)
{
    const int ncb = (Ci + CB - 1) / CB;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int cb = 0; cb < ncb; ++cb)
            for (int ih = 0; ih < Hp; ++ih) {
                const int ci0 = cb * CB, nci = (Ci - ci0 < CB) ? Ci - ci0 : CB;
                REAL *dxr[CB];
                for (int c = 0; c < nci; ++c) {
                    dxr[c] = dxp + IDX4(n, ci0 + c, ih, 0, Ci, Hp, Wp);
                    memset(dxr[c], 0, sizeof(REAL) * (size_t)Wp);
                }
                for (int co = 0; co < Co; ++co)
                    for (int kh = 0; kh < K; ++kh) {
                        const int t = ih - kh;
                        if (t < 0 || t % s) continue;
                        const int oh = t / s;
                        if (oh >= Ho) continue;
                        const REAL *restrict dyr = dy + IDX4(n, co, oh, 0, Co, Ho, Wo);
                        for (int kw = 0; kw < K; ++kw)
                            for (int c = 0; c < nci; ++c) {
                                const REAL wv = w[IDX4(co, ci0 + c, kh, kw, Ci, K, K)];
                                REAL *restrict dxc = dxr[c] + kw;
                                if (s == 1)
                                    for (int ow = 0; ow < Wo; ++ow) dxc[ow] += wv * dyr[ow];
                                else
                                    for (int ow = 0; ow < Wo; ++ow) dxc[ow * s] += wv * dyr[ow];
                            }
                    }
            }
}

#define KMAX 7 /* largest kernel on the path (networks.py:159, 187) */

/* weight gradient: xp[N,Ci,Hp,Wp], dy[N,Co,Ho,Wo] -> dw[Co,Ci,K,K]  (overwrites dw)
 * dw[co,ci,kh,kw] = sum over (n, oh) in lexicographic order of the row dot products, accumulated in double. */
void conv_wgrad(const REAL *xp, const REAL *dy, REAL *dw,
                int N, int Ci, int Hp, int Wp, int Co, int K, int s, int Ho, int Wo)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci) {
            double acc[KMAX][KMAX]; /* long reduction over N*Ho*Wo: keep it in double in both builds */
            for (int kh = 0; kh < K; ++kh)
                for (int kw = 0; kw < K; ++kw) acc[kh][kw] = 0.0;
            for (int n = 0; n < N; ++n)
                for (int oh = 0; oh < Ho; ++oh) {
                    const REAL *dyr = dy + IDX4(n, co, oh, 0, Co, Ho, Wo);
                    for (int kh = 0; kh < K; ++kh) {
                        const REAL *xrow = xp + IDX4(n, ci, oh * s + kh, 0, Ci, Hp, Wp);
                        for (int kw = 0; kw < K; ++kw) {
                            const REAL *xr = xrow + kw;
                            REAL row = (REAL)0;
                            if (s == 1) {
#pragma omp simd reduction(+ : row)
                                for (int ow = 0; ow < Wo; ++ow) row += xr[ow] * dyr[ow];
                            } else {
#pragma omp simd reduction(+ : row)
                                for (int ow = 0; ow < Wo; ++ow) row += xr[ow * s] * dyr[ow];
                            }
                            acc[kh][kw] += (double)row;
                        }
                    }
                }
            for (int kh = 0; kh < K; ++kh)
                for (int kw = 0; kw < K; ++kw) dw[IDX4(co, ci, kh, kw, Ci, K, K)] = (REAL)acc[kh][kw];
        }
}

int conv_ref_real_bytes(void) { return (int)sizeof(REAL); }
