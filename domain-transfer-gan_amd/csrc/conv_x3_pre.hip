// Wave-specialised implicit-GEMM convolution (bf16x3 arithmetic, 128x128 tiles) on PRE-SPLIT activations.
//
// conv_x3.hip gathers fp32 activations, splits every value into bf16 hi + lo in registers (20 VALU per 8 values) and
// stores the halves to LDS with ds_write_b128 — all of it producer-wave work beside the MFMAs.  Here the gathered tensor
// is stored pre-split ("S16": per pixel and 8-channel group, 16 bytes of hi followed by 16 bytes of lo — the same 4 bytes
// per element as fp32, written once by the kernel that produces the activation), so BOTH operands travel global -> LDS
// by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR staging, no split, no LDS stores; the four producer waves only
// issue DMA pieces and wait for them.
//
// A image ("row patch", as in conv_x3.hip: the K taps of a kernel row read the same input rows shifted by one pixel, so
// the patch — the tile's row segments with their K-1 halo pixels — is fetched once per kernel row and 32-channel chunk).
// A DMA piece is 64 lanes x 16 B of CONTIGUOUS LDS, so padding has to be bought with lanes: an image row is 80 bytes =
// the four 16-byte chunks {hi, lo} x {k-group q, k-group q+2} of one pixel plus one pad chunk (a masked lane), and the
// k-groups of different parity live in two images a multiple of 256 B apart.  A ds_read_b128 lane group (8 lanes of
// k-group 2j on rows r..r+3, r+12..r+15 and 8 lanes of k-group 2j+1 on rows r+4..r+11) then reads the SAME in-row offset
// of 16 consecutive rows of the two images: 16 distinct 16-byte slots of the 256-byte bank row for every tap shift
// (5 slots per row, 5 odd).  The source side stays line-friendly: the four data lanes of a row read 64 of the 128
// contiguous bytes a pixel's 32-channel chunk occupies, the other image's piece the other 64.
#include "common.h"
#include "conv_internal.h"
#include <cstdlib>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BM = 128, BN = 128, KC = 32, NK8 = KC / 8;
constexpr int BPL = BN * 8;                 // B plane stride (bf16 elements), planes XOR-permuted as in conv_x3.hip
constexpr int B_IMG = NK8 * BPL;            // one hi (or lo) B image, elements
__device__ __forceinline__ int lds_at(int plane, int row, int pl) { return plane * pl + ((row ^ (2 * plane)) * 8); }
constexpr int PROW = 80;                    // bytes per A image row
constexpr int PPIECES = 11;                 // 1 KB DMA pieces per parity image: 140 rows
constexpr int PIMG = PPIECES * 1024;        // bytes per parity image (a multiple of 256)
constexpr int ABUF = 2 * PIMG;              // one A buffer: parity images 0 and 1
constexpr int A_BYTES = 2 * ABUF;           // two A buffers (kernel rows alternate)
constexpr int BBUF = 2 * B_IMG * 2;         // one B buffer: hi image, lo image (bytes)
constexpr int LDS_BYTES = A_BYTES + 2 * BBUF;
constexpr int NPW = 6;                      // A pieces per producer wave and kernel row: 22 over 4 waves = 6, 6, 5, 5
constexpr unsigned NO_PIX = 0xFFFFFFFFu;
}

#ifdef ACG_STAMP
// diagnostic build only: per consumer wave (barrier wait, rest) cycles of the main loop, [workgroup][wave][2]
__device__ unsigned long long g_pre_stamps[8192 * 8 * 3];   // [workgroup][wave][3]
__device__ unsigned long long g_pre_tile[8192 * 4];        // [workgroup]: cycles of set-up, main loop, epilogue (wave 0); start (100 MHz clock)
extern "C" int acg_debug_pre_tile(unsigned long long *host, size_t n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pre_tile), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int acg_debug_pre_stamps(unsigned long long *host, size_t n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pre_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// SUMS: the launch also emits the backward sums of the norm in front of it (Geom.ns_part) — a separate instantiation so that
// kernel traces tell the two apart and the plain epilogue carries none of it
template <bool REFLECT, bool STATS, bool SUMS = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void
igemm_conv_x3_pre(const char *__restrict__ in, const __bf16 *__restrict__ wp, const float *__restrict__ bias,
                  float *__restrict__ out, Geom g, Taps taps, unsigned in_bytes, unsigned w_bytes, unsigned w_lo_bytes,
                  float *__restrict__ stats, int kdim, int dxmin, int kstep)
{
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    // output row of every tile pixel in 16-byte units; bit 31: the row lies in Geom.out2 (the un-padded tensor of a reflect
    // data gradient: side inputs apply there), NO_PIX past the end
    __shared__ unsigned pix_off[BM];
    __shared__ float red[2][2][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, bid = blockIdx.x;
#ifdef ACG_STAMP
    const unsigned long long tl_entry = __builtin_amdgcn_s_memtime(), tl_real = __builtin_amdgcn_s_memrealtime();
    unsigned long long tl_loop0 = 0, tl_loop1 = 0;
#endif
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tiles_n = g.ncols_pad / BN;
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int n0 = tile_n * BN;
    // 32-bit pixel arithmetic: the launcher bounds the gathered tensor to 4 GiB, i.e. Mtot < 2^25 (64-bit divisions are
    // ~150 instructions each on this chip, and nine of them per producer thread stood in front of the first DMA of a tile)
    const int m0 = tile_m * BM;
    const int Mt = (int)g.Mtot;
    const int GHW = g.GH * g.GW;
    const int S = taps.n * (g.Cin / KC);

    if (tid < BM) {
        const int m = m0 + tid;
        unsigned po = NO_PIX;
        if (m < Mt) {
            const int n = m / GHW;
            const int r = m - n * GHW;
            const int gy = r / g.GW, gx = r - gy * g.GW;
            po = (unsigned)(((((long long)n * g.Hout + gy) * g.Wout + gx) * g.Cout) >> 2);
            if (g.fold_p > 0) { // reflect data gradient: pixels nothing is mirrored onto bypass the fold
                const int p = g.fold_p, iy = gy - p, ix = gx - p;
                const bool cy = iy >= 0 && iy < g.fold_H && !(iy >= 1 && iy <= p) && !(iy >= g.fold_H - 1 - p && iy <= g.fold_H - 2);
                const bool cx = ix >= 0 && ix < g.fold_W && !(ix >= 1 && ix <= p) && !(ix >= g.fold_W - 1 - p && ix <= g.fold_W - 2);
                if (cy && cx) po = (unsigned)(((((long long)n * g.fold_H + iy) * g.fold_W + ix) * g.Cout) >> 2) | 0x80000000u;
            }
            if (g.unpad) po |= 0x80000000u; // un-padded reflect data gradient: every pixel is final here (out2 == out)
        }
        pix_off[tid] = po;
    }

    typedef __attribute__((address_space(3))) void lds_void;
    if (wave >= 4) {
        // ------------------------------------------------------------------ producers: DMA issue only
        const int pt = tid - 256, pw = wave - 4;
        constexpr int BCH = NK8 * BN, BL = BCH / 256;               // 16-byte chunks of the B tile: 2 per thread
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, w_bytes, 0x00020000);
        unsigned b_voff[BL];
        int b_lds[BL];
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = pt + 256 * i;
            const int plane = idx / BN, slot = idx - plane * BN, col = slot ^ (2 * plane);
            b_voff[i] = (unsigned)((((plane >> 1) * g.ncols_pad + n0 + col) * 16 + (plane & 1) * 8) * 2);
            b_lds[i] = __builtin_amdgcn_readfirstlane((plane * BPL + (slot & ~63) * 8) * 2);   // byte offset of the wave's piece
        }
        const int tap_v = taps.pk[lane < taps.n ? lane : 0];
        auto tap_pk = [&](int t) { return __builtin_amdgcn_readlane(tap_v, t); };
        int bt = 0, bc0 = 0; // tap and first input channel of the next B stage
        // Geom.unpad: the tile of grid row 1 / H-2 reads the mirrored dy row through kernel row 2 / 0, whose slabs 6..8 /
        // 0..2 become the summed slabs 9..11
        int sub_lo = -100, sub_add = 0;
        if (g.unpad) {
            const int gy_t = (m0 / g.GW) % g.GH;
            if (gy_t == 1) { sub_lo = 6; sub_add = 3; }
            else if (gy_t == g.GH - 2) { sub_lo = 0; sub_add = 9; }
        }
        auto dma_b = [&](int buf) { // 2 * BL pieces per thread: hi and lo image of stage (bt, bc0) into B buffer `buf`
            char *Bb = lds + A_BYTES + buf * BBUF;
            int slab = tap_pk(bt) >> 16;
            slab += (unsigned)(slab - sub_lo) < 3u ? sub_add : 0;
            const unsigned soff = (unsigned)(((slab * (g.Cin >> 4) + (bc0 >> 4)) * g.ncols_pad) * 16) * 2u;
#pragma unroll
            for (int i = 0; i < BL; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Bb + b_lds[i]), 16, b_voff[i], soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Bb + B_IMG * 2 + b_lds[i]), 16, b_voff[i], soff + w_lo_bytes, 0, 0);
            }
            if (++bt == taps.n) { bt = 0; bc0 += KC; }
        };
        static_assert(BL == 2, "the vmcnt immediates below count 2 * BL = 4 B pieces per stage");
        dma_b(0);   // the first weight tile is on its way while the patch geometry below is worked out

        // The tile's 128 consecutive output pixels form SEGMENTS, one per grid row it touches (the first starts at column
        // x0, the others at 0); segment s occupies image rows [row0(s), row0(s) + len(s) + K-1).  Piece e = pw + 4 j of a
        // kernel row fills 1 KB of parity image e / 11; its lane L holds chunk L % 5 of image row L / 5.
        const int grow0 = m0 / g.GW;                           // global grid row (image * GH + gy) of the first pixel
        const int x0 = m0 - grow0 * g.GW;
        const int first = g.GW - x0 < BM ? g.GW - x0 : BM;     // pixels in segment 0
        const int RW = g.GW + kdim - 1;
        const int nseg = first >= BM ? 1 : 1 + (BM - first + g.GW - 1) / g.GW;
        const int nrows = BM + nseg * (kdim - 1);
        const int grows = Mt / g.GW;                           // grid rows in the whole tensor
        int pa_gy[NPW], pa_nb[NPW], pa_lds[NPW];
        unsigned pa_col[NPW];
        bool pa_ok[NPW];
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int e = pw + 4 * j;
            const int q = e >= PPIECES ? 1 : 0, pc = e - q * PPIECES;
            const int L = pc * 64 + lane;
            const int row = L / 5, ch = L - row * 5;
            int seg = 0, px = row;
            if (row >= first + kdim - 1) {
                const int qq = row - (first + kdim - 1);
                seg = 1 + qq / RW;
                px = qq - (seg - 1) * RW;
            }
            const int grow = grow0 + seg;
            const int n_img = grow / g.GH;
            int ix = (seg == 0 ? x0 : 0) + px + dxmin;
            bool ok = e < 2 * PPIECES && ch < 4 && row < nrows && grow < grows;
            if (REFLECT) {
                ix = ix < 0 ? -ix : ix;
                ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
            } else {
                ok = ok && (unsigned)ix < (unsigned)g.Win;
            }
            pa_gy[j] = grow - n_img * g.GH;
            pa_nb[j] = n_img * g.Hin;
            pa_col[j] = (unsigned)(ix * g.Cin * 4 + (q + 2 * (ch >> 1)) * 32 + (ch & 1) * 16);
            pa_ok[j] = ok;
            pa_lds[j] = __builtin_amdgcn_readfirstlane(q * PIMG + pc * 1024);
        }
        int ar_t = 0, ar_c0 = 0; // first tap and first input channel of the next kernel row to fetch
        auto dma_a = [&](int buf) { // the patch of one kernel row (all of its taps share dy), from dx = dxmin
            const int ty = (tap_pk(ar_t) << 24) >> 24;
            char *Ab = lds + buf * ABUF;
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                if (pw + 4 * j < 2 * PPIECES) { // wave-uniform: waves 2 and 3 have five pieces
                    int iy = pa_gy[j] + ty;
                    bool ok = pa_ok[j];
                    if (REFLECT) {
                        iy = iy < 0 ? -iy : iy;
                        iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                    } else {
                        ok = ok && (unsigned)iy < (unsigned)g.Hin;
                    }
                    const unsigned off = (unsigned)((pa_nb[j] + iy) * g.Win) * (unsigned)(g.Cin * 4) + pa_col[j] + (unsigned)(ar_c0 * 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(Ab + pa_lds[j]), 16, acg_masked_off(off, ok), 0, 0, 0);
                }
            }
            ar_t += kdim;
            if (ar_t >= taps.n) { ar_t = 0; ar_c0 += KC; }
        };
        // Schedule: iteration s (which ends with barrier s, releasing consumer stage s) issues the B tile of stage s; the A
        // patch of kernel row r+1 is issued in the SECOND iteration of row r — its buffer was last read in the final stage
        // of row r-1, which the consumers have left when they pass the barrier before this iteration — and has to have
        // landed by the end of the first iteration of row r+1.  vmcnt counts in issue order: the second iteration waits for
        // its B pieces and leaves the (five or six) A pieces behind them in flight, every other iteration drains the queue.
        // (Measured against it and dropped, one box each: every iteration draining its queue first and issuing a third of
        // the patch afterwards, so that patch pieces never queue ahead of a B tile: +2 % time.  Timing-only ablations of
        // this kernel, resblock forward 0.372 ms: no DMA at all after the first row 0.290; weight pieces only 0.311; patch
        // pieces only 0.329; not waiting for the weight pieces 0.340; weight pieces from one cache-hot tile 0.365; patch
        // pieces reading 1 KB contiguous each 0.346 — the loop is paced by the DMA traffic itself, the patch costing more
        // per byte than the weights, not by the latency of one stage of look-ahead.  The patch in two halves behind the B
        // pieces of the second and third iteration (4, 7, 7 pieces per iteration instead of 4, 10, 4): +-0.  In-kernel stamps
        // (-DACG_STAMP): a consumer wave's main loop is 67 k cycles per tile = 1 870 per stage, 27 % of it at the barrier;
        // the two consumer waves of a SIMD need 1 536 matrix cycles per stage, i.e. the loop itself runs the pipe at 82 %
        // and the tile's set-up and epilogue (a quarter of a tile's 90 k cycles) account for the rest of the 65 % busy.)
        const int rows = S / kdim;
        dma_a(0);
        int pk_k = 0, row = 0;
#ifdef ACG_STAMP
        unsigned long long p_issue = 0, p_land = 0, p_bar = 0, p_t = __builtin_amdgcn_s_memtime();
#define PSTAMP(acc) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - p_t; p_t = t_; }
#else
#define PSTAMP(acc)
#endif
        for (int s = 0; s < S; ++s) {
            if (s > 0) dma_b(s & 1);
            if (pk_k == 1 && row + 1 < rows) {
                dma_a((row + 1) & 1);
                PSTAMP(p_issue)
                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                PSTAMP(p_issue)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PSTAMP(p_land)
            if (++pk_k == kdim) { pk_k = 0; ++row; }
            __builtin_amdgcn_s_barrier();
            PSTAMP(p_bar)
        }
#ifdef ACG_STAMP
        if (lane == 0 && blockIdx.x < 8192) {
            g_pre_stamps[(blockIdx.x * 8 + wave) * 3] = p_issue;
            g_pre_stamps[(blockIdx.x * 8 + wave) * 3 + 1] = p_land;
            g_pre_stamps[(blockIdx.x * 8 + wave) * 3 + 2] = p_bar;
        }
#endif
#undef PSTAMP
    }   // (no return: the producer waves take half of the epilogue's rows)

    // ---------------------------------------------------------------------- consumers
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    constexpr int TM = 64, TN = 64;
    f32x4 acc[4][4];
    const int pl = lane >> 4, lr = lane & 15;
    if (wave < 4) {
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int a_base[4]; // byte offset of this lane's hi chunk in row-tile i at tap column 0 (lo: + 16)
    {
        const int x0c = m0 % g.GW, firstc = g.GW - x0c < BM ? g.GW - x0c : BM, RWc = g.GW + kdim - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tp = wm * TM + i * 16 + lr;
            int arow;
            if (tp < firstc) {
                arow = tp;
            } else {
                const int sg = 1 + (tp - firstc) / g.GW;
                arow = (firstc + kdim - 1) + (sg - 1) * RWc + (tp - firstc - (sg - 1) * g.GW);
            }
            a_base[i] = (pl & 1) * PIMG + arow * PROW + (pl >> 1) * 32;
        }
    }
    const int b_base = 2 * lds_at(pl, wn * TN + lr, BPL);
    const char *ldsb = (const char *)lds;
    auto stage = [&](const char *pa_off, const char *pb) { // one K stage: 16 fragment reads, 48 MFMAs
        bf16x8 a[4], al[4], b[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char *pa = pa_off + a_base[i];
            a[i] = *(const bf16x8 *)pa;
            al[i] = *(const bf16x8 *)(pa + 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j] = *(const bf16x8 *)(pb + j * 256);
            bl[j] = *(const bf16x8 *)(pb + 2 * B_IMG + j * 256);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#ifndef ACG_ABL_HIONLY   // (timing-only ablation: one MFMA per product; the lo fragments are not even read then)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], b[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
#endif
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
    };
    const char *pb0 = ldsb + A_BYTES + b_base, *pb1 = pb0 + BBUF;
#ifdef ACG_STAMP
    unsigned long long st_wait = 0, st_work = 0, st_t = __builtin_amdgcn_s_memtime();
    tl_loop0 = st_t;
#endif
    bool done = false;
    if (kdim == 3 && S % 6 == 0) {
        // 3 x 3 layers: six stages (two kernel rows) per trip, so buffer, tap column and B buffer of every stage are
        // compile-time constants that ride in the ds_read offset field
        auto run = [&](auto ks) {
            constexpr int KS = decltype(ks)::value;
            for (int it = 0; it < S / 6; ++it) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
#ifdef ACG_STAMP
                    { const unsigned long long t = __builtin_amdgcn_s_memtime(); st_work += t - st_t; st_t = t; }
#endif
                    __syncthreads();
#ifdef ACG_STAMP
                    { const unsigned long long t = __builtin_amdgcn_s_memtime(); st_wait += t - st_t; st_t = t; }
#endif
                    const int kx = KS > 0 ? k % 3 : 2 - k % 3;
                    stage(ldsb + (k / 3) * ABUF + kx * PROW, (k & 1) ? pb1 : pb0);
                }
            }
        };
        if (kstep > 0) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, -1>{});
        done = true;
    }
    if (!done) {
        int kk = 0, abuf = 0, kx = kstep > 0 ? 0 : kdim - 1;
        for (int s = 0; s < S; ++s) {
            __syncthreads();
            stage(ldsb + abuf * ABUF + kx * PROW, (s & 1) ? pb1 : pb0);
            kx += kstep;
            if (++kk == kdim) { kk = 0; abuf ^= 1; kx = kstep > 0 ? 0 : kdim - 1; }
        }
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef ACG_STAMP
    tl_loop1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x < 8192) {
        g_pre_stamps[(blockIdx.x * 8 + wave) * 3] = st_wait;
        g_pre_stamps[(blockIdx.x * 8 + wave) * 3 + 1] = st_work + (__builtin_amdgcn_s_memtime() - st_t);
        g_pre_stamps[(blockIdx.x * 8 + wave) * 3 + 2] = 0;
    }
#endif
    }   // consumers
    // Epilogue through LDS: the tile (accumulator + bias, activation) is staged in the LDS the main loop no longer needs
    // and leaves in coalesced rows.  ALL EIGHT waves share its rows: a workgroup's tile time is loop + epilogue (the other
    // workgroup of the CU only fills the matrix pipe meanwhile), and the epilogue is a chain of load round trips — with 512
    // threads a thread has half the rows, i.e. half the round trips.
#ifdef ACG_STAMP
#define TILE_STAMP_END()                                                                                              \
    if (tid == 0 && blockIdx.x < 8192) {                                                                              \
        const unsigned long long te_ = __builtin_amdgcn_s_memtime();                                                  \
        g_pre_tile[blockIdx.x * 4] = tl_loop0 - tl_entry; g_pre_tile[blockIdx.x * 4 + 1] = tl_loop1 - tl_loop0;       \
        g_pre_tile[blockIdx.x * 4 + 2] = te_ - tl_loop1; g_pre_tile[blockIdx.x * 4 + 3] = tl_real;                    \
    }
#else
#define TILE_STAMP_END()
#endif
    constexpr int TS = BN; // row stride (floats)
    constexpr int ET = 512; // threads of the output loops
    static_assert(BM * TS * 4 <= LDS_BYTES, "the staged tile must fit the LDS buffers");
    float *tile = (float *)lds;
    // fp32 side inputs of a data-gradient tile (skip gradient + its sign bitmask; the input and the activation mask of the
    // norm whose backward sums leave with the tile): item u of a thread = row (tid >> 5) + 16 u, channels 4 cq .. 4 cq + 3.
    // They do not depend on the tile, so ALL of a thread's side loads are issued in one go as early as its registers are free:
    // by the producer waves right behind their last DMA wait — in flight while the consumers run the last stage and stage
    // their accumulators — and by the consumer waves as soon as the accumulators are in LDS: one memory round trip per tile
    // instead of two dependent ones (the side streams are 2 x 268 MB per launch at batch 32: what they cost is latency the
    // other workgroup of the CU cannot hide, not bandwidth).  The two roles run two COPIES of this epilogue (side_epilogue
    // below) — joined in one region, the compiler keeps the producers' prefetched values live across the consumers'
    // accumulator staging and spills — and pass its barriers as raw s_barrier: __syncthreads() would drain the vector-memory
    // queue, i.e. the prefetch, in front of every barrier.
    // (data-gradient instantiations only: the forward kernels keep their registers)
    const bool side32 = !REFLECT && !STATS && !g.out_s16 && (SUMS || g.addend != nullptr);
    auto stage_tile = [&]() {   // consumer waves: accumulator + bias, activation -> LDS
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cl = wn * TN + j * 16 + lr;
            const float bv = (bias != nullptr && n0 + cl < g.Cout) ? bias[n0 + cl] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    tile[(wm * TM + i * 16 + 4 * pl + r) * TS + cl] = acg_apply_act(acc[i][j][r] + bv, STATS ? (int)ACG_ACT_NONE : g.act);
        }
    };
    auto lds_barrier = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); };   // lgkmcnt(0) only
    auto side_epilogue = [&](auto early_c) {
        constexpr bool EARLY = decltype(early_c)::value;   // producer waves: loads first, then the barriers
        constexpr int NI = BM * (BN / 4) / ET;   // 8
        const int cq = tid & 31;
        // (launches with side inputs are bounded to 4 GiB tensors by the launcher: 32-bit offsets into buffer resources; a
        // null side pointer gives a resource of zero records, whose loads return zeros — as do lanes masked to offset ~0)
        const unsigned tb = 0xFFFFFFF0u;
        const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc((void *)g.addend, 0, g.addend != nullptr ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_am = __builtin_amdgcn_make_buffer_rsrc((void *)g.addend_mask, 0, g.addend_mask != nullptr ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void *)g.ns_x, 0, (SUMS && g.ns_x != nullptr) ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_nm = __builtin_amdgcn_make_buffer_rsrc((void *)g.ns_mask, 0, (SUMS && g.ns_mask != nullptr) ? tb : 0u, 0x00020000);
        auto side_off = [&](unsigned p) { return (p & 0x7fffffffu) * 16u + (unsigned)(n0 + cq * 4) * 4u; };
        unsigned s_aw[NI], s_nw[NI];   // mask WORDS (the nibble is picked where it is used, as are the rows' output offsets)
        f32x4 s_av[NI], s_xv[NI];
        // un-padded reflect data gradient: the mirrored pad columns land on pixels 1 and W-2 (threads 0 .. 255: consumer
        // waves).  Its load goes FIRST: vmcnt counts in order, so waiting for a load issued behind the side loads would
        // wait for all of them in front of the barrier.
        const int cf_grow = m0 / g.GW, cf_x0 = m0 - cf_grow * g.GW;
        const int cf_sd = tid >> 7, cf_c = tid & (BN - 1), cf_row = cf_sd == 0 ? 1 - cf_x0 : g.GW - 2 - cf_x0;
        const bool cf_on = !EARLY && g.colfix != nullptr && tid < 256 && (unsigned)cf_row < (unsigned)BM && n0 + cf_c < g.Cout;
        float cf = 0.f;
        if (!EARLY) {
            __syncthreads();   // every consumer wave is done with the last LDS buffer, every DMA piece has landed
            stage_tile();      // ... and the accumulators' registers are free
            if (cf_on) cf = g.colfix[((size_t)cf_grow * 2 + cf_sd) * g.Cout + n0 + cf_c];
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const unsigned po = pix_off[(tid >> 5) + (ET / 32) * u];
            const bool sd = po != NO_PIX && (po >> 31);
            const unsigned bo = side_off(po);
            const unsigned mo = acg_masked_off((bo >> 7) << 2, sd);   // float index bo / 4, word index / 32, byte offset * 4
            s_av[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_add, acg_masked_off(bo, sd), 0, 0));
            if (SUMS) s_xv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, acg_masked_off(bo, sd), 0, 0));
            s_aw[u] = __builtin_amdgcn_raw_buffer_load_b32(r_am, mo, 0, 0);
            if (SUMS) s_nw[u] = __builtin_amdgcn_raw_buffer_load_b32(r_nm, mo, 0, 0);
        }
        if (EARLY) __builtin_amdgcn_s_barrier();   // (the consumers' __syncthreads above)
        lds_barrier();                             // the tile is staged
        if (g.colfix != nullptr) {
            if (cf_on) tile[cf_row * TS + cf_c] += cf;
            lds_barrier();
        }
        // Geom.ns_part: the tile is one 128-pixel chunk of the gradient w.r.t. a norm's output, and its share of that norm's
        // backward sums (norm.hip norm_bwd_partial: gy = dy * act'(y), S1 = sum gy, S2 = sum gy * xhat) leaves with it: the
        // norm's input rides along as one more side stream (same addresses as the output), a thread sums its 8 rows of 4
        // channels in registers (rows in ascending order: the sums of the two-batch loop this replaces, bit for bit) and
        // the 16 row groups meet in LDS behind the loop.
        char *const base0 = (char *)out, *const base1 = (char *)g.out2;
        constexpr bool sums = SUMS;
        const bool remask = sums && g.ns_act != ACG_ACT_NONE && g.ns_mask == nullptr;
        const int img = m0 / GHW;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, mu = s1, rs = s1;
        const bool cok = n0 + cq * 4 < g.Cout;
        if (sums && cok) {
            mu = *(const f32x4 *)(g.ns_mean + (size_t)img * g.Cout + n0 + cq * 4);
            rs = *(const f32x4 *)(g.ns_rstd + (size_t)img * g.Cout + n0 + cq * 4);
        }
        // (trimmed for instruction count — with 16 waves on the CU the loop is issue-bound, not latency-bound: masks as
        // sign-extended one-bit fields ANDed onto the value, the un-padded grid's stores through a buffer resource with the
        // 32-bit offset the loads used)
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((void *)g.out2, 0, tb, 0x00020000);
        const unsigned cq4 = (unsigned)(n0 >> 2) + (unsigned)cq;
        const bool add_on = g.addend != nullptr, act_on = g.ns_act != ACG_ACT_NONE;
        const unsigned all_add = g.addend_mask == nullptr ? 15u : 0u;
        auto keep = [](float val, unsigned word, int bit) {   // val where bit `bit` of word is set, else +0
            const int m = __builtin_amdgcn_sbfe((int)word, bit, 1);   // 0 or -1
            return __builtin_bit_cast(float, __builtin_bit_cast(int, val) & m);
        };
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int row = (tid >> 5) + (ET / 32) * u;   // idx = tid + ET u: row idx / 32, channel group idx % 32 = cq
            const unsigned po = pix_off[row];
            f32x4 v = *(const f32x4 *)&tile[row * TS + cq * 4];
            const bool sd = po != NO_PIX && (po >> 31), live = po != NO_PIX && cok;
            const int sh = 4 * (int)(((po & 0x7fffffffu) + cq4) & 7u);   // these 4 elements' nibble of the mask words
            if (sd && add_on) {
                const unsigned nb = (s_aw[u] >> sh) | all_add;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += keep(s_av[u][q], nb, q);
            }
            if (live) {
                if (g.unpad) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, side_off(po), 0, 0);
                else *(f32x4 *)(((po >> 31) ? base1 : base0) + (size_t)side_off(po)) = v;
            }
            if (sums && live) {
                f32x4 gy = v;
                const f32x4 xh = (s_xv[u] - mu) * rs;
                if (remask) {   // no stored sign bitmask (rare): the mask from the norm's own expression (norm_apply_kernel);
                    // scale and shift are fetched per item — cache hits — instead of held across the loop
                    const f32x4 ga = *(const f32x4 *)(g.ns_gamma + (size_t)img * g.ns_gstride + n0 + cq * 4);
                    const f32x4 be = *(const f32x4 *)(g.ns_beta + (size_t)img * g.ns_gstride + n0 + cq * 4);
                    const f32x4 yy = xh * ga + be;
#pragma unroll
                    for (int q = 0; q < 4; ++q) gy[q] = yy[q] > 0.f ? gy[q] : 0.f;
                } else if (act_on) {
                    const unsigned nm = s_nw[u] >> sh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) gy[q] = keep(gy[q], nm, q);
                }
                s1 += gy;
                s2 += gy * xh;
            }
        }
        if (sums) { // thread (cq, tid >> 5) holds rows (tid >> 5) + 16 j of channels 4 cq .. 4 cq + 3
            __syncthreads(); // every thread is done reading the staged tile: its LDS becomes the scratch [2][16 row groups][BN]
            constexpr int RG = ET / 32;
            float *sc = (float *)lds;
            const int rg = tid >> 5;
            *(f32x4 *)&sc[(0 * RG + rg) * BN + cq * 4] = s1;
            *(f32x4 *)&sc[(1 * RG + rg) * BN + cq * 4] = s2;
            __syncthreads();
            if (tid < 64) {
                const int k = tid >> 5, cc = tid & 31, chunk = (m0 - img * GHW) / BM;
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < RG; ++r) a += *(const f32x4 *)&sc[(k * RG + r) * BN + cc * 4];
                if (n0 + cc * 4 < g.Cout)
                    *(f32x4 *)(g.ns_part + ((size_t)(img * (GHW / BM) + chunk) * 2 + k) * g.Cout + n0 + cc * 4) = a;
            }
        }
    };
    if (side32) {   // (wave-uniform)
        if (wave >= 4) side_epilogue(std::integral_constant<bool, true>{});
        else side_epilogue(std::integral_constant<bool, false>{});
        TILE_STAMP_END()
        return;
    }
    __syncthreads(); // every consumer wave is done with the last LDS buffer, every DMA piece has landed
    if (wave < 4) stage_tile();
    __syncthreads();
    if (g.colfix != nullptr) { // un-padded reflect data gradient: the mirrored pad columns land on pixels 1 and W-2
        const int grow = m0 / g.GW, x0c = m0 - grow * g.GW;
        const int sd = tid >> 7, c = tid & (BN - 1), row = sd == 0 ? 1 - x0c : g.GW - 2 - x0c;
        if (tid < 256 && (unsigned)row < (unsigned)BM && n0 + c < g.Cout)
            tile[row * TS + c] += g.colfix[((size_t)grow * 2 + sd) * g.Cout + n0 + c];
        __syncthreads();
    }
    if constexpr (STATS) {
        // per-tile (mean, M2) of the 128 output pixels of every channel for the InstanceNorm that follows (conv_x3.hip)
        // (the first four waves, in the summation order of conv_x3.hip: results stay bit-identical; everybody keeps the barriers)
        float *redf = &red[0][0][0]; // 256 floats
        const int c = tid & (BN - 1), h = (tid >> 7) & 1; // column c, rows h*64 .. h*64+63
        const bool sw = tid < 256;
        float sum = 0.f;
        if (sw) {
#pragma unroll 8
            for (int r = 0; r < BM / 2; ++r) sum += tile[(h * (BM / 2) + r) * TS + c];
            redf[h * BN + c] = sum;
        }
        __syncthreads();
        const float mu = (redf[c] + redf[BN + c]) * (1.f / BM);
        float sq = 0.f;
        if (sw) {
#pragma unroll 8
            for (int r = 0; r < BM / 2; ++r) {
                const float dlt = tile[(h * (BM / 2) + r) * TS + c] - mu;
                sq += dlt * dlt;
            }
        }
        __syncthreads();
        if (sw) redf[h * BN + c] = sq;
        __syncthreads();
        if (sw && h == 0 && n0 + c < g.Cout) {
            float *o = stats + ((m0 / BM) * 2) * g.Cout + n0 + c; // chunk = image * (GH*GW/128) + tile within the image
            o[0] = mu;
            o[g.Cout] = redf[c] + redf[BN + c];
        }
    }
    char *const base0 = (char *)out, *const base1 = (char *)g.out2;
    if (g.out_s16) {
        // pre-split output: a thread converts 8 channels of one pixel (hi at +0, lo at +16 of the group's 32 bytes); the
        // optional ReLU source is pre-split too and only its sign is needed: the sign of the hi halves
        const char *rsrc = (const char *)g.relu_src;
        constexpr int NK = BM * (BN / 8) / ET;   // 4 (pixel, 8-channel group) items per thread
        u32x4 sv[NK];
        // the sign words of all items first: one round trip (loaded inside the loop they were NK dependent ones).  With a
        // sign BITMASK as the ReLU source (Geom.relu_mask) an item needs one byte of a 32-bit word: sv[k][0]
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = tid + ET * k, row = idx / (BN / 8), c8 = idx - row * (BN / 8);
            const unsigned po = pix_off[row];
            const bool live = po != NO_PIX && (po >> 31) && n0 + c8 * 8 < g.Cout;
            const size_t boff = (size_t)(po & 0x7fffffffu) * 16 + (size_t)(n0 + c8 * 8) * 4;
            if (g.relu_mask != nullptr) sv[k][0] = live ? g.relu_mask[boff >> 7] : 0u;   // boff / 4 = float index, / 32 = word
            else sv[k] = *(const u32x4 *)(rsrc != nullptr && live ? rsrc + boff : in);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int idx = tid + ET * k, row = idx / (BN / 8), c8 = idx - row * (BN / 8);
            const unsigned po = pix_off[row];
            const bool live = po != NO_PIX && n0 + c8 * 8 < g.Cout;
            const size_t boff = (size_t)(po & 0x7fffffffu) * 16 + (size_t)(n0 + c8 * 8) * 4;
            const f32x4 t0 = *(const f32x4 *)&tile[row * TS + c8 * 8], t1 = *(const f32x4 *)&tile[row * TS + c8 * 8 + 4];
            float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
            if (g.relu_mask != nullptr && (po >> 31)) {
                const unsigned bits = sv[k][0] >> (8 * (c8 & 3));
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (bits >> q) & 1u ? v[q] : 0.f;
            } else if (rsrc != nullptr && (po >> 31)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned a = sv[k][q] & 0xffffu, b = sv[k][q] >> 16;
                    v[2 * q] = (a - 1u) < 0x7fffu ? v[2 * q] : 0.f;         // positive, non-zero bf16
                    v[2 * q + 1] = (b - 1u) < 0x7fffu ? v[2 * q + 1] : 0.f;
                }
            }
            if (g.mask_out != nullptr) { // (value > 0) bits of these 8 channels; the four lanes of a 32-channel word meet by shuffle
                unsigned w = 0u;
#pragma unroll
                for (int q = 0; q < 8; ++q) w |= (v[q] > 0.f ? 1u : 0u) << q;
                w = live ? w << (8 * (c8 & 3)) : 0u;
                w |= __shfl_xor(w, 1);
                w |= __shfl_xor(w, 2);
                if (live && (c8 & 3) == 0) g.mask_out[boff >> 7] = w;
            }
            if (!live) continue;
            if (g.fold_p > 0 && !(po >> 31)) { // frame pixel of a reflect data gradient: fp32 into the padded scratch, the
                *(f32x4 *)(base0 + boff) = t0; // fold kernel sums, masks and splits it
                *(f32x4 *)(base0 + boff + 16) = t1;
                continue;
            }
            acg_u32x4 hi, lo;
            acg_split8(v, hi, lo);
            char *dst = ((po >> 31) ? base1 : base0) + boff;
            *(acg_u32x4 *)dst = hi;
            *(acg_u32x4 *)(dst + 16) = lo;
        }
        TILE_STAMP_END()
        return;
    }
    {
#pragma unroll 4
        for (int k = 0; k < BM * (BN / 4) / ET; ++k) { // 8 float4 per thread, consecutive lanes on consecutive channels
            const int idx = tid + ET * k, row = idx / (BN / 4), c4 = idx - row * (BN / 4);
            const unsigned po = pix_off[row];
            if (po != NO_PIX && n0 + c4 * 4 < g.Cout)
                *(f32x4 *)(((po >> 31) ? base1 : base0) + (size_t)(po & 0x7fffffffu) * 16 + (size_t)(n0 + c4 * 4) * 4) = *(const f32x4 *)&tile[row * TS + c4 * 4];
        }
    }
    TILE_STAMP_END()
}
#undef TILE_STAMP_END

// Row-patch geometry the pre-split kernel needs: K x K tap lists in kernel-row order (forward: dx ascending, stride-1 data
// gradient: descending) whose patch fills exactly the eleven 1 KB pieces per parity image the kernel issues.
static bool pre_rowp(const Geom &g, const Taps &t, int *kdim, int *dxmin, int *kstep)
{
    if (g.is != 1 || g.os != 1 || g.oy0 != 0 || g.ox0 != 0 || g.Hout != g.GH || g.Wout != g.GW) return false;
    int k = 1;
    while (k * k < t.n) ++k;
    if (k * k != t.n || k < 2) return false;
    const int nseg_max = 2 + BM / g.GW, rows_max = BM + nseg_max * (k - 1);
    if (rows_max * 5 > PPIECES * 64 || (BM + (k - 1)) * 5 <= (PPIECES - 1) * 64) return false; // all eleven pieces, no more
    int mn = t.dx[0];
    for (int i = 1; i < t.n; ++i) mn = t.dx[i] < mn ? t.dx[i] : mn;
    for (int i = 0; i < t.n; ++i)
        if (t.dy[i] != t.dy[(i / k) * k] || t.dx[i] < mn || t.dx[i] >= mn + k) return false;
    const int ks = t.dx[0] == mn ? 1 : -1;
    for (int i = 0; i < t.n; ++i)
        if (t.dx[i] != (ks > 0 ? mn + i % k : mn + k - 1 - i % k)) return false;
    *kdim = k; *dxmin = mn; *kstep = ks;
    return true;
}

bool acg_igemm_x3_pre_ok(const Geom &g, const Taps &t)
{
    int a, b, c;
    return g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA && !g.thin && g.Cout >= 128 && g.Cin % 32 == 0 &&
           pre_rowp(g, t, &a, &b, &c);
}

// `in` is a pre-split (S16) tensor of g.Cin channels; Geom.out_s16 / relu_src describe the output side
int acg_igemm_x3_pre_launch(const void *in, const void *wp, const float *bias, float *out, const Geom &g0, const Taps &t,
                            long long n_w_elems, hipStream_t st, float *stats)
{
    Geom g = g0;
    g.thin = 0;
    if (acg_igemm_x3_pp_ok(g, t)) return acg_igemm_x3_pp_launch(in, wp, bias, out, g, t, n_w_elems, st, stats);   // the persistent form
    int kdim = 0, dxmin = 0, kstep = 1;
    ACG_REQUIRE(acg_igemm_x3_pre_ok(g, t) && pre_rowp(g, t, &kdim, &dxmin, &kstep), "igemm_conv_x3_pre: unsupported geometry");
    dim3 grid(acg_cdiv(g.Mtot, BM) * (g.ncols_pad / BN));
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    const long long out_bytes = nimg * g.Hout * g.Wout * g.Cout * 4;
    const long long w_bytes = n_w_elems * 2 * 2;
    ACG_REQUIRE(in_bytes < (1LL << 32) && w_bytes < (1LL << 32) && out_bytes < (1LL << 34) && g.Mtot < (1LL << 30),
                "igemm_conv_x3_pre: operand exceeds the buffer-addressing limit");
    ACG_REQUIRE(stats == nullptr || (((long long)g.GH * g.GW) % BM == 0 && g.act == ACG_ACT_NONE && !g.out_s16),
                "igemm_conv_x3_pre: per-tile statistics need whole 128-pixel tiles per image, no activation, fp32 output");
    ACG_REQUIRE(!g.out_s16 || (g.addend == nullptr && g.Cout % 8 == 0), "igemm_conv_x3_pre: pre-split output takes no addend");
    ACG_REQUIRE(g.relu_src == nullptr || ((g.fold_p > 0 || g.unpad) && g.out_s16), "igemm_conv_x3_pre: the ReLU source needs the frame path and pre-split output");
    ACG_REQUIRE((g.mask_out == nullptr && g.relu_mask == nullptr) || (g.out_s16 && g.Cout % 32 == 0 && g.ncols_pad == g.Cout && g.fold_p == 0 &&
                                                                        (g.relu_mask == nullptr || (g.unpad && g.relu_src == nullptr))),
                "igemm_conv_x3_pre: sign bitmasks go with pre-split output, 32-multiple channels and the un-padded grid");
    ACG_REQUIRE((g.ns_part == nullptr && (g.addend == nullptr || g.out_s16)) || out_bytes < (1LL << 32),
                "igemm_conv_x3_pre: side inputs (skip addend / norm sums) on a tensor of 4 GiB or more");
    ACG_REQUIRE(g.ns_part == nullptr || (g.unpad && !g.out_s16 && g.Cout % 4 == 0 && (g.ns_act == ACG_ACT_NONE || g.ns_act == ACG_ACT_RELU) &&
                                       g.ns_x != nullptr && g.ns_mean != nullptr && g.ns_rstd != nullptr &&
                                       (g.ns_act == ACG_ACT_NONE || g.ns_mask != nullptr || (g.ns_gamma != nullptr && g.ns_beta != nullptr))),
                "igemm_conv_x3_pre: the norm sums ride on the un-padded fp32 data gradient (act NONE / RELU)");
    ACG_REQUIRE(!g.unpad || (g.GW % BM == 0 && g.GH >= 4 && kdim == 3 && !g.reflect && g.fold_p == 0 && g.out2 == out && g.act == ACG_ACT_NONE),
                "igemm_conv_x3_pre: the un-padded reflect data gradient needs whole-row tiles of a 3x3 layer");
    const unsigned inb = (unsigned)in_bytes, wb = (unsigned)w_bytes, wlo = (unsigned)(n_w_elems * 2);
    const Taps tp = acg_taps_pack(t);
#define X3_PRE(R, S, U) hipLaunchKernelGGL((igemm_conv_x3_pre<R, S, U>), grid, dim3(512), 0, st, (const char *)in, (const __bf16 *)wp, bias, out, g, tp, inb, wb, wlo, stats, kdim, dxmin, kstep)
    ACG_REQUIRE(g.ns_part == nullptr || (!g.reflect && stats == nullptr), "igemm_conv_x3_pre: norm sums on a forward launch");
    if (g.reflect) { if (stats) X3_PRE(true, true, false); else X3_PRE(true, false, false); }
    else if (g.ns_part != nullptr) X3_PRE(false, false, true);
    else { if (stats) X3_PRE(false, true, false); else X3_PRE(false, false, false); }
#undef X3_PRE
    ACG_CHECK_LAUNCH("igemm_conv_x3_pre");
    acg_note_kernel("igemm_conv_x3_pre<REFLECT=%d,STATS=%d%s>", g.reflect ? 1 : 0, stats ? 1 : 0, g.ns_part != nullptr ? ",SUMS=1" : "");
    return ACG_OK;
}
