// Wave-specialised implicit-GEMM convolution for the bf16x3 arithmetic (ACG_PREC_BF16X3), 128x128 output tiles.
//
// The single-role kernel (conv_bf16.hip) runs its phases one after the other inside every wave — split the fp32
// operands (VALU), store them to LDS, barrier, read fragments, issue MFMAs — and rocprofv3 counters show the three
// pipes each busy about a third of the time.  Here a workgroup of 8 waves divides the work:
//   waves 0-3  consumers: ds_read fragments + v_mfma_f32_16x16x32_bf16 only (2x2 waves x 64x64 outputs; this shape
//              ran 3-4 % faster than 32x32x16 at equal cycles: the chip holds a higher clock on it);
//   waves 4-7  producers: buffer loads, hi/lo split, LDS stores for the NEXT stage into the other LDS buffer.
// One s_barrier per stage hands a filled buffer to the consumers and a drained one back to the producers, so the
// matrix pipe, the VALU and the LDS work of consecutive stages overlap on every SIMD (one wave of each role per SIMD
// and per resident workgroup; two workgroups per CU).
#include "common.h"
#include "conv_internal.h"
#include <cstdlib>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BM = 128, BN = 128, KC = 32, NK8 = KC / 8;
// LDS images are planes of 8 consecutive k: [k8][row][8 bf16], rows 16 B apart, planes 2 KB (256-B aligned).
// v_mfma_f32_16x16x32_bf16 wants row l&15 of plane l>>4 in lane l, and ds_read_b128 serves a wave in four 16-lane
// groups ({0-3,12-15,20-27}, ...) over a 256-B bank row, so a group spans two planes: XOR-permuting the rows of plane p
// by 2p keeps {rows 0-3,12-15 of plane 2q} U {rows 4-11 of plane 2q+1} on 16 distinct 16-byte slots, and the 8-lane
// groups of ds_write_b128 (2 rows x 4 planes for A, 8 columns of one plane for B) on 8 distinct slots of their 128-B row:
// SQ_LDS_BANK_CONFLICT = 0 (a [row][16 k] image with 32-B rows is 2-way conflicted: 42 % of the LDS cycles).
constexpr int APL = BM * 8, BPL = BN * 8;             // plane strides (bf16 elements)
constexpr int A_IMG = NK8 * APL, B_IMG = NK8 * BPL;   // one hi (or lo) image
__device__ __forceinline__ int lds_at(int plane, int row, int pl) { return plane * pl + ((row ^ (2 * plane)) * 8); }

// ROWP ("row patch") variant, stride-1 K x K convolutions: the 128 consecutive output pixels of a tile are a few
// SEGMENTS of grid rows, and the K taps of one kernel row read the SAME input rows shifted by one pixel.  The A image then
// holds the segments with their K-1 halo pixels and is refreshed once per kernel ROW instead of once per tap — a third of
// the A loads, splits and LDS stores for a 3x3 — while a tap's fragment is the 16 image rows of its pixels, kx further
// (consecutive except where a 16-pixel group crosses a segment end: those few lanes are 2-way).  Any shift must stay conflict-free, so the planes are not XOR-permuted here:
// planes 2q and 2q+1 (the pair a ds_read_b128 lane group spans) sit a multiple of 256 B apart (16 consecutive rows ->
// 16 distinct slots in both), and the pairs are offset by 64 B so that the 8-lane store groups (2 pixels x 4 planes) land
// 2-way (16 array cycles against the 13-cycle store).
constexpr int RP_ROWS = 160;                          // row capacity per plane: 128 + segments * (K-1)
constexpr int RP_PL = RP_ROWS * 8;                    // plane size (bf16 elements)
constexpr int RP_IMG = NK8 * RP_PL + 64;              // one hi (or lo) image incl. the pair offset
__device__ __forceinline__ int rp_at(int plane, int row) { return plane * RP_PL + (plane >> 1) * 32 + row * 8; }
template <bool ROWP> struct WsLds {
    static constexpr int AIMG = ROWP ? RP_IMG : A_IMG;
    static constexpr int TOTAL = 4 * AIMG + 4 * B_IMG;            // [A buf 0/1][hi|lo] then [B buf 0/1][hi|lo]
    static __device__ __forceinline__ int a_off(int buf) { return buf * 2 * AIMG; }
    static __device__ __forceinline__ int b_off(int buf) { return 4 * AIMG + buf * 2 * B_IMG; }
};
}

#ifdef ACG_STAMP
// diagnostic build only (tools/ws_stamps.py): cycles of set-up, main loop, epilogue (consumer wave 0) and the start time of
// every workgroup
__device__ unsigned long long g_ws_tile[8192 * 4];
extern "C" int acg_debug_ws_tile(unsigned long long *host, size_t n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_tile), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#define WS_TILE_STAMP_END()                                                                                           \
    if (tid == 0 && blockIdx.x < 8192) {                                                                              \
        const unsigned long long te_ = __builtin_amdgcn_s_memtime();                                                  \
        g_ws_tile[blockIdx.x * 4] = tl_loop0 - tl_entry; g_ws_tile[blockIdx.x * 4 + 1] = tl_loop1 - tl_loop0;         \
        g_ws_tile[blockIdx.x * 4 + 2] = te_ - tl_loop1; g_ws_tile[blockIdx.x * 4 + 3] = tl_real;                      \
    }
#else
#define WS_TILE_STAMP_END()
#endif

// 4 waves per SIMD = two 8-wave workgroups per CU: the register budget is 128 VGPRs (all four instances allocate 122,
// no scratch).  STATS: also emit the per-tile (mean, M2) for an InstanceNorm behind the convolution (see the epilogue).
template <bool REFLECT, bool STATS, bool ROWP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void
igemm_conv_x3_ws(const float *__restrict__ in, const __bf16 *__restrict__ wp,
                                                        const float *__restrict__ bias, float *__restrict__ out,
                                                        Geom g, Taps taps, unsigned in_bytes, unsigned w_bytes,
                                                        unsigned w_lo_bytes, float *__restrict__ stats, int kdim, int dxmin, int kstep)
{
    typedef WsLds<ROWP> L;
    constexpr int AIMG = L::AIMG;
    __shared__ __attribute__((aligned(16))) __bf16 lds[L::TOTAL];
    __shared__ float *out_ptr[BM]; // output row of every tile pixel (nullptr past the end)
    __shared__ const float *add_ptr[BM]; // row of Geom.addend to add on the way out (nullptr: none)
    __shared__ const float *msk_ptr[BM]; // row of Geom.relu_src whose sign masks the result (nullptr: none)
    __shared__ float red[2][2][64]; // [wn][wm][column]: cross-wave fold of the per-tile statistics

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const long long m0 = (long long)tile_m * BM;
    const int GHW = g.GH * g.GW;
    const int S = taps.n * (g.Cin / KC);

    if (tid < BM) {
        const long long m = m0 + tid;
        float *ptr = nullptr;
        const float *aptr = nullptr, *mptr = nullptr;
        if (m < g.Mtot) {
            const int n = (int)(m / GHW);
            const int r = (int)(m - (long long)n * GHW);
            const int gy = r / g.GW, gx = r - gy * g.GW;
            const int oy = gy * g.os + g.oy0, ox = gx * g.os + g.ox0;
            ptr = out + (((long long)n * g.Hout + oy) * g.Wout + ox) * g.Cout;
            if (g.fold_p > 0) { // reflect data gradient: pixels nothing is mirrored onto bypass the fold
                const int p = g.fold_p, iy = oy - p, ix = ox - p;
                const bool cy = iy >= 0 && iy < g.fold_H && !(iy >= 1 && iy <= p) && !(iy >= g.fold_H - 1 - p && iy <= g.fold_H - 2);
                const bool cx = ix >= 0 && ix < g.fold_W && !(ix >= 1 && ix <= p) && !(ix >= g.fold_W - 1 - p && ix <= g.fold_W - 2);
                if (cy && cx) {
                    const long long o2 = (((long long)n * g.fold_H + iy) * g.fold_W + ix) * g.Cout;
                    ptr = g.out2 + o2;
                    if (g.addend != nullptr) aptr = g.addend + o2;
                    if (g.relu_src != nullptr) mptr = g.relu_src + o2;
                }
            }
        }
        out_ptr[tid] = ptr;
        add_ptr[tid] = aptr;
        msk_ptr[tid] = mptr;
    }

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pt = tid - 256;
        constexpr int UPR = KC / 8, RPP = 256 / UPR, AL = BM / RPP; // 4 units per pixel row, 64 rows per pass, 2 passes
        constexpr int BCH = NK8 * BN, BL = BCH / 256;               // 16-byte chunks of the B tile: 2 per thread
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, w_bytes, 0x00020000);
        const int u = pt % UPR, rrow = pt / UPR;
        int a_row[AL], a_by[AL], a_bx[AL];
        bool a_ok[AL];
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            const long long m = m0 + rrow + RPP * j;
            a_ok[j] = m < g.Mtot;
            const long long mm = a_ok[j] ? m : 0;
            const int n = (int)(mm / GHW);
            const int r = (int)(mm - (long long)n * GHW);
            const int gy = r / g.GW, gx = r - gy * g.GW;
            a_by[j] = gy * g.is;
            a_bx[j] = gx * g.is;
            a_row[j] = REFLECT ? n * g.Hin : (n * g.Hin + a_by[j]) * g.Win + a_bx[j];
        }
        // The weight tile goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write): lane
        // L of a wave lands on the 16-byte slot (slot0 + L) of the wave's 1 KB piece of a plane, so it fetches the column the
        // XOR permutation of that plane puts there.  Measured against register staging (two loads and two ds_write_b128
        // per chunk): -2 % on the resblock forward.
        typedef __attribute__((address_space(3))) void lds_void;
        unsigned b_voff[BL];
        int b_lds[BL];
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = pt + 256 * i;
            const int plane = idx / BN, slot = idx - plane * BN, col = slot ^ (2 * plane);
            b_voff[i] = (unsigned)((((plane >> 1) * g.ncols_pad + n0 + col) * 16 + (plane & 1) * 8) * 2);
            b_lds[i] = __builtin_amdgcn_readfirstlane((plane * BPL + (slot & ~63) * 8) * 2);   // byte offset of the wave's piece
        }
        // the tap list lives in one VGPR (lane t = tap t) and is read with v_readlane: indexing the kernel argument with a
        // run-time tap number is a scalar memory load per stage (neutral in the step: the producers have that slack)
        const int tap_v = taps.pk[lane < taps.n ? lane : 0];
        auto tap_pk = [&](int t) { return __builtin_amdgcn_readlane(tap_v, t); };
        int bt = 0, bc0 = 0; // tap and first input channel of the next B stage
        // Timing-only gates (wrong results; tools/build_variant.sh <name> -DWS_ABL_...; DESIGN_LOG.md R6.3): what would a tile cost
        // that streamed half / none of its weight pieces (two accumulator sets sharing one weight stage), or half / none of its
        // gathered pixels (a de-interleaved stride-2 row patch)?  Skipped loads are not issued (weights) or masked off (gathers).
#if defined(WS_ABL_HALFW)
        const bool ws_skip_w = (tile_m & 1) != 0;
#elif defined(WS_ABL_NOW)
        const bool ws_skip_w = true;
#else
        const bool ws_skip_w = false;
#endif
#if defined(WS_ABL_HALFA)
        const bool ws_skip_a = (tile_m & 1) != 0;
#elif defined(WS_ABL_NOA)
        const bool ws_skip_a = true;
#else
        const bool ws_skip_a = false;
#endif
        auto dma_b = [&](int buf) { // 2 * BL pieces per thread: hi and lo image of stage (bt, bc0) into B buffer `buf`
            char *Bb = (char *)(lds + L::b_off(buf));
            const unsigned soff = (unsigned)((((tap_pk(bt) >> 16) * (g.Cin >> 4) + (bc0 >> 4)) * g.ncols_pad) * 16) * 2u;
            if (!ws_skip_w)
#pragma unroll
            for (int i = 0; i < BL; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Bb + b_lds[i]), 16, b_voff[i], soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Bb + B_IMG * 2 + b_lds[i]), 16, b_voff[i], soff + w_lo_bytes, 0, 0);
            }
            if (++bt == taps.n) { bt = 0; bc0 += KC; }
        };
        static_assert(BL == 2, "the vmcnt immediates below count 2 * BL = 4 DMA pieces per stage");
        f32x8 ra[ROWP ? 3 : 2 * AL]; // ROWP: up to 3 of the <= 160*4 patch units per thread; else two stages of AL units in flight
        auto store_a = [&](int j, int a_at, __bf16 *As) {
            const float v[8] = {ra[j][0], ra[j][1], ra[j][2], ra[j][3], ra[j][4], ra[j][5], ra[j][6], ra[j][7]};
            acg_u32x4 hi, lo;
            acg_split8(v, hi, lo);
            *(acg_u32x4 *)&As[a_at] = hi;
            *(acg_u32x4 *)&As[AIMG + a_at] = lo;
        };
        auto load_a = [&](int j, unsigned off, bool ok) {
            ok = ok && !ws_skip_a;
            const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off, ok), 0, 0);
            const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off + 16u, ok), 0, 0);
            const f32x4 flo = __builtin_bit_cast(f32x4, lo), fhi = __builtin_bit_cast(f32x4, hi);
            ra[j] = (f32x8){flo[0], flo[1], flo[2], flo[3], fhi[0], fhi[1], fhi[2], fhi[3]};
        };
        if constexpr (ROWP) {
            // The tile's 128 consecutive output pixels form SEGMENTS, one per grid row it touches (the first starts at column
            // x0, the others at 0); segment s occupies image rows [row0(s), row0(s) + len(s) + K-1).
            const long long grow0 = m0 / g.GW;                     // global grid row (image * GH + gy) of the first pixel
            const int x0 = (int)(m0 - grow0 * g.GW);
            const int first = g.GW - x0 < BM ? g.GW - x0 : BM;     // pixels in segment 0
            const int RW = g.GW + kdim - 1;
            const int nseg = first >= BM ? 1 : 1 + (BM - first + g.GW - 1) / g.GW;
            const int npu = (BM + nseg * (kdim - 1)) * 4;
            const long long grows = g.Mtot / g.GW;                 // grid rows in the whole tensor
            // the patch unit (pixel pp, 8-channel group uu) a thread gathers does not depend on the kernel row except through
            // iy = gy + dy: segment, image, reflected column and validity are fixed per thread, so only the row term is left
            // inside the loop (the 64-bit divisions of the segment arithmetic used to run once per kernel row and unit)
            int rp_gy[3], rp_nb[3], rp_ix[3];
            bool rp_ok[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int q = pt + 256 * j;
                const int pp = q >> 2;
                int seg = 0, px = pp;
                if (pp >= first + kdim - 1) {
                    const int qq = pp - (first + kdim - 1);
                    seg = 1 + qq / RW;
                    px = qq - (seg - 1) * RW;
                }
                const long long grow = grow0 + seg;
                const int n_img = (int)(grow / g.GH);
                int ix = (seg == 0 ? x0 : 0) + px + dxmin;
                bool ok = q < npu && grow < grows;
                if (REFLECT) {
                    ix = ix < 0 ? -ix : ix;
                    ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                } else {
                    ok = ok && (unsigned)ix < (unsigned)g.Win;
                }
                rp_gy[j] = (int)(grow - (long long)n_img * g.GH);
                rp_nb[j] = n_img * g.Hin;
                rp_ix[j] = ix;
                rp_ok[j] = ok;
            }
            int ar_t = 0, ar_c0 = 0; // first tap and first input channel of the next kernel row to gather
            auto load_a_row = [&]() { // the R x (GW+K-1) patch of one kernel row, from dx = dxmin
                const int ty = (tap_pk(ar_t) << 24) >> 24;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int uu = (pt + 256 * j) & 3;
                    int iy = rp_gy[j] + ty;
                    bool ok = rp_ok[j];
                    if (REFLECT) {
                        iy = iy < 0 ? -iy : iy;
                        iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                    } else {
                        ok = ok && (unsigned)iy < (unsigned)g.Hin;
                    }
                    load_a(j, (unsigned)(((rp_nb[j] + iy) * g.Win + rp_ix[j]) * g.Cin + ar_c0 + 8 * uu) * 4u, ok);
                }
                ar_t += kdim;
                if (ar_t >= taps.n) { ar_t = 0; ar_c0 += KC; }
            };
            const int rows = S / kdim;
            load_a_row();
            int pk_k = 0, row = 0; // position of stage s in its kernel row, and that row (its A buffer: row & 1)
            for (int s = 0; s < S; ++s) {
                dma_b(s & 1);
                if (pk_k == 0) { // first stage of a kernel row: its patch (loaded a row ago, older than the DMA pieces just issued)
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int q = pt + 256 * j;
                        if (q < npu) store_a(j, rp_at(q & 3, q >> 2), lds + L::a_off(row & 1));
                    }
                }
                if (pk_k == kdim - 1 && row + 1 < rows) {
                    load_a_row();
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // this stage's B pieces have landed; the next patch stays in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (++pk_k == kdim) { pk_k = 0; ++row; }
                __syncthreads(); // B buffer s&1 (and the A buffer of its kernel row) is full; the consumers have drained the other
            }
        } else {
            int st_t = 0, st_c0 = 0; // tap and first input channel of the next A stage to load
            auto load_a_stage = [&](auto PC) { // into register set PC::value
                constexpr int P = decltype(PC)::value;
                const int pk = tap_pk(st_t);
                const int ty = (pk << 24) >> 24, tx = (pk << 16) >> 24;
#pragma unroll
                for (int j = 0; j < AL; ++j) {
                    int pix;
                    bool ok = a_ok[j];
                    if (REFLECT) {
                        int iy = a_by[j] + ty, ix = a_bx[j] + tx;
                        iy = iy < 0 ? -iy : iy;
                        iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                        ix = ix < 0 ? -ix : ix;
                        ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                        pix = (a_row[j] + iy) * g.Win + ix;
                    } else {
                        const int iy = a_by[j] + ty, ix = a_bx[j] + tx;
                        ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
                        pix = a_row[j] + ty * g.Win + tx;
                    }
                    load_a(j + AL * P, (unsigned)(pix * g.Cin + st_c0 + 8 * u) * 4u, ok);
                }
                if (++st_t == taps.n) { st_t = 0; st_c0 += KC; }
            };
            static_assert(AL == 2, "the vmcnt immediates below count 2 * AL = 4 A loads per stage");
            // A tiles travel TWO stages ahead in registers (set s & 1): a stage is 768 MFMA cycles per workgroup, shorter than the
            // gather's latency — with one stage ahead the producers stood at the A wait every stage
            // (Round 5, tools/ws_stamps.py: hipcc puts an s_waitcnt vmcnt(0) of its own in front of the LDS stores below — a
            // wave's LDS stores are ordered behind its pending LDS-DMA, it cannot tell the A image from the B image — so the
            // second stage in flight is waited for early.  A two-barrier pipeline without that wait, with the gathers truly
            // two stages ahead and the weight pieces issued by the consumer waves, measured the same 0.29 ms on the stride-2
            // 64 -> 128 layer: what paces the stage is the ISSUE of the four gather loads, 1.7 k cycles per stage of
            // back-pressure from the CU's vector-memory path; DESIGN_LOG.md R5.4.)
            const std::integral_constant<int, 0> c0;
            const std::integral_constant<int, 1> c1;
            load_a_stage(c0);
            if (S > 1) load_a_stage(c1);
            auto stage = [&](int s, auto PC) {
                constexpr int P = decltype(PC)::value;
                dma_b(s & 1);
                // in flight, oldest first: A(s), A(s+1), the 4 DMA pieces of B(s)
                if (s + 1 < S) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
                for (int j = 0; j < AL; ++j) store_a(j + AL * P, lds_at(u, rrow + RPP * j, APL), lds + L::a_off(s & 1));
                if (s + 2 < S) {
                    load_a_stage(PC);
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); // the B pieces (and the older A(s+1)) have landed; A(s+2) stays in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads(); // buffers s&1 are full; the consumers have drained the others
            };
            for (int s = 0; s < S; s += 2) {
                stage(s, c0);
                if (s + 1 < S) stage(s + 1, c1);
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    __builtin_amdgcn_s_setprio(2);
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int TM = 64, TN = 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int pl = lane >> 4, lr = lane & 15;
    int arow[4]; // ROWP: image row of this lane's pixel in row-tile i at column offset 0 of its segment's patch
    if constexpr (ROWP) {
        const int x0c = (int)(m0 % g.GW), firstc = g.GW - x0c < BM ? g.GW - x0c : BM, RWc = g.GW + kdim - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tp = wm * TM + i * 16 + lr;
            if (tp < firstc) {
                arow[i] = tp;
            } else {
                const int sg = 1 + (tp - firstc) / g.GW;
                arow[i] = (firstc + kdim - 1) + (sg - 1) * RWc + (tp - firstc - (sg - 1) * g.GW);
            }
        }
    }
    // Fragment addresses: a per-lane byte offset fixed for the whole tile plus a wave-uniform stage offset (buffer, and
    // in ROWP the tap's column: the taps of a kernel row step through the patch columns by kstep, +1 forward, -1 data
    // gradient) kept in scalar counters — five v_add per stage (the index arithmetic written out per fragment cost 18
    // VALU instructions per stage beside the MFMAs; a per-stage scalar load of the tap behind the barrier stalled the
    // wave on its latency).  The four B fragments of a lane are 16 columns = 256 B apart (the XOR permutation only
    // touches the low column bits).
    int a_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_base[i] = 2 * (ROWP ? rp_at(pl, arow[i]) : lds_at(pl, wm * TM + i * 16 + lr, APL));
    const int b_base = 2 * lds_at(pl, wn * TN + lr, BPL);
    const char *ldsb = (const char *)lds;
    auto stage = [&](const char *pa_off, const char *pb) { // one K stage: 16 fragment reads, 48 MFMAs
        bf16x8 a[4], al[4], b[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char *pa = pa_off + a_base[i];
            a[i] = *(const bf16x8 *)pa;
            al[i] = *(const bf16x8 *)(pa + 2 * AIMG);
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
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], b[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
    };
    const char *pb0 = ldsb + b_base + 2 * L::b_off(0), *pb1 = ldsb + b_base + 2 * L::b_off(1);
    bool done = false;
#ifdef ACG_STAMP
    tl_loop0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (ROWP) {
        if (kdim == 3 && S % 6 == 0) {
            // 3 x 3 layers: six stages (two kernel rows) per trip, so buffer, tap column and B buffer of every stage are
            // compile-time constants that ride in the ds_read offset field: no vector and hardly any scalar bookkeeping is
            // left beside the MFMAs
            auto run = [&](auto ks) {
                constexpr int KS = decltype(ks)::value;
                for (int it = 0; it < S / 6; ++it) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        __syncthreads();
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int kx = KS > 0 ? k % 3 : 2 - k % 3;
                        stage(ldsb + 2 * (L::a_off(k / 3) + kx * 8), (k & 1) ? pb1 : pb0);
                    }
                }
            };
            if (kstep > 0) run(std::integral_constant<int, 1>{});
            else run(std::integral_constant<int, -1>{});
            done = true;
        }
    }
    if (!done) {
        int kk = 0, abuf = 0, kx = ROWP ? (kstep > 0 ? 0 : kdim - 1) : 0;
        for (int s = 0; s < S; ++s) {
            __syncthreads();
            stage(ldsb + 2 * (L::a_off(ROWP ? abuf : s & 1) + (ROWP ? kx * 8 : 0)), (s & 1) ? pb1 : pb0);
            if constexpr (ROWP) {
                kx += kstep;
                if (++kk == kdim) { kk = 0; abuf ^= 1; kx = kstep > 0 ? 0 : kdim - 1; }
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef ACG_STAMP
    tl_loop1 = __builtin_amdgcn_s_memtime();
#endif
    // Epilogue through LDS: the tile (accumulator + bias, activation) is staged in the LDS the main loop no longer needs
    // and leaves in coalesced 512-byte rows — 37 us per launch faster than storing the 16-column MFMA fragments
    // directly (64-byte segments), and the accumulators die early.
    constexpr int TS = BN; // row stride (floats)
    static_assert(BM * TS * 4 <= L::TOTAL * 2, "the staged tile must fit the LDS buffers");
    float *tile = (float *)lds;
    __syncthreads(); // every consumer wave is done with the last LDS buffer (the producers have exited)
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
    __syncthreads();
    if constexpr (STATS) {
        // Per-tile (mean, M2) of the 128 output pixels of every channel, for the InstanceNorm that follows: the norm's
        // statistics pass then merges 128-pixel partials (Chan) instead of re-reading the tensor.  A thread per (column,
        // row half) walks its column of the staged tile twice (mean, then squared deviations); the halves meet in LDS.
        // (A shuffle version on the accumulators spilled 23 dwords per thread next to the 128-VGPR budget: +100 MB of
        // memory traffic per launch.)  Full tiles inside one image and act == NONE are guaranteed by the launcher.
        float *redf = &red[0][0][0]; // 256 floats
        const int c = tid & (BN - 1), h = tid >> 7; // tid < 256 here: column c, rows h*64 .. h*64+63
        float sum = 0.f;
#pragma unroll 8
        for (int r = 0; r < BM / 2; ++r) sum += tile[(h * (BM / 2) + r) * TS + c];
        redf[h * BN + c] = sum;
        __syncthreads();
        const float mu = (redf[c] + redf[BN + c]) * (1.f / BM);
        float sq = 0.f;
#pragma unroll 8
        for (int r = 0; r < BM / 2; ++r) {
            const float dlt = tile[(h * (BM / 2) + r) * TS + c] - mu;
            sq += dlt * dlt;
        }
        __syncthreads();
        redf[h * BN + c] = sq;
        __syncthreads();
        if (h == 0 && n0 + c < g.Cout) {
            float *o = stats + ((m0 / BM) * 2) * g.Cout + n0 + c; // chunk = image * (GH*GW/128) + tile within the image
            o[0] = mu;
            o[g.Cout] = redf[c] + redf[BN + c];
        }
    }
    if (g.addend == nullptr && g.relu_src == nullptr) {
#pragma unroll 4
        for (int k = 0; k < BM * (BN / 4) / 256; ++k) { // 16 float4 per thread, consecutive lanes on consecutive channels
            const int idx = tid + 256 * k, row = idx / (BN / 4), c4 = idx - row * (BN / 4);
            float *dst = out_ptr[row];
            if (dst != nullptr && n0 + c4 * 4 < g.Cout) *(f32x4 *)(dst + n0 + c4 * 4) = *(const f32x4 *)&tile[row * TS + c4 * 4];
        }
        WS_TILE_STAMP_END()
        return;
    }
    // Data-gradient epilogues with side inputs (ReLU source, skip gradient and its sign bitmask): four rows at a time, every
    // side load issued before the first is used.  Rows without a side input (fold targets, the frame) read a dummy address
    // instead of branching: with the loads inside per-row branches the compiler drained the queue (vmcnt(0)) three times
    // per row, 48 dependent round trips per thread.
    const float *dummy = in;
    const unsigned *amask = g.addend_mask != nullptr ? g.addend_mask : (const unsigned *)in;
    constexpr int EB = 4;
#pragma unroll 1
    for (int kb = 0; kb < BM * (BN / 4) / 256; kb += EB) {
        float *dst[EB];
        const float *mp[EB], *ap[EB];
        f32x4 v[EB], mv[EB], av[EB];
        unsigned nb[EB];
        int col[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            const int idx = tid + 256 * (kb + u), row = idx / (BN / 4), c4 = idx - row * (BN / 4);
            col[u] = n0 + c4 * 4;
            dst[u] = out_ptr[row];
            mp[u] = msk_ptr[row];
            ap[u] = add_ptr[row];
            v[u] = *(const f32x4 *)&tile[row * TS + c4 * 4];
        }
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            mv[u] = *(const f32x4 *)((mp[u] != nullptr ? mp[u] : dummy) + col[u]);
            av[u] = *(const f32x4 *)((ap[u] != nullptr ? ap[u] : dummy) + col[u]);
            const long long f = ap[u] != nullptr ? ((ap[u] - g.addend) + col[u]) >> 2 : 0;   // float4 index of these 4 elements
            nb[u] = (amask[f >> 3] >> (4 * (int)(f & 7))) & 15u;
        }
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            if (mp[u] != nullptr) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][q] = mv[u][q] > 0.f ? v[u][q] : 0.f;
            }
            if (ap[u] != nullptr) {
                if (g.addend_mask == nullptr) nb[u] = 15u;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][q] += (nb[u] >> q) & 1u ? av[u][q] : 0.f;
            }
            if (dst[u] != nullptr && col[u] < g.Cout) *(f32x4 *)(dst[u] + col[u]) = v[u];
        }
    }
    WS_TILE_STAMP_END()
}

// same contract as acg_igemm_bf16_launch for bn == 128, Cin % 32 == 0, ACG_PREC_BF16X3
int acg_igemm_x3_ws_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g0, const Taps &t,
                           long long n_w_elems, hipStream_t st, float *stats)
{
    Geom g = g0;
    g.thin = 0;
    dim3 grid(acg_cdiv(g.Mtot, BM) * (g.ncols_pad / BN));
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    const long long w_bytes = n_w_elems * 2 * 2;
    ACG_REQUIRE(in_bytes < (1LL << 32) && w_bytes < (1LL << 32), "igemm_conv_x3_ws: operand exceeds the 4 GiB buffer-addressing limit");
    ACG_REQUIRE(stats == nullptr || (((long long)g.GH * g.GW) % BM == 0 && g.act == ACG_ACT_NONE && g.os == 1),
                "igemm_conv_x3_ws: per-tile statistics need whole 128-pixel tiles per image and no activation");
    const unsigned inb = (unsigned)in_bytes, wb = (unsigned)w_bytes, wlo = (unsigned)(n_w_elems * 2);
    // row-patch variant: K x K tap lists in kernel-row order (forward: dx ascending, stride-1 data gradient: descending) on
    // grids whose width divides the 128-pixel tile
    int kdim = 0, dxmin = 0, kstep = 1;
    static const bool no_rowp = acg_debug_switch("ACG_NO_ROWP"); // A/B switch
    if (!no_rowp && g.is == 1 && g.os == 1 && g.oy0 == 0 && g.ox0 == 0) {
        int k = 1;
        while (k * k < t.n) ++k;
        bool ok = k * k == t.n && k >= 2 && BM + (2 + BM / g.GW) * (k - 1) <= RP_ROWS && g.Hout == g.GH && g.Wout == g.GW;
        int mn = t.dx[0];
        for (int i = 1; i < t.n; ++i) mn = t.dx[i] < mn ? t.dx[i] : mn;
        for (int i = 0; ok && i < t.n; ++i) // one dy per kernel row, its dx within [mn, mn + k) (ascending: forward, descending: data gradient)
            ok = t.dy[i] == t.dy[(i / k) * k] && t.dx[i] >= mn && t.dx[i] < mn + k;
        kstep = t.dx[0] == mn ? 1 : -1;
        for (int i = 0; ok && i < t.n; ++i) // and the dx of every row run mn .. mn+k-1 in steps of kstep
            ok = t.dx[i] == (kstep > 0 ? mn + i % k : mn + k - 1 - i % k);
        if (ok) { kdim = k; dxmin = mn; }
    }
#define X3_WS(R, S, P) hipLaunchKernelGGL((igemm_conv_x3_ws<R, S, P>), grid, dim3(512), 0, st, in, (const __bf16 *)wp, bias, out, g, t, inb, wb, wlo, stats, kdim, dxmin, kstep)
#define X3_WS2(R, S) do { if (kdim) X3_WS(R, S, true); else X3_WS(R, S, false); } while (0)
    if (g.reflect) { if (stats) X3_WS2(true, true); else X3_WS2(true, false); }
    else { if (stats) X3_WS2(false, true); else X3_WS2(false, false); }
#undef X3_WS2
#undef X3_WS
    ACG_CHECK_LAUNCH("igemm_conv_x3_ws");
    acg_note_kernel("igemm_conv_x3_ws<REFLECT=%d,STATS=%d,ROWP=%d>", g.reflect ? 1 : 0, stats ? 1 : 0, kdim ? 1 : 0);
    return ACG_OK;
}

