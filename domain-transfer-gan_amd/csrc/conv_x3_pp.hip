// Persistent, role-pipelined form of conv_x3_pre.hip's tile for the residual trunk (modules.py:139-235: the 3x3 reflect
// convolutions of ResnetBlock / CINResnetBlock, 128 channels on both sides), bf16x3 arithmetic on pre-split (S16) operands.
//
// conv_x3_pre.hip runs one 128 x 128 tile per workgroup, two workgroups per CU: a tile pays 1.9 k cycles of set-up and
// 10-24 k cycles of epilogue (LDS transpose, side streams, statistics / norm sums, stores) next to a 66-68 k cycle loop, and
// the co-resident workgroup does not absorb them — its own loop is paced by its own barrier / LDS chain (DESIGN_LOG.md A.3).
// Here ONE 896-thread workgroup per CU walks over its tiles and never leaves the loop:
//   * waves 0-7 (two per SIMD) read fragments and issue MFMAs, 64 pixels x 32 channels each.  Fragments of stage s+1 are read
//     while the MFMAs of stage s issue (two register sets), so no LDS latency opens a stage; every wave also issues its
//     two 1 KB pieces of the weight tile of stage s+3 (LDS-DMA, three B buffers: the pieces have two stages to land);
//   * waves 12-13 issue the row patch (A) of the kernel row after next (two A buffers, three stages to land) — the ring runs
//     across tile boundaries, so the first stages of tile t+1 are in LDS before tile t's last MFMA.  (Their own waves: vmcnt
//     counts in order, and an L2-hit DMA piece must not queue behind an HBM-latency side load of the drain.)
//   * waves 8-11 DRAIN, one per SIMD (beside MFMAs a SIMD has about two free vector-issue slots per MFMA and the epilogue is
//     a few thousand vector instructions per tile — on two SIMDs it paced the loop): at the end of a tile the MFMA waves
//     drop their raw accumulators into a 64 KB staging tile (their only epilogue work: 32 ds_write_b32) and go straight
//     on; the drain waves turn the staged tile into the layer's output during the NEXT tile's loop, one small slice per
//     stage — bias / activation, the mirrored-column term, the masked skip gradient, per-tile norm statistics,
//     norm-backward sums, the pre-split form and its sign bitmask.  Every global load of a slice is issued FIVE stages
//     ahead of its use (a ring of three register sets; the first items' loads go out in the last stages of the previous
//     tile): under load an HBM round trip is several thousand cycles, more than a stage.
//   * one raw s_barrier per stage (14 waves), lgkmcnt-only waits in front of it; vector memory stays in flight across it.
// Arithmetic, summation order and every output bit are those of conv_x3_pre.hip (the tests compare the two exactly).
#include "common.h"
#include "conv_internal.h"
#include <cstdlib>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef PP_DMA_SPLIT
#define PP_DMA_SPLIT 1
#endif
#ifndef PP_DRAIN_DELAY
#define PP_DRAIN_DELAY 0   // s_sleep units (64 cycles) between a stage's barrier and the drain's slice
#endif
#ifndef PP_A_DELAY
#define PP_A_DELAY 0       // ... and the row-patch DMA issue
#endif
#ifndef PP_DRAIN_PRIO
#define PP_DRAIN_PRIO 0
#endif
#ifndef PP_N1
#define PP_N1 9      // fragment reads of the next stage issued beside the first twelve MFMAs (the rest beside the next ones)
#endif
namespace {
constexpr int BM = 128, BN = 128, KC = 32, NK8 = KC / 8;
constexpr int BPL = BN * 8;                 // B plane stride (bf16 elements), planes XOR-permuted as in conv_x3_pre.hip
constexpr int B_IMG = NK8 * BPL;            // one hi (or lo) B image, elements
__device__ __forceinline__ int lds_at(int plane, int row, int pl) { return plane * pl + ((row ^ (2 * plane)) * 8); }
constexpr int PROW = 80;                    // bytes per A image row (conv_x3_pre.hip: four 16-byte chunks + one pad chunk)
constexpr int PPIECES = 11;                 // 1 KB DMA pieces per parity image: 140 rows
constexpr int PIMG = PPIECES * 1024;
constexpr int ABUF = 2 * PIMG;              // one A buffer: parity images 0 and 1
constexpr int NA = 2, NB = 3;
constexpr int A_BYTES = NA * ABUF;
constexpr int BBUF = 2 * B_IMG * 2;         // one B buffer: hi image, lo image (bytes)
constexpr int B_BYTES = NB * BBUF;
constexpr int T_BYTES = BM * BN * 4;        // the staged tile (raw accumulators)
constexpr int R_BYTES = 2048;                // drain scratch: the two half-column sums of the tile statistics
constexpr int LDS_BYTES = A_BYTES + B_BYTES + T_BYTES + R_BYTES;   // 161 792 of the CU's 163 840
constexpr int NPW = 11;                     // A pieces per A wave and kernel row: 22 over 2 waves
constexpr int NTHREADS = 896;               // 8 MFMA waves, 4 drain waves, 2 A waves
enum { PP_F32 = 0, PP_STATS = 1, PP_S16 = 2, PP_SUMS = 3 };
// float index of (row, 4-channel group c4) in the staged tile: the 16-byte slots of a row are XOR-permuted by the row's
// accumulator quad so that the four rows a ds_write_b32 of the MFMA layout touches fall on four different bank groups
__device__ __forceinline__ int t_at(int row, int c4) { return row * BN + ((c4 ^ (((row >> 2) & 3) << 2)) << 2); }
}

#ifdef ACG_STAMP
__device__ unsigned long long g_pp_stamps[256 * 14 * 4];   // [workgroup][wave][wait, work, -, -] cycles over the whole kernel
extern "C" int acg_debug_pp_stamps(unsigned long long *host, size_t n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_stamps), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
__device__ unsigned long long g_pp_hist[256 * 36];         // [workgroup][stage of the tile]: cycles between barrier exits, MFMA wave 0, summed over tiles
extern "C" int acg_debug_pp_hist(unsigned long long *host, size_t n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_hist), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

template <bool REFLECT, int MODE>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void
igemm_conv_x3_pp(const char *__restrict__ in, const __bf16 *__restrict__ wp, const float *__restrict__ bias,
                 float *__restrict__ out, Geom g, unsigned long long slabs, int dys, unsigned in_bytes, unsigned w_bytes,
                 unsigned w_lo_bytes, float *__restrict__ stats, int kstep, int ntiles, int abl)
{
    // ONE shared object (a second one beside an LDS-DMA target makes hipcc drain vmcnt in front of ds_reads)
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    typedef __attribute__((address_space(3))) void lds_void;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this workgroup's tiles: XCD x (workgroups are dealt to the 8 XCDs round-robin) owns a contiguous range of tiles and
    // its workgroups walk it side by side, so vertically adjacent tiles — which share halo rows — meet in one L2
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, jx = bid >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int cnt = q8 + (xcd < r8 ? 1 : 0), start = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int Gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int nt = jx < cnt ? (cnt - jx + Gx - 1) / Gx : 0;
    auto tile_of = [&](int k) { return start + jx + k * Gx; };
    const int S = 9 * (g.Cin / KC);                  // stages per tile (36: the launcher checks Cin == 128)
    const int GHW = g.GH * g.GW;
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0) only: this wave's LDS traffic is done, vector memory stays in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if (nt == 0) return;
#ifdef ACG_STAMP
    unsigned long long st_wait = 0, st_work = 0, st_t = __builtin_amdgcn_s_memtime(), hist_t = 0;
    if (tid < 36 && blockIdx.x < 256) g_pp_hist[blockIdx.x * 36 + tid] = 0;
#define PP_STAMP(acc) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - st_t; st_t = t_; }
#define PP_STAMP_END() if (lane == 0 && blockIdx.x < 256) { g_pp_stamps[(blockIdx.x * 14 + wave) * 4] = st_wait; g_pp_stamps[(blockIdx.x * 14 + wave) * 4 + 1] = st_work; }
#else
#define PP_STAMP(acc)
#define PP_STAMP_END()
#endif

    if (wave < 8) {
        // ------------------------------------------------------------------------------------------------ MFMA waves
        __builtin_amdgcn_s_setprio(2);
        const int wm = wave >> 2, wn = wave & 3;     // 64 pixels x 32 channels
        const int pl = lane >> 4, lr = lane & 15;
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int a_base[4];   // byte offset of this lane's hi chunk in row-tile i at tap column 0 (lo: + 16); one row segment per tile
#pragma unroll
        for (int i = 0; i < 4; ++i) a_base[i] = (pl & 1) * PIMG + (wm * 64 + i * 16 + lr) * PROW + (pl >> 1) * 32;
        const char *ldsb = (const char *)lds;
        const char *pb_base = ldsb + A_BYTES + 2 * lds_at(pl, wn * 32 + lr, BPL);
        // this wave's two DMA pieces of a weight stage: 64 consecutive columns of plane `bpl` (hi image and lo image)
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, w_bytes, 0x00020000);
        // (the per-lane part of the source offset is recomputed from the lane id at every issue: held in a register across the
        // loop it is spilled, and the reload's vmcnt(0) drains the DMA queue)
        const int bpl = wave >> 1;
        const unsigned b_soff0 = (unsigned)((((bpl >> 1) * g.ncols_pad) * 16 + (bpl & 1) * 8) * 2);
        const int b_lds = __builtin_amdgcn_readfirstlane(bpl * 2048 + (wave & 1) * 1024);
        // state of the DMA stream (three stages ahead of the MFMAs): tap, first channel, tile, stages left to issue
        int d_t = 0, d_c0 = 0, d_k = 0, d_left = nt * S;
        int d_sub_lo = -100, d_sub_add = 0;
        // Geom.unpad: the tile of grid row 1 / H-2 reads the mirrored dy row through kernel row 2 / 0, whose slabs 6..8 /
        // 0..2 become the summed slabs 9..11 (conv_x3_pre.hip)
        auto tile_sub = [&](int k) {
            d_sub_lo = -100; d_sub_add = 0;
            if (g.unpad && k < nt) {
                const int gy_t = ((tile_of(k) * BM) / g.GW) % g.GH;
                if (gy_t == 1) { d_sub_lo = 6; d_sub_add = 3; }
                else if (gy_t == g.GH - 2) { d_sub_lo = 0; d_sub_add = 9; }
            }
        };
        tile_sub(0);
        const int cin16 = g.Cin >> 4;
        auto dma_b = [&](int buf) {
            char *Bb = lds + A_BYTES + buf * BBUF;
            int slab = (int)((slabs >> (4 * d_t)) & 15ull);
            slab += (unsigned)(slab - d_sub_lo) < 3u ? d_sub_add : 0;
            const unsigned soff = (unsigned)(((slab * cin16 + (d_c0 >> 4)) * g.ncols_pad) * 16) * 2u + b_soff0;
            unsigned ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const unsigned b_voff = ((((unsigned)(wave & 1) * 64u + ln) ^ (unsigned)(2 * bpl)) * 16u) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Bb + b_lds), 16, b_voff, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Bb + B_IMG * 2 + b_lds), 16, b_voff, soff + w_lo_bytes, 0, 0);
            --d_left;
            if (++d_t == 9) {
                d_t = 0; d_c0 += KC;
                if (d_c0 == g.Cin) { d_c0 = 0; tile_sub(++d_k); }
            }
        };
        dma_b(0); dma_b(1); dma_b(2);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // stages 0 and 1 have landed (this wave's pieces)
        bar();                                                 // barrier P: ... everybody's, and the A rows 0 and 1
        bf16x8 fa[2][4], fal[2][4], fb[2][2], fbl[2][2];
        // fragment reads lo .. hi - 1 of 12, in the order the next stage needs them: row tile 0, the two column tiles, row tiles
        // 1 .. 3 (lo, hi: constants at every call site — the loop unrolls and the indices fold)
        auto read_range = [&](auto setc, const int lo, const int hi, const char *pa, const char *pb) {
            constexpr int SET = decltype(setc)::value;
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                if (q < lo || q >= hi) continue;
                const int i = q < 2 ? 0 : (q - 4) / 2;
                if (q == 2) fb[SET][0] = *(const bf16x8 *)(pb);
                else if (q == 3) fbl[SET][0] = *(const bf16x8 *)(pb + 2 * B_IMG);
                else if (q == 4) fb[SET][1] = *(const bf16x8 *)(pb + 256);
                else if (q == 5) fbl[SET][1] = *(const bf16x8 *)(pb + 2 * B_IMG + 256);
                else if (q % 2 == 0) fa[SET][i] = *(const bf16x8 *)(pa + a_base[i]);
                else fal[SET][i] = *(const bf16x8 *)(pa + a_base[i] + 16);
            }
        };
        auto read_frags = [&](auto setc, const char *pa, const char *pb) { read_range(setc, 0, 12, pa, pb); };
        auto mfmas = [&](auto setc, auto halfc) {   // row tiles 2 h, 2 h + 1: twelve MFMAs
            constexpr int SET = decltype(setc)::value, H = decltype(halfc)::value;
#pragma unroll
            for (int i = 2 * H; i < 2 * H + 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[SET][i], fbl[SET][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);
                }
        };
        float *const T = (float *)(lds + A_BYTES + B_BYTES);
        auto run = [&](auto ksc) {
            constexpr int KS = decltype(ksc)::value;
            read_frags(std::integral_constant<int, 0>{}, ldsb + (KS > 0 ? 0 : 2) * PROW, pb_base);   // stage 0 of the first tile
            for (int tk = 0; tk < nt; ++tk) {
                for (int it = 0; it < S / 6; ++it) {
                    // six stages (two kernel rows) per trip: A buffer, tap column, B buffer and register set of every stage
                    // are compile-time constants
                    auto stage = [&](auto kc) {
                        constexpr int k = decltype(kc)::value, k1 = (k + 1) % 6;
                        constexpr int kxn = KS > 0 ? k1 % 3 : 2 - k1 % 3;
                        PP_STAMP(st_work)
                        bar();                               // barrier of stage s: stage s+1 is complete in LDS, everybody has
                        PP_STAMP(st_wait)                    // read stage s into registers (its B buffer is free)
#ifdef ACG_STAMP
                        if (wave == 0 && lane == 0 && blockIdx.x < 256 && S == 36) {
                            const int sidx = (it * 6 + k + 35) % 36;   // the stage that just ended
                            if (tk > 0 || it * 6 + k > 0) g_pp_hist[blockIdx.x * 36 + sidx] += st_t - hist_t;
                            hist_t = st_t;
                        }
#endif
                        // The weight tile of stage s+3.  Issuing a DMA piece stalls the wave for 60-185 cycles: the OLDER wave of a
                        // SIMD (waves 0-3) issues at the head of the stage, while its partner feeds the matrix pipe; the partner
                        // issues in the middle, behind its first twelve MFMAs
                        const bool more = d_left > 0;
                        const bool head = PP_DMA_SPLIT == 0 || wave < 4 || (abl & 32);
                        if (head && more && !(abl & 1)) dma_b(k % 3);
                        __builtin_amdgcn_sched_barrier(0);
                        // the fragment reads of stage s+1 go out beside the MFMAs of stage s, as early as the registers the MFMAs free
                        // allow (both sets complete would be 96 registers + 32 accumulators, the budget is 128): PP_N1 beside the
                        // first twelve MFMAs, the rest beside the next ones — all landed well before the end of the stage
                        const std::integral_constant<int, (k + 1) & 1> nx;
                        const std::integral_constant<int, k & 1> cu;
                        const char *pan = ldsb + (k1 / 3) * ABUF + kxn * PROW, *pbn = pb_base + (k1 % 3) * BBUF;
#ifndef PP_ABL_NOREADS   // (timing-only build: the fragments of stage 0 are used throughout)
                        read_range(nx, 0, PP_N1, pan, pbn);
#endif
                        mfmas(cu, std::integral_constant<int, 0>{});
#pragma unroll
                        for (int q = 0; q < PP_N1; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 12 - PP_N1, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (!head && more && !(abl & 1)) dma_b(k % 3);
                        __builtin_amdgcn_sched_barrier(0);
#ifndef PP_ABL_NOREADS
                        read_range(nx, PP_N1, 12, pan, pbn);
#endif
                        mfmas(cu, std::integral_constant<int, 1>{});
#pragma unroll
                        for (int q = 0; q < 12 - PP_N1; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, PP_N1, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        // this wave's pieces of stage s+2 have landed (those of s+3, just issued, stay in flight)
                        if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    };
                    stage(std::integral_constant<int, 0>{});
                    stage(std::integral_constant<int, 1>{});
                    stage(std::integral_constant<int, 2>{});
                    stage(std::integral_constant<int, 3>{});
                    stage(std::integral_constant<int, 4>{});
                    stage(std::integral_constant<int, 5>{});
                }
                // the tile is done: raw accumulators -> staging tile (the service waves take it from there), and on to the next
                // (addresses from a fresh lane id: kept in registers across the loop they are spilled and reloaded behind vmcnt(0))
                unsigned ln;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                const int tpl = (int)(ln >> 4), tlr = (int)(ln & 15);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = wn * 32 + j * 16 + tlr;
                    const int pc = (((col >> 2) ^ (tpl << 2)) << 2) + (col & 3);   // t_at(): (row >> 2) & 3 == tpl for every row below
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            T[(wm * 64 + i * 16 + 4 * tpl + r) * BN + pc] = acc[i][j][r];
                            acc[i][j][r] = 0.f;
                        }
                }
            }
        };
        if (kstep > 0) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, -1>{});
        PP_STAMP(st_work)
        bar();                                                 // barrier F: the last tile is staged
        PP_STAMP_END()
        return;
    }

    if (wave >= 12) {
        // ------------------------------------------------------------------------------------------------ A (row patch) waves
        // Piece e = pw + 2 j of a kernel row fills 1 KB of parity image e / 11; its lane L holds chunk L % 5 of image row L / 5
        // (conv_x3_pre.hip).  A tile is one run of 128 pixels of a grid row: image row = pixel x0 - 1 + row.
        const int pw = wave - 12;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
        unsigned pa_col[NPW];
        bool pa_ok[NPW];
        int pa_lds[NPW];
        int a_gy = 0, a_nb = 0;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int e = pw + 2 * j, q = e >= PPIECES ? 1 : 0, pc = e - q * PPIECES;
            pa_lds[j] = __builtin_amdgcn_readfirstlane(q * PIMG + pc * 1024);
        }
        auto geom = [&](int k) {
            const int m0 = tile_of(k) * BM;
            const int grow = m0 / g.GW, x0 = m0 - grow * g.GW;
            const int n_img = grow / g.GH;
            a_gy = grow - n_img * g.GH;
            a_nb = n_img * g.Hin;
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int e = pw + 2 * j, q = e >= PPIECES ? 1 : 0, pc = e - q * PPIECES;
                const int L = pc * 64 + lane;
                const int row = L / 5, ch = L - row * 5;
                int ix = x0 + row - 1;
                bool ok = ch < 4 && row < BM + 2;
                if (REFLECT) {
                    ix = ix < 0 ? -ix : ix;
                    ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                } else {
                    ok = ok && (unsigned)ix < (unsigned)g.Win;
                }
                pa_col[j] = (unsigned)(ix * g.Cin * 4 + (q + 2 * (ch >> 1)) * 32 + (ch & 1) * 16);
                pa_ok[j] = ok;
            }
        };
        auto dma_a = [&](int rr, int buf) {   // kernel row rr % 3, 32-channel chunk rr / 3 of the current geometry
            const int ky = rr % 3, c0 = (rr / 3) * KC;
            const int ty = (dys << (24 - 8 * ky)) >> 24;   // sign-extended byte ky
            char *Ab = lds + buf * ABUF;
            int iy = a_gy + ty;
            bool rok = true;
            if (REFLECT) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
            } else {
                rok = (unsigned)iy < (unsigned)g.Hin;
            }
            const unsigned rbase = (unsigned)((a_nb + iy) * g.Win) * (unsigned)(g.Cin * 4) + (unsigned)(c0 * 4);
#pragma unroll
            for (int j = 0; j < NPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(Ab + pa_lds[j]), 16, acg_masked_off(rbase + pa_col[j], pa_ok[j] && rok), 0, 0, 0);
        };

        const int rows = S / 3;               // kernel rows x chunks per tile (even)
        geom(0);
        dma_a(0, 0);
        dma_a(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();                                // barrier P
        int a_rr = 2, a_k = 0;                // next row to issue, and its tile
        for (int tk = 0; tk < nt; ++tk)
            for (int it = 0; it < S / 6; ++it) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    PP_STAMP(st_work)
                    bar();
                    PP_STAMP(st_wait)
                    if (k % 3 == 2) {
                        // every MFMA wave has read the last stage of the row two back: its buffer takes the row after next
                        if (PP_A_DELAY > 0) __builtin_amdgcn_s_sleep(PP_A_DELAY);
                        if (a_rr == rows) { a_rr = 0; ++a_k; if (a_k < nt) geom(a_k); }
                        if (a_k < nt && !(abl & 8)) dma_a(a_rr, k / 3);
                        ++a_rr;
                    }
                    if (k % 3 == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the row issued two stages ago: read from the next stage on
                }
            }
        PP_STAMP(st_work)
        bar();                                // barrier F
        PP_STAMP_END()
        return;
    }

    // ---------------------------------------------------------------------------------------------------- drain waves
    {
        __builtin_amdgcn_s_setprio(PP_DRAIN_PRIO);
        const int dt = tid - 512;             // 0 .. 255
        float *const T = (float *)(lds + A_BYTES + B_BYTES);
        float *const red = (float *)(lds + A_BYTES + B_BYTES + T_BYTES);   // [2][256]
        const unsigned tb = 0xFFFFFFF0u;
        const bool sd = g.unpad != 0;         // side inputs (skip gradient, ReLU sign, norm sums, mirrored columns) apply
        // Every global load and store below is issued UNCONDITIONALLY, through a buffer resource, with the offset of a slot that
        // has nothing to do masked to ~0 (loads return zeros, stores are dropped): the vector-memory instruction stream of a
        // stage is then the same on every path, which is what lets the compiler place counted vmcnt waits — with a load inside a
        // branch it falls back to vmcnt(0) at every use, and a load issued five stages ahead waits for the one issued last.
        // fp32 output (PP_F32 / PP_STATS / PP_SUMS): thread (cq, rg) takes rows rg + 8 u of channels 4 cq .. 4 cq + 3: item u in
        // stage 2 u + 1, its side loads five stages earlier (stage 2 u - 4: items 0 and 1 in stages 32 and 34 of the tile before)
        const int cq = dt & 31, rg = dt >> 5;
        const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc((void *)g.addend, 0, g.addend != nullptr ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_am = __builtin_amdgcn_make_buffer_rsrc((void *)g.addend_mask, 0, g.addend_mask != nullptr ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void *)g.ns_x, 0, (MODE == PP_SUMS && g.ns_x != nullptr) ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_nm = __builtin_amdgcn_make_buffer_rsrc((void *)g.ns_mask, 0, (MODE == PP_SUMS && g.ns_mask != nullptr) ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_dst = __builtin_amdgcn_make_buffer_rsrc(sd ? (void *)g.out2 : (void *)out, 0, tb, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_cf = __builtin_amdgcn_make_buffer_rsrc((void *)g.colfix, 0, g.colfix != nullptr ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_mean = __builtin_amdgcn_make_buffer_rsrc((void *)g.ns_mean, 0, (MODE == PP_SUMS && g.ns_mean != nullptr) ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rstd = __builtin_amdgcn_make_buffer_rsrc((void *)g.ns_rstd, 0, (MODE == PP_SUMS && g.ns_rstd != nullptr) ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rm = __builtin_amdgcn_make_buffer_rsrc((void *)g.relu_mask, 0, g.relu_mask != nullptr ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc((void *)g.relu_src, 0, g.relu_src != nullptr ? tb : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_mo = __builtin_amdgcn_make_buffer_rsrc((void *)g.mask_out, 0, g.mask_out != nullptr ? tb : 0u, 0x00020000);
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr && MODE != PP_S16) bias4 = *(const f32x4 *)(bias + cq * 4);
        const int act = (MODE == PP_STATS) ? (int)ACG_ACT_NONE : g.act;
        const bool add_on = g.addend != nullptr, nact_on = g.ns_act != ACG_ACT_NONE;
        const unsigned all_add = g.addend_mask == nullptr ? 15u : 0u;
        const bool remask = MODE == PP_SUMS && g.ns_act != ACG_ACT_NONE && g.ns_mask == nullptr;
        auto keep = [](float val, unsigned word, int bit) {   // val where bit `bit` of word is set, else +0
            const int m = __builtin_amdgcn_sbfe((int)word, bit, 1);
            return __builtin_bit_cast(float, __builtin_bit_cast(int, val) & m);
        };
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const u32x4 zu = {0u, 0u, 0u, 0u};
        f32x4 s_av[3] = {z4, z4, z4}, s_xv[3] = {z4, z4, z4};
        unsigned s_aw[3] = {0u, 0u, 0u}, s_nw[3] = {0u, 0u, 0u};
        f32x4 s1a = z4, s1b = z4, s2a = z4, s2b = z4, mu = z4, rs = z4, mu_n = z4, rs_n = z4;
        float cf_n = 0.f;                     // the mirrored-column term of the NEXT tile to drain (loaded a tile ahead)
        // statistics (PP_STATS): thread (c, h) walks rows 64 h .. 64 h + 63 of column c in ascending order (conv_x3_pre.hip's
        // two threads per column): sums, mean, then squared deviations; the halves meet in `red`
        const int sc_c = dt & (BN - 1), sc_h = dt >> 7;
        float st_sum = 0.f, st_mu = 0.f, st_sq = 0.f;
        const float bias_c = (bias != nullptr && MODE == PP_STATS) ? bias[sc_c] : 0.f;
        // pre-split output (PP_S16): thread (c8, rg16) takes rows rg16 + 16 u of channels 8 c8 .. 8 c8 + 7: item u in stage
        // 4 u + 3, its ReLU sign source five stages earlier (stage 4 u - 2: item 0 in stage 34 of the tile before), slot u % 3
        const int c8 = dt & 15, rg16 = dt >> 4;
        f32x4 bias8a = z4, bias8b = z4;
        if (bias != nullptr && MODE == PP_S16) { bias8a = *(const f32x4 *)(bias + c8 * 8); bias8b = *(const f32x4 *)(bias + c8 * 8 + 4); }
        u32x4 sv[3] = {zu, zu, zu};
        float pv8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // an item between its two halves
        f32x4 pv4 = z4;

        // one slice of the drain per stage.  tl: the tile being drained (valid if `live`); tn: the tile to drain next (valid if
        // `next`: its first loads go out here); s = 12 m + K, K at compile time
        auto step = [&](int tl, bool live, int tn, bool next, int s, auto kc) {
            constexpr int K = decltype(kc)::value;
            const int m0 = tl * BM;
            if (s == 0 && live) {
                // un-padded reflect data gradient: the mirrored pad columns land on pixels 1 and W-2 of the grid row (no bias,
                // no activation on that path: adding to the raw accumulators is adding to the tile)
                if (g.colfix != nullptr) {
                    const int grow = m0 / g.GW, x0 = m0 - grow * g.GW;
                    const int side = dt >> 7, c = dt & (BN - 1), row = side == 0 ? 1 - x0 : g.GW - 2 - x0;
                    if ((unsigned)row < (unsigned)BM) T[t_at(row, c >> 2) + (c & 3)] += cf_n;
                }
                if (MODE == PP_SUMS) { s1a = s1b = s2a = s2b = z4; mu = mu_n; rs = rs_n; }
                if (MODE == PP_STATS) { st_sum = 0.f; st_sq = 0.f; }
            }
            if (K == 10) {
                // stage 34: what the next tile's drain needs before its accumulators arrive — the mirrored-column term, the
                // statistics of the norm whose sums it emits
                const bool on = s == 34 && next;
                const int mn = tn * BM;
                const int grow = mn / g.GW, x0 = mn - grow * g.GW;
                const int side = dt >> 7, c = dt & (BN - 1), row = side == 0 ? 1 - x0 : g.GW - 2 - x0;
                cf_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_cf, acg_masked_off((unsigned)(((grow * 2 + side) * g.Cout + c) * 4), on && (unsigned)row < (unsigned)BM), 0, 0));
                if (MODE == PP_SUMS) {
                    const unsigned o = acg_masked_off((unsigned)(((mn / GHW) * g.Cout + cq * 4) * 4), on);
                    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_mean, o, 0, 0));
                    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rstd, o, 0, 0));
                    mu_n = a; rs_n = b;   // (the masked instances of stages 10 and 22 leave zeros that nobody reads)
                }
            }
            if (MODE == PP_S16) {
                // item u in two halves, so that no stage carries a whole item: stage 4 u + 2 reads the tile, applies bias,
                // activation and the ReLU sign; stage 4 u + 3 forms the sign bitmask and the pre-split halves and stores them
                if (K % 4 == 2) {
                    {   // first half of item u = (s - 2) / 4 from sign slot u % 3
                        constexpr int SLOT = ((K - 2) / 4) % 3;
                        const int u = s >> 2;
                        const u32x4 sg = sv[SLOT];
                        const int row = rg16 + 16 * (u & 7);
                        const f32x4 t0 = *(const f32x4 *)&T[t_at(row, 2 * c8)] + bias8a, t1 = *(const f32x4 *)&T[t_at(row, 2 * c8 + 1)] + bias8b;
                        float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = acg_apply_act(v[q], act);
                        if (g.relu_mask != nullptr && sd) {
                            const unsigned bits = sg[0] >> (8 * (c8 & 3));
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = (bits >> q) & 1u ? v[q] : 0.f;
                        } else if (g.relu_src != nullptr && sd) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const unsigned a = sg[q] & 0xffffu, b = sg[q] >> 16;
                                v[2 * q] = (a - 1u) < 0x7fffu ? v[2 * q] : 0.f;         // positive, non-zero bf16
                                v[2 * q + 1] = (b - 1u) < 0x7fffu ? v[2 * q + 1] : 0.f;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) pv8[q] = v[q];
                    }
                    {   // stage 4 ul - 2: the sign source of item ul (of the next tile to drain in stage 34) into slot ul % 3
                        constexpr int SLOT = ((K + 2) / 4) % 3;
                        const int ul = (s + 2) >> 2;          // 1 .. 9
                        const bool nx = ul == 9 && next, on = ((ul < 8 && live) || nx) && !(abl & 64);
                        const unsigned boff = (unsigned)((nx ? tn : tl) * BM + rg16 + 16 * (nx ? 0 : ul)) * (unsigned)(g.Cout * 4) + (unsigned)c8 * 32u;
                        u32x4 v = zu;
                        if (g.relu_mask != nullptr) v[0] = __builtin_amdgcn_raw_buffer_load_b32(r_rm, acg_masked_off((boff >> 7) << 2, on && sd), 0, 0);
                        else v = __builtin_amdgcn_raw_buffer_load_b128(r_rs, acg_masked_off(boff, on && sd), 0, 0);
                        sv[SLOT] = v;
                    }
                } else if (K % 4 == 3) {
                    // second half of item u = (s - 3) / 4
                    const int u = s >> 2;
                    const bool on = u < 8 && live;
                    const int row = rg16 + 16 * (u & 7);
                    const unsigned boff = (unsigned)(m0 + row) * (unsigned)(g.Cout * 4) + (unsigned)c8 * 32u;
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = pv8[q];
                    // (value > 0) bits of these 8 channels; the four lanes of a 32-channel word meet by shuffle
                    unsigned w = 0u;
#pragma unroll
                    for (int q = 0; q < 8; ++q) w |= (v[q] > 0.f ? 1u : 0u) << q;
                    w = w << (8 * (c8 & 3));
                    w |= __shfl_xor(w, 1);
                    w |= __shfl_xor(w, 2);
                    __builtin_amdgcn_raw_buffer_store_b32(w, r_mo, acg_masked_off((boff >> 7) << 2, on && (c8 & 3) == 0), 0, 0);
                    acg_u32x4 hi, lo;
                    acg_split8(v, hi, lo);
                    const unsigned so = acg_masked_off(boff, on && !(abl & 128));
                    __builtin_amdgcn_raw_buffer_store_b128(hi, r_dst, so, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(lo, r_dst, so, 16, 0);
                }
                return;
            }
            if (K % 2 == 0) {
                if (MODE == PP_SUMS) {
                    // even stage 2 u + 2: the norm-backward sums of item u, stored in the stage before (half of an item's work per
                    // stage: a whole item in one stage made the drain waves the last to reach that stage's barrier)
                    constexpr int PS = (K / 2 + 2) % 3;       // slot (u % 3) of item u = s / 2 - 1
                    const int u = (s >> 1) - 1;
                    const bool on = u >= 0 && u < 16 && live;
                    const int row = rg + 8 * (u & 15);
                    const unsigned ob = (unsigned)(m0 + row) * (unsigned)(g.Cout * 4) + (unsigned)cq * 16u;
                    const int sh = 4 * (int)((ob >> 4) & 7u);
                    f32x4 gy = pv4;
                    const f32x4 xh = (s_xv[PS] - mu) * rs;
                    if (remask) {   // no stored sign bitmask (rare): the mask from the norm's own expression (norm_apply_kernel)
                        const int img = m0 / GHW;
                        const f32x4 ga = *(const f32x4 *)(g.ns_gamma + (size_t)img * g.ns_gstride + cq * 4);
                        const f32x4 be = *(const f32x4 *)(g.ns_beta + (size_t)img * g.ns_gstride + cq * 4);
                        const f32x4 yy = xh * ga + be;
#pragma unroll
                        for (int q = 0; q < 4; ++q) gy[q] = yy[q] > 0.f ? gy[q] : 0.f;
                    } else if (nact_on) {
                        const unsigned nm = s_nw[PS] >> sh;
#pragma unroll
                        for (int q = 0; q < 4; ++q) gy[q] = keep(gy[q], nm, q);
                    }
                    // u % 2: conv_x3_pre.hip's row group rg + 8 (u % 2), whose rows it sums in this order (fused multiply-adds, as there)
                    if (on) {
                        if (u & 1) {
                            s1b += gy;
#pragma unroll
                            for (int q = 0; q < 4; ++q) s2b[q] = __builtin_fmaf(gy[q], xh[q], s2b[q]);
                        } else {
                            s1a += gy;
#pragma unroll
                            for (int q = 0; q < 4; ++q) s2a[q] = __builtin_fmaf(gy[q], xh[q], s2a[q]);
                        }
                    }
                }
                // the side loads of item s / 2 + 2 (of the next tile to drain from stage 32 on) into slot (K / 2 + 2) % 3 — the one
                // the sums above have just read
                if (MODE == PP_SUMS || MODE == PP_F32) {
                    constexpr int SLOT = (K / 2 + 2) % 3;
                    const int ul = (s >> 1) + 2;
                    const bool nx = ul >= 18 && next, on = ((ul < 16 && live) || nx) && sd && !(abl & 64);
                    const unsigned bo = (unsigned)((nx ? tn : tl) * BM + rg + 8 * (nx ? ul - 18 : ul)) * (unsigned)(g.Cout * 4) + (unsigned)cq * 16u;
                    const unsigned mo = acg_masked_off((bo >> 7) << 2, on);   // float index bo / 4, word index / 32, byte offset * 4
                    s_av[SLOT] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_add, acg_masked_off(bo, on), 0, 0));
                    if (MODE == PP_SUMS) s_xv[SLOT] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, acg_masked_off(bo, on), 0, 0));
                    s_aw[SLOT] = __builtin_amdgcn_raw_buffer_load_b32(r_am, mo, 0, 0);
                    if (MODE == PP_SUMS) s_nw[SLOT] = __builtin_amdgcn_raw_buffer_load_b32(r_nm, mo, 0, 0);
                }
            } else {
                // odd stage 2 u + 1: item u from slot u % 3 = ((K - 1) / 2) % 3
                constexpr int PS = ((K - 1) / 2) % 3;
                const int u = s >> 1, row = rg + 8 * (u & 15);
                const bool on = u < 16 && live;
                const unsigned ob = (unsigned)(m0 + row) * (unsigned)(g.Cout * 4) + (unsigned)cq * 16u;
                f32x4 v = *(const f32x4 *)&T[t_at(row, cq)] + bias4;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acg_apply_act(v[q], act);
                if (MODE == PP_SUMS || MODE == PP_F32) {
                    const int sh = 4 * (int)((ob >> 4) & 7u);   // these 4 elements' nibble of the mask words
                    if (sd && add_on) {
                        const unsigned nb = (s_aw[PS] >> sh) | all_add;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += keep(s_av[PS][q], nb, q);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_dst, acg_masked_off(ob, on && !(abl & 128)), 0, 0);
                pv4 = v;
            }
            if (MODE == PP_STATS && live) {
                // column sc_c, rows 64 sc_h + 4 (s - 1) .. + 3 in stages 1 .. 16 (sums); the halves meet in stage 17 / 18; rows
                // 64 sc_h + 4 (s - 18) .. + 3 in stages 18 .. 33 (squared deviations); the halves meet in stage 34 / 35
                const int cb = (sc_c & 3), c4 = sc_c >> 2;
                if (s >= 1 && s <= 16) {
                    const int r0 = 64 * sc_h + 4 * (s - 1);
                    float a = st_sum;
#pragma unroll
                    for (int r = 0; r < 4; ++r) a += T[t_at(r0 + r, c4) + cb] + bias_c;
                    st_sum = a;
                } else if (s == 17) {
                    red[sc_h * BN + sc_c] = st_sum;
                } else if (s >= 18 && s <= 33) {
                    if (s == 18) st_mu = (red[sc_c] + red[BN + sc_c]) * (1.f / BM);
                    const int r0 = 64 * sc_h + 4 * (s - 18);
                    float q2 = st_sq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dlt = (T[t_at(r0 + r, c4) + cb] + bias_c) - st_mu;
                        q2 = __builtin_fmaf(dlt, dlt, q2);
                    }
                    st_sq = q2;
                } else if (s == 34) {
                    red[256 + sc_h * BN + sc_c] = st_sq;
                } else if (s == 35 && sc_h == 0) {
                    float *o = stats + ((size_t)(m0 / BM) * 2) * g.Cout + sc_c;   // chunk = image * (GH * GW / 128) + tile within the image
                    o[0] = st_mu;
                    o[g.Cout] = red[256 + sc_c] + red[256 + BN + sc_c];
                }
            }
            if (MODE == PP_SUMS && live) {
                // the 16 row groups of conv_x3_pre.hip meet in LDS in its order: group rg + 8 q, scratch [2][16 groups][BN]
                float *sc = T;
                if (s == 33) {
                    *(f32x4 *)&sc[(0 * 16 + rg) * BN + cq * 4] = s1a;
                    *(f32x4 *)&sc[(0 * 16 + rg + 8) * BN + cq * 4] = s1b;
                    *(f32x4 *)&sc[(1 * 16 + rg) * BN + cq * 4] = s2a;
                    *(f32x4 *)&sc[(1 * 16 + rg + 8) * BN + cq * 4] = s2b;
                } else if (s == 34 && dt < 64) {
                    const int k = dt >> 5, cc = dt & 31, img = m0 / GHW, chunk = (m0 - img * GHW) / BM;
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; ++r) a += *(const f32x4 *)&sc[(k * 16 + r) * BN + cc * 4];
                    *(f32x4 *)(g.ns_part + ((size_t)(img * (GHW / BM) + chunk) * 2 + k) * g.Cout + cc * 4) = a;
                }
            }
        };

        // Period tk (the loop of tile tk) drains tile tk - 1; period nt drains the last tile after the other waves have left
        // (their closing barrier is this loop's first of period nt; from then on a barrier counts the drain waves only)
        bar();                                // barrier P
        for (int tk = 0; tk <= nt; ++tk) {
            const int tl = tk > 0 ? tile_of(tk - 1) : 0, tn = tk < nt ? tile_of(tk) : 0;
            const bool live = tk > 0, next = tk < nt;
            for (int s12 = 0; s12 < S; s12 += 12) {
#define PP_STEP(KK) PP_STAMP(st_work) bar(); PP_STAMP(st_wait) if (PP_DRAIN_DELAY > 0) __builtin_amdgcn_s_sleep(PP_DRAIN_DELAY); if (!(abl & 4)) step(tl, live, tn, next, s12 + KK, std::integral_constant<int, KK>{});
                PP_STEP(0) PP_STEP(1) PP_STEP(2) PP_STEP(3) PP_STEP(4) PP_STEP(5)
                PP_STEP(6) PP_STEP(7) PP_STEP(8) PP_STEP(9) PP_STEP(10) PP_STEP(11)
#undef PP_STEP
            }
        }
        PP_STAMP_END()
    }
#undef PP_STAMP
#undef PP_STAMP_END
}

// the row-patch tap order conv_x3_pre.hip takes (kernel rows in order, dx ascending or descending), 3 x 3 only
static bool pp_taps(const Taps &t, int *dxmin, int *kstep)
{
    if (t.n != 9) return false;
    int mn = t.dx[0];
    for (int i = 1; i < 9; ++i) mn = t.dx[i] < mn ? t.dx[i] : mn;
    const int ks = t.dx[0] == mn ? 1 : -1;
    for (int i = 0; i < 9; ++i) {
        if (t.dy[i] != t.dy[(i / 3) * 3]) return false;
        if (t.dx[i] != (ks > 0 ? mn + i % 3 : mn + 2 - i % 3)) return false;
    }
    *dxmin = mn; *kstep = ks;
    return true;
}

// Geometries of conv_x3_pre.hip that the persistent kernel takes: tiles that are one run of a grid row, 128 output columns,
// exactly 128 input channels (the drain's prefetch of the next tile's side data is scheduled on the S = 36 stages of a
// 128-channel tile: stages 10 + 12 k and 18 .. 34), no frame path
bool acg_igemm_x3_pp_ok(const Geom &g, const Taps &t)
{
    // A/B switches, read per call (tools/pp_check.py flips them inside one process).  Until the persistent kernel beats the
    // one-tile-per-workgroup kernel on every launch kind it is opt-in: ACG_PP=1 (with ACG_DEBUG_SWITCHES)
    const bool off = acg_debug_switch("ACG_NO_PP") || !acg_debug_switch("ACG_PP");
    int a, b;
    if (off || !acg_igemm_x3_pre_ok(g, t) || !pp_taps(t, &a, &b) || a != -1) return false;
    if (g.GW % BM != 0 || g.Cout != BN || g.ncols_pad != BN || g.Cin != 128 || g.fold_p != 0) return false;
    if (g.is != 1 || g.os != 1 || g.Hout != g.GH || g.Wout != g.GW || g.Hin != g.GH || g.Win != g.GW) return false;
    if (g.out_s16 && g.addend != nullptr) return false;
    return true;
}

int acg_igemm_x3_pp_launch(const void *in, const void *wp, const float *bias, float *out, const Geom &g0, const Taps &t,
                           long long n_w_elems, hipStream_t st, float *stats)
{
    Geom g = g0;
    g.thin = 0;
    int dxmin = 0, kstep = 1;
    ACG_REQUIRE(acg_igemm_x3_pp_ok(g, t) && pp_taps(t, &dxmin, &kstep), "igemm_conv_x3_pp: unsupported geometry");
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    const long long out_bytes = nimg * g.Hout * g.Wout * g.Cout * 4;
    const long long w_bytes = n_w_elems * 2 * 2;
    ACG_REQUIRE(in_bytes < (1LL << 32) && w_bytes < (1LL << 32) && out_bytes < (1LL << 32) && g.Mtot < (1LL << 30),
                "igemm_conv_x3_pp: operand exceeds the buffer-addressing limit");
    ACG_REQUIRE(stats == nullptr || (g.act == ACG_ACT_NONE && !g.out_s16), "igemm_conv_x3_pp: per-tile statistics need no activation, fp32 output");
    ACG_REQUIRE(g.relu_src == nullptr || (g.unpad && g.out_s16), "igemm_conv_x3_pp: the ReLU source needs the un-padded grid and pre-split output");
    ACG_REQUIRE((g.mask_out == nullptr && g.relu_mask == nullptr) || (g.out_s16 && (g.relu_mask == nullptr || (g.unpad && g.relu_src == nullptr))),
                "igemm_conv_x3_pp: sign bitmasks go with pre-split output");
    ACG_REQUIRE(g.ns_part == nullptr || (g.unpad && !g.out_s16 && !g.reflect && stats == nullptr && (g.ns_act == ACG_ACT_NONE || g.ns_act == ACG_ACT_RELU) &&
                                       g.ns_x != nullptr && g.ns_mean != nullptr && g.ns_rstd != nullptr &&
                                       (g.ns_act == ACG_ACT_NONE || g.ns_mask != nullptr || (g.ns_gamma != nullptr && g.ns_beta != nullptr))),
                "igemm_conv_x3_pp: the norm sums ride on the un-padded fp32 data gradient (act NONE / RELU)");
    ACG_REQUIRE(!g.unpad || (g.GH >= 4 && !g.reflect && g.out2 == out && g.act == ACG_ACT_NONE && bias == nullptr),
                "igemm_conv_x3_pp: the un-padded reflect data gradient takes no bias / activation");
    ACG_REQUIRE(g.unpad || (g.addend == nullptr && g.colfix == nullptr), "igemm_conv_x3_pp: side inputs on a forward launch");
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu = v;
    }
    // timing ablations (wrong results): 1 no weight DMA, 4 no drain, 8 no row-patch DMA, 32 all weight DMA at the stage head, 64 drain loads masked off, 128 drain stores masked off
    const int abl = (getenv("ACG_PP_ABL") && acg_debug_switch("ACG_PP_ABL")) ? atoi(getenv("ACG_PP_ABL")) : 0;
    const int ntiles = (int)(g.Mtot / BM);
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    const unsigned inb = (unsigned)in_bytes, wb = (unsigned)w_bytes, wlo = (unsigned)(n_w_elems * 2);
    unsigned long long slabs = 0;
    int dys = 0;
    for (int i = 0; i < 9; ++i) {
        ACG_REQUIRE(t.w[i] >= 0 && t.w[i] < 16, "igemm_conv_x3_pp: weight slab index");
        slabs |= (unsigned long long)t.w[i] << (4 * i);
    }
    for (int ky = 0; ky < 3; ++ky) dys |= (t.dy[3 * ky] & 0xff) << (8 * ky);
#define X3_PP(R, M) hipLaunchKernelGGL((igemm_conv_x3_pp<R, M>), dim3(grid), dim3(NTHREADS), 0, st, (const char *)in, (const __bf16 *)wp, bias, out, g, slabs, dys, inb, wb, wlo, stats, kstep, ntiles, abl)
    const int mode = g.out_s16 ? PP_S16 : g.ns_part != nullptr ? PP_SUMS : stats != nullptr ? PP_STATS : PP_F32;
    if (g.reflect) {
        if (mode == PP_S16) X3_PP(true, PP_S16); else if (mode == PP_STATS) X3_PP(true, PP_STATS); else X3_PP(true, PP_F32);
    } else {
        if (mode == PP_S16) X3_PP(false, PP_S16); else if (mode == PP_SUMS) X3_PP(false, PP_SUMS);
        else if (mode == PP_STATS) X3_PP(false, PP_STATS); else X3_PP(false, PP_F32);
    }
#undef X3_PP
    ACG_CHECK_LAUNCH("igemm_conv_x3_pp");
    acg_note_kernel("igemm_conv_x3_pp<REFLECT=%d,%s>", g.reflect ? 1 : 0, mode == PP_S16 ? "S16" : mode == PP_SUMS ? "SUMS" : mode == PP_STATS ? "STATS" : "F32");
    return ACG_OK;
}
