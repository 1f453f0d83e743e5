// detect.hip -- a1-a6 of the YSMR hot path on gfx950: fused (BGR ->) gray -> 3x3 blur -> 11x11
// Gaussian adaptive double threshold (k_threshold_strip; k_threshold for widths that are not a multiple
// of 4); 4-connected hysteresis + 8-connected component labelling, for small islands of foreground in
// registers as 64 x 64 bit windows (k_windows) and for the rest on a lock-free union-find that lives in
// the label map (k_residue: pass_union4 .. pass_bbox_euler); RETR_EXTERNAL ordering / nesting (k_rank,
// nested_component in k_geometry's launch); per-component minAreaRect (k_geometry, k_compact).
// Reference call sites: ysmr/track_eval.py:180-303.
//
// All kernels are batched over frames (detection is frame-parallel) and launched as RESIDENT grids
// that stride over their work (a grid larger than the chip holds starves every other HIP stream,
// i.e. the per-frame link kernels, until it has drained) -- except k_windows where no such kernel runs
// beside it (the batch link is one workgroup that has long been seated): there it is a wave per work
// item, handed out by the dispatcher.  Only k_threshold_strip, k_windows and the dense fallback of
// k_clear touch every pixel.
//
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
#include "common.h"
#include "thr_mfma.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <algorithm>
#include <cstdlib>

#ifdef YSMR_STAMPS
// begin stamps of the detection kernels (100 MHz realtime counter), read by scripts/ring_gaps.py
__device__ unsigned long long g_det_ring[1024 * 2];
__device__ unsigned int g_det_ring_n;
#define DET_RING(tag) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); unsigned int i_ = atomicAdd(&g_det_ring_n, 1u) & 1023u; g_det_ring[2 * i_] = (unsigned long long)(tag); g_det_ring[2 * i_ + 1] = t_; } } while (0)
extern "C" int ysmr_debug_read_det_ring(unsigned long long *out, unsigned int *n) { hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_det_ring), sizeof(unsigned long long) * 2048); if (e == hipSuccess) e = hipMemcpyFromSymbol(n, HIP_SYMBOL(g_det_ring_n), 4); return (int)e; }
#else
#define DET_RING(tag) do {} while (0)
#endif

namespace {

// ------------------------------------------------------------------------------------------
// k_threshold: a1-a3
// ------------------------------------------------------------------------------------------
constexpr int TW = 64, TH = 32;            // output tile per 256-thread block
constexpr int GW = TW + 12, GH = TH + 12;  // gray halo   (blur 1 + gaussian 5 on each side)
constexpr int BW = TW + 10, BH = TH + 10;  // blurred halo

struct Gauss11 { float k[11]; };

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int CH>
__global__ __launch_bounds__(256) void k_threshold(const uint8_t *__restrict__ frames,
                                                   uint8_t *__restrict__ cls, int H, int W,
                                                   Gauss11 gk, int inv, int t_low, int t_high,
                                                   int use_high, ysmr::GrayCoef gc)
{
    __shared__ uint8_t s_gray[GH][GW + 4];
    __shared__ uint8_t s_blur[BH][BW + 2];
    __shared__ float s_row[BH][TW + 1];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t frame_px = (size_t)H * W;
    const uint8_t *src = frames + (size_t)blockIdx.z * frame_px * CH;

    // gray halo at absolute coordinates (y0-6.., x0-6..); out-of-image entries are never read
    for (int i = tid; i < GH * GW; i += 256) {
        int r = i / GW, c = i - r * GW;
        int gy = y0 - 6 + r, gx = x0 - 6 + c;
        uint8_t v = 0;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            if (CH == 1) {
                v = src[(size_t)gy * W + gx];
            } else {
                const uint8_t *p = src + ((size_t)gy * W + gx) * 3;
                v = (uint8_t)bgr2gray(gc, p[0], p[1], p[2]);
            }
        }
        s_gray[r][c] = v;
    }
    __syncthreads();

    // 3x3 blur (REFLECT_101) evaluated at coordinates clamped to the image (REPLICATE for the 11x11)
    for (int i = tid; i < BH * BW; i += 256) {
        int r = i / BW, c = i - r * BW;
        int yy = clampi(y0 - 5 + r, 0, H - 1), xx = clampi(x0 - 5 + c, 0, W - 1);
        int ym = reflect101(yy - 1, H) - (y0 - 6), yc = yy - (y0 - 6), yp = reflect101(yy + 1, H) - (y0 - 6);
        int xm = reflect101(xx - 1, W) - (x0 - 6), xc = xx - (x0 - 6), xp = reflect101(xx + 1, W) - (x0 - 6);
        int s = s_gray[ym][xm] + 2 * s_gray[ym][xc] + s_gray[ym][xp] +
                2 * (s_gray[yc][xm] + 2 * s_gray[yc][xc] + s_gray[yc][xp]) +
                s_gray[yp][xm] + 2 * s_gray[yp][xc] + s_gray[yp][xp];
        s_blur[r][c] = (uint8_t)((s + 8) >> 4);
    }
    __syncthreads();

    // row pass: ascending FMA chain from zero
    for (int i = tid; i < BH * TW; i += 256) {
        int r = i / TW, c = i - r * TW;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 11; ++k) s = fmaf((float)s_blur[r][c + k], gk.k[k], s);
        s_row[r][c] = s;
    }
    __syncthreads();

    // column pass (symmetric form), round-half-even, compare
    for (int i = tid; i < TH * TW; i += 256) {
        int r = i / TW, c = i - r * TW;
        int y = y0 + r, x = x0 + c;
        if (y >= H || x >= W) continue;
        float s = fmaf(s_row[r + 5][c], gk.k[5], 0.0f);
#pragma unroll
        for (int j = 1; j <= 5; ++j) s = fmaf(s_row[r + 5 + j][c] + s_row[r + 5 - j][c], gk.k[5 + j], s);
        int m = clampi((int)rintf(s), 0, 255);
        int d = (int)s_blur[r + 5][c + 5] - m;
        int lo = inv ? (d <= t_low) : (d > t_low);
        int hi = use_high ? (inv ? (d <= t_high) : (d > t_high)) : lo;
        cls[(size_t)blockIdx.z * frame_px + (size_t)y * W + x] = (uint8_t)(lo | (hi << 1));
    }
}

// ------------------------------------------------------------------------------------------
// k_threshold_strip: a1-a3 for gray input, the HBM-facing kernel of the path.
//
// One WAVE owns a vertical strip: lane L holds the 4 pixels (one dword) at columns
// xs - 12 + 4L .. +3 and marches down the rows.  Lanes 0-2 and 61-63 are halo (the 3x3 blur needs
// 1 and the 11-tap row filter 5 more pixels on each side).  Per row:
//   gray dword (HBM, read once, coalesced 256 B / wave)  -> horizontal 1-2-1 sums as SWAR on two
//   16-bit fields -> vertical 1-2-1 with a 3-row sliding window -> blurred dword (exact u8)
//   -> neighbour dwords by DPP, 14 x v_cvt_f32_ubyte -> 4 x 11-tap ascending FMA chains (row filter)
//   -> 11-row ring of row-filtered values in REGISTERS (static slots, loop unrolled by 11)
//   -> symmetric column filter, round-half-even, SWAR compare, 4 class bytes = one dword store.
// Nothing but the input frame and the class map touches HBM: 1 B/px read + 1 B/px written.
// ------------------------------------------------------------------------------------------
constexpr int STRIP_HALO_LANES = 3;
constexpr int STRIP_MAX_OUT_LANES = 64 - 2 * STRIP_HALO_LANES;   // 58 lanes = 232 columns

struct StripParams {
    int H, W, batch;
    int out_lanes;     // output lanes per strip (<= 58), strips_x * out_lanes * 4 >= W
    int strips_x;
    int seg_h;         // output rows per wave
    int segs_y;
    int inv, t_low, t_high;
    int by_xcd;        // frames are dealt to the 8 XCDs (needs batch % 8 == 0 and a grid that is a multiple of 8)
    ysmr::GrayCoef gc; // BGR input only
};

typedef short short2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

struct HSum { uint32_t a, b; };   // horizontal 1-2-1 sums of 4 pixels as 2 x (16-bit, 16-bit)

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v)   // value of lane-1 (0 into lane 0)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v)   // value of lane+1 (0 into lane 63)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

__device__ __forceinline__ HSum hsum4(uint32_t g)
{
    const uint32_t gp = wave_shr1(g), gn = wave_shl1(g);
    // v_perm_b32(S0, S1, sel): result bytes picked from {S1: 0-3, S0: 4-7}, 0x0c = zero
    uint32_t p0 = __builtin_amdgcn_perm(gp, g, 0x0C000C07u);   // (b-1, b0)
    uint32_t p1 = __builtin_amdgcn_perm(g, g, 0x0C010C00u);    // (b0, b1)
    uint32_t p2 = __builtin_amdgcn_perm(g, g, 0x0C020C01u);    // (b1, b2)
    uint32_t p3 = __builtin_amdgcn_perm(g, g, 0x0C030C02u);    // (b2, b3)
    uint32_t p4 = __builtin_amdgcn_perm(gn, g, 0x0C040C03u);   // (b3, b4)
    HSum h;
    h.a = p0 + (p1 << 1) + p2;   // fields <= 1020: no carry between the 16-bit halves
    h.b = p2 + (p3 << 1) + p4;
    return h;
}

__device__ __forceinline__ uint32_t vblur4(const HSum &u, const HSum &c, const HSum &d)
{
    uint32_t ta = (u.a + (c.a << 1) + d.a + 0x00080008u) >> 4;
    uint32_t tb = (u.b + (c.b << 1) + d.b + 0x00080008u) >> 4;
    return __builtin_amdgcn_perm(tb, ta, 0x06040200u);   // bytes 0,2 of ta then bytes 0,2 of tb
}

// Per-wave constants of a strip
struct StripCtx {
    const uint8_t *src;   // frame base
    uint8_t *dst;
    int H, W, lane, c0, c0_base, y0, y1;   // c0 = c0_base + 4 * lane
    uint32_t ld_col;      // column actually loaded by this lane (clamped into the row)
    bool left_edge, right_edge;
    int edge_lane;        // lane holding the last image column (W % 4 == 0: in its byte 3)
    bool writes;
    // classification (see strip_body): margin = sgn * (s - m) + k_max; thresh = margin >= c_lo, markers = margin >= c_hi
    float sgn, nsgn, k_p;          // k_p = k_max + sgn * 1.5 * 2^23
    uint32_t swar_lo, swar_hi;     // (128 - c) in every byte
    float *xrow;             // this wave's LDS exchange row (with 8 floats of slack on each side)
    const uint32_t *q;       // this wave's LDS ring of gray rows in flight, and its LDS byte address
    uint32_t q_lds;
    ysmr::GrayCoef gc;
};

// The 4 pixels of `row` held by this lane, as loaded (lanes outside the image load a clamped, valid
// address).  Gray input: one dword.  BGR input: three dwords, turned into the gray dword only when
// the row is consumed (converting at load time would wait for the load right behind its issue).
template <int CH> struct RawRow;
template <> struct RawRow<1> { uint32_t w0; };
template <> struct RawRow<3> { uint32_t w0, w1, w2; };

template <int CH>
__device__ __forceinline__ RawRow<CH> load_row(const StripCtx &c, int row)
{
    const uint8_t *p = c.src + ((size_t)((uint32_t)row * (uint32_t)c.W + c.ld_col)) * CH;
    RawRow<CH> r;
    r.w0 = load_u32_unaligned(p);
    if constexpr (CH == 3) { r.w1 = load_u32_unaligned(p + 4); r.w2 = load_u32_unaligned(p + 8); }
    return r;
}
__device__ __forceinline__ uint32_t gray_of(const StripCtx &, const RawRow<1> &r) { return r.w0; }
__device__ __forceinline__ uint32_t gray_of(const StripCtx &c, const RawRow<3> &r)
{
    const uint32_t g0 = bgr2gray(c.gc, r.w0 & 0xFFu, (r.w0 >> 8) & 0xFFu, (r.w0 >> 16) & 0xFFu);
    const uint32_t g1 = bgr2gray(c.gc, r.w0 >> 24, r.w1 & 0xFFu, (r.w1 >> 8) & 0xFFu);
    const uint32_t g2 = bgr2gray(c.gc, (r.w1 >> 16) & 0xFFu, r.w1 >> 24, r.w2 & 0xFFu);
    const uint32_t g3 = bgr2gray(c.gc, (r.w2 >> 8) & 0xFFu, (r.w2 >> 16) & 0xFFu, r.w2 >> 24);
    return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

// XEDGE: the single out-of-image byte that matters -- the REFLECT_101 neighbour of the first/last
// image column -- is patched in from the neighbouring lane when the row is CONSUMED (patching at
// load time would force a vmcnt(0) right behind every prefetch).
template <bool XEDGE>
__device__ __forceinline__ uint32_t patch_row(const StripCtx &c, uint32_t g)
{
    if (XEDGE) {
        if (c.left_edge) {   // column -1 (byte 3 of lane 2) := column 1 (byte 1 of lane 3)
            uint32_t g3 = (uint32_t)__builtin_amdgcn_readlane((int)g, STRIP_HALO_LANES);
            if (c.lane == STRIP_HALO_LANES - 1) g = (g3 << 16) & 0xFF000000u;
        }
        if (c.right_edge) {  // column W (byte 0 of lane edge+1) := column W-2 (byte 2 of lane edge)
            uint32_t ge = (uint32_t)__builtin_amdgcn_readlane((int)g, c.edge_lane);
            if (c.lane == c.edge_lane + 1) g = (ge >> 16) & 0xFFu;
        }
    }
    return g;
}

// ---- asynchronous row loads -----------------------------------------------------------------------------
// The gray rows are fetched STRIP_PF steps ahead.  Issued from C++ into registers, the compiler's wait-count
// pass merges its bookkeeping at every control-flow join of the unrolled body and ends up waiting for (nearly)
// all loads in flight at every step, which exposes the memory latency of every row (ISA of round 1:
// s_waitcnt vmcnt(3) with six operations younger than the row it needed; vmcnt(1) once the body had one more
// branch).  So the rows travel by LDS DMA (global_load_lds_dword: no destination registers, nothing for the
// register allocator to move while a load is in flight) into a per-wave ring of STRIP_PF rows, issued by
// inline assembly, invisible to that pass, and are waited for with an explicit s_waitcnt vmcnt(N): vmcnt
// retires in order and the row wanted next always has STRIP_PF - 1 younger rows behind it, so "at most
// STRIP_PF - 1 rows' worth of operations outstanding" implies that it has landed, however many class-map
// stores were issued in between.  The row is then read from LDS one step before it is used.
constexpr int STRIP_PF = 8;   // rows in flight per wave (a power of two)

template <int CH>
struct StripState {
    float ring[11][4];    // row-filtered values of the last 11 blurred rows (slot = step mod 11)
    uint32_t bring[11];   // the blurred pixels themselves, packed (only the last 7 rows are live)
    HSum hc, hd;          // horizontal 1-2-1 sums of the two newest gray rows under the blur
    RawRow<CH> next;      // the gray row that enters the window at the next fresh step
    int k;                // its index in feed order (ring slot k % STRIP_PF)
    int row0;             // fed row k is reflect101(row0 + k): the rows fed are consecutive (see strip_body)
};

// LDS ring of one wave: slot t, dword j of a row (1 for gray, 3 for BGR), lane l at q + ((t * CH + j) * 64 + l)
template <int CH>
__device__ __forceinline__ void request_row(const StripCtx &c, const StripState<CH> &st, int k)
{
    // (rows past the image are never consumed; min() keeps their address inside the frame)
    const int r = reflect101(min(st.row0 + k, c.H), c.H);
    const uint32_t off = ((uint32_t)r * (uint32_t)c.W + c.ld_col) * CH;
    const uint32_t lds = c.q_lds + (uint32_t)((k & (STRIP_PF - 1)) * CH) * 256u;   // wave-uniform
    if constexpr (CH == 1) {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds), "v"(off), "s"(c.src) : "memory");
    } else {
        // (no instruction offsets: the offset field of an LDS-DMA load moves the LDS address as well)
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %3, %4\n\t"
                     "s_mov_b32 m0, %1\n\tglobal_load_lds_dword %3, %5\n\t"
                     "s_mov_b32 m0, %2\n\tglobal_load_lds_dword %3, %6"
                     ::"s"(lds), "s"(lds + 256u), "s"(lds + 512u), "v"(off), "s"(c.src), "s"(c.src + 4), "s"(c.src + 8) : "memory");
    }
}
// wait until fed row k has landed (k + 1 .. k + STRIP_PF - 1 may still be in flight) and read it
template <int CH>
__device__ __forceinline__ RawRow<CH> take_row(const StripCtx &c, int k)
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STRIP_PF - 1) * CH) : "memory");
    const uint32_t *q = c.q + (uint32_t)((k & (STRIP_PF - 1)) * CH) * 64u + c.lane;
    RawRow<CH> r;
    r.w0 = q[0];
    if constexpr (CH == 3) { r.w1 = q[64]; r.w2 = q[128]; }
    asm volatile("" ::: "memory");   // (the ring is rewritten behind the compiler's back: no reuse of these loads)
    return r;
}

// One step of a strip = one blurred row rb entering (feed) and one finished output row y = rb - 6 leaving (out_row).
//   * feed: the next gray row slides into the 3-row window -- or (hold) the previous blurred row is repeated:
//     rows above / below the image are BORDER_REPLICATE for the Gaussian; the blurred pixels are converted
//     once and sent to the neighbour lanes through one LDS row;
//   * out_row, issued while those floats travel: column filter, rounding, classification and store of row y,
//     which only needs rows up to rb - 1, all of them in the register ring;
//   * then the row filter of rb overwrites the ring slot of row rb - 11, which the column filter has just read
//     for the last time.
// (Round 1 issued the column filter of row rb - 5 behind the row filter of rb and sat out the LDS round trip of
// every row.)  S, the ring slot, is a compile-time constant.  In the steady-state loop the three flags are
// literal constants too and the body has no branch at all; the first and last steps of a strip pass them as
// run-time (wave-uniform) values.
template <int CH, bool XEDGE, int S>
__device__ __forceinline__ void strip_step(const StripCtx &c, const float (&kw)[6], StripState<CH> &st, int rb, bool feed,
                                           bool hold, bool out_row)
{
    const int lane = c.lane;
    float4 mine = make_float4(0.f, 0.f, 0.f, 0.f), l1 = mine, r1 = mine;
    float l2 = 0.f, r2 = 0.f;
    uint32_t b_new = 0u;
    const bool fresh = feed && !hold;
    if (fresh) {
        // the row read from the ring in the step before enters the window; its ring slot is requested again
        // (row k + STRIP_PF), and the row for the next step is read as soon as it is known to have landed
        const RawRow<CH> row = st.next;
        const HSum hu = st.hc;
        st.hc = st.hd;
#ifdef STRIP_DBG_NOBLUR   // (the STRIP_DBG_* builds delete one part each to time the rest: scripts/thr_parts.sh)
        st.hd.a = row.w0; st.hd.b = row.w0 >> 3;
#else
        st.hd = hsum4(patch_row<XEDGE>(c, gray_of(c, row)));
#endif
#ifndef STRIP_DBG_NOLOAD
        request_row<CH>(c, st, st.k + STRIP_PF);
        st.next = take_row<CH>(c, st.k + 1);
#else
        st.next.w0 += 0x01010101u;
#endif
        ++st.k;
#ifdef STRIP_DBG_NOBLUR
        uint32_t b = hu.a + st.hc.b + st.hd.a;
#else
        uint32_t b = vblur4(hu, st.hc, st.hd);
#endif
        if (XEDGE) {
            if (c.left_edge) {
                uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)b, STRIP_HALO_LANES) & 0xFFu;
                if (lane < STRIP_HALO_LANES) b = e * 0x01010101u;
            }
            if (c.right_edge) {
                uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)b, c.edge_lane) >> 24;
                if (lane > c.edge_lane) b = e * 0x01010101u;
            }
        }
        // convert once, exchange floats with the neighbour lanes through one LDS row
        // (a wave's LDS accesses complete in order; the fences only pin the compiler)
        b_new = b;
        mine = make_float4((float)(b & 0xFFu), (float)((b >> 8) & 0xFFu), (float)((b >> 16) & 0xFFu), (float)(b >> 24));
#ifdef STRIP_DBG_NOEXCH
        l1 = mine; r1 = mine; l2 = mine.x; r2 = mine.w;
        if (false) {
#else
        {
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        *reinterpret_cast<float4 *>(c.xrow + 4 * lane) = mine;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        l1 = *reinterpret_cast<const float4 *>(c.xrow + 4 * lane - 4);
        r1 = *reinterpret_cast<const float4 *>(c.xrow + 4 * lane + 4);
        l2 = c.xrow[4 * lane - 5];
        r2 = c.xrow[4 * lane + 8];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    // ---- column filter for output row y = rb - 6 (slot of row y + j is (S + 5 + j) mod 11; slot S still holds
    // row rb - 11 = y - 5)
    if (out_row) {
        constexpr int sc = (S + 5) % 11;   // slot of the centre row y
        // On gfx950 v_fma / v_add / v_sub and two-operand integer ops issue in ~2.6 cycles per wave, three-operand
        // integer ops, conversions, v_med3 / v_max in ~4.4 (scripts/ubench/mfma_probe.hip), so:
        // R = acc + 1.5 * 2^23 is round-half-even of the mean (as a float whose value is 1.5 * 2^23 + m);
        // P = sgn * s + k + sgn * 1.5 * 2^23 (the centre pixels are kept packed, one register per row, and
        // converted here: four floats per row for seven rows do not fit beside the ring at 128 VGPRs);
        // margin = P - sgn * R = sgn * (s - m) + k, by which the pixel passes the LOOSER of the two comparisons, is
        // an exact integer; v_cvt_pk_u8_f32 clamps it to a byte; the two class bits of four pixels then come from
        // two per-byte comparisons on the packed dword (bit = margin >= c, c <= 128).
        uint32_t margin = 0;
        const uint32_t bc = st.bring[sc];
        const float sf[4] = {(float)(bc & 0xFFu), (float)((bc >> 8) & 0xFFu), (float)((bc >> 16) & 0xFFu), (float)(bc >> 24)};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float acc = fmaf(st.ring[sc][o], kw[5], 0.0f);
#ifndef STRIP_DBG_NOCOL
#pragma unroll
            for (int j = 1; j <= 5; ++j)
                acc = fmaf(st.ring[(sc + j) % 11][o] + st.ring[(sc + 11 - j) % 11][o], kw[5 - j], acc);
#else
            acc += st.ring[(sc + 5) % 11][o] + st.ring[(sc + 6) % 11][o];
#endif
            const float r = acc + 12582912.0f;                      // rintf for 0 <= acc < 2^22
            margin = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(r, c.nsgn, fmaf(sf[o], c.sgn, c.k_p)), o, margin);
        }
        const uint32_t low7 = margin & 0x7F7F7F7Fu;
        const uint32_t ge_lo = (low7 + c.swar_lo) | margin, ge_hi = (low7 + c.swar_hi) | margin;   // bit 7 of each byte
        uint32_t out = ((ge_lo >> 7) & 0x01010101u) | ((ge_hi >> 6) & 0x02020202u);
        // (computed by every lane: tucked under the store's lane mask, the whole block above becomes a branch
        // target and the steady-state body is no longer straight-line code)
        asm volatile("" : "+v"(out));
#ifdef STRIP_DBG_NOSTORE
        if (c.writes && out == 0x12345678u) {
#else
        if (c.writes) {
#endif
            // the lane's column is rebuilt from a fresh lane id: kept live across the loop it is
            // the one value that got spilled
            uint32_t l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            const int y = rb - 6;
            *reinterpret_cast<uint32_t *>(c.dst + ((uint32_t)y * (uint32_t)c.W + (uint32_t)c.c0_base + 4u * l)) = out;
        }
    }
    if (fresh) {
        // ---- row filter of row rb (ascending FMA chain) into slot S
        const float v[14] = {l2, l1.x, l1.y, l1.z, l1.w, mine.x, mine.y, mine.z, mine.w, r1.x, r1.y, r1.z, r1.w, r2};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float acc = 0.0f;
#ifndef STRIP_DBG_NOROW
#pragma unroll
            for (int k = 0; k < 11; ++k) acc = fmaf(v[o + k], kw[k <= 5 ? k : 10 - k], acc);
#else
            acc = v[o] + v[o + 5] * kw[2] + v[o + 10];
#endif
            st.ring[S][o] = acc;
        }
        st.bring[S] = b_new;
    } else if (feed) {
        // BORDER_REPLICATE: blurred row rb is the blurred row of the step before
        constexpr int prev = (S + 10) % 11;
#pragma unroll
        for (int o = 0; o < 4; ++o) st.ring[S][o] = st.ring[prev][o];
        st.bring[S] = st.bring[prev];
    }
}

template <int CH, bool XEDGE, int S = 0>
__device__ __forceinline__ void strip_group(const StripCtx &c, const float (&gk)[6], StripState<CH> &st, int rb0, bool out_rows)
{
    // eleven branch-free steps: every row is fresh; out_rows is a literal at both call sites
    if constexpr (S < 11) {
        strip_step<CH, XEDGE, S>(c, gk, st, rb0 + S, true, false, out_rows);
        strip_group<CH, XEDGE, S + 1>(c, gk, st, rb0, out_rows);
    }
}

template <int CH, bool XEDGE, int S = 0>
__device__ __forceinline__ void strip_group_checked(const StripCtx &c, const float (&gk)[6], StripState<CH> &st, int rb0, int rb_hi)
{
    // eleven steps with run-time flags: the first group of a strip at the top of the image and the last steps
    // of every strip (rows beyond the image are repeats; one extra step flushes the last output row)
    if constexpr (S < 11) {
        const int rb = rb0 + S;
        if (rb <= rb_hi + 1) {
            const bool feed = rb <= rb_hi;
            const bool hold = (rb <= 0 && rb > c.y0 - 5) || rb > c.H - 1;
            strip_step<CH, XEDGE, S>(c, gk, st, rb, feed, hold, rb - 6 >= c.y0);
        }
        strip_group_checked<CH, XEDGE, S + 1>(c, gk, st, rb0, rb_hi);
    }
}

// One strip: blurred rows rb_lo = y0 - 5 .. rb_hi = y1 + 4 enter one per step, output row rb - 6 leaves.
// The gray rows fed into the blur's 3-row window are CONSECUTIVE: the first step sets the window up so that row
// g0 = max(rb_lo, 0) + 1 completes it (at the top of the image REFLECT_101 makes that window (1, 0, 1)), every
// later fresh step feeds the next row, and the image's last row is followed by its reflection, row H - 2.
// XEDGE: the strip touches the left / right image border.
template <int CH, bool XEDGE>
__device__ __forceinline__ void strip_body(const StripCtx &c, const Gauss11 &gauss)
{
    const int H = c.H;
    // The 88 multiply-adds of the two filter passes take their weight from a VECTOR register: on gfx950 a
    // scalar-register source operand halves the issue rate of v_fma / v_fmac / v_add (4.6 instead of 2.8 cycles per
    // wave-instruction, scripts/ubench/sgpr_operand.hip), which is where the compiler puts kernel arguments.  The
    // kernel is symmetric (k[i] == k[10 - i] bit for bit, make_gauss11), so six registers hold it; the empty asm
    // hides where the values came from, or the operands would be folded back into scalar registers.
    float gk[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        gk[i] = gauss.k[i];
        asm volatile("" : "+v"(gk[i]));
    }
    StripState<CH> st;
#pragma unroll
    for (int s = 0; s < 11; ++s) {
        st.bring[s] = 0u;
#pragma unroll
        for (int o = 0; o < 4; ++o) st.ring[s][o] = 0.f;
    }
    const int rb_lo = c.y0 - 5, rb_hi = c.y1 - 1 + 5;
    {   // the two gray rows above the first fed one, and the rows in flight
        const int rbc = max(rb_lo, 0);
        st.hc = hsum4(patch_row<XEDGE>(c, gray_of(c, load_row<CH>(c, reflect101(rbc - 1, H)))));
        st.hd = hsum4(patch_row<XEDGE>(c, gray_of(c, load_row<CH>(c, rbc))));
        st.row0 = rbc + 1;
        st.k = 0;
#pragma unroll
        for (int d = 0; d < STRIP_PF; ++d) request_row<CH>(c, st, d);
        st.next = take_row<CH>(c, 0);
    }
    // steady state: whole groups of eleven steps whose rows are all fresh (rb <= min(rb_hi, H - 1)); the first
    // group produces no output (y = rb - 6 < y0) and, at the top of the image, repeats rows
    const int last_fresh = min(rb_hi, H - 1);
    int rb0 = rb_lo;
    if (rb0 + 10 <= last_fresh && rb_lo >= 0) {
        strip_group<CH, XEDGE>(c, gk, st, rb0, false);
        rb0 += 11;
    } else {
        strip_group_checked<CH, XEDGE>(c, gk, st, rb0, rb_hi);
        rb0 += 11;
    }
    for (; rb0 + 10 <= last_fresh; rb0 += 11) strip_group<CH, XEDGE>(c, gk, st, rb0, true);
    for (; rb0 <= rb_hi + 1; rb0 += 11) strip_group_checked<CH, XEDGE>(c, gk, st, rb0, rb_hi);
    // rows requested beyond the last one consumed are still on their way: they must not land in the ring of
    // the wave's next strip
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Requires W % 4 == 0 and W >= 16 (else the tile kernel above is used).
// (resident grid: gray 3 blocks per CU at 128 VGPRs -- with 4 blocks and 58-row items the kernel alone is
// 15 % faster (101 us), but the link kernel's waves then find no registers while it runs and the
// end-to-end rate is no better; the BGR variant needs ~140 VGPRs and runs 2 blocks per CU)
template <int CH>
__global__ __launch_bounds__(256, 4) void k_threshold_strip(const uint8_t *__restrict__ frames,
                                                         uint8_t *__restrict__ cls, StripParams P, Gauss11 gk)
{
    DET_RING(1);
    __shared__ float s_x[4][64 * 4 + 16];
    __shared__ uint32_t s_q[4][STRIP_PF * CH * 64];
    // everything derived from the wave index is wave-uniform: say so (readfirstlane), or the compiler
    // keeps row counters in VGPRs and turns every loop test into a divergent branch + vmcnt(0)
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_frame = P.strips_x * P.segs_y;
    // resident grid, strided over the (frame, strip, segment) items: a grid larger than the chip
    // can hold keeps the dispatcher busy and starves every other stream until it has drained.
    // Workgroups b and b + 8 run on the same XCD (observed dispatch order; a matter of speed only): with
    // by_xcd every XCD works through whole frames, so the segment above and below an item -- whose ten halo
    // rows it re-reads -- was fetched into the same L2, not into another XCD's.
    long long first, step, count;
    int f_mul, f_add;
    if (P.by_xcd) {
        first = (long long)(blockIdx.x >> 3) * 4 + wave_in_block; step = (long long)(gridDim.x >> 3) * 4;
        count = (long long)(P.batch >> 3) * per_frame; f_mul = 8; f_add = (int)(blockIdx.x & 7u);
    } else {
        first = (long long)blockIdx.x * 4 + wave_in_block; step = (long long)gridDim.x * 4;
        count = (long long)P.batch * per_frame; f_mul = 1; f_add = 0;
    }
    for (long long wave = first; wave < count; wave += step) {
    const int f = (int)(wave / per_frame) * f_mul + f_add;
    const int rem = (int)(wave % per_frame);
    const int sx = rem % P.strips_x, sy = rem / P.strips_x;
    StripCtx c;
    c.H = P.H; c.W = P.W; c.lane = lane; c.gc = P.gc;
    c.xrow = s_x[wave_in_block] + 8;
    c.q = s_q[wave_in_block];
    c.q_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)s_q[wave_in_block]);
    const int xs = sx * P.out_lanes * 4;
    c.c0_base = xs - 4 * STRIP_HALO_LANES;
    c.c0 = c.c0_base + 4 * lane;
    c.y0 = sy * P.seg_h;
    c.y1 = min(c.y0 + P.seg_h, P.H);
    c.src = frames + (size_t)f * P.H * P.W * CH;
    c.dst = cls + (size_t)f * P.H * P.W;
    c.ld_col = (uint32_t)clampi(c.c0, 0, P.W - 4);
    c.left_edge = (xs == 0);
    const int last_col_rel = (P.W - 1) - (xs - 4 * STRIP_HALO_LANES);
    c.right_edge = last_col_rel < 252;           // column W falls inside this wave's lanes
    c.edge_lane = min(last_col_rel >> 2, 63);
    c.writes = lane >= STRIP_HALO_LANES && lane < STRIP_HALO_LANES + P.out_lanes && c.c0 < P.W;
    // THRESH_BINARY: bit = (d > t) = (d - t >= 1);  THRESH_BINARY_INV: bit = (d <= t) = (t + 1 - d >= 1)
    c.sgn = P.inv ? -1.0f : 1.0f;
    c.nsgn = -c.sgn;
    {
        const int k_lo = P.inv ? P.t_low + 1 : -P.t_low, k_hi = P.inv ? P.t_high + 1 : -P.t_high;
        const int k_max = k_lo > k_hi ? k_lo : k_hi;
        c.k_p = (float)k_max + c.sgn * 12582912.0f;
        c.swar_lo = (uint32_t)(128 - (1 + k_max - k_lo)) * 0x01010101u;    // (the launcher guarantees |t_low - t_high| <= 127)
        c.swar_hi = (uint32_t)(128 - (1 + k_max - k_hi)) * 0x01010101u;
    }
    if (c.left_edge || c.right_edge) strip_body<CH, true>(c, gk);
    else strip_body<CH, false>(c, gk);
    }
}

// ------------------------------------------------------------------------------------------
// Union-find in the label map.  labels[p] holds (parent index + 1); 0 = not a node.
// The smaller raster index always becomes the root, so a component's root is its first pixel.
// ------------------------------------------------------------------------------------------
constexpr uint32_t DEAD = 0xFFFFFFFFu;
constexpr int NR_STRIDE = 32;   // ints between per-frame counters

__device__ __forceinline__ uint32_t ld_node(const uint32_t *L, uint32_t i)
{
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t find_root(const uint32_t *L, uint32_t p)
{
    uint32_t v = ld_node(L, p);
    while (true) {
        if (v == 0) return DEAD;
        if (v - 1 == p) return p;
        p = v - 1;
        v = ld_node(L, p);
    }
}

// Same walk with ordinary cached loads: for passes in which no union runs concurrently (parents
// then only change by path compression towards the root, so a stale value is still an ancestor).
__device__ __forceinline__ uint32_t find_root_cached(const uint32_t *L, uint32_t p)
{
    uint32_t v = L[p];
    while (true) {
        if (v == 0) return DEAD;
        if (v - 1 == p) return p;
        p = v - 1;
        v = L[p];
    }
}

__device__ __forceinline__ void unite(uint32_t *L, uint32_t a, uint32_t b)
{
    while (true) {
        a = find_root(L, a);
        b = find_root(L, b);
        if (a == b || a == DEAD || b == DEAD) return;
        if (a < b) { uint32_t t = a; a = b; b = t; }
        uint32_t old = atomicMin(&L[a], b + 1);
        if (old == a + 1) return;
        a = old - 1;
    }
}

struct Geo {
    int H, W;
    uint32_t HW;
    size_t total;  // batch * H * W
};

// flat batch index -> frame, pixel, y, x
__device__ __forceinline__ void locate(const Geo &g, size_t flat, uint32_t &f, uint32_t &p, int &y, int &x)
{
    f = (uint32_t)(flat / g.HW);
    p = (uint32_t)(flat - (size_t)f * g.HW);
    y = (int)(p / (uint32_t)g.W);
    x = (int)(p - (uint32_t)y * (uint32_t)g.W);
}

// Load the 16 class bytes of chunk `c` (zero beyond the end of the batch).
__device__ __forceinline__ uint4 load_chunk(const uint8_t *cls, size_t c, size_t total)
{
    size_t base = c * 16;
    if (base + 16 <= total) return *reinterpret_cast<const uint4 *>(cls + base);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16 && base + i < total; ++i) w[i >> 2] |= (uint32_t)cls[base + i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t chunk_byte(const uint4 &v, int i)
{
    uint32_t w = (i < 8) ? (i < 4 ? v.x : v.y) : (i < 12 ? v.z : v.w);
    return (w >> (8 * (i & 3))) & 0xFFu;
}

// The foreground is sparse (bacteria cover ~1 % of a frame) and made of small islands.  k_windows (below)
// settles every island whose bounding box is at most 16 x 16 pixels in registers; the pixels of larger ones
// -- the RESIDUE -- are listed, one entry per pixel, and go through the union-find passes k_union4 ..
// k_bbox_euler, one lane per listed pixel.  The list has room for 1/8 of the batch; a denser residue makes
// the passes fall back to walking every pixel (`count` keeps counting past `cap`, which is how they know), and
// then everything k_windows settled is done again by them (k_residue).
// The workspace remembers, across calls, where the previous call may have written (round 5: one word per core and
// row, left by k_windows, WIN_CLEARS below; before that the components' bounding boxes, still the WIN_CLEARS_N=0
// build): if the next call gets the same buffers and geometry, zeroing the label map and the final mask there
// replaces a dense memset (362 MB per 64-frame batch at 1228x922 -- more HBM traffic than the threshold kernel
// itself).
constexpr unsigned long long WS_MAGIC = 0x59534D5248495032ull;   // "YSMRHIP2"
constexpr int WS_BIG = 16;              // components too large for the per-component clear that a header can name
constexpr int WS_BIG_AREA = 64 * 64;    // bounding-box area from which a component counts as large
struct WsHeader {
    unsigned long long magic;    // WS_MAGIC while the fields below describe a completed call
    unsigned long long labels;   // label map / final mask the component tables refer to
    unsigned long long mask;
    unsigned long long total;    // batch * H * W of that call
    uint32_t fault;              // YSMR_WS_FAULT_RESIDUE_STALL (written by a TEST into its own workspace): the next call's
                                 // barrier kernel behaves as if one workgroup never became resident; cleared by that call
    uint32_t count[2];           // [0] residue pixels found (may exceed cap: list incomplete)
    uint32_t old_bits_valid;     // written by k_clear for k_windows: the header vouched, `oldbits` describes what is to be zeroed
    int32_t batch, H, W, max_det;
    uint32_t dense;              // the tables do not cover everything written (a frame overflowed max_det, or
                                 // more than WS_BIG large components): the next call clears everything
    uint32_t n_big;
    int32_t big[WS_BIG][2];      // (frame, rank) of the large components
};
static_assert(sizeof(WsHeader) <= 256, "the header is the first 256 bytes of the workspace");
struct PixelList {
    uint32_t *idx[2];  // [0]: flat pixel indices (frame * H * W + y * W + x) of the residue, unordered
    WsHeader *hdr;
    uint32_t cap;      // entries the list can hold
};

constexpr int SPARSE_BLOCKS = 768;  // resident grid of the list-driven passes (grid-stride over the list)

// The passes are bound by chains of dependent L2 round trips (union-find), so they want many
// short threads: one lane per pixel, grid-stride.
// (nb: the number of blocks that take part, see k_residue)
// Where a pass takes its pixels from: entries `first, first + step, ...` of a list (the residue list in HBM, or one frame's
// part of it gathered in LDS), optionally only those of one frame (a frame's workgroup reading the whole list), or -- `dense`,
// the list overflowed -- every pixel from `dense_base` on.
struct ResidueIter {
    const uint32_t *idx;
    size_t n, first, step, dense_base;
    int only_frame;        // -1: any
    bool dense;
};
__device__ __forceinline__ ResidueIter residue_iter_grid(const PixelList &pl, const Geo &g, unsigned nb)
{
    const size_t cnt = pl.hdr->count[0];
    const bool dense = cnt > pl.cap;
    return ResidueIter{pl.idx[0], dense ? g.total : cnt, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)nb * 256, 0, -1, dense};
}
#define FOR_LISTED_PIXELS(it, g, flat)                                                                                    \
    for (size_t li_ = (it).first; li_ < (it).n; li_ += (it).step)                                                         \
        if (const size_t flat = (it).dense ? (it).dense_base + li_ : (size_t)(it).idx[li_];                               \
            (it).only_frame < 0 || flat / (g).HW == (size_t)(it).only_frame)

struct CompTables {
    int32_t *nroots;   // [B * NR_STRIDE]: one counter per 128-byte line (same-line atomics serialise)
    int32_t *roots;    // [B][max_det] unordered
    int32_t *order;    // [B][max_det] roots sorted descending (= findContours order)
    int32_t *bbox;     // [B][max_det][4] minx, maxx, miny, maxy
    int32_t *euler4;   // [B][max_det] 4 * Euler number (8-connectivity)
    int32_t *nested;   // [B][max_det] component lies in a hole of another one
    int32_t *max_roots; // largest per-frame component count of the batch (k_rank)
    int32_t *prev_n;   // [B] component counts of the previous call (k_compact -> k_clear)
    int32_t *bbox_tmp; // [B][max_det][4], [B][max_det]: box and Euler number of the components k_windows
    int32_t *euler_tmp; //   settled, in the order of `roots` (k_rank moves them to their rank)
    int max_det;
};

// Clears the label map and the final mask (both are written sparsely afterwards): everything, unless the header
// vouches for these buffers -- then k_windows does it, core by core (WIN_CLEARS; in a WIN_CLEARS_N=0 build this
// kernel does, inside the bounding boxes of the previous call's components).  A
// resident grid instead of hipMemsetAsync: the runtime's fill kernels use grids far larger than the chip
// holds, and such a grid starves every other stream (the link) until it has drained.
// (The status words are zeroed here.  The per-call counters are left at zero by k_compact, the last kernel of a
// call, which also keeps the component counts for this kernel (prev_n) and vouches for the buffers; k_windows,
// the next launch, marks the header invalid until then, so a chain that was cut short is followed by a full clear.)
constexpr int CLEAR_BLOCKS = 512;
// k_windows also CLEARS (round 5, last hours): the previous call's label map and mask are zeroed by the wave that works on a core
// -- the pixels that call may have written (one word per core and row, left in the workspace by k_windows itself) and this one
// will not -- instead of by k_clear beforehand (which dirtied the same ~1.45 M lines k_windows dirties again, and had them
// written back in between: 50-57 us per batch).
#ifndef WIN_CLEARS_N
#define WIN_CLEARS_N 1
#endif
constexpr bool WIN_CLEARS = WIN_CLEARS_N != 0;
__device__ __forceinline__ void clear_box(uint32_t *__restrict__ lab, uint8_t *__restrict__ mask, int W, int x0, int x1, int y0,
                                          int y1, int first, int step)
{
    const int bw = x1 - x0 + 1, cells = bw * (y1 - y0 + 1);
    for (int i = first; i < cells; i += step) {
        const int r = i / bw, c = i - r * bw;
        const size_t at = (size_t)(y0 + r) * W + (x0 + c);
        if (lab[at] != 0u) {
            lab[at] = 0u;
            if (mask) mask[at] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void k_clear(PixelList pl, CompTables t, uint8_t *labels, uint8_t *mask, size_t total,
                                               int n_counters, int32_t *status, int batch, int H, int W)
{
    DET_RING(2);
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    const WsHeader *hp = pl.hdr;
    struct { unsigned long long magic, labels, mask, total; int32_t batch, H, W, max_det; uint32_t dense, n_big; } h =
        {hp->magic, hp->labels, hp->mask, hp->total, hp->batch, hp->H, hp->W, hp->max_det, hp->dense, hp->n_big};
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < batch; i += 256) status[i] = 0;
    const size_t HW = (size_t)H * W;
    if (h.magic == WS_MAGIC && h.labels == (unsigned long long)labels && h.mask == (unsigned long long)mask &&
        h.total == total && h.batch == batch && h.H == H && h.W == W && h.max_det == t.max_det && !h.dense &&
        (WIN_CLEARS || h.n_big <= (uint32_t)WS_BIG)) {      // (the boxes' limit: k_windows' words cover components of any size)
        // k_windows zeroes what the previous call left, core by core, from the words it wrote then (WIN_CLEARS)
        if (WIN_CLEARS) {
            if (blockIdx.x == 0 && threadIdx.x == 0) pl.hdr->old_bits_valid = 1u;
            return;
        }
        // Work items are (16 consecutive ranks, frame), frame fastest: the populated ranks come first in every
        // frame, so the live items are spread evenly over the waves.  16 lanes per component (one per column of
        // its box), four components at a time, all box loads of an item first; blind stores (test-then-store is a
        // dependent L2 round trip per cell: 92 us per batch that way)
        const int lane = threadIdx.x & 63;
        const long long items = (long long)((t.max_det + 15) / 16) * batch, waves = (long long)(stride / 64);
        for (long long it = (long long)(tid / 64); it < items; it += waves) {
            const int kb = (int)(it / batch), f = (int)(it - (long long)kb * batch);
            const int n = t.prev_n[f];
            if (kb * 16 >= n) break;   // (items are rank-major: every later item of this wave is empty too)
            uint32_t *lab = reinterpret_cast<uint32_t *>(labels) + (size_t)f * HW;
            uint8_t *msk = mask ? mask + (size_t)f * HW : nullptr;
            int4 box[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kb * 16 + i * 4 + (lane >> 4);
                box[i] = k < n ? *reinterpret_cast<const int4 *>(t.bbox + ((size_t)f * t.max_det + k) * 4) : make_int4(0, -1, 0, -1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bw = box[i].y - box[i].x + 1, bh = box[i].w - box[i].z + 1;
                if (bw <= 0 || bh <= 0 || bw * bh >= WS_BIG_AREA) continue;   // (large ones are in h.big: every block helps below)
                for (int r = 0; r < bh; ++r)
                    for (int c = lane & 15; c < bw; c += 16) {
                        const size_t at = (size_t)(box[i].z + r) * W + (box[i].x + c);
                        lab[at] = 0u;
                        if (msk) msk[at] = 0;
                    }
            }
        }
        for (uint32_t b = 0; b < h.n_big; ++b) {
            const int f = hp->big[b][0];
            const size_t s = (size_t)f * t.max_det + hp->big[b][1];
            clear_box(reinterpret_cast<uint32_t *>(labels) + (size_t)f * HW, mask ? mask + (size_t)f * HW : nullptr, W, t.bbox[s * 4 + 0],
                      t.bbox[s * 4 + 1], t.bbox[s * 4 + 2], t.bbox[s * 4 + 3], (int)tid, (int)stride);
        }
    } else {
        // (nobody reads the counters on this path, and every block takes it -- the header's fields that say whether it
        // vouches are not written here)
        if (blockIdx.x == 0 && threadIdx.x == 0) pl.hdr->old_bits_valid = 0u;
        if (blockIdx.x == 0) {
            for (int i = threadIdx.x; i < n_counters; i += 256) t.nroots[i] = 0;
            for (int i = threadIdx.x; i < batch; i += 256) t.prev_n[i] = 0;
            if (threadIdx.x == 0) pl.hdr->count[0] = 0;
        }
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int which = 0; which < 2; ++which) {
            uint8_t *p = which ? mask : labels;
            const size_t bytes = which ? total : total * sizeof(uint32_t);
            if (!p) continue;
            uint4 *q = reinterpret_cast<uint4 *>(p);       // (16-byte aligned: checked by the caller)
            const size_t n16 = bytes / 16;
            for (size_t i = tid; i < n16; i += stride) q[i] = z;
            for (size_t i = n16 * 16 + tid; i < bytes; i += stride) p[i] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_windows: a4 (binary_propagation) and the labelling of a5 for every small island, in registers.
//
// An ISLAND is an 8-connected component of the pixels that carry any class bit.  The final mask
// R = binary_propagation(markers, mask=thresh) lies inside the islands, a thresh 4-component and an
// 8-component of R each lie inside one island, and nothing about an island depends on pixels outside it: the
// islands are independent problems.  An island is SMALL if its bounding box is at most 16 x 16 pixels.
//
// The frame is tiled by 32 x 32 CORES; the WINDOW of a core is the core grown by 16 pixels on every side:
// 64 rows x 64 columns, held by one wave as a 64-bit row mask per lane (bit k = column k) for each class
// bit.  Dilations are shifts within the lane and wave_shr:1 / wave_shl:1 moves between lanes; "until nothing
// changes" is a ballot.  From a seed pixel of its core a wave flood-fills (8-connected, over the class pixels
// it sees) a set F:
//   * F touches the outermost ring of the window, or its box exceeds 16 x 16: the island is large -- seen from
//     every window alike, because a small island is completely visible, ring-free, from every window whose core
//     holds one of its pixels.  The wave lists F's pixels inside its OWN core as residue (cores tile the
//     frame: every residue pixel is listed exactly once) and seeds their labels for the union-find passes.
//   * otherwise F is a whole small island.  The window whose core holds the island's first pixel (raster
//     order) owns it: hysteresis = iterate R |= dilate4(R) & thresh from R = markers (scipy's
//     binary_propagation, markers outside the mask included); 8-components of R by flood fill (normally R is
//     the whole island: one component, no second fill); per component the label map, the final mask, the
//     bounding box and the Euler number from bit-quad counts (three popcounts per lane).
// The class bytes of a macro-tile of 4 x 4 cores are turned into row masks once, in LDS (v_dot4_u32_u8 gathers
// the class bit of four pixels), and read by the 16 windows of the tile.
// ------------------------------------------------------------------------------------------
constexpr int WIN_CORE = 32, WIN_MARGIN = 16;
#ifndef WIN_GROUP_N
#define WIN_GROUP_N 8
#endif
constexpr int WIN_GROUP = WIN_GROUP_N;                           // cores per work item, side by side (4, or 8: round 5)
constexpr int WIN_HALVES = 2 * (WIN_GROUP + 1);                  // 16-pixel half-words per staged row (5 dwords, odd: lanes = rows hit 32 banks)
// Resident grid: 4 blocks per CU.  The kernel's own time hardly depends on it (1536 .. 2560 blocks: 62 .. 66 us), but
// every block holds 10 KB of LDS, and with 8 per CU only one of k_frame's 59 KB blocks fits next to them
// (scripts/sweep_e2e_blocks.sh: end to end 83.7 k frames/s with 2048 blocks here and 1536 in k_geometry, 85.7 k
// with 1024 and 512).
#ifndef WINDOW_BLOCKS_N
#define WINDOW_BLOCKS_N 1024
#endif
constexpr int WINDOW_BLOCKS = WINDOW_BLOCKS_N;
#ifndef WINDOW_BLOCKS_FREE_N
#define WINDOW_BLOCKS_FREE_N 8192
#endif
constexpr int WINDOW_BLOCKS_FREE = WINDOW_BLOCKS_FREE_N;   // k_windows' grid where no per-frame link kernel runs beside it (components_batch)

__device__ __forceinline__ uint64_t row_above(uint64_t v)   // lane r: the mask of lane r-1 (0 into lane 0)
{
    return (uint64_t)wave_shr1((uint32_t)v) | ((uint64_t)wave_shr1((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint64_t row_below(uint64_t v)   // lane r: the mask of lane r+1 (0 into lane 63)
{
    return (uint64_t)wave_shl1((uint32_t)v) | ((uint64_t)wave_shl1((uint32_t)(v >> 32)) << 32);
}
// m << 1 / m >> 1 as two 32-bit operations each (v_lshlrev_b64 is a quarter-rate instruction)
__device__ __forceinline__ uint64_t shl1(uint64_t m)
{
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    return (uint64_t)(lo << 1) | ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 31) << 32);
}
__device__ __forceinline__ uint64_t shr1(uint64_t m)
{
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    return (uint64_t)__builtin_amdgcn_alignbit(hi, lo, 1) | ((uint64_t)(hi >> 1) << 32);
}
__device__ __forceinline__ uint64_t grow8(uint64_t m)
{
    const uint64_t h = m | shl1(m) | shr1(m);
    return h | row_above(h) | row_below(h);
}
__device__ __forceinline__ uint64_t grow4(uint64_t m) { return m | shl1(m) | shr1(m) | row_above(m) | row_below(m); }
__device__ __forceinline__ uint64_t lane_mask(uint64_t v, int src)   // the mask held by lane `src` (wave-uniform)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32);
}
// 8-connected flood fill of `seed` inside `within`
__device__ __forceinline__ uint64_t flood8(uint64_t seed, uint64_t within)
{
    uint64_t f = seed;
    while (true) {
        uint64_t n = grow8(f) & within;
        n = grow8(n) & within;
        const bool changed = __ballot(n != f) != 0ull;
        f = n;
        if (!changed) return f;
    }
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_imin(int v) { return min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, CTRL, ROW_MASK, 0xF, false)); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_imax(int v) { return max(v, __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, ROW_MASK, 0xF, false)); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_iadd(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false); }
#define WAVE_REDUCE(fn, v)                                                                                      \
    (v = fn<0x111, 0xF>(v), v = fn<0x112, 0xF>(v), v = fn<0x114, 0xF>(v), v = fn<0x118, 0xF>(v), v = fn<0x142, 0xA>(v), \
     v = fn<0x143, 0xC>(v), __builtin_amdgcn_readlane(v, 63))

struct WindowBox { int x0, x1, y0, y1; };   // window coordinates (column = bit, row = lane); wave-uniform

// ---- the same in 32 bits.  A small island lies within 15 columns of any of its pixels, so once a seed in column sx
// is chosen the work moves to the 32-column SLICE [sx - 16, sx + 15] of the window (one funnel shift per row mask):
// half the instructions of the 64-bit forms in every step below.
__device__ __forceinline__ uint32_t grow8(uint32_t m)
{
    const uint32_t h = m | (m << 1) | (m >> 1);
    return h | wave_shr1(h) | wave_shl1(h);
}
__device__ __forceinline__ uint32_t grow4(uint32_t m) { return m | (m << 1) | (m >> 1) | wave_shr1(m) | wave_shl1(m); }
__device__ __forceinline__ uint32_t flood8(uint32_t seed, uint32_t within)
{
    uint32_t f = seed;
    while (true) {
        uint32_t n = grow8(f) & within;
        n = grow8(n) & within;
        const bool changed = __ballot(n != f) != 0ull;
        f = n;
        if (!changed) return f;
    }
}
__device__ __forceinline__ WindowBox box_of(uint32_t m)
{
    WindowBox b;
    const unsigned long long rows = __ballot(m != 0u);
    b.y0 = rows ? __builtin_ctzll(rows) : 64;
    b.y1 = rows ? 63 - __builtin_clzll(rows) : -1;
    int lo = m ? __builtin_ctz(m) : 32, hi = m ? 31 - __builtin_clz(m) : -1;
    b.x0 = WAVE_REDUCE(dpp_imin, lo);
    b.x1 = WAVE_REDUCE(dpp_imax, hi);
    return b;
}

struct WindowOut {
    uint8_t *cls;        // frame base
    uint32_t *labels;
    uint8_t *mask;       // may be null
    int W, H, wx0, wy0;  // frame geometry; frame coordinates of window column 0 / row 0
    uint32_t fbase;      // flat index of the frame's first pixel
};

// A component whose table row is still to be written: its slot comes back from an atomic (a round trip of 1-2 us), and the
// wave does not wait for it -- the row is written when the NEXT component is found (or behind the wave's last item), by
// which time the slot has arrived.  Wave-uniform but for `idx` (lane 0's).
struct PendingComp {
    int idx;
    int f;
    uint32_t root;
    int x0, x1, y0, y1, euler4;
    bool valid;
};
__device__ __forceinline__ void flush_component(PendingComp &pc, int lane, const CompTables &t)
{
    if (pc.valid && lane == 0 && pc.idx < t.max_det) {
        const size_t s = (size_t)pc.f * t.max_det + pc.idx;
        t.roots[s] = (int32_t)pc.root;
        t.bbox_tmp[s * 4 + 0] = pc.x0; t.bbox_tmp[s * 4 + 1] = pc.x1;
        t.bbox_tmp[s * 4 + 2] = pc.y0; t.bbox_tmp[s * 4 + 3] = pc.y1;
        t.euler_tmp[s] = pc.euler4;
    }
    pc.valid = false;
}

// One component C of the final mask (rows in lanes, columns of the slice that starts at frame column sx0;
// wave-uniform box in slice coordinates): label map, mask, tables.
__device__ __forceinline__ void window_component(const WindowOut &o, uint32_t c, const WindowBox &b, int sx0, int lane, int f,
                                                 const CompTables &t, PendingComp &pc)
{
    flush_component(pc, lane, t);
    int idx = 0;   // slot of the component in the frame's tables
    if (lane == 0) idx = atomicAdd(&t.nroots[(size_t)f * NR_STRIDE], 1);
    const int rx = __builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)c, b.y0));   // first pixel: top row, leftmost column
    const uint32_t root = (uint32_t)(o.wy0 + b.y0) * (uint32_t)o.W + (uint32_t)(sx0 + rx);
    const uint32_t row = (uint32_t)(o.wy0 + lane) * (uint32_t)o.W + (uint32_t)sx0;
    for (uint32_t rest = c; rest; rest &= rest - 1) {
        const uint32_t p = row + (uint32_t)__builtin_ctz(rest);
        o.labels[p] = root + 1u;
        if (o.mask) o.mask[p] = 255;
    }
    // bit quads: window (rows r, r+1; columns k, k+1) is counted at bit k of lane r.  C touches neither row 0 of
    // the window nor column 0 / 31 of the slice, so every 2 x 2 window that meets C has its top-left cell inside
    const uint32_t qa = c, qb = c >> 1, qc = wave_shl1(c), qd = qc >> 1;
    const uint32_t odd = qa ^ qb ^ qc ^ qd, pair = (qa & qb) | (qc & qd);
    const uint32_t diag = (qa & qd & ~(qb | qc)) | (qb & qc & ~(qa | qd));
    int q = __popc(odd & ~pair) - __popc(odd & pair) - 2 * __popc(diag);
    const int euler4 = WAVE_REDUCE(dpp_iadd, q);
    pc.idx = idx; pc.f = f; pc.root = root;
    pc.x0 = sx0 + b.x0; pc.x1 = sx0 + b.x1; pc.y0 = o.wy0 + b.y0; pc.y1 = o.wy0 + b.y1;
    pc.euler4 = euler4;
    pc.valid = true;
}

// The islands a window's core holds a pixel of (see above).  T / M: thresh / marker rows of the window.
// keep: the core's pixels (bit j = core column j; rows = lanes 16..47) that are somebody's to write in this call -- the small
// islands, whoever owns them (the owner writes the final mask's pixels and zeroes the others), and the listed pixels of the
// large ones (the residue passes write those).
__device__ __forceinline__ void window_islands(uint64_t T, uint64_t M, const WindowOut &o, int lane, int f, uint8_t *cf,
                                               const PixelList &pl, const CompTables &t, PendingComp &pc, uint32_t &keep)
{
    const uint64_t core_rows = (lane >= WIN_MARGIN && lane < WIN_MARGIN + WIN_CORE) ? 0x0000FFFFFFFF0000ull : 0ull;
    const uint64_t A = T | M;
    uint64_t todo = A & core_rows;
    while (true) {
        const unsigned long long rows = __ballot(todo != 0ull);
        if (!rows) break;
        const int sy = __builtin_ctzll(rows);
        const int sx = __builtin_ctzll(lane_mask(todo, sy));
        // the slice: window columns sx - 16 .. sx + 15 (sx is a core column, 16 .. 47: the slice lies inside the window)
        const int sh = sx - WIN_MARGIN;
        const uint32_t a32 = __builtin_amdgcn_alignbit((uint32_t)(A >> 32), (uint32_t)A, sh);
        const uint32_t F = flood8(lane == sy ? (1u << WIN_MARGIN) : 0u, a32);
        // LARGE (seen alike from every window): the island reaches the first or the last row of the window, column
        // sx - 16, a class pixel in column sx + 16 next to its own in column sx + 15, or its box exceeds 16 x 16
        const uint32_t beyond = (uint32_t)(A >> 32) >> sh & 1u;                     // class pixel in window column sx + 16
        const uint32_t spill = (F >> 31) & (beyond | wave_shr1(beyond) | wave_shl1(beyond));
        const WindowBox b = box_of(F);
        const bool large = __ballot(((F & 1u) | spill) != 0u || ((lane == 0 || lane == 63) && F != 0u)) != 0ull ||
                           b.x1 - b.x0 >= WIN_MARGIN || b.y1 - b.y0 >= WIN_MARGIN;
        if (large) {
            // residue: this core's pixels of the island (as far as the window shows it) go on the list, each its own
            // root for the union-find passes
            const uint64_t F64 = flood8(lane == sy ? (1ull << sx) : 0ull, A);
            todo &= ~F64;
            const uint64_t E = F64 & core_rows;
            keep |= (uint32_t)(E >> WIN_MARGIN);
            const uint32_t cnt = (uint32_t)__popcll(E);
            const unsigned long long below = (1ull << lane) - 1ull;
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int bit = 0; bit < 6; ++bit) {
                const unsigned long long m = __ballot((cnt >> bit) & 1u);
                before += (uint32_t)__popcll(m & below) << bit;
                total += (uint32_t)__popcll(m) << bit;
            }
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&pl.hdr->count[0], total);
            base = __builtin_amdgcn_readfirstlane(base);
            uint32_t slot = base + before;
            const uint32_t row = (uint32_t)(o.wy0 + lane) * (uint32_t)o.W + (uint32_t)o.wx0;   // (wraps above the frame: never used there)
            for (uint64_t rest = E; rest; rest &= rest - 1, ++slot) {
                const int k = __builtin_ctzll(rest);
                const uint32_t p = row + (uint32_t)k;
                if (slot < pl.cap) pl.idx[0][slot] = o.fbase + p;
                o.labels[p] = p + 1u;
                cf[p] = (uint8_t)(((T >> k) & 1ull) | (((M >> k) & 1ull) << 1));   // (drops stale flag bits of a caller-supplied map)
            }
            continue;
        }
        todo &= ~((uint64_t)F << sh);
        // (what of a small island lies in this core is its owner's to write, all of it: `keep`)
        keep |= (uint32_t)(((uint64_t)F << sh) >> WIN_MARGIN);      // slice column k = window column sh + k = core column sh + k - 16
        if (sy != b.y0 || __builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)F, b.y0)) != WIN_MARGIN) continue;   // first pixel in another core: not ours
        // the island is settled here
        const uint32_t thresh = __builtin_amdgcn_alignbit((uint32_t)(T >> 32), (uint32_t)T, sh) & F;
        uint32_t R = __builtin_amdgcn_alignbit((uint32_t)(M >> 32), (uint32_t)M, sh) & F;
        bool whole = false;   // R is the whole island (the usual end: one step from the markers)
        while (true) {
            const uint32_t n = R | (grow4(R) & thresh);
            const bool changed = __ballot(n != R) != 0ull;
            R = n;
            whole = __ballot(R != F) == 0ull;
            if (!changed || whole) break;
        }
        const int sx0 = o.wx0 + sh;   // frame column of the slice's column 0
        if (whole) {
            window_component(o, F, b, sx0, lane, f, t, pc);
        } else {
            if (WIN_CLEARS) {
                // the island's pixels outside the final mask: zero, whatever the previous call left there (the cores' waves
                // leave a small island to its owner)
                const uint32_t row = (uint32_t)(o.wy0 + lane) * (uint32_t)o.W + (uint32_t)sx0;
                for (uint32_t rest = F & ~R; rest; rest &= rest - 1) {
                    const uint32_t p = row + (uint32_t)__builtin_ctz(rest);
                    o.labels[p] = 0u;
                    if (o.mask) o.mask[p] = 0;
                }
            }
            while (true) {
                const unsigned long long rr = __ballot(R != 0u);
                if (!rr) break;
                const int cy0 = __builtin_ctzll(rr);
                const int cx0 = __builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)R, cy0));
                const uint32_t C = flood8(lane == cy0 ? (1u << cx0) : 0u, R);
                R &= ~C;
                window_component(o, C, box_of(C), sx0, lane, f, t, pc);
            }
        }
    }
}

// 16 bytes of row y from column xs on, zeros where the frame is not (any alignment: W need not be a multiple of 4)
__device__ __forceinline__ uint4 load16_clipped(const uint8_t *cf, int y, int xs, int H, int W)
{
    uint4 vj = make_uint4(0, 0, 0, 0);
    if (y >= 0 && y < H && xs < W && xs + 16 > 0) {
        const uint8_t *src = cf + (size_t)y * W + xs;
        if (xs >= 0 && xs + 16 <= W) {
            __builtin_memcpy(&vj, src, 16);
        } else if (xs >= 0 && W >= 16) {
            // the chunk sticks out of the row on the right: the row's last 16 bytes, shifted down.  (A loop
            // of guarded byte loads here was a chain of round trips in every item of the last column group,
            // and those items were the kernel's duration.)
            uint4 l;
            __builtin_memcpy(&l, src - (xs + 16 - W), 16);
            const int drop = xs + 16 - W, a = drop >> 2, b = drop & 3;   // 1..15 bytes
            const uint32_t e0 = a == 0 ? l.x : a == 1 ? l.y : a == 2 ? l.z : l.w;
            const uint32_t e1 = a == 0 ? l.y : a == 1 ? l.z : a == 2 ? l.w : 0u;
            const uint32_t e2 = a == 0 ? l.z : a == 1 ? l.w : 0u;
            const uint32_t e3 = a == 0 ? l.w : 0u;
            vj = make_uint4(__builtin_amdgcn_alignbyte(e1, e0, b), __builtin_amdgcn_alignbyte(e2, e1, b),
                            __builtin_amdgcn_alignbyte(e3, e2, b), __builtin_amdgcn_alignbyte(0u, e3, b));
        } else {   // frames narrower than 16 pixels
            uint32_t d[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; ++k)
                if (xs + k >= 0 && xs + k < W) d[k >> 2] |= (uint32_t)src[k] << (8 * (k & 3));
            vj = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
    return vj;
}

// Work item of a wave: WIN_GROUP cores side by side = 64 rows x 160 columns of class bytes, turned into row masks
// through the wave's own slice of LDS (adjacent lanes load adjacent 16-byte chunks of a row; lane = row reads the
// words back): no block barrier, the waves of a block do not wait for each other.
__global__ __launch_bounds__(256) void k_windows(uint8_t *__restrict__ cls, uint32_t *__restrict__ labels,
                                                 uint8_t *__restrict__ mask, Geo g, int batch, PixelList pl, CompTables t,
                                                 uint32_t *__restrict__ oldbits)
{
    DET_RING(4);
    // (k_clear's word: the header vouched for these buffers, so the words this kernel left in `oldbits` in the previous call
    // say where the label map and the mask may be non-zero; otherwise k_clear has zeroed everything)
    const bool use_old = WIN_CLEARS && pl.hdr->old_bits_valid != 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // (k_clear, the previous launch, was the last reader of these)
        pl.hdr->magic = 0;
        pl.hdr->dense = 0;
        pl.hdr->n_big = 0;
    }
    constexpr int CHUNKS = 64 * WIN_HALVES, ROUNDS = CHUNKS / 64;
    __shared__ __attribute__((aligned(8))) uint16_t s_bits[4][2][CHUNKS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = g.H, W = g.W;
    constexpr int SPAN = WIN_GROUP * WIN_CORE;   // 128
    const int groups_x = (W + SPAN - 1) / SPAN, rows_y = (H + WIN_CORE - 1) / WIN_CORE;
    const long long per_frame = (long long)groups_x * rows_y, items = per_frame * batch;
    uint16_t *st = s_bits[wave][0], *sm = s_bits[wave][1];
    // Workgroups b and b + 8 run on the same XCD (as in k_threshold_strip): every XCD works through whole frames
    // -- the frames it thresholded, if the batch is a multiple of 8 -- or else through one contiguous eighth of the
    // items, so that the rows an item shares with its neighbours (each class byte is read by 2.5 items) come out
    // of the XCD's own L2.  (Items dealt round-robin over the chip: 93 us per batch for the loads alone.)
    long long first = (long long)blockIdx.x * 4 + wave, step = (long long)gridDim.x * 4, count = items, offset = 0;
    int f_mul = 1, f_add = 0;
    if ((gridDim.x & 7u) == 0) {
        first = (long long)(blockIdx.x >> 3) * 4 + wave; step = (long long)(gridDim.x >> 3) * 4;
        if ((batch & 7) == 0) { count = (long long)(batch >> 3) * per_frame; f_mul = 8; f_add = (int)(blockIdx.x & 7u); }
        else {
            const long long share = (items + 7) / 8;
            offset = share * (blockIdx.x & 7u);
            count = max(0ll, min(share, items - offset));
        }
    }
    // (static dealing: handing further items out by a ticket counter per group of blocks, requested an item ahead,
    // was slower -- 61 against 55 us -- and 230 us with the eight counters on one cache line)
    PendingComp pc;
    pc.valid = false;
    for (long long j = first; j < count; j += step) {
        const long long it = offset + j;
        const int f = (int)(it / per_frame) * f_mul + f_add, rem = (int)(it % per_frame);
        const int Y0 = rem / groups_x * WIN_CORE, X0 = rem % groups_x * SPAN;
        uint8_t *cf = cls + (size_t)f * g.HW;
        // what the previous call may have left in the label map and the mask inside this item's cores: one word per core and row
        // (rows in lanes 16..47, the cores' rows of the window), written by this kernel in that call (`keep` below)
        static_assert(WIN_GROUP % 4 == 0, "a row's words are loaded four at a time");
        uint32_t oldw[WIN_GROUP];
#pragma unroll
        for (int k = 0; k < WIN_GROUP; ++k) oldw[k] = 0u;
        uint32_t *item_bits = oldbits + ((size_t)f * per_frame + rem) * (WIN_CORE * WIN_GROUP) + (size_t)((lane - WIN_MARGIN) & (WIN_CORE - 1)) * WIN_GROUP;   // (used by lanes 16..47: the core's rows)
        if (use_old && lane >= WIN_MARGIN && lane < WIN_MARGIN + WIN_CORE) {
#pragma unroll
            for (int k = 0; k < WIN_GROUP / 4; ++k) {
                const uint4 q = reinterpret_cast<const uint4 *>(item_bits)[k];
                oldw[4 * k] = q.x; oldw[4 * k + 1] = q.y; oldw[4 * k + 2] = q.z; oldw[4 * k + 3] = q.w;
            }
        }
        uint32_t old_any = 0u;      // bit k: the previous call left something in core k (most cores of most items: nothing)
#pragma unroll
        for (int k = 0; k < WIN_GROUP; ++k) old_any |= (__ballot(oldw[k] != 0u) != 0ull ? 1u : 0u) << k;
        constexpr int PARTS = WIN_GROUP == 4 ? 2 : 3, PER_PART = ROUNDS / PARTS;
        static_assert(ROUNDS % PARTS == 0, "equal part-batches of loads");
#pragma unroll 1
        for (int half = 0; half < PARTS; ++half) {   // (all ten loads in flight at once cost 40 VGPRs and a wave per SIMD)
        uint4 v[PER_PART];
#pragma unroll
        for (int jj = 0; jj < PER_PART; ++jj) {
            const int j = jj + half * PER_PART;
            const int q = lane + 64 * j, r = q / WIN_HALVES, c = q - r * WIN_HALVES;
            const int y = Y0 - WIN_MARGIN + r, xs = X0 - WIN_MARGIN + 16 * c;
            v[jj] = load16_clipped(cf, y, xs, H, W);
        }
#pragma unroll
        for (int jj = 0; jj < PER_PART; ++jj) {
            const int j = jj + half * PER_PART;
            const uint4 &vj = v[jj];
            const uint32_t m = 0x01010101u, lo = 0x08040201u, hi = 0x80402010u;   // v_dot4_u32_u8: 4 class bits -> a nibble
            const uint32_t t0 = __builtin_amdgcn_udot4(vj.x & m, lo, __builtin_amdgcn_udot4(vj.y & m, hi, 0u, false), false);
            const uint32_t t1 = __builtin_amdgcn_udot4(vj.z & m, lo, __builtin_amdgcn_udot4(vj.w & m, hi, 0u, false), false);
            const uint32_t m0 = __builtin_amdgcn_udot4((vj.x >> 1) & m, lo, __builtin_amdgcn_udot4((vj.y >> 1) & m, hi, 0u, false), false);
            const uint32_t m1 = __builtin_amdgcn_udot4((vj.z >> 1) & m, lo, __builtin_amdgcn_udot4((vj.w >> 1) & m, hi, 0u, false), false);
            st[lane + 64 * j] = (uint16_t)(t0 | (t1 << 8));
            sm[lane + 64 * j] = (uint16_t)(m0 | (m1 << 8));
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // this wave's LDS writes before its LDS reads
        __builtin_amdgcn_wave_barrier();
        uint32_t tw[WIN_GROUP == 4 ? WIN_GROUP + 1 : 1], mw[WIN_GROUP == 4 ? WIN_GROUP + 1 : 1];
        if constexpr (WIN_GROUP == 4) {
#pragma unroll
            for (int k = 0; k <= WIN_GROUP; ++k) {
                tw[k] = reinterpret_cast<const uint32_t *>(st)[lane * (WIN_GROUP + 1) + k];
                mw[k] = reinterpret_cast<const uint32_t *>(sm)[lane * (WIN_GROUP + 1) + k];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // ... and the reads before the next item's writes
            __builtin_amdgcn_wave_barrier();
        }
        WindowOut o;
        o.cls = cf; o.labels = labels + (size_t)f * g.HW; o.mask = mask ? mask + (size_t)f * g.HW : nullptr;
        o.W = W; o.H = H; o.wy0 = Y0 - WIN_MARGIN;
        o.fbase = (uint32_t)((size_t)f * g.HW);
        static_assert(WIN_GROUP == 4 || WIN_GROUP == 8, "four cores (words in registers, picked by selects) or eight (words read from LDS core by core)");
#pragma unroll 1
        for (int cx = 0; cx < WIN_GROUP; ++cx) {   // (one copy of the window code: the words are picked by selects)
            if (X0 + cx * WIN_CORE >= W) break;   // core outside the frame
            uint32_t t_lo, t_hi, m_lo, m_hi;
            if constexpr (WIN_GROUP == 4) {
                t_lo = cx == 0 ? tw[0] : cx == 1 ? tw[1] : cx == 2 ? tw[2] : tw[3];
                t_hi = cx == 0 ? tw[1] : cx == 1 ? tw[2] : cx == 2 ? tw[3] : tw[4];
                m_lo = cx == 0 ? mw[0] : cx == 1 ? mw[1] : cx == 2 ? mw[2] : mw[3];
                m_hi = cx == 0 ? mw[1] : cx == 1 ? mw[2] : cx == 2 ? mw[3] : mw[4];
            } else {
                const uint32_t *tp = reinterpret_cast<const uint32_t *>(st) + lane * (WIN_GROUP + 1) + cx;
                const uint32_t *mp = reinterpret_cast<const uint32_t *>(sm) + lane * (WIN_GROUP + 1) + cx;
                t_lo = tp[0]; t_hi = tp[1]; m_lo = mp[0]; m_hi = mp[1];
            }
            const uint64_t T = (uint64_t)t_lo | ((uint64_t)t_hi << 32), M = (uint64_t)m_lo | ((uint64_t)m_hi << 32);
            const uint64_t core_rows = (lane >= WIN_MARGIN && lane < WIN_MARGIN + WIN_CORE) ? 0x0000FFFFFFFF0000ull : 0ull;
            uint32_t keep = 0u;
            const bool has_class = __ballot(((T | M) & core_rows) != 0ull) != 0ull;
            // nothing here now and nothing before: the core's words stay zero (a call that does not go by the words writes them all:
            // what it finds there may be anything)
            if (use_old && !has_class && !((old_any >> cx) & 1u)) continue;
            if (has_class) {
                o.wx0 = X0 + cx * WIN_CORE - WIN_MARGIN;
#ifdef WIN_DBG_STAGE_ONLY
                if (T == 0x123456789ull) o.labels[0] = 1;
#else
                window_islands(T, M, o, lane, f, cf, pl, t, pc, keep);
#endif
            }
            if (WIN_CLEARS) {
                if (lane >= WIN_MARGIN && lane < WIN_MARGIN + WIN_CORE) item_bits[cx] = keep;   // for the next call
                // the previous call's pixels of this core that nobody writes in this one: zero in the label map and in the mask
                uint32_t old_cx = oldw[0];
#pragma unroll
                for (int k = 1; k < WIN_GROUP; ++k) old_cx = cx == k ? oldw[k] : old_cx;
                const uint32_t z = old_cx & ~keep;
                if (__ballot(z != 0u) != 0ull) {
                    const uint32_t row = (uint32_t)(Y0 + lane - WIN_MARGIN) * (uint32_t)W + (uint32_t)(X0 + cx * WIN_CORE);   // (lanes 16..47 only: z is zero elsewhere)
                    for (uint32_t rest = z; rest; rest &= rest - 1) {
                        const uint32_t p = row + (uint32_t)__builtin_ctz(rest);
                        o.labels[p] = 0u;
                        if (o.mask) o.mask[p] = 0;
                    }
                }
            }
        }
        if constexpr (WIN_GROUP != 4) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the cores' reads of the words before the next item's writes
            __builtin_amdgcn_wave_barrier();
        }
    }
    flush_component(pc, lane, t);
}

// Pass B: 4-connected components of the `thresh` bit (the mask of binary_propagation).
__device__ __forceinline__ void pass_union4(const uint8_t *__restrict__ cls, uint32_t *labels, const Geo &g, const ResidueIter &it)
{
    FOR_LISTED_PIXELS(it, g, flat) {
        if (flat >= g.total || !(cls[flat] & 1u)) continue;
        uint32_t f, p; int y, x;
        locate(g, flat, f, p, y, x);
        uint32_t *L = labels + (size_t)f * g.HW;
        const bool w = x > 0 && (cls[flat - 1] & 1u), n = y > 0 && (cls[flat - g.W] & 1u);
        if (w) unite(L, p, p - 1);
        // with W, NW and N all set, W-NW and NW-N are united by W's and N's own threads
        if (n && !(w && (cls[flat - g.W - 1] & 1u))) unite(L, p, p - (uint32_t)g.W);
    }
}

__device__ __forceinline__ void set_flag(uint8_t *cls, size_t flat)
{
    if (cls[flat] & 4u) return;   // (a stale 0 only costs a redundant atomic)
    uint32_t *w = reinterpret_cast<uint32_t *>(cls) + (flat >> 2);
    atomicOr(w, 4u << (8 * (flat & 3)));
}

// Pass C: flag (bit2 on the root's class byte) every thresh-component that holds a marker pixel
// or touches (4-neighbourhood) a marker pixel lying outside the mask.
__device__ __forceinline__ void pass_flag(uint8_t *cls, uint32_t *labels, const Geo &g, const ResidueIter &it)
{
    FOR_LISTED_PIXELS(it, g, flat) {
        {
            if (flat >= g.total) continue;
            uint32_t b = cls[flat];
            if (!(b & 2u)) continue;
            uint32_t f, p; int y, x;
            locate(g, flat, f, p, y, x);
            uint32_t *L = labels + (size_t)f * g.HW;
            size_t fbase = (size_t)f * g.HW;
            if (b & 1u) {
                const uint32_t r = find_root_cached(L, p);
                set_flag(cls, fbase + r);
                // no union runs in this pass: point the pixel straight at its root, so that the finds of
                // the next two passes are one hop (a stale reader still sees an ancestor)
                if (r != p && r != DEAD) L[p] = r + 1u;
            } else {
                if (x > 0 && (cls[flat - 1] & 1u)) set_flag(cls, fbase + find_root_cached(L, p - 1));
                if (x < g.W - 1 && (cls[flat + 1] & 1u)) set_flag(cls, fbase + find_root_cached(L, p + 1));
                if (y > 0 && (cls[flat - g.W] & 1u)) set_flag(cls, fbase + find_root_cached(L, p - g.W));
                if (y < g.H - 1 && (cls[flat + g.W] & 1u)) set_flag(cls, fbase + find_root_cached(L, p + g.W));
            }
        }
    }
}

// Membership in the final mask R = binary_propagation(markers, mask=thresh).
__device__ __forceinline__ bool in_result(const uint8_t *cls_frame, const uint32_t *L, uint32_t p, uint32_t b)
{
    if (b & 2u) return true;
    if (!(b & 1u)) return false;
    uint32_t r = find_root(L, p);
    return r != DEAD && (cls_frame[r] & 6u) != 0;
}

// Pass D: 8-connected components of R (what cv2.findContours traces).
__device__ __forceinline__ void pass_union8(const uint8_t *__restrict__ cls, uint32_t *labels, const Geo &g, const ResidueIter &it)
{
    FOR_LISTED_PIXELS(it, g, flat) {
    {
        if (flat >= g.total) continue;
        uint32_t b = cls[flat];
        if (!(b & 3u)) continue;
        uint32_t f, p; int y, x;
        locate(g, flat, f, p, y, x);
        uint32_t *L = labels + (size_t)f * g.HW;
        const uint8_t *cf = cls + (size_t)f * g.HW;
        if (!in_result(cf, L, p, b)) continue;
        const int W = g.W;
        // W, NW, N, NE
        if (x > 0) { uint32_t q = p - 1, bq = cf[q]; if ((bq & 3u) && !((b & bq) & 1u) && in_result(cf, L, q, bq)) unite(L, p, q); }
        if (y > 0) {
            uint32_t q = p - W, bq = cf[q];
            if ((bq & 3u) && !((b & bq) & 1u) && in_result(cf, L, q, bq)) unite(L, p, q);
            // a diagonal pair of thresh pixels with a thresh pixel on a shared side is already one
            // 4-component of pass B: nothing to unite (the usual case inside a blob)
            const uint32_t bn = bq;
            if (x > 0) {
                q = p - W - 1; bq = cf[q];
                if ((bq & 3u) && !((b & bq & 1u) && ((bn | cf[p - 1]) & 1u)) && in_result(cf, L, q, bq)) unite(L, p, q);
            }
            if (x < W - 1) {
                q = p - W + 1; bq = cf[q];
                if ((bq & 3u) && !((b & bq & 1u) && ((bn | cf[p + 1]) & 1u)) && in_result(cf, L, q, bq)) unite(L, p, q);
            }
        }
    }
    }
}

// Pass E: final labels (root + 1), final mask (cleared by k_clear beforehand), roots per frame.
__device__ __forceinline__ void pass_flatten(const uint8_t *__restrict__ cls, uint32_t *labels, uint8_t *__restrict__ mask,
                                             const Geo &g, const ResidueIter &it, const CompTables &t)
{
    FOR_LISTED_PIXELS(it, g, flat) {
        if (flat >= g.total) continue;
        uint32_t b = cls[flat];
        if (!(b & 3u)) continue;
        uint32_t f = (uint32_t)(flat / g.HW);
        uint32_t p = (uint32_t)(flat - (size_t)f * g.HW);
        uint32_t *L = labels + (size_t)f * g.HW;
        const uint8_t *cf = cls + (size_t)f * g.HW;
        uint32_t r = find_root_cached(L, p);
        bool inr = (r != DEAD) && ((b & 2u) || (cf[r] & 6u));
        if (inr) {
            if (r != p) L[p] = r + 1;   // plain store: write-through agent-scope stores cost a fabric write each
            if (mask) mask[flat] = 255;
            if (r == p) {
                int idx = atomicAdd(&t.nroots[(size_t)f * NR_STRIDE], 1);
                if (idx < t.max_det) {   // box and Euler number: k_bbox_euler, once the component has its rank
                    const size_t s = (size_t)f * t.max_det + idx;
                    t.roots[s] = (int32_t)p;
                    t.bbox_tmp[s * 4 + 0] = g.W; t.bbox_tmp[s * 4 + 1] = -1; t.bbox_tmp[s * 4 + 2] = g.H; t.bbox_tmp[s * 4 + 3] = -1;
                    t.euler_tmp[s] = 0;
                }
            }
        } else {
            L[p] = 0u;
            if (mask) mask[flat] = 0;   // (k_windows left the listed pixels to this pass: WIN_CLEARS)
        }
    }
}

// ------------------------------------------------------------------------------------------
// Ordering (reverse raster order of first pixels), bounding boxes, Euler numbers
// ------------------------------------------------------------------------------------------

// Rank of every root among its frame's roots (descending pixel index = findContours order); four
// lanes share a root and split the comparisons.  A component takes its box and Euler number along to its rank
// (k_windows / pass_bbox_euler filed them under the slot the root was appended at), gets its root's label back
// (pass_tag_roots), and is queued for nested_component if it has holes.
constexpr int RANK_THREADS = 1024;
constexpr int RANK_BANDS = 2048;      // bands of image rows (LDS histogram)
constexpr int RANK_BUCKET = 8192;     // roots of a frame the banded ranking holds in LDS
constexpr int HOLED_CAP = 4096;

__global__ __launch_bounds__(RANK_THREADS) void k_rank(CompTables t, uint32_t *labels, uint32_t HW, int W, int H, int32_t *status,
                                                       WsHeader *hdr, int32_t *n_holed, int2 *holed)
{
    DET_RING(9);
    const int f = blockIdx.x;
    // the test hook of the residue kernels (include/ysmr_hip.h) is good for one call: cleared here, behind the launch that read it
    if (f == 0 && blockIdx.y == 0 && threadIdx.x == 0 && hdr->fault == YSMR_WS_FAULT_RESIDUE_STALL) hdr->fault = 0u;
    int n = t.nroots[(size_t)f * NR_STRIDE];
    if (n > t.max_det) {
        if (threadIdx.x == 0 && blockIdx.y == 0) {
            atomicOr(&status[f], YSMR_DET_OVERFLOW);
            hdr->dense = 1u;   // pixels of the components beyond max_det are in no table: the next call clears everything
        }
        n = t.max_det;
    }
    if (threadIdx.x == 0 && blockIdx.y == 0) atomicMax(t.max_roots, n);
    // (the grid is sized for max_det: a workgroup whose roots do not exist leaves before the histogram -- six of the eight per
    // frame on the bench stream; 3 us of the chain)
    if ((int)blockIdx.y * (RANK_THREADS / 4) >= n) return;
    const int32_t *roots = t.roots + (size_t)f * t.max_det;
    // rank of a root = roots in the bands of image rows below its own + roots of its own band with a larger index:
    // a histogram over the bands, a suffix sum, the roots bucketed by band in LDS, and each root compared with its
    // own bucket only (all pairs were 25 M comparisons per 4K frame: 68 us per batch)
    __shared__ int32_t s_hist[RANK_BANDS], s_start[RANK_BANDS], s_bucket[RANK_BUCKET], s_wave[RANK_THREADS / 64];
    const bool banded = n <= RANK_BUCKET;
    const int band_h = (H + RANK_BANDS - 1) / RANK_BANDS;
    if (banded) {
        for (int b = threadIdx.x; b < RANK_BANDS; b += RANK_THREADS) s_hist[b] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += RANK_THREADS) atomicAdd(&s_hist[(roots[j] / W) / band_h], 1);
        __syncthreads();
        // s_start[b] = roots in bands > b: thread k owns bands RANK_BANDS-1-2k and RANK_BANDS-2-2k (scan from the bottom)
        static_assert(RANK_BANDS == 2 * RANK_THREADS, "two bands per thread");
        const int b0 = RANK_BANDS - 1 - 2 * (int)threadIdx.x, h0 = s_hist[b0], h1 = s_hist[b0 - 1];
        int incl = h0 + h1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if ((int)(threadIdx.x & 63) >= d) incl += o;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
        __syncthreads();
        int before = incl - (h0 + h1);
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += s_wave[w];
        s_start[b0] = before;
        s_start[b0 - 1] = before + h0;
        s_hist[b0] = 0;          // reused as the fill count of the bucket
        s_hist[b0 - 1] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += RANK_THREADS) {
            const int32_t r = roots[j];
            const int b = (r / W) / band_h;
            s_bucket[s_start[b] + atomicAdd(&s_hist[b], 1)] = r;
        }
        __syncthreads();
    }
    int32_t *tile = s_bucket;   // (the all-pairs path for tables beyond RANK_BUCKET roots stages 1024 roots at a time)
    const int sub = threadIdx.x & 3;
    for (int i0 = (int)blockIdx.y * (RANK_THREADS / 4); i0 < n; i0 += (int)gridDim.y * (RANK_THREADS / 4)) {
        const int i = i0 + (threadIdx.x >> 2);
        const int32_t mine = i < n ? roots[i] : -1;
        int rank = 0;
        if (banded) {
            if (i < n) {
                const int b = (mine / W) / band_h, first = s_start[b], cnt = s_hist[b];
                for (int j = sub; j < cnt; j += 4) rank += s_bucket[first + j] > mine;
                if (sub == 0) rank += first;
            }
        } else {
            for (int j0 = 0; j0 < n; j0 += 1024) {
                __syncthreads();
                for (int j = threadIdx.x; j < 1024 && j0 + j < n; j += RANK_THREADS) tile[j] = roots[j0 + j];
                __syncthreads();
                const int m = min(1024, n - j0);
                for (int j = sub; j < m; j += 4) rank += tile[j] > mine;
            }
        }
        rank += __shfl_xor(rank, 1);
        rank += __shfl_xor(rank, 2);
        if (i < n && sub == 0) {
            size_t o = (size_t)f * t.max_det + rank;
            t.order[o] = mine;
            const size_t from = (size_t)f * t.max_det + i;
            int bb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t.bbox[o * 4 + q] = bb[q] = t.bbox_tmp[from * 4 + q];
            const int e4 = t.euler_tmp[from];
            t.euler4[o] = e4;
            t.nested[o] = 0;
            labels[(size_t)f * HW + (uint32_t)mine] = (uint32_t)mine + 1u;
            // components whose Euler number is not 1 have holes and may enclose other components (RETR_EXTERNAL
            // skips those): queued for nested_component
            if (e4 != 4) {
                const int at = atomicAdd(n_holed, 1);
                if (at < HOLED_CAP) holed[at] = make_int2(f, rank);
                else atomicOr(&status[f], YSMR_DET_ARENA);
            }
            // large boxes are named in the header: the next call's k_clear has every block work on them
            if ((bb[1] - bb[0] + 1) * (bb[3] - bb[2] + 1) >= WS_BIG_AREA) {
                const uint32_t at = atomicAdd(&hdr->n_big, 1u);
                if (at < (uint32_t)WS_BIG) { hdr->big[at][0] = f; hdr->big[at][1] = rank; }
            }
        }
    }
}

// index of `root` in the descending list order[0..n)
__device__ __forceinline__ int find_rank(const int32_t *order, int n, int32_t root)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (order[mid] > root) lo = mid + 1; else hi = mid;
    }
    return (lo < n && order[lo] == root) ? lo : -1;
}

// While pass_bbox_euler runs, the label of a residue component's root holds SLOT_TAG | (slot the root was appended
// at) instead of root + 1, so that every pixel finds its component's table row with one dependent load; k_rank
// puts root + 1 back.
constexpr uint32_t SLOT_TAG = 0x80000000u;

// Per final-mask pixel: bounding box of its component; bit-quad counts for the Euler number
// (E8 = (Q1 - Q3 - 2 QD) / 4 over all 2x2 windows, each window counted by its first set pixel).
__device__ __forceinline__ void pass_bbox_euler(const uint8_t *__restrict__ cls, const uint32_t *labels, const Geo &g,
                                                const ResidueIter &it, const CompTables &t)
{
    // The list keeps the pixels of a 16-pixel chunk on adjacent lanes, so a horizontal run of a
    // component sits on consecutive lanes: its lanes pool their y-extent candidates and quad counts
    // with ballots, and only the run's first lane issues those atomics.  (The loop is kept
    // wave-uniform for the ballots: lanes past the end of the list carry valid = false.)
    const uint32_t *idx = it.idx;
    const bool dense = it.dense;
    const size_t ln = it.n;
    const int lane = threadIdx.x & 63;
    const size_t stride = it.step;
    for (size_t base = it.first - (size_t)lane; base < ln; base += stride) {
        const size_t li = base + lane;
        bool valid = li < ln;
        size_t flat = 0;
        if (valid) flat = dense ? it.dense_base + li : (size_t)idx[li];
        if (valid && it.only_frame >= 0) valid = flat / g.HW == (size_t)it.only_frame;
        uint32_t lab_i = 0;
        if (valid) valid = (cls[flat] & 3u) != 0;
        if (valid) { lab_i = labels[flat]; valid = lab_i != 0; }
        uint32_t f = 0, p = 0; int y = 0, x = 0;
        if (valid) locate(g, flat, f, p, y, x);
        const uint32_t *L = labels + (size_t)f * g.HW;
        if (valid && !(lab_i & SLOT_TAG)) lab_i = L[lab_i - 1];   // the root's label carries its slot (pass_tag_roots)
        valid = valid && (lab_i & SLOT_TAG);   // else: component beyond max_det (k_rank flags the overflow)
        const uint32_t o = valid ? f * (uint32_t)t.max_det + (lab_i & ~SLOT_TAG) : 0xFFFFFFFFu;
        const int W = g.W, H = g.H;
        auto at = [&](int yy, int xx) -> int {
            return (valid && yy >= 0 && yy < H && xx >= 0 && xx < W && L[(size_t)yy * W + xx] != 0) ? 1 : 0;
        };
        const int nw = at(y - 1, x - 1), nn = at(y - 1, x), ne = at(y - 1, x + 1);
        const int ww = at(y, x - 1), ee = at(y, x + 1);
        const int sw = at(y + 1, x - 1), ss = at(y + 1, x), se = at(y + 1, x + 1);
        // the four 2x2 windows containing (y,x); window order of pixels: TL, TR, BL, BR
        int q = 0;
        auto quad = [&](int tl, int tr, int bl, int br) {
            int c4 = tl + tr + bl + br;
            if (c4 == 1) q += 1;
            else if (c4 == 3) q -= 1;
            else if (c4 == 2 && ((tl && br) || (tr && bl))) q -= 2;
        };
        quad(1, ee, ss, se);                        // (y,x) is TL: always first
        if (!ww) quad(ww, 1, sw, ss);               // TR: first iff TL clear
        if (!nn && !ne) quad(nn, ne, 1, ee);        // BL: first iff TL, TR clear
        if (!nw && !nn && !ww) quad(nw, nn, ww, 1); // BR: first iff all others clear
        // runs: lane l continues lane l-1's run if that lane holds the west neighbour of the same component
        const uint32_t prev_flat = wave_shr1((uint32_t)flat), prev_o = wave_shr1(o);
        const bool head = valid && !(lane > 0 && x > 0 && prev_o == o && prev_flat + 1u == (uint32_t)flat);
        const unsigned long long heads = __ballot(head);
        const unsigned long long upto = (2ull << lane) - 1ull;                  // lanes 0..lane
        const unsigned long long later = heads & ~upto;
        const int end = later ? __ffsll((long long)later) - 2 : 63;            // last lane before the next run
        const unsigned long long run = ((2ull << end) - 1ull) & ~((1ull << lane) - 1ull);   // valid for head lanes
        const unsigned long long top = __ballot(valid && !nn), bottom = __ballot(valid && !ss);
        const uint32_t qb = valid ? (uint32_t)(q + 8) : 0u;                     // q in [-8, 4] -> 4 bit planes
        const unsigned long long members = __ballot(valid);
        int run_q = -8 * (int)__popcll(members & run);
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) run_q += (int)__popcll(__ballot((qb >> bit) & 1u) & run) << bit;
        if (valid) {   // bbox: only extreme candidates issue atomics
            if (!ww) atomicMin(&t.bbox_tmp[(size_t)o * 4 + 0], x);
            if (!ee) atomicMax(&t.bbox_tmp[(size_t)o * 4 + 1], x);
        }
        if (head) {
            if (top & run) atomicMin(&t.bbox_tmp[(size_t)o * 4 + 2], y);
            if (bottom & run) atomicMax(&t.bbox_tmp[(size_t)o * 4 + 3], y);
            if (run_q) atomicAdd(&t.euler_tmp[o], run_q);
        }
    }
}

// The union-find passes over the residue as ONE launch: usually there is no residue (every block returns at
// once; five near-empty launches cost 25 us per batch), otherwise the passes are separated by a software grid
// barrier (all RESIDUE_BLOCKS blocks are resident: 128 x 256 threads, no LDS to speak of).
constexpr int RESIDUE_BLOCKS = 128;

__device__ __forceinline__ bool grid_barrier(uint32_t *counter, uint32_t target, uint32_t spin_limit = 20000000u)
{
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        uint32_t spins = 0;
        int ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > spin_limit) { ok = 0; break; }   // (a block that cannot become resident: give up rather than hang)
        }
        __threadfence();
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

__device__ __forceinline__ void pass_tag_roots(uint32_t *labels, const Geo &g, const CompTables &t, int batch, unsigned nb)
{
    // (roots appended by k_windows get the tag too -- none of their pixels is visited by pass_bbox_euler -- and
    // k_rank writes root + 1 back over every one)
    const size_t slots = (size_t)batch * t.max_det;
    for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < slots; s += (size_t)nb * 256) {
        const int f = (int)(s / t.max_det), k = (int)(s - (size_t)f * t.max_det);
        if (k < min(t.nroots[(size_t)f * NR_STRIDE], t.max_det))
            labels[(size_t)f * g.HW + (uint32_t)t.roots[s]] = SLOT_TAG | (uint32_t)k;
    }
}

__global__ __launch_bounds__(256) void k_residue(uint8_t *cls, uint32_t *labels, uint8_t *mask, Geo g, int batch, PixelList pl,
                                                 CompTables t, uint32_t *barrier, int32_t *status)
{
    DET_RING(5);
    const uint32_t listed = pl.hdr->count[0];   // (written by k_windows, the previous launch: every block sees the same)
    // (fault injection by the library's own test, through the workspace it owns: include/ysmr_hip.h)
    const bool fault = pl.hdr->fault == YSMR_WS_FAULT_RESIDUE_STALL;
    if (listed == 0u && !fault) return;
    // (fault injection: block 0 never arrives at the first barrier, as a workgroup that found no compute unit would not)
    if (fault && blockIdx.x == 0) return;
    const uint32_t spin_limit = fault ? 20000u : 20000000u;
    // (a barrier among fewer blocks is cheaper -- 128 blocks: 2.7 us, 16: 1.2 us -- but letting only one block per 512
    // listed pixels take part made the launch slower, 21 against 16 us: the passes are chains of round trips and
    // want the threads)
    const bool dense = listed > pl.cap;
    const unsigned nb = gridDim.x;
    bool ok = true;
    uint32_t phase = 0;
    if (dense) {
        // more residue than the list holds: the passes walk every pixel, so everything k_windows settled is done
        // again here -- its components are dropped, every class pixel becomes its own root again.  (Its final mask
        // stays: the same pixels get the same 255.)
        for (size_t flat = (size_t)blockIdx.x * 256 + threadIdx.x; flat < g.total; flat += (size_t)nb * 256) {
            const uint32_t b = cls[flat];
            if (b & 3u) {
                labels[flat] = (uint32_t)(flat % g.HW) + 1u;
                if (b & ~3u) cls[flat] = (uint8_t)(b & 3u);
            }
        }
        if (blockIdx.x == 0)
            for (int f = threadIdx.x; f < batch; f += 256) t.nroots[(size_t)f * NR_STRIDE] = 0;
        ok = grid_barrier(barrier, ++phase * nb, spin_limit);
    }
    const ResidueIter it = residue_iter_grid(pl, g, nb);
    if (ok) pass_union4(cls, labels, g, it);
    ok = ok && grid_barrier(barrier, ++phase * nb, spin_limit);
    if (ok) pass_flag(cls, labels, g, it);
    ok = ok && grid_barrier(barrier, ++phase * nb, spin_limit);
    if (ok) pass_union8(cls, labels, g, it);
    ok = ok && grid_barrier(barrier, ++phase * nb, spin_limit);
    if (ok) pass_flatten(cls, labels, mask, g, it, t);
    ok = ok && grid_barrier(barrier, ++phase * nb, spin_limit);
    if (ok) pass_tag_roots(labels, g, t, batch, nb);
    ok = ok && grid_barrier(barrier, ++phase * nb, spin_limit);
    if (ok) pass_bbox_euler(cls, labels, g, it, t);
    if (!ok && threadIdx.x == 0) {
        // The passes stopped half way: labels hold union-find parents, tables do not cover what was written.  The call's
        // results are void (status), and the header must not vouch for them either: dense = 1 makes the next call on
        // these buffers clear everything -- label map, mask, counters and this barrier word -- instead of walking the
        // component boxes (k_clear; k_compact keeps the flag).
        pl.hdr->dense = 1u;
        for (int f = 0; f < batch; ++f) atomicOr(&status[f], YSMR_DET_STALLED);
    }
}

// The same passes with ONE WORKGROUP PER FRAME (round 4): nothing about a frame's residue depends on another frame, so a
// batch of many frames needs no grid barrier at all -- a frame's workgroup gathers its entries of the residue list into LDS
// (in list order: pass_bbox_euler pools runs of adjacent entries) and runs the passes with workgroup barriers between them.
// Five grid barriers among 128 blocks were 13 of k_residue's 16 us on the bench batch.  A frame with more entries than
// the LDS list holds reads the whole list in every pass and keeps its own; an overflowed list (dense) walks the frame.
// Batches of fewer than eight frames keep k_residue (few workgroups would serve whole frames; from a batch of 16 4K frames on this form wins: 342 -> 310 us of labelling chain, 22.5 -> 23.2 k frames/s).
#ifndef RESF_MIN_BATCH_N
#define RESF_MIN_BATCH_N 8
#endif
constexpr int RESF_THREADS = 1024, RESF_CAP = 8192, RESF_MIN_BATCH = RESF_MIN_BATCH_N;
__global__ __launch_bounds__(RESF_THREADS) void k_residue_frames(uint8_t *cls, uint32_t *labels, uint8_t *mask, Geo g, int batch,
                                                                 PixelList pl, CompTables t, int32_t *status)
{
    DET_RING(5);
    __shared__ uint32_t s_list[RESF_CAP];
    __shared__ uint32_t s_wave[RESF_THREADS / 64];
    const uint32_t listed = pl.hdr->count[0];   // (written by k_windows, the previous launch)
    const bool fault = pl.hdr->fault == YSMR_WS_FAULT_RESIDUE_STALL;
    if (listed == 0u && !fault) return;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (fault) {
        // (the test's stalled barrier, include/ysmr_hip.h: this form has no barrier that could stall -- it reports what
        // k_residue's bail-out reports, so that the caller's recovery is the same whichever kernel a batch size takes)
        if (tid == 0) {
            atomicOr(&status[f], YSMR_DET_STALLED);
            if (f == 0) pl.hdr->dense = 1u;     // (the fault word itself is cleared by the NEXT kernel of the chain, k_rank:
                                                // a workgroup of this launch that starts late must still see it)
        }
        return;
    }
    const bool dense = listed > pl.cap;
    uint32_t n_own = 0;
    if (!dense) {
        for (uint32_t i0 = 0; i0 < listed; i0 += RESF_THREADS) {
            const uint32_t i = i0 + (uint32_t)tid;
            const uint32_t e = i < listed ? pl.idx[0][i] : 0u;
            const bool mine = i < listed && e / g.HW == (uint32_t)f;
            const unsigned long long b = __ballot(mine);
            if (lane == 0) s_wave[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < RESF_THREADS / 64; ++w) { const uint32_t c = s_wave[w]; before += w < wave ? c : 0u; total += c; }
            const uint32_t pos = n_own + before + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            if (mine && pos < (uint32_t)RESF_CAP) s_list[pos] = e;
            n_own += total;
            __syncthreads();
        }
        if (n_own == 0u) return;
    }
    ResidueIter it;
    if (dense) it = ResidueIter{nullptr, (size_t)g.HW, (size_t)tid, (size_t)RESF_THREADS, (size_t)f * g.HW, -1, true};
    else if (n_own <= (uint32_t)RESF_CAP) it = ResidueIter{s_list, (size_t)n_own, (size_t)tid, (size_t)RESF_THREADS, 0, -1, false};
    else it = ResidueIter{pl.idx[0], (size_t)listed, (size_t)tid, (size_t)RESF_THREADS, 0, f, false};
    if (dense) {
        // (as in k_residue: everything k_windows settled in this frame is done again)
        for (size_t p = (size_t)tid; p < (size_t)g.HW; p += RESF_THREADS) {
            const size_t flat = (size_t)f * g.HW + p;
            const uint32_t b = cls[flat];
            if (b & 3u) {
                labels[flat] = (uint32_t)p + 1u;
                if (b & ~3u) cls[flat] = (uint8_t)(b & 3u);
            }
        }
        if (tid == 0) t.nroots[(size_t)f * NR_STRIDE] = 0;
        __syncthreads();
    }
    pass_union4(cls, labels, g, it);
    __syncthreads();
    pass_flag(cls, labels, g, it);
    __syncthreads();
    pass_union8(cls, labels, g, it);
    __syncthreads();
    pass_flatten(cls, labels, mask, g, it, t);
    __syncthreads();
    {   // pass_tag_roots for this frame
        const int n = min(t.nroots[(size_t)f * NR_STRIDE], t.max_det);
        for (int k = tid; k < n; k += RESF_THREADS)
            labels[(size_t)f * g.HW + (uint32_t)t.roots[(size_t)f * t.max_det + k]] = SLOT_TAG | (uint32_t)k;
    }
    __syncthreads();
    pass_bbox_euler(cls, labels, g, it, t);
}

// ------------------------------------------------------------------------------------------
// RETR_EXTERNAL nesting, one workgroup per holed component D.  Window = D's bounding box grown
// by one cell (cells outside the image count as outside background).  Background cells that
// cannot be reached (4-connected) from the window margin are enclosed by foreground; a component
// whose first pixel has such a cell as its west neighbour lies in a hole and is not a detection.
// Cell states: 0 foreground, 1 reached background, 2 unreached background.
// ------------------------------------------------------------------------------------------
// (s_cell / lds_cells: the workgroup's LDS for the window -- k_geometry's own arrays, which its nesting workgroups do not use)
__device__ void nested_component(const uint32_t *__restrict__ labels, const Geo &g, const CompTables &t, int idx,
                                 const int2 *holed, float *arena, uint32_t arena_floats, uint32_t *arena_used,
                                 int32_t *status, uint8_t *s_cell, int lds_cells)
{
    __shared__ int s_changed;
    __shared__ uint32_t s_off;
    const int f = holed[idx].x, k = holed[idx].y;
    const size_t o = (size_t)f * t.max_det + k;
    const int wx0 = t.bbox[o * 4 + 0] - 1, wy0 = t.bbox[o * 4 + 2] - 1;
    const int ww = t.bbox[o * 4 + 1] - t.bbox[o * 4 + 0] + 3, wh = t.bbox[o * 4 + 3] - t.bbox[o * 4 + 2] + 3;
    const int cells = ww * wh;
    const uint32_t *L = labels + (size_t)f * g.HW;
    const int W = g.W, H = g.H;
    uint8_t *cell = s_cell;
    if (cells > lds_cells) {
        if (threadIdx.x == 0) {
            uint32_t need = (uint32_t)(cells + 3) / 4;
            uint32_t off = atomicAdd(arena_used, need);
            s_off = (off + need > arena_floats) ? 0xFFFFFFFFu : off;
        }
        __syncthreads();
        if (s_off == 0xFFFFFFFFu) {
            if (threadIdx.x == 0) atomicOr(&status[f], YSMR_DET_ARENA);
            return;
        }
        cell = reinterpret_cast<uint8_t *>(arena + s_off);
    }
    for (int i = threadIdx.x; i < cells; i += 256) {
        int r = i / ww, c = i - r * ww;
        int y = wy0 + r, x = wx0 + c;
        bool inimg = (y >= 0 && y < H && x >= 0 && x < W);
        bool fg = inimg && L[(size_t)y * W + x] != 0;
        bool margin = (r == 0 || c == 0 || r == wh - 1 || c == ww - 1);
        cell[i] = fg ? 0 : ((!inimg || margin) ? 1 : 2);
    }
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
        bool any = false;
        for (int r = threadIdx.x; r < wh; r += 256) {
            uint8_t *row = cell + (size_t)r * ww;
            uint8_t prev = 0;
            for (int c = 0; c < ww; ++c) { uint8_t v = row[c]; if (v == 2 && prev == 1) { row[c] = 1; v = 1; any = true; } prev = v; }
            prev = 0;
            for (int c = ww - 1; c >= 0; --c) { uint8_t v = row[c]; if (v == 2 && prev == 1) { row[c] = 1; v = 1; any = true; } prev = v; }
        }
        __threadfence_block();
        __syncthreads();
        for (int c = threadIdx.x; c < ww; c += 256) {
            uint8_t prev = 0;
            for (int r = 0; r < wh; ++r) { uint8_t v = cell[(size_t)r * ww + c]; if (v == 2 && prev == 1) { cell[(size_t)r * ww + c] = 1; v = 1; any = true; } prev = v; }
            prev = 0;
            for (int r = wh - 1; r >= 0; --r) { uint8_t v = cell[(size_t)r * ww + c]; if (v == 2 && prev == 1) { cell[(size_t)r * ww + c] = 1; v = 1; any = true; } prev = v; }
        }
        if (any) s_changed = 1;
        __threadfence_block();
        __syncthreads();
        if (!s_changed) break;
    }
    const int n = min(t.nroots[(size_t)f * NR_STRIDE], t.max_det);
    for (int i = threadIdx.x; i < cells; i += 256) {
        int r = i / ww, c = i - r * ww;
        if (c == 0 || cell[i] != 0) continue;
        int y = wy0 + r, x = wx0 + c;
        uint32_t p = (uint32_t)y * (uint32_t)W + (uint32_t)x;
        if (L[p] != p + 1u || cell[i - 1] != 2) continue;
        int kk = find_rank(t.order + (size_t)f * t.max_det, n, (int32_t)p);
        if (kk >= 0) t.nested[(size_t)f * t.max_det + kk] = 1;
    }
}

// The queued components (usually a handful) are worked on by the first NEST_BLOCKS workgroups of k_geometry's launch (round 5:
// nothing in k_geometry depends on the outcome -- k_compact drops the nested components -- and as a launch of its own,
// between k_rank and k_geometry, this was 10-16 us of the labelling chain for a handful of dependent loads).
constexpr int NEST_BLOCKS = 128;

// ------------------------------------------------------------------------------------------
// k_geometry: a6, a 16-lane group per component (column scan), then its lane 0 (hull, calipers)
// ------------------------------------------------------------------------------------------
constexpr int GEO_THREADS = 256;
constexpr int LDS_POINTS = 2 * 16 + 3;  // hull capacity of the in-LDS fast path (components <= 16 columns wide)

struct HullStore {
    float *base;
    int stride;  // in floats
    __device__ __forceinline__ float &px(int i) const { return base[(size_t)(i * 5 + 0) * stride]; }
    __device__ __forceinline__ float &py(int i) const { return base[(size_t)(i * 5 + 1) * stride]; }
    __device__ __forceinline__ float &vx(int i) const { return base[(size_t)(i * 5 + 2) * stride]; }
    __device__ __forceinline__ float &vy(int i) const { return base[(size_t)(i * 5 + 3) * stride]; }
    __device__ __forceinline__ float &il(int i) const { return base[(size_t)(i * 5 + 4) * stride]; }
};

// Monotone-chain stack with strict turns (collinear points are dropped).  The two topmost points
// live in registers, so a push costs one LDS write and no dependent LDS reads; a pop reloads one
// point.  Coordinates are pixel indices < 16384 (check_geometry), so the cross product is exact in
// 32-bit integers.
struct Chain {
    int start, n;       // stack occupies store slots [start, n)
    int x1, y1, x2, y2; // top and second-from-top (valid when n - start >= 1 / >= 2)
};

__device__ __forceinline__ void chain_push(const HullStore &s, Chain &c, int x, int y)
{
    if (c.n > c.start && c.x1 == x && c.y1 == y) return;
    while (c.n - c.start >= 2) {
        int cr = (c.x1 - c.x2) * (y - c.y2) - (c.y1 - c.y2) * (x - c.x2);
        if (cr > 0) break;
        --c.n;                      // pop: the second becomes the top, reload the new second
        c.x1 = c.x2; c.y1 = c.y2;
        if (c.n - c.start >= 2) { c.x2 = (int)s.px(c.n - 2); c.y2 = (int)s.py(c.n - 2); }
    }
    s.px(c.n) = (float)x;
    s.py(c.n) = (float)y;
    c.x2 = c.x1; c.y2 = c.y1;
    c.x1 = x; c.y1 = y;
    ++c.n;
}

#define YSMR_PI 3.1415926535897932384626433832795

// vectors_ready: vx / vy / il of every edge are in the store already (k_geometry's lanes work them out side by side)
template <bool VECTORS_READY = false>
__device__ void min_area_rect_hull(const HullStore &s, int n, float *rect)
{
    float cx = 0.f, cy = 0.f, bw = 0.f, bh = 0.f, ang = 0.f;
    if (n > 2) {
        float minarea = FLT_MAX;
        int best_left = 0, best_bottom = 0;
        float best_a = 0.f, best_b = 0.f, best_w = 0.f, best_h = 0.f;
        int left = 0, bottom = 0, right = 0, top = 0;
        float ptx = s.px(0), pty = s.py(0);
        float left_x = ptx, right_x = ptx, top_y = pty, bottom_y = pty;
        for (int i = 0; i < n; ++i) {
            if (ptx < left_x) { left_x = ptx; left = i; }
            if (ptx > right_x) { right_x = ptx; right = i; }
            if (pty > top_y) { top_y = pty; top = i; }
            if (pty < bottom_y) { bottom_y = pty; bottom = i; }
            int nx = (i + 1 < n) ? i + 1 : 0;
            float qx = s.px(nx), qy = s.py(nx);
            if (!VECTORS_READY) {
                double dx = (double)qx - (double)ptx, dy = (double)qy - (double)pty;
                s.vx(i) = (float)dx;
                s.vy(i) = (float)dy;
                s.il(i) = (float)(1. / sqrt(dx * dx + dy * dy));
            }
            ptx = qx; pty = qy;
        }
        float orientation = 0.f;
        {
            double ax = s.vx(n - 1), ay = s.vy(n - 1);
            for (int i = 0; i < n; ++i) {
                double bx = s.vx(i), by = s.vy(i);
                double convexity = ax * by - ay * bx;
                if (convexity != 0) { orientation = (convexity > 0) ? 1.f : -1.f; break; }
                ax = bx; ay = by;
            }
        }
        float base_a = orientation, base_b = 0.f;
        int seq[4] = {bottom, right, top, left};
        for (int k = 0; k < n; ++k) {
            float dp0 = +base_a * s.vx(seq[0]) + base_b * s.vy(seq[0]);
            float dp1 = -base_b * s.vx(seq[1]) + base_a * s.vy(seq[1]);
            float dp2 = -base_a * s.vx(seq[2]) - base_b * s.vy(seq[2]);
            float dp3 = +base_b * s.vx(seq[3]) - base_a * s.vy(seq[3]);
            float maxcos = dp0 * s.il(seq[0]);
            int main_element = 0;
            float c1 = dp1 * s.il(seq[1]);
            if (c1 > maxcos) { main_element = 1; maxcos = c1; }
            float c2 = dp2 * s.il(seq[2]);
            if (c2 > maxcos) { main_element = 2; maxcos = c2; }
            float c3 = dp3 * s.il(seq[3]);
            if (c3 > maxcos) { main_element = 3; maxcos = c3; }
            int pindex = main_element == 0 ? seq[0] : main_element == 1 ? seq[1] : main_element == 2 ? seq[2] : seq[3];
            float lead_x = s.vx(pindex) * s.il(pindex);
            float lead_y = s.vy(pindex) * s.il(pindex);
            if (main_element == 0) { base_a = lead_x; base_b = lead_y; }
            else if (main_element == 1) { base_a = lead_y; base_b = -lead_x; }
            else if (main_element == 2) { base_a = -lead_x; base_b = -lead_y; }
            else { base_a = -lead_y; base_b = lead_x; }
            int nxt = pindex + 1;
            if (nxt == n) nxt = 0;
            if (main_element == 0) seq[0] = nxt; else if (main_element == 1) seq[1] = nxt;
            else if (main_element == 2) seq[2] = nxt; else seq[3] = nxt;

            float dx = s.px(seq[1]) - s.px(seq[3]);
            float dy = s.py(seq[1]) - s.py(seq[3]);
            float width = dx * base_a + dy * base_b;
            dx = s.px(seq[2]) - s.px(seq[0]);
            dy = s.py(seq[2]) - s.py(seq[0]);
            float height = -dx * base_b + dy * base_a;
            float area = width * height;
            if (area <= minarea) {
                minarea = area;
                best_left = seq[3]; best_a = base_a; best_w = width;
                best_b = base_b; best_h = height; best_bottom = seq[0];
            }
        }
        float A1 = best_a, B1 = best_b, A2 = -best_b, B2 = best_a;
        float C1 = A1 * s.px(best_left) + s.py(best_left) * B1;
        float C2 = A2 * s.px(best_bottom) + s.py(best_bottom) * B2;
        float idet = 1.f / (A1 * B2 - A2 * B1);
        float o0 = (C1 * B2 - C2 * B1) * idet;
        float o1 = (A1 * C2 - A2 * C1) * idet;
        float o2 = A1 * best_w, o3 = B1 * best_w, o4 = A2 * best_h, o5 = B2 * best_h;
        cx = o0 + (o2 + o4) * 0.5f;
        cy = o1 + (o3 + o5) * 0.5f;
        bw = (float)sqrt((double)o2 * o2 + (double)o3 * o3);
        bh = (float)sqrt((double)o4 * o4 + (double)o5 * o5);
        ang = (float)atan2((double)o3, (double)o2);
    } else if (n == 2) {
        cx = (s.px(0) + s.px(1)) * 0.5f;
        cy = (s.py(0) + s.py(1)) * 0.5f;
        double dx = s.px(1) - s.px(0), dy = s.py(1) - s.py(0);
        bw = (float)sqrt(dx * dx + dy * dy);
        bh = 0.f;
        ang = (float)atan2(dy, dx);
    } else if (n == 1) {
        cx = s.px(0);
        cy = s.py(0);
    }
    ang = (float)((double)(ang * 180.f) / YSMR_PI);
    rect[0] = cx; rect[1] = cy; rect[2] = bw; rect[3] = bh; rect[4] = ang;
}

// det slots are compacted afterwards (nested components dropped) by k_compact.
#ifdef YSMR_STAMPS
__device__ unsigned long long g_geo_stamps[16];
#define GEOSTAMP(k) do { if (blockIdx.x == 3 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_geo_stamps[k] = t_; } } while (0)
extern "C" int ysmr_debug_read_geo_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_geo_stamps), sizeof(unsigned long long) * 16); }
#else
#define GEOSTAMP(k) do {} while (0)
#endif

// GEO_GROUP lanes per component: lane `sub` scans columns minx+sub and minx+sub+GEO_GROUP of the bounding box (then the next
// two, for boxes wider than 2 GEO_GROUP) for the top-most and bottom-most pixel of the component (the only hull candidates of a
// column; all loads are independent, so two columns cost one round trip); lanes 0 and 1 then build the two chains from the
// (top, bottom) pairs side by side, all lanes the edge vectors, lane 0 runs the calipers.  The kernel is bound by that chain of
// dependent steps per component, i.e. by how many components are in flight: 16 -> 8 lanes per component took it from 61 to 41 us
// per 64 frames (round 2), 8 -> 4 (round 5, with the chains on two lanes: 64 components per block, 53 KB of LDS, three blocks per
// unit) the labelling chain from 366-380 to 344-349 us per 256 frames.  Components wider than GEO_COLS columns take the serial
// path with arena storage.
#ifndef GEO_GROUP_N
#define GEO_GROUP_N 4
#endif
constexpr int GEO_GROUP = GEO_GROUP_N;
constexpr int GEO_COLS = 16;
constexpr int GEO_COMPS = GEO_THREADS / GEO_GROUP;          // components per block
constexpr int GEO_LDS_STRIDE = LDS_POINTS * 5 + 2;          // 177 floats: the 8 active lanes of a wave start on 8 different banks

__device__ __forceinline__ void column_extent(const uint32_t *L, int W, int x, int miny, int maxy, uint32_t want,
                                              int &top, int &bot)
{
    top = -1; bot = -1;
    for (int y0 = miny; y0 <= maxy; y0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = L[(size_t)min(y0 + u, maxy) * W + x];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (y0 + u <= maxy && v[u] == want) { if (top < 0) top = y0 + u; bot = y0 + u; }
    }
}

// two columns at once (their loads in flight together)
__device__ __forceinline__ void column_extent2(const uint32_t *L, int W, int xa, int xb, int miny, int maxy, uint32_t want,
                                               int (&top)[2], int (&bot)[2])
{
    top[0] = top[1] = -1; bot[0] = bot[1] = -1;
    for (int y0 = miny; y0 <= maxy; y0 += 8) {
        uint32_t va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t row = (size_t)min(y0 + u, maxy) * W;
            va[u] = L[row + xa];
            vb[u] = L[row + xb];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (y0 + u <= maxy && va[u] == want) { if (top[0] < 0) top[0] = y0 + u; bot[0] = y0 + u; }
            if (y0 + u <= maxy && vb[u] == want) { if (top[1] < 0) top[1] = y0 + u; bot[1] = y0 + u; }
        }
    }
}

__device__ void geometry_group(const uint32_t *__restrict__ labels, const Geo &g, const CompTables &t, int f, int k,
                               float *det_tmp, float *arena, uint32_t arena_floats, uint32_t *arena_used,
                               int32_t *status, float *lds, int (*s_top)[GEO_COLS], int (*s_bot)[GEO_COLS])
{
    GEOSTAMP(0);
    const int grp = threadIdx.x / GEO_GROUP, sub = threadIdx.x % GEO_GROUP;
    // (the component's table entries are requested beside the count that says whether rank k exists, not behind it: the
    // kernel is a chain of round trips per component, and this was one of them)
    const size_t o = (size_t)f * t.max_det + min(k, t.max_det - 1);
    const int n_f = t.nroots[(size_t)f * NR_STRIDE];
    const int order = t.order[o];
    const int4 box = *reinterpret_cast<const int4 *>(t.bbox + o * 4);
    const bool live = k < min(n_f, t.max_det);   // (nested components too: the nesting workgroups of this launch are still at it)
    uint32_t want = 0;
    int minx = 0, maxx = -1, miny = 0, maxy = -1;
    if (live) {
        want = (uint32_t)order + 1u;
        minx = box.x; maxx = box.y; miny = box.z; maxy = box.w;
    }
    const int bwid = maxx - minx + 1;
    const uint32_t *L = labels + (size_t)(live ? f : 0) * g.HW;
    const int W = g.W;
    const bool narrow = live && bwid <= GEO_COLS;
    GEOSTAMP(1);
    if (narrow) {
        // lane `sub` takes columns sub, sub + GEO_GROUP, ... of the box, two at a time
        for (int c = sub; c < bwid; c += 2 * GEO_GROUP) {
            if (c + GEO_GROUP < bwid) {
                int top[2], bot[2];
                column_extent2(L, W, minx + c, minx + c + GEO_GROUP, miny, maxy, want, top, bot);
                s_top[grp][c] = top[0]; s_bot[grp][c] = bot[0];
                s_top[grp][c + GEO_GROUP] = top[1]; s_bot[grp][c + GEO_GROUP] = bot[1];
            } else {
                int top, bot;
                column_extent(L, W, minx + c, miny, maxy, want, top, bot);
                s_top[grp][c] = top;
                s_bot[grp][c] = bot;
            }
        }
    }
    // (a group's columns are written and read by lanes of ONE wave, like its hull store: no workgroup barrier anywhere in this
    // kernel -- a wave does not wait for the tallest box of the other three's components)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    GEOSTAMP(2);
    if (!live) return;
    if (narrow) {
        // Round 5: the two monotone chains side by side -- lane 0 of the group the max-y side (columns right -> left, within a
        // column bottom first then top) into the point columns of the store, lane 1 the min-y side (left -> right, top then
        // bottom) into the store's vector columns, which nobody needs yet -- then all eight lanes move the second chain behind the
        // first and work out the edge vectors (a float64 square root and a division per edge, which lane 0 used to do one after the
        // other), and lane 0 runs the calipers.  The same points, the same arithmetic, in the same places as the serial form.
        HullStore s;
        s.base = lds + grp * GEO_LDS_STRIDE;
        s.stride = 1;
        const int lane = threadIdx.x & 63, g0 = lane & ~(GEO_GROUP - 1);
        int n_mine = 0;
        if (sub < 2) {
            HullStore sc = s;
            if (sub == 1) sc.base = s.base + 2;               // px / py of this view are the store's vx / vy
            Chain c{0, 0, 0, 0, 0, 0};
            for (int k = 0; k < bwid; ++k) {
                const int col = sub == 0 ? bwid - 1 - k : k;
                const int top = s_top[grp][col], bot = s_bot[grp][col];
                if (top < 0) continue;
                chain_push(sc, c, minx + col, sub == 0 ? bot : top);
                chain_push(sc, c, minx + col, sub == 0 ? top : bot);
            }
            n_mine = c.n;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nu = __shfl(n_mine, g0), nl = __shfl(n_mine, g0 + 1);
        // the min-y side starts at the max-y side's last point (the lexicographically first pixel): its points 1 .. nl - 1 follow
        for (int j = 1 + sub; j < nl; j += GEO_GROUP) {
            const float x = s.vx(j), y = s.vy(j);
            s.px(nu - 1 + j) = x;
            s.py(nu - 1 + j) = y;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // drop the closing point (== point 0) unless the hull is a single point
        int hn = nu - 1 + nl - 1;
        if (hn < 1) hn = 1;
        for (int i = sub; i < hn; i += GEO_GROUP) {
            const int nx = (i + 1 < hn) ? i + 1 : 0;
            const float ptx = s.px(i), pty = s.py(i), qx = s.px(nx), qy = s.py(nx);
            const double dx = (double)qx - (double)ptx, dy = (double)qy - (double)pty;
            s.vx(i) = (float)dx;
            s.vy(i) = (float)dy;
            s.il(i) = (float)(1. / sqrt(dx * dx + dy * dy));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GEOSTAMP(3);
        if (sub == 0) min_area_rect_hull<true>(s, hn, det_tmp + o * 5);
        GEOSTAMP(4);
        return;
    }
    if (sub != 0) return;

    HullStore s;
    {
        uint32_t need = (uint32_t)(2 * bwid + 3) * 5u;
        uint32_t off = atomicAdd(arena_used, need);
        if (off + need > arena_floats) {
            atomicOr(&status[f], YSMR_DET_ARENA);
            float *d = det_tmp + o * 5;
            d[0] = d[1] = d[2] = d[3] = d[4] = 0.f;
            return;
        }
        s.base = arena + off;
        s.stride = 1;
    }
    // max-y side: columns right -> left, within a column bottom first then top
    Chain up{0, 0, 0, 0, 0, 0};
    for (int x = maxx; x >= minx; --x) {
        int top, bot;
        column_extent(L, W, x, miny, maxy, want, top, bot);
        if (top < 0) continue;
        chain_push(s, up, x, bot);
        chain_push(s, up, x, top);
    }
    // up.n >= 1; its last point is the lexicographically first pixel, where the min-y side starts
    Chain lo{up.n - 1, up.n - 1, 0, 0, 0, 0};
    for (int x = minx; x <= maxx; ++x) {
        int top, bot;
        column_extent(L, W, x, miny, maxy, want, top, bot);
        if (top < 0) continue;
        chain_push(s, lo, x, top);
        chain_push(s, lo, x, bot);
    }
    // drop the closing point (== point 0) unless the hull is a single point
    int hn = lo.n - 1;
    if (hn < 1) hn = 1;
    GEOSTAMP(3);
    min_area_rect_hull(s, hn, det_tmp + o * 5);
    GEOSTAMP(4);
}

// Resident grid.  Work items are (block of GEO_COMPS ranks, frame), frame fastest: the populated
// ranks come first in every frame, so the live items are spread evenly over the blocks and the
// loop stops at the largest component count of the batch (t.max_roots, from k_rank).  2 blocks per CU: a block
// holds 27 KB of LDS, and six of them per CU left no room for a k_frame block (see WINDOW_BLOCKS); the benchmark
// batch takes two rounds this way (detection alone 229 k instead of 243 k frames/s, end to end +2 %).
constexpr int GEO_BLOCKS = 512;
__global__ __launch_bounds__(GEO_THREADS) void k_geometry(const uint32_t *__restrict__ labels, Geo g, CompTables t,
                                                          int batch, float *det_tmp, float *arena,
                                                          uint32_t arena_floats, uint32_t *arena_used, int32_t *status,
                                                          const int32_t *n_holed, const int2 *holed)
{
    DET_RING(13);
    __shared__ float lds[GEO_COMPS * GEO_LDS_STRIDE];
    __shared__ int s_top[GEO_COMPS][GEO_COLS], s_bot[GEO_COMPS][GEO_COLS];
    static_assert(GEO_THREADS == 256, "nested_component strides by 256 threads");
    if (blockIdx.x < (unsigned)NEST_BLOCKS) {      // RETR_EXTERNAL nesting: the components with holes, one workgroup each
        const int nh = min(*n_holed, HOLED_CAP);
        for (int idx = blockIdx.x; idx < nh; idx += NEST_BLOCKS) {
            nested_component(labels, g, t, idx, holed, arena, arena_floats, arena_used, status, reinterpret_cast<uint8_t *>(lds),
                             (int)sizeof(lds));
            __syncthreads();
        }
        return;
    }
    const int most = min(*t.max_roots, t.max_det);
    const long long items = (long long)((most + GEO_COMPS - 1) / GEO_COMPS) * batch;
    for (long long it = blockIdx.x - NEST_BLOCKS; it < items; it += gridDim.x - NEST_BLOCKS) {
        const int kb = (int)(it / batch), f = (int)(it - (long long)kb * batch);
        geometry_group(labels, g, t, f, kb * GEO_COMPS + threadIdx.x / GEO_GROUP, det_tmp, arena, arena_floats,
                       arena_used, status, lds, s_top, s_bot);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (this item's LDS reads before the next item's writes, wave by wave)
        __builtin_amdgcn_wave_barrier();
    }
}

// Drop nested components, write final detection list / count / anchors.
constexpr int COMPACT_THREADS = 1024;
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact(CompTables t, const float *__restrict__ det_tmp, float *det,
                                                 int32_t *det_count, int32_t *anchors, PixelList pl, const uint8_t *labels,
                                                 const uint8_t *mask, size_t total, int batch, int H, int W, bool angle_pre451)
{
    DET_RING(14);
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // last kernel of the call: the current list describes these buffers
        WsHeader *h = pl.hdr;
        h->labels = (unsigned long long)labels;
        h->mask = (unsigned long long)mask;
        h->total = total;
        h->batch = batch; h->H = H; h->W = W; h->max_det = t.max_det;
        h->count[0] = 0;
        __threadfence();
        h->magic = WS_MAGIC;
    }
    const int f = blockIdx.x;
    const int n = min(t.nroots[(size_t)f * NR_STRIDE], t.max_det);
    __syncthreads();   // (every thread has its n)
    if (threadIdx.x == 0) {
        t.prev_n[f] = n;
        t.nroots[(size_t)f * NR_STRIDE] = 0;
        if (f == 0)
            for (int i = 0; i < 8; ++i) t.nroots[(size_t)batch * NR_STRIDE + i] = 0;   // n_holed, arena_used, barrier, max_roots
    }
    // stable compaction: ranks of the kept components from a ballot per wave and the wave totals of a round in LDS
    // (two buffers used alternately: one barrier per round of COMPACT_THREADS components)
    __shared__ int s_cnt[2][COMPACT_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int i0 = 0, round = 0; i0 < n; i0 += COMPACT_THREADS, ++round) {
        const int i = i0 + threadIdx.x;
        const size_t o = (size_t)f * t.max_det + i;
        const bool keep = i < n && !t.nested[o];
        float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int anchor = 0;
        if (keep) {
#pragma unroll
            for (int j = 0; j < 5; ++j) r[j] = det_tmp[o * 5 + j];
            anchor = t.order[o];
        }
        const unsigned long long kept = __ballot(keep);
        int *cnt = s_cnt[round & 1];
        if (lane == 0) cnt[wave] = __popcll(kept);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < COMPACT_THREADS / 64; ++w) {
            const int c = cnt[w];
            before += w < wave ? c : 0;
            total += c;
        }
        if (keep) {
            const size_t q = (size_t)f * t.max_det + base + before + __popcll(kept & ((1ull << lane) - 1ull));
            // k_geometry's hull order + rotating calipers give the angle in [0, 90], cv::minAreaRect's range from
            // OpenCV 4.5.1 on.  Earlier releases report the same rectangle with its angle in [-90, 0) and the
            // sides named the other way round (YSMR_CV_ANGLE_PRE451; only rectangles from the rotating
            // calipers, i.e. with two non-zero sides)
            if (angle_pre451 && r[3] > 0.f) {
                if (r[4] == 90.f) r[4] = -90.f;
                else { const float w = r[2]; r[2] = r[3]; r[3] = w; r[4] -= 90.f; }
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) det[q * 5 + j] = r[j];
            if (anchors) anchors[q] = anchor;
        }
        base += total;
    }
    if (threadIdx.x == 0) det_count[f] = base;
}

Gauss11 make_gauss11()
{
    Gauss11 g;
    const double sigma = 0.3 * ((11 - 1) * 0.5 - 1.0) + 0.8;
    const double scale2x = -0.5 / (sigma * sigma);
    double t[11], sum = 0.0;
    for (int i = 0; i < 11; ++i) {
        double x = i - 5.0;
        t[i] = std::exp(scale2x * x * x);
        sum += t[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < 11; ++i) g.k[i] = (float)(t[i] * sum);
    return g;
}

struct Workspace {
    int32_t *nroots, *roots, *order, *bbox, *euler4, *nested, *n_holed, *max_roots, *prev_n, *bbox_tmp, *euler_tmp;
    int2 *holed;
    PixelList pixels;
    uint32_t *arena_used, *oldbits;
    float *det_tmp, *arena;
    uint32_t arena_floats;
    size_t bytes;
};

Workspace carve(void *base, int batch, int H, int W, int max_det)
{
    Workspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = ysmr::align_up(off + bytes, 256); return (char *)base + o; };
    size_t bm = (size_t)batch * max_det;
    w.pixels.hdr = (WsHeader *)take(256);   // persists across calls (ysmr_detect_workspace_init zeroes it)
    // counters first: k_clear zeroes nroots, n_holed, arena_used, max_roots in one go
    w.nroots = (int32_t *)take(sizeof(int32_t) * ((size_t)batch * NR_STRIDE + 8));
    w.n_holed = w.nroots + (size_t)batch * NR_STRIDE;
    w.arena_used = (uint32_t *)(w.n_holed + 1);
    w.max_roots = w.n_holed + 3;
    w.prev_n = (int32_t *)take(sizeof(int32_t) * (size_t)batch);
    w.pixels.cap = (uint32_t)(((size_t)batch * H * W + 7) / 8);
    w.pixels.idx[0] = (uint32_t *)take(sizeof(uint32_t) * w.pixels.cap);
    w.pixels.idx[1] = nullptr;
    w.holed = (int2 *)take(sizeof(int2) * HOLED_CAP);
    w.roots = (int32_t *)take(sizeof(int32_t) * bm);
    w.order = (int32_t *)take(sizeof(int32_t) * bm);
    w.bbox = (int32_t *)take(sizeof(int32_t) * bm * 4);
    w.euler4 = (int32_t *)take(sizeof(int32_t) * bm);
    w.nested = (int32_t *)take(sizeof(int32_t) * bm);
    w.bbox_tmp = (int32_t *)take(sizeof(int32_t) * bm * 4);
    w.euler_tmp = (int32_t *)take(sizeof(int32_t) * bm);
    w.det_tmp = (float *)take(sizeof(float) * bm * 5);
    // scratch arena shared by k_geometry (hulls wider than the LDS fast path) and its nesting workgroups (windows
    // larger than LDS): room for 16 full-width hulls per frame, and at least one full-frame window
    size_t arena_floats = (size_t)batch * std::max<size_t>(16 * 5 * (2 * (size_t)W + 3), ((size_t)(H + 2) * (W + 2) + 3) / 4);
    if (arena_floats > 0x7FFFFFFFull) arena_floats = 0x7FFFFFFFull;
    w.arena_floats = (uint32_t)arena_floats;
    w.arena = (float *)take(sizeof(float) * arena_floats);
    // k_windows' words for the next call's clearing: per work item (WIN_GROUP cores side by side) 32 rows x WIN_GROUP words
    {
        constexpr int SPAN = WIN_GROUP * WIN_CORE;
        const size_t items = (size_t)batch * ((H + WIN_CORE - 1) / WIN_CORE) * ((W + SPAN - 1) / SPAN);
        w.oldbits = (uint32_t *)take(sizeof(uint32_t) * items * WIN_CORE * WIN_GROUP);
    }
    w.bytes = off;
    return w;
}

int check_geometry(int batch, int H, int W, int channels, int max_det)
{
    if (batch <= 0 || H <= 0 || W <= 0 || max_det <= 0)
        return ysmr::fail(YSMR_ERR_ARG, "batch, height, width, max_det must be positive (got %d, %d, %d, %d)", batch, H, W, max_det);
    if (channels != 1 && channels != 3)
        return ysmr::fail(YSMR_ERR_ARG, "channels must be 1 (gray) or 3 (BGR), got %d", channels);
    if (H > 16384 || W > 16384 || (size_t)batch * H * W > (1ull << 32) - 64)
        return ysmr::fail(YSMR_ERR_ARG, "batch too large (height and width are limited to 16384, batch*height*width to 2^32 - 1)");
    return YSMR_OK;
}

// Tuning knobs: resident grid sizes of the detection kernels and the strip kernel's segment height; values
// <= 0 keep the defaults.  The shipped library has none (a drop-in library does not read its caller's
// environment); a build with EXTRA=-DYSMR_TUNING reads them once per process for the sweeps in scripts/.
struct Knobs {
    int seg_h, thr_blocks, collect_blocks, sparse_blocks, clear_blocks, geo_blocks;
};
const Knobs &knobs()
{
#ifdef YSMR_TUNING
    static const Knobs k = [] {
        auto get = [](const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; };
        return Knobs{get("YSMR_SEG_H"), get("YSMR_THR_BLOCKS"), get("YSMR_COLLECT_BLOCKS"), get("YSMR_SPARSE_BLOCKS"),
                     get("YSMR_CLEAR_BLOCKS"), get("YSMR_GEO_BLOCKS")};
    }();
#else
    static const Knobs k = {0, 0, 0, 0, 0, 0};
#endif
    return k;
}

// workgroups of the matrix-pipe threshold kernel under the caller's hints (0: one per compute unit, thr_mfma's own choice)
static int threshold_workgroups(int cv_flavour)
{
    return (cv_flavour & YSMR_BESIDE_BATCH_LINK) ? 248 : ((cv_flavour & YSMR_BESIDE_SPLIT_LINK) ? 160 : 0);
}


thread_local hipEvent_t t_thr_events[2] = {nullptr, nullptr};   // ysmr_threshold_timing -> the next launch_threshold of this thread

// variant: 0 = the shipped choice of kernel; 1 = strip / tile kernels only (bit-exact float32 chain for every pixel);
// 2, 3 = the matrix-pipe kernel's diagnostic builds (thr_mfma.h)
int launch_threshold(hipStream_t st, const uint8_t *frames, int batch, int H, int W, int channels, int inv, int t_low,
                     int t_high, int use_high, uint8_t *cls, int cv_flavour, int variant = 0)
{
    Gauss11 gk = make_gauss11();
    const ysmr::GrayCoef gc = ysmr::gray_coef(cv_flavour);
    // (ysmr_threshold_timing: this call's kernel dispatch updates the caller's two events)
    const hipEvent_t ev0 = t_thr_events[0], ev1 = t_thr_events[1];
    t_thr_events[0] = t_thr_events[1] = nullptr;
    const bool timed = ev0 || ev1;
    if (variant != 1 && ysmr_thr::supported(H, W, channels, t_low, t_high, use_high))
        return ysmr_thr::launch(st, frames, cls, batch, H, W, inv, t_low, t_high, use_high, gk.k,
                                knobs().thr_blocks > 0 ? knobs().thr_blocks : threshold_workgroups(cv_flavour),
                                variant >= 2 ? variant - 1 : 0, ev0, ev1);
    if (variant >= 2) return ysmr::fail(YSMR_ERR_ARG, "the matrix-pipe threshold kernel does not serve this geometry");
    const int t_gap = use_high ? (t_high > t_low ? t_high - t_low : t_low - t_high) : 0;
    if ((W & 3) == 0 && W >= 16 && H >= 2 && t_low > -100000 && t_low < 100000 && t_high > -100000 && t_high < 100000 && t_gap <= 127) {
        StripParams P;
        P.H = H; P.W = W; P.batch = batch;
        const int quads = (W + 3) / 4;
        P.strips_x = (quads + STRIP_MAX_OUT_LANES - 1) / STRIP_MAX_OUT_LANES;
        P.out_lanes = (quads + P.strips_x - 1) / P.strips_x;
        int resident = 768;   // 3 blocks per CU (105 / 111 VGPRs, allocated as 112): a wave slot and 176 VGPRs per
                              // SIMD stay free for the link kernel's waves
        if (knobs().thr_blocks > 0) resident = knobs().thr_blocks;
        // rows per work item: every item re-reads and re-filters 10 halo rows, so items are as tall as they can
        // be while there is still one for every resident wave (1228x922, 64 frames: 384 strip columns x 8
        // segments of 116 rows = 3072 items = 768 blocks x 4 waves; it was 45 rows, 2.6 items per wave); never
        // shorter than 32 rows
        {
            const long long columns = (long long)batch * P.strips_x;
            const long long segs = std::max<long long>(1, (long long)resident * 4 / columns);
            P.seg_h = (int)std::max<long long>(32, (H + segs - 1) / segs);
        }
        if (knobs().seg_h > 0) P.seg_h = knobs().seg_h;
        P.segs_y = (H + P.seg_h - 1) / P.seg_h;
        P.inv = inv; P.t_low = t_low; P.t_high = use_high ? t_high : t_low; P.gc = gc;
        const long long waves = (long long)batch * P.strips_x * P.segs_y;
        long long blocks = (waves + 3) / 4;
        if (resident > 0 && blocks > resident) blocks = resident;
        P.by_xcd = (batch % 8 == 0 && blocks % 8 == 0) ? 1 : 0;
        if (P.by_xcd && blocks / 8 * 4 > (long long)(batch / 8) * P.strips_x * P.segs_y)    // (no more waves than an XCD has items)
            blocks = (((long long)(batch / 8) * P.strips_x * P.segs_y + 3) / 4) * 8;
        if (timed) {
            if (channels == 1) hipExtLaunchKernelGGL(k_threshold_strip<1>, dim3((unsigned)blocks), dim3(256), 0u, st, ev0, ev1, 0u, frames, cls, P, gk);
            else hipExtLaunchKernelGGL(k_threshold_strip<3>, dim3((unsigned)blocks), dim3(256), 0u, st, ev0, ev1, 0u, frames, cls, P, gk);
        } else if (channels == 1) hipLaunchKernelGGL(k_threshold_strip<1>, dim3((unsigned)blocks), dim3(256), 0, st, frames, cls, P, gk);
        else hipLaunchKernelGGL(k_threshold_strip<3>, dim3((unsigned)blocks), dim3(256), 0, st, frames, cls, P, gk);
    } else {
        dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, batch);
        if (timed) {
            if (channels == 1)
                hipExtLaunchKernelGGL(k_threshold<1>, grid, dim3(256), 0u, st, ev0, ev1, 0u, frames, cls, H, W, gk, inv, t_low, t_high, use_high, gc);
            else
                hipExtLaunchKernelGGL(k_threshold<3>, grid, dim3(256), 0u, st, ev0, ev1, 0u, frames, cls, H, W, gk, inv, t_low, t_high, use_high, gc);
        } else if (channels == 1)
            hipLaunchKernelGGL(k_threshold<1>, grid, dim3(256), 0, st, frames, cls, H, W, gk, inv, t_low, t_high, use_high, gc);
        else
            hipLaunchKernelGGL(k_threshold<3>, grid, dim3(256), 0, st, frames, cls, H, W, gk, inv, t_low, t_high, use_high, gc);
    }
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

}  // namespace

extern "C" {

size_t ysmr_detect_workspace_bytes(int batch, int height, int width, int max_det)
{
    if (batch <= 0 || height <= 0 || width <= 0 || max_det <= 0) return 0;
    return carve(nullptr, batch, height, width, max_det).bytes;
}

int ysmr_detect_workspace_init(void *stream, void *workspace_dev, size_t workspace_bytes)
{
    if (!workspace_dev || workspace_bytes < 256) return ysmr::fail(YSMR_ERR_ARG, "workspace_dev is NULL or smaller than its header");
    YSMR_HIP_CHECK(hipMemsetAsync(workspace_dev, 0, 256, (hipStream_t)stream));
    return YSMR_OK;
}

int ysmr_threshold_batch(void *stream, const uint8_t *frames_dev, int batch, int height, int width, int channels,
                         int inv, int t_low, int t_high, int use_high, uint8_t *cls_dev, int cv_flavour)
{
    if (int rc = check_geometry(batch, height, width, channels, 1)) return rc;
    if (!frames_dev || !cls_dev) return ysmr::fail(YSMR_ERR_ARG, "frames_dev and cls_dev must not be NULL");
    if (cv_flavour & ~YSMR_CV_FLAVOUR_MASK) return ysmr::fail(YSMR_ERR_ARG, "unknown cv_flavour bits 0x%x", cv_flavour);
    return launch_threshold((hipStream_t)stream, frames_dev, batch, height, width, channels, inv, t_low, t_high,
                            use_high, cls_dev, cv_flavour, (cv_flavour & YSMR_BESIDE_LINK) ? 1 : 0);
}

int ysmr_threshold_timing(void *start_event, void *stop_event)
{
    t_thr_events[0] = (hipEvent_t)start_event;
    t_thr_events[1] = (hipEvent_t)stop_event;
    return YSMR_OK;
}

int ysmr_threshold_workgroups(int cv_flavour)
{
    const int n = threshold_workgroups(cv_flavour);
    return n > 0 ? n : 256;
}

int ysmr_threshold_batch_variant(void *stream, const uint8_t *frames_dev, int batch, int height, int width, int channels,
                                 int inv, int t_low, int t_high, int use_high, uint8_t *cls_dev, int cv_flavour, int variant)
{
    if (int rc = check_geometry(batch, height, width, channels, 1)) return rc;
    if (!frames_dev || !cls_dev) return ysmr::fail(YSMR_ERR_ARG, "frames_dev and cls_dev must not be NULL");
    if (cv_flavour & ~YSMR_CV_FLAVOUR_MASK) return ysmr::fail(YSMR_ERR_ARG, "unknown cv_flavour bits 0x%x", cv_flavour);
    if (variant < 0 || variant > 3) return ysmr::fail(YSMR_ERR_ARG, "variant must be 0..3, got %d", variant);
    if (variant == 0 && (cv_flavour & YSMR_BESIDE_LINK)) variant = 1;
    return launch_threshold((hipStream_t)stream, frames_dev, batch, height, width, channels, inv, t_low, t_high,
                            use_high, cls_dev, cv_flavour, variant);
}

int ysmr_components_batch(void *stream, int batch, int height, int width, void *workspace_dev, size_t workspace_bytes,
                          uint8_t *cls_dev, uint8_t *mask_dev, int32_t *labels_dev, int32_t *det_count_dev,
                          float *det_dev, int32_t *anchors_dev, int max_det, int32_t *status_dev, int cv_flavour)
{
    if (int rc = check_geometry(batch, height, width, 1, max_det)) return rc;
    if (cv_flavour & ~YSMR_CV_FLAVOUR_MASK) return ysmr::fail(YSMR_ERR_ARG, "unknown cv_flavour bits 0x%x", cv_flavour);
    if (!cls_dev || !labels_dev || !det_count_dev || !det_dev || !status_dev || !workspace_dev)
        return ysmr::fail(YSMR_ERR_ARG, "a required device pointer is NULL");
    if (((uintptr_t)cls_dev & 15) || ((uintptr_t)labels_dev & 15) || (mask_dev && ((uintptr_t)mask_dev & 15)))
        return ysmr::fail(YSMR_ERR_ARG, "cls_dev, labels_dev and mask_dev must be 16-byte aligned");
    Workspace w = carve(workspace_dev, batch, height, width, max_det);
    if (workspace_bytes < w.bytes)
        return ysmr::fail(YSMR_ERR_CAPACITY, "workspace too small: %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    Geo g{height, width, (uint32_t)((size_t)height * width), (size_t)batch * height * width};
    uint32_t *labels = reinterpret_cast<uint32_t *>(labels_dev);

    const Knobs &kn = knobs();
    // (the LDS reserve matters to k_frame, the one-launch link, which tables of more than 2456 detections do not use
    // and which the caller announces with YSMR_BESIDE_LINK)
    const bool small_tables = max_det <= 2456 && (cv_flavour & YSMR_BESIDE_LINK);
    // k_windows' grid.  Beside the per-frame link kernels (one-launch or split) it stays a RESIDENT grid that leaves them their LDS
    // and wave slots.  Beside the batch link -- one workgroup that has long been seated -- or with no link at all, a wave per work item
    // and the dispatcher hands the workgroups out as units become free: the work items differ (a core with twenty islands, a core
    // with none), and 2048 resident workgroups with 4.4 items per wave ended as late as the unluckiest of them (round 5, last hours:
    // 293 -> 267-270 us of labelling chain per 256 frames with 7168-8192 workgroups, scripts/sweep_window_blocks.sh; the link's
    // time does not change)
    const bool per_frame_link = (cv_flavour & (YSMR_BESIDE_LINK | YSMR_BESIDE_SPLIT_LINK)) != 0;
    const unsigned window_blocks = kn.collect_blocks > 0 ? (unsigned)kn.collect_blocks
                                   : (unsigned)(small_tables ? WINDOW_BLOCKS : (per_frame_link ? 2 * WINDOW_BLOCKS : WINDOW_BLOCKS_FREE));
    const unsigned sparse_blocks = kn.sparse_blocks > 0 ? (unsigned)kn.sparse_blocks : (unsigned)SPARSE_BLOCKS;
    const unsigned clear_blocks = kn.clear_blocks > 0 ? (unsigned)kn.clear_blocks : (unsigned)CLEAR_BLOCKS;
    const unsigned geo_blocks = kn.geo_blocks > 0 ? (unsigned)kn.geo_blocks : (unsigned)(small_tables ? GEO_BLOCKS : 3 * GEO_BLOCKS);
    CompTables t{w.nroots, w.roots, w.order, w.bbox, w.euler4, w.nested, w.max_roots, w.prev_n, w.bbox_tmp, w.euler_tmp, max_det};
    hipLaunchKernelGGL(k_clear, dim3(clear_blocks), dim3(256), 0, st, w.pixels, t, reinterpret_cast<uint8_t *>(labels), mask_dev,
                       g.total, batch * NR_STRIDE + 8, status_dev, batch, height, width);
    const dim3 sg(sparse_blocks), tb(256);
    {
        constexpr int SPAN = WIN_GROUP * WIN_CORE;
        const long long items = (long long)batch * ((height + WIN_CORE - 1) / WIN_CORE) * ((width + SPAN - 1) / SPAN);
        long long wblocks = std::min<long long>((items + 3) / 4, window_blocks);
        if (wblocks >= 8) wblocks &= ~7ll;   // (a multiple of 8: the kernel deals its items to the XCDs)
        hipLaunchKernelGGL(k_windows, dim3((unsigned)wblocks), tb, 0, st, cls_dev, labels, mask_dev, g,
                           batch, w.pixels, t, w.oldbits);
    }
    if (batch >= RESF_MIN_BATCH)
        hipLaunchKernelGGL(k_residue_frames, dim3(batch), dim3(RESF_THREADS), 0, st, cls_dev, labels, mask_dev, g, batch, w.pixels, t,
                           status_dev);
    else
        hipLaunchKernelGGL(k_residue, dim3(RESIDUE_BLOCKS), tb, 0, st, cls_dev, labels, mask_dev, g, batch, w.pixels, t, w.arena_used + 1,
                           status_dev);
    YSMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rank, dim3(batch, (max_det + RANK_THREADS / 4 - 1) / (RANK_THREADS / 4) < 32 ? (max_det + RANK_THREADS / 4 - 1) / (RANK_THREADS / 4) : 32), dim3(RANK_THREADS), 0, st, t, labels, g.HW,
                       width, height, status_dev, w.pixels.hdr, w.n_holed, w.holed);
    hipLaunchKernelGGL(k_geometry, dim3(geo_blocks + NEST_BLOCKS), dim3(GEO_THREADS), 0, st,
                       labels, g, t, batch, w.det_tmp, w.arena, w.arena_floats, w.arena_used, status_dev, w.n_holed, w.holed);
    hipLaunchKernelGGL(k_compact, dim3(batch), dim3(COMPACT_THREADS), 0, st, t, w.det_tmp, det_dev, det_count_dev, anchors_dev, w.pixels,
                       reinterpret_cast<const uint8_t *>(labels), mask_dev, g.total, batch, height, width,
                       (cv_flavour & YSMR_CV_ANGLE_PRE451) != 0);
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

int ysmr_detect_batch(void *stream, const uint8_t *frames_dev, int batch, int height, int width, int channels, int inv,
                      int t_low, int t_high, int use_high, void *workspace_dev, size_t workspace_bytes,
                      uint8_t *cls_dev, uint8_t *mask_dev, int32_t *labels_dev, int32_t *det_count_dev, float *det_dev,
                      int32_t *anchors_dev, int max_det, int32_t *status_dev, int cv_flavour)
{
    if (int rc = ysmr_threshold_batch(stream, frames_dev, batch, height, width, channels, inv, t_low, t_high, use_high,
                                      cls_dev, cv_flavour))
        return rc;
    return ysmr_components_batch(stream, batch, height, width, workspace_dev, workspace_bytes, cls_dev, mask_dev,
                                 labels_dev, det_count_dev, det_dev, anchors_dev, max_det, status_dev, cv_flavour);
}

}  // extern "C"
