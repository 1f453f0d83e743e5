// libysmr_hip.so -- the mean-gray threshold branch of the detection path (ysmr/track_eval.py:219-253,
// taken when 'adaptive double threshold' < 0): instead of the Gaussian adaptive threshold, every
// frame is compared against ONE level derived from the gray image's mean and standard deviation,
// averaged over the last 5 s of frames.  Three launches per batch:
//   k_gray_sums        sum and sum of squares of the gray frame (exact integers; cv2.meanStdDev)
//   k_mean_levels      mean/stddev -> per-frame level -> moving average in the reference's summation
//                      order -> int(level); carries the list of the last levels across batches
//   k_level_strip      cvtColor + GaussianBlur(3x3) + cv2.threshold(blurred, level) -> class map
//                      (k_level_threshold for widths that are not a multiple of 4)
// The class map uses value 3 (thresh and marker bit) for foreground so that ysmr_components_batch
// keeps every component, as the reference does in this branch (no binary_propagation).
// All grids are resident-sized and stride over their work (see detect.hip on why).
#include "common.h"
#include <cmath>

namespace {

constexpr int MG_BLOCKS = 1024;
constexpr int LEVEL_SEG = 32;      // output rows per thread item of k_level_threshold
constexpr int LEVEL_PF = 4;        // rows in flight per thread

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}
// source index of a BORDER_REFLECT_101 access, forced into the image (indices more than one step
// outside only occur for padding columns of a partial group, whose results are never stored)
__device__ __forceinline__ int border_index(int i, int n)
{
    i = reflect101(i, n);
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
__device__ __forceinline__ uint32_t load_u32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// gray value of pixel p of a frame
template <int CH>
__device__ __forceinline__ uint32_t gray_at(const ysmr::GrayCoef &gc, const uint8_t *frame, uint32_t p)
{
    if constexpr (CH == 1) return frame[p];
    else return bgr2gray(gc, frame[3 * (size_t)p], frame[3 * (size_t)p + 1], frame[3 * (size_t)p + 2]);
}
// the 4 gray pixels p .. p+3 (all inside the frame) packed into one dword
template <int CH>
__device__ __forceinline__ uint32_t gray4_at(const ysmr::GrayCoef &gc, const uint8_t *frame, uint32_t p)
{
    if constexpr (CH == 1) return load_u32(frame + p);
    else {
        const uint8_t *q = frame + 3 * (size_t)p;
        const uint32_t w0 = load_u32(q), w1 = load_u32(q + 4), w2 = load_u32(q + 8);
        const uint32_t g0 = bgr2gray(gc, w0 & 0xFFu, (w0 >> 8) & 0xFFu, (w0 >> 16) & 0xFFu);
        const uint32_t g1 = bgr2gray(gc, w0 >> 24, w1 & 0xFFu, (w1 >> 8) & 0xFFu);
        const uint32_t g2 = bgr2gray(gc, (w1 >> 16) & 0xFFu, w1 >> 24, w2 & 0xFFu);
        const uint32_t g3 = bgr2gray(gc, (w2 >> 8) & 0xFFu, (w2 >> 16) & 0xFFu, w2 >> 24);
        return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    }
}

// sums[f][0] += sum of gray, sums[f][1] += sum of gray^2 (u64; the caller zeroes them).
// A block step covers one of `parts` contiguous ranges of one frame: 16-byte loads (gray) or 3 dwords
// per 4 pixels (BGR), several in flight, ONE block reduction and one pair of atomics per step.
template <int CH>
__global__ __launch_bounds__(256) void k_gray_sums(const uint8_t *__restrict__ frames, uint32_t HW, int batch, int parts,
                                                   unsigned long long *__restrict__ sums, int stride, ysmr::GrayCoef gc)
{
    __shared__ unsigned long long s_part[2][4];
    const uint32_t span = ((HW + parts - 1) / parts + 15u) & ~15u;   // pixels per part
    const long long items = (long long)batch * parts;
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int f = (int)(it / parts);
        const uint32_t lo = (uint32_t)(it - (long long)f * parts) * span;
        const uint32_t hi = lo + span < HW ? lo + span : HW;
        const uint8_t *frame = frames + (size_t)f * HW * CH;
        unsigned long long s = 0, q = 0;
        auto add4 = [&](uint32_t g, uint32_t &ps, uint32_t &pq) {
#pragma unroll
            for (int o = 0; o < 4; ++o) { const uint32_t v = (g >> (8 * o)) & 0xFFu; ps += v; pq += v * v; }
        };
        if (lo < hi) {
            uint32_t body_lo = lo, body_hi = hi;
            if constexpr (CH == 1) {
                // 16-byte aligned body; the unaligned ends go pixel by pixel
                const uintptr_t a0 = (uintptr_t)(frame + lo);
                body_lo = lo + (uint32_t)((16u - (a0 & 15u)) & 15u);
                if (body_lo > hi) body_lo = hi;
                body_hi = body_lo + ((hi - body_lo) & ~15u);
                const uint4 *src = reinterpret_cast<const uint4 *>(frame + body_lo);
                const uint32_t n16 = (body_hi - body_lo) >> 4;
#pragma unroll 4
                for (uint32_t i = threadIdx.x; i < n16; i += 256) {
                    const uint4 v = src[i];
                    uint32_t ps = 0, pq = 0;
                    add4(v.x, ps, pq); add4(v.y, ps, pq); add4(v.z, ps, pq); add4(v.w, ps, pq);
                    s += ps; q += pq;
                }
            } else {
                body_hi = lo + ((hi - lo) & ~3u);
                const uint32_t n4 = (body_hi - lo) >> 2;
#pragma unroll 4
                for (uint32_t i = threadIdx.x; i < n4; i += 256) {
                    uint32_t ps = 0, pq = 0;
                    add4(gray4_at<CH>(gc, frame, lo + 4u * i), ps, pq);
                    s += ps; q += pq;
                }
            }
            const uint32_t ends = (body_lo - lo) + (hi - body_hi);   // < 32 pixels
            if (threadIdx.x < ends) {
                const uint32_t e = threadIdx.x < body_lo - lo ? lo + threadIdx.x : body_hi + (threadIdx.x - (body_lo - lo));
                const uint32_t v = gray_at<CH>(gc, frame, e);
                s += v; q += v * v;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
        if ((threadIdx.x & 63) == 0) { s_part[0][threadIdx.x >> 6] = s; s_part[1][threadIdx.x >> 6] = q; }
        __syncthreads();
        if (threadIdx.x < 2 && lo < hi) {
            const unsigned long long *v = s_part[threadIdx.x];
            atomicAdd(&sums[(size_t)f * stride + threadIdx.x], v[0] + v[1] + v[2] + v[3]);
        }
        __syncthreads();
    }
}

// State of the reference's `threshold_list` (track_eval.py:120, 239-242) between batches:
// ring[g % window] = level of frame g for the last `window` frames, then the number of frames seen.
struct LevelState {
    double *ring;
    long long *seen;
};

// One block.  stats[f] = {sum, sum of squares} as u64 on entry;
// {mean, stddev, level of this frame, averaged integer level} as f64 on exit.
constexpr int LEVEL_LDS = 6000;   // doubles: the list before the batch (< window) followed by the batch's levels
__global__ __launch_bounds__(256) void k_mean_levels(double *stats, int batch, double n_px, int inv, double offset,
                                                     int window, LevelState st, int32_t *levels)
{
    __shared__ double s_seq[LEVEL_LDS];
    const long long seen = *st.seen;
    const int before = (int)(seen < window - 1 ? seen : window - 1);   // list entries the batch's averages can reach
    const bool staged = before + batch <= LEVEL_LDS;
    if (staged)
        for (int i = threadIdx.x; i < before; i += 256) s_seq[i] = st.ring[(seen - before + i) % window];
    for (int f = threadIdx.x; f < batch; f += 256) {
        const unsigned long long s = (unsigned long long)__double_as_longlong(stats[4 * (size_t)f + 0]);
        const unsigned long long q = (unsigned long long)__double_as_longlong(stats[4 * (size_t)f + 1]);
        // cv2.meanStdDev: scale = 1/N; mean = s*scale; stddev = sqrt(max(sq*scale - mean*mean, 0))
        const double scale = 1.0 / n_px;
        const double mean = (double)s * scale;
        const double var = (double)q * scale - mean * mean;
        const double sd = sqrt(var > 0.0 ? var : 0.0);
        // track_eval.py:222-229; `offset` arrives with the sign it has at that point (negated at :132
        // for dark-on-bright videos)
        const double level = inv ? (mean - sd) - offset : (mean + sd) + offset;
        stats[4 * (size_t)f + 0] = mean;
        stats[4 * (size_t)f + 1] = sd;
        stats[4 * (size_t)f + 2] = level;
        if (staged) s_seq[before + f] = level;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < batch; f += 256) {
        // int(sum(threshold_list) / len(threshold_list)) with the list as it is after this frame's append:
        // the last min(g + 1, window) levels, summed oldest first (Python's sum starts from int 0)
        const long long g = seen + f;
        const int cnt = (int)(g + 1 < window ? g + 1 : window);
        double acc = 0.0;
        if (staged) {
            const double *v = s_seq + (before + f - cnt + 1);
#pragma unroll 8
            for (int j = 0; j < cnt; ++j) acc += v[j];
        } else {
            for (long long j = g - cnt + 1; j <= g; ++j)
                acc += j >= seen ? stats[4 * (size_t)(j - seen) + 2] : st.ring[j % window];
        }
        const double avg = acc / (double)cnt;
        double t = trunc(avg);
        t = t < -1.0 ? -1.0 : (t > 256.0 ? 256.0 : t);   // beyond [0, 255] every value behaves the same
        levels[f] = (int32_t)t;
        stats[4 * (size_t)f + 3] = t;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < batch; f += 256)
        if (f >= batch - window) st.ring[(seen + f) % window] = stats[4 * (size_t)f + 2];
    if (threadIdx.x == 0) *st.seen = seen + batch;
}

struct LevelGeo {
    int H, W, batch, groups_x, segs_y;
    int inv;
};

// horizontal 1-2-1 sums of 4 pixels as 2 x (16-bit, 16-bit); left/right are the neighbour pixels
__device__ __forceinline__ void hsum4(uint32_t g, uint32_t left, uint32_t right, uint32_t &a, uint32_t &b)
{
    const uint32_t p0 = left | ((g & 0xFFu) << 16);                        // (b-1, b0)
    const uint32_t p1 = __builtin_amdgcn_perm(g, g, 0x0C010C00u);          // (b0, b1)
    const uint32_t p2 = __builtin_amdgcn_perm(g, g, 0x0C020C01u);          // (b1, b2)
    const uint32_t p3 = __builtin_amdgcn_perm(g, g, 0x0C030C02u);          // (b2, b3)
    const uint32_t p4 = (g >> 24) | (right << 16);                         // (b3, b4)
    a = p0 + (p1 << 1) + p2;
    b = p2 + (p3 << 1) + p4;
}

// One thread: 4 adjacent columns x LEVEL_SEG rows, sliding a 3-row window of horizontal sums.
template <int CH>
__global__ __launch_bounds__(256) void k_level_threshold(const uint8_t *__restrict__ frames, uint8_t *__restrict__ cls,
                                                         LevelGeo G, const int32_t *__restrict__ levels, ysmr::GrayCoef gc)
{
    const int H = G.H, W = G.W;
    const long long per_frame = (long long)G.groups_x * G.segs_y;
    const long long items = per_frame * G.batch;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int f = (int)(it / per_frame);
        const int rem = (int)(it - (long long)f * per_frame);
        const int sy = rem / G.groups_x, gx = rem - sy * G.groups_x;
        const int c = 4 * gx, y0 = sy * LEVEL_SEG, y1 = min(y0 + LEVEL_SEG, H);
        const uint8_t *frame = frames + (size_t)f * H * W * CH;
        uint8_t *out = cls + (size_t)f * H * W;
        const int level = levels[f];
        const bool whole = c + 3 < W;                 // all 4 columns exist
        const bool inner = c >= 1 && c + 4 < W;       // ... and so do both neighbours
        const int cl = border_index(c - 1, W), cr = border_index(c + 4, W);

        // a row as loaded; turned into gray / horizontal sums only when it is consumed, so that the
        // loads of the next LEVEL_PF rows stay in flight meanwhile
        struct Raw { uint32_t w0, w1, w2, left, right; };
        auto load_row = [&](int r) -> Raw {
            const uint32_t base = (uint32_t)r * (uint32_t)W;
            Raw v{0, 0, 0, 0, 0};
            if (inner) {
                const uint8_t *q = frame + (size_t)(base + c) * CH;
                v.w0 = load_u32(q);
                if constexpr (CH == 3) { v.w1 = load_u32(q + 4); v.w2 = load_u32(q + 8); }
                v.left = gray_at<CH>(gc, frame, base + c - 1);
                v.right = gray_at<CH>(gc, frame, base + c + 4);
            } else {
                // border groups: columns beyond the image take their BORDER_REFLECT_101 source (columns
                // that only exist as padding of a partial group are never stored)
                for (int o = 0; o < 4; ++o)
                    v.w0 |= gray_at<CH>(gc, frame, base + border_index(c + o, W)) << (8 * o);
                v.left = gray_at<CH>(gc, frame, base + cl);
                v.right = gray_at<CH>(gc, frame, base + cr);
            }
            return v;
        };
        auto row_sums = [&](const Raw &v, uint32_t &a, uint32_t &b) {
            uint32_t g = v.w0;
            if constexpr (CH == 3) {
                if (inner) {
                    const uint32_t g0 = bgr2gray(gc, v.w0 & 0xFFu, (v.w0 >> 8) & 0xFFu, (v.w0 >> 16) & 0xFFu);
                    const uint32_t g1 = bgr2gray(gc, v.w0 >> 24, v.w1 & 0xFFu, (v.w1 >> 8) & 0xFFu);
                    const uint32_t g2 = bgr2gray(gc, (v.w1 >> 16) & 0xFFu, v.w1 >> 24, v.w2 & 0xFFu);
                    const uint32_t g3 = bgr2gray(gc, (v.w2 >> 8) & 0xFFu, (v.w2 >> 16) & 0xFFu, v.w2 >> 24);
                    g = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
                }
            }
            hsum4(g, v.left, v.right, a, b);
        };

        uint32_t ua, ub, ca, cb, da, db;
        row_sums(load_row(border_index(y0 - 1, H)), ua, ub);
        row_sums(load_row(y0), ca, cb);
        Raw ahead[LEVEL_PF];
#pragma unroll
        for (int d = 0; d < LEVEL_PF; ++d) ahead[d] = load_row(border_index(min(y0 + 1 + d, H), H));
        for (int yb = y0; yb < y1; yb += LEVEL_PF) {
#pragma unroll
        for (int slot = 0; slot < LEVEL_PF; ++slot) {   // (compile-time slots: see k_level_strip)
            const int y = yb + slot;
            if (y >= y1) break;
            row_sums(ahead[slot], da, db);
            ahead[slot] = load_row(border_index(min(y + 1 + LEVEL_PF, H), H));
            const uint32_t ta = (ua + (ca << 1) + da + 0x00080008u) >> 4;   // GaussianBlur 3x3: (sum + 8) >> 4
            const uint32_t tb = (ub + (cb << 1) + db + 0x00080008u) >> 4;
            const int b0 = ta & 0xFFu, b1 = (ta >> 16) & 0xFFu, b2 = tb & 0xFFu, b3 = (tb >> 16) & 0xFFu;
            // cv2.threshold: THRESH_BINARY sets src > level, THRESH_BINARY_INV the others
            const uint32_t o0 = ((b0 > level) != (G.inv != 0)) ? 3u : 0u, o1 = ((b1 > level) != (G.inv != 0)) ? 3u : 0u;
            const uint32_t o2 = ((b2 > level) != (G.inv != 0)) ? 3u : 0u, o3 = ((b3 > level) != (G.inv != 0)) ? 3u : 0u;
            uint8_t *dst = out + (uint32_t)y * (uint32_t)W + c;
            if (whole) {
                const uint32_t v = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
                __builtin_memcpy(dst, &v, 4);
            } else {
                dst[0] = (uint8_t)o0;
                if (c + 1 < W) dst[1] = (uint8_t)o1;
                if (c + 2 < W) dst[2] = (uint8_t)o2;
            }
            ua = ca; ub = cb; ca = da; cb = db;
        }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_level_strip: the same result for W % 4 == 0, W >= 8 with the data flow of detect.hip's
// k_threshold_strip: one WAVE owns a vertical strip, lane L holds the dword of 4 pixels at columns
// xs - 4 + 4L .. +3 (lanes 0 and 63 are halo), rows are read once with coalesced dword loads, the
// horizontal neighbours come from the adjacent lanes (DPP), three rows stay in flight.
// (k_level_threshold above issues three loads per 4 pixels and is bound by the texture path.)
// ------------------------------------------------------------------------------------------
constexpr int LSTRIP_PF = 8;   // rows in flight per lane: the kernel is a stream, it needs ~6 MB in flight to cover the HBM latency
constexpr int LSTRIP_OUT_LANES = 62;

struct LevelStrip {
    int H, W, batch, out_lanes, strips_x, seg_h, segs_y, inv;
};

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v)   // value of lane-1 (0 into lane 0)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v)   // value of lane+1 (0 into lane 63)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

template <int CH>
__global__ __launch_bounds__(256) void k_level_strip(const uint8_t *__restrict__ frames, uint8_t *__restrict__ cls,
                                                     LevelStrip P, const int32_t *__restrict__ levels, ysmr::GrayCoef gc)
{
    // everything derived from the wave index is wave-uniform: say so, or row counters end up in VGPRs
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int H = P.H, W = P.W;
    const int per_frame = P.strips_x * P.segs_y;
    const long long n_items = (long long)P.batch * per_frame;
    for (long long item = (long long)blockIdx.x * 4 + wave_in_block; item < n_items; item += (long long)gridDim.x * 4) {
        const int f = (int)(item / per_frame);
        const int rem = (int)(item - (long long)f * per_frame);
        const int sx = rem % P.strips_x, sy = rem / P.strips_x;
        const int xs = sx * P.out_lanes * 4, c0 = xs - 4 + 4 * lane;
        const int y0 = sy * P.seg_h, y1 = min(y0 + P.seg_h, H);
        const uint8_t *src = frames + (size_t)f * H * W * CH;
        uint8_t *dst = cls + (size_t)f * H * W;
        const uint32_t ld_col = (uint32_t)(c0 < 0 ? 0 : (c0 > W - 4 ? W - 4 : c0));   // lanes outside load a valid address
        const bool left_edge = xs == 0;                       // lane 0 holds columns -4 .. -1
        const int last_col_rel = (W - 1) - (xs - 4);
        const bool right_edge = last_col_rel < 252;           // column W falls inside this wave's lanes
        const int edge_lane = __builtin_amdgcn_readfirstlane(min(last_col_rel >> 2, 62));   // holds column W-1 in its byte 3
        const bool writes = lane >= 1 && lane <= P.out_lanes && c0 < W;
        const int level = min(levels[f], 255);                          // (-1 .. 255; nothing exceeds 255)
        const uint32_t above = (uint32_t)(255 - level) * 0x00010001u;   // per field: 0 .. 256
        const uint32_t flip = P.inv ? 0x03030303u : 0u;                 // THRESH_BINARY_INV sets the other pixels

        struct Raw { uint32_t w0, w1, w2; };
        auto load_row = [&](int r) -> Raw {
            const uint8_t *q = src + ((size_t)((uint32_t)r * (uint32_t)W + ld_col)) * CH;
            Raw v{load_u32(q), 0, 0};
            if constexpr (CH == 3) { v.w1 = load_u32(q + 4); v.w2 = load_u32(q + 8); }
            return v;
        };
        // horizontal 1-2-1 sums of the lane's 4 pixels (converted and border-patched when the row is consumed)
        auto row_sums = [&](const Raw &v, uint32_t &a, uint32_t &b) {
            uint32_t g = v.w0;
            if constexpr (CH == 3) {
                const uint32_t g0 = bgr2gray(gc, v.w0 & 0xFFu, (v.w0 >> 8) & 0xFFu, (v.w0 >> 16) & 0xFFu);
                const uint32_t g1 = bgr2gray(gc, v.w0 >> 24, v.w1 & 0xFFu, (v.w1 >> 8) & 0xFFu);
                const uint32_t g2 = bgr2gray(gc, (v.w1 >> 16) & 0xFFu, v.w1 >> 24, v.w2 & 0xFFu);
                const uint32_t g3 = bgr2gray(gc, (v.w2 >> 8) & 0xFFu, (v.w2 >> 16) & 0xFFu, v.w2 >> 24);
                g = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
            }
            if (left_edge) {    // column -1 (byte 3 of lane 0) := column 1 (byte 1 of lane 1)
                const uint32_t g1 = (uint32_t)__builtin_amdgcn_readlane((int)g, 1);
                if (lane == 0) g = (g1 << 16) & 0xFF000000u;
            }
            if (right_edge) {   // column W (byte 0 of lane edge+1) := column W-2 (byte 2 of lane edge)
                const uint32_t ge = (uint32_t)__builtin_amdgcn_readlane((int)g, edge_lane);
                if (lane == edge_lane + 1) g = (ge >> 16) & 0xFFu;
            }
            const uint32_t gp = wave_shr1(g), gn = wave_shl1(g);
            hsum4(g, gp >> 24, gn & 0xFFu, a, b);
        };

        uint32_t ua, ub, ca, cb, da, db;
        row_sums(load_row(border_index(y0 - 1, H)), ua, ub);
        row_sums(load_row(y0), ca, cb);
        Raw ahead[LSTRIP_PF];
#pragma unroll
        for (int d = 0; d < LSTRIP_PF; ++d) {
            ahead[d] = load_row(border_index(min(y0 + 1 + d, H), H));
            // keep the issue order = the order of use: the wait for slot 0 at the top of the loop below is
            // the merge of this path and the back edge, and a last-issued slot 0 would turn it into vmcnt(0)
            asm volatile("" ::: "memory");
        }
        // (slot indices are compile-time constants: shifting the queue with moves instead would make every
        // move wait for the load it carries; whole groups run without guards so that the compiler's wait
        // counts stay exact, the last < LSTRIP_PF rows find their sources already in the queue)
        auto emit = [&](int y) {
            const uint32_t ta = ((ua + (ca << 1) + da + 0x00080008u) >> 4) & 0x00FF00FFu;   // GaussianBlur 3x3: (sum + 8) >> 4
            const uint32_t tb = ((ub + (cb << 1) + db + 0x00080008u) >> 4) & 0x00FF00FFu;
            // cv2.threshold on both 16-bit fields at once: t > level  <=>  bit 8 of t + (255 - level)
            const uint32_t ma = ((ta + above) >> 8) & 0x00010001u, mb = ((tb + above) >> 8) & 0x00010001u;
            // (ma | mb << 8) holds pixels 0, 2, 1, 3 in its bytes: swap the middle ones; x3 = thresh and marker bit
            const uint32_t set = __builtin_amdgcn_perm(0u, ma | (mb << 8), 0x03010200u) * 3u;
            if (writes)
                *reinterpret_cast<uint32_t *>(dst + ((uint32_t)y * (uint32_t)W + (uint32_t)c0)) = set ^ flip;
            ua = ca; ub = cb; ca = da; cb = db;
        };
        int y = y0;
        for (; y + LSTRIP_PF <= y1; y += LSTRIP_PF) {
#pragma unroll
            for (int slot = 0; slot < LSTRIP_PF; ++slot) {
                row_sums(ahead[slot], da, db);
                ahead[slot] = load_row(border_index(min(y + slot + 1 + LSTRIP_PF, H), H));
                emit(y + slot);
            }
        }
#pragma unroll
        for (int slot = 0; slot < LSTRIP_PF - 1; ++slot) {
            if (y + slot < y1) {
                row_sums(ahead[slot], da, db);
                emit(y + slot);
            }
        }
    }
}

}  // namespace

extern "C" {

size_t ysmr_mean_threshold_state_bytes(int window)
{
    return window > 0 ? sizeof(double) * (size_t)window + sizeof(long long) : 0;
}

int ysmr_mean_threshold_batch(void *stream, const uint8_t *frames_dev, int batch, int height, int width, int channels,
                              int inv, double offset, int window, void *state_dev, double *stats_dev,
                              int32_t *levels_dev, uint8_t *cls_dev, int cv_flavour)
{
    if (cv_flavour & ~YSMR_CV_FLAVOUR_MASK) return ysmr::fail(YSMR_ERR_ARG, "unknown cv_flavour bits 0x%x", cv_flavour);
    const ysmr::GrayCoef gc = ysmr::gray_coef(cv_flavour);
    if (batch <= 0 || height <= 0 || width <= 0)
        return ysmr::fail(YSMR_ERR_ARG, "batch, height, width must be positive (got %d, %d, %d)", batch, height, width);
    if (channels != 1 && channels != 3) return ysmr::fail(YSMR_ERR_ARG, "channels must be 1 (gray) or 3 (BGR), got %d", channels);
    if (height > 16384 || width > 16384 || (size_t)batch * height * width > (1ull << 32) - 64)
        return ysmr::fail(YSMR_ERR_ARG, "batch too large (height and width are limited to 16384, batch*height*width to 2^32 - 1)");
    if (window < 1) return ysmr::fail(YSMR_ERR_ARG, "window must be >= 1 (got %d)", window);
    if (!frames_dev || !state_dev || !stats_dev || !levels_dev || !cls_dev)
        return ysmr::fail(YSMR_ERR_ARG, "a required device pointer is NULL");
    if (((uintptr_t)state_dev & 7) || ((uintptr_t)stats_dev & 7))
        return ysmr::fail(YSMR_ERR_ARG, "state_dev and stats_dev must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t HW = (uint32_t)height * (uint32_t)width;
    YSMR_HIP_CHECK(hipMemsetAsync(stats_dev, 0, sizeof(double) * 4 * (size_t)batch, st));
    // k_gray_sums: enough parts per frame to fill the resident grid, each at least 4096 pixels
    int parts = (MG_BLOCKS + batch - 1) / batch;
    const int max_parts = (int)((HW + 4095u) / 4096u);
    parts = parts < 1 ? 1 : (parts > max_parts ? max_parts : parts);
    const long long sum_items = (long long)batch * parts;
    const unsigned sum_blocks = (unsigned)(sum_items < MG_BLOCKS ? sum_items : MG_BLOCKS);
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(stats_dev);
    if (channels == 1) hipLaunchKernelGGL(k_gray_sums<1>, dim3(sum_blocks), dim3(256), 0, st, frames_dev, HW, batch, parts, sums, 4, gc);
    else hipLaunchKernelGGL(k_gray_sums<3>, dim3(sum_blocks), dim3(256), 0, st, frames_dev, HW, batch, parts, sums, 4, gc);
    LevelState ls{reinterpret_cast<double *>(state_dev), reinterpret_cast<long long *>(reinterpret_cast<double *>(state_dev) + window)};
    hipLaunchKernelGGL(k_mean_levels, dim3(1), dim3(256), 0, st, stats_dev, batch, (double)HW, inv, offset, window, ls, levels_dev);
    if ((width & 3) == 0 && width >= 8 && (((uintptr_t)cls_dev | (uintptr_t)frames_dev) & 3) == 0) {
        LevelStrip P;
        P.H = height; P.W = width; P.batch = batch; P.inv = inv;
        const int quads = width / 4;
        P.strips_x = (quads + LSTRIP_OUT_LANES - 1) / LSTRIP_OUT_LANES;
        P.out_lanes = (quads + P.strips_x - 1) / P.strips_x;
        // rows per item: the tallest segment <= 64 rows for which the items fill the resident waves a whole
        // number of times (a last, nearly empty round costs as much as a full one)
        const int resident = 768;            // 3 blocks per CU, as k_threshold_strip
        const long long columns = (long long)batch * P.strips_x;
        P.seg_h = 32;
        for (int k = 1; k <= 16; ++k) {
            const long long segs = (long long)k * resident * 4 / columns;
            if (segs < 1) continue;
            const int h = (int)((height + segs - 1) / segs);
            if (h <= 64) { P.seg_h = h < 8 ? 8 : h; break; }
        }
        P.segs_y = (height + P.seg_h - 1) / P.seg_h;
        const long long waves = columns * P.segs_y;
        const long long want = (waves + 3) / 4;
        const unsigned blocks = (unsigned)(want < resident ? want : resident);
        if (channels == 1) hipLaunchKernelGGL(k_level_strip<1>, dim3(blocks), dim3(256), 0, st, frames_dev, cls_dev, P, levels_dev, gc);
        else hipLaunchKernelGGL(k_level_strip<3>, dim3(blocks), dim3(256), 0, st, frames_dev, cls_dev, P, levels_dev, gc);
    } else {
        LevelGeo G{height, width, batch, (width + 3) / 4, (height + LEVEL_SEG - 1) / LEVEL_SEG, inv};
        const long long thr_items = (long long)G.groups_x * G.segs_y * batch;
        const long long want = (thr_items + 255) / 256;
        const unsigned thr_blocks = (unsigned)(want < MG_BLOCKS ? want : MG_BLOCKS);
        if (channels == 1) hipLaunchKernelGGL(k_level_threshold<1>, dim3(thr_blocks), dim3(256), 0, st, frames_dev, cls_dev, G, levels_dev, gc);
        else hipLaunchKernelGGL(k_level_threshold<3>, dim3(thr_blocks), dim3(256), 0, st, frames_dev, cls_dev, G, levels_dev, gc);
    }
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

}  // extern "C"
