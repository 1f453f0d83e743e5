// libysmr_hip.so -- select_tracks / find_good_tracks (ysmr/track_eval.py:408-843) on the sorted table:
// the consumer of the detect-and-link path's output.  Input: the columns of the
// (TRACK_ID, POSITION_T)-ordered table in device memory.  Output: which rows the reference's
// selection keeps (and their index in the cleaned table, the 'index' column of its result).
//
//   clean-up   area = W*H; per-track median (two stable radix sorts: by area, then by track);
//              rows of tracks whose median area is out of bounds, rows above factor x median, rows of
//              area 0, rows of tracks shorter than the minimum -> dropped (stream compaction)
//   bounds     area quantiles q / 1-q and the 25 / 75 % distance quartiles by a radix sort each and
//              numpy's linear interpolation; distance outliers beyond the outer fence
//   tracks     one WAVE per track walks the reference's recursion (split at the largest hole / the
//              first distance outlier) with an explicit stack; hole / outlier scans are wave-parallel;
//              the four means of a surviving segment reproduce numpy's pairwise summation bit for bit
//              (leaf blocks of <= 128 values with 8 accumulators each on separate lanes, then the
//              recursion's own combination order)
//   output     rows of each track's longest surviving segment (optionally cut at the length limit)
//
// The call is synchronous (it sizes its later stages from counts it reads back); it runs once per
// video.  All grids are resident-sized.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"
#include "prim.h"

namespace {

constexpr int SEL_BLOCKS = 1024;
constexpr int TRACK_BLOCKS = 256;           // k_sel_tracks: 4 waves per block, one track per wave at a time

__device__ __forceinline__ long long gtid() { return (long long)blockIdx.x * 256 + threadIdx.x; }
__device__ __forceinline__ long long gstride() { return (long long)gridDim.x * 256; }

// ---- clean-up ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sel_area(const uint32_t *__restrict__ id, const double *__restrict__ w,
                                                  const double *__restrict__ h, long long n, double *__restrict__ area,
                                                  uint32_t *__restrict__ start)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        area[i] = w[i] * h[i];
        start[i] = (i == 0 || id[i] != id[i - 1]) ? 1u : 0u;
    }
}

// seg[i] = index of row i's track; first/last row of every track
__global__ __launch_bounds__(256) void k_sel_segments(const uint32_t *__restrict__ start, const uint32_t *__restrict__ incl,
                                                      long long n, uint32_t *__restrict__ seg, uint32_t *__restrict__ first,
                                                      uint32_t *__restrict__ last)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const uint32_t s = incl[i] - 1u;
        seg[i] = s;
        if (start[i]) first[s] = (uint32_t)i;
        if (i == n - 1 || start[i + 1]) last[s] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_sel_bits(const double *__restrict__ v, long long n, unsigned long long *__restrict__ bits)
{
    // (non-negative doubles order like their bit patterns)
    for (long long i = gtid(); i < n; i += gstride()) bits[i] = (unsigned long long)__double_as_longlong(v[i]);
}

// pandas groupby median (median_linear): n odd -> a[n/2]; n even -> (a[n/2] + a[n/2 - 1]) / 2
__global__ __launch_bounds__(256) void k_sel_median(const unsigned long long *__restrict__ sorted_bits,
                                                    const uint32_t *__restrict__ first, const uint32_t *__restrict__ last,
                                                    uint32_t n_tracks, double *__restrict__ median)
{
    for (long long s = gtid(); s < n_tracks; s += gstride()) {
        const uint32_t f = first[s], n = last[s] - f + 1u;
        const double hi = __longlong_as_double((long long)sorted_bits[f + n / 2]);
        median[s] = (n & 1u) ? hi : (hi + __longlong_as_double((long long)sorted_bits[f + n / 2 - 1])) / 2.0;
    }
}

struct CleanParams {
    double lo, hi, factor;
    int use_factor;
    uint32_t min_len;
};

__global__ __launch_bounds__(256) void k_sel_keep(const double *__restrict__ area, const uint32_t *__restrict__ seg,
                                                  const double *__restrict__ median, const uint32_t *__restrict__ first,
                                                  const uint32_t *__restrict__ last, const uint32_t *__restrict__ t,
                                                  long long n, CleanParams p, uint32_t *__restrict__ keep)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const uint32_t s = seg[i];
        const double a = area[i], m = median[s];
        // (last - first + 1).astype(np.uint16): the subtraction runs in uint32, the cast keeps 16 bits
        const uint32_t length = (uint32_t)(uint16_t)(t[last[s]] - t[first[s]] + 1u);
        bool ok = m >= p.lo && m <= p.hi;
        if (p.use_factor) ok = ok && a <= m * p.factor;
        ok = ok && a != 0.0 && length >= p.min_len;
        keep[i] = ok ? 1u : 0u;
    }
}

struct Table {   // the cleaned table (reset_index): row i2 <- original row orig[i2]
    uint32_t *id, *t, *orig;
    double *x, *y, *area, *ratio, *dist;
};

__global__ __launch_bounds__(256) void k_sel_compact(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos,
                                                     const uint32_t *__restrict__ id, const uint32_t *__restrict__ t,
                                                     const double *__restrict__ x, const double *__restrict__ y,
                                                     const double *__restrict__ w, const double *__restrict__ h,
                                                     const double *__restrict__ area, long long n, Table o)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        if (!keep[i]) continue;
        const uint32_t j = pos[i] - 1u;   // inclusive scan
        o.id[j] = id[i]; o.t[j] = t[i]; o.orig[j] = (uint32_t)i;
        o.x[j] = x[i]; o.y[j] = y[i]; o.area[j] = area[i];
        // short side / long side (np.where(HEIGHT <= WIDTH, HEIGHT / WIDTH, WIDTH / HEIGHT))
        o.ratio[j] = (h[i] <= w[i]) ? h[i] / w[i] : w[i] / h[i];
    }
}

// start flags of the cleaned table + per-row distance: sqrt(dx^2 + dy^2) / dT, 0 at a track's first row
__global__ __launch_bounds__(256) void k_sel_dist(Table c, long long n, uint32_t *__restrict__ start)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const bool st = i == 0 || c.id[i] != c.id[i - 1];
        start[i] = st ? 1u : 0u;
        double d = 0.0;
        if (!st) {
            const double dx = c.x[i] - c.x[i - 1], dy = c.y[i] - c.y[i - 1];
            d = sqrt(dx * dx + dy * dy) / ((double)c.t[i] - (double)c.t[i - 1]);
        }
        c.dist[i] = d;
    }
}

// ---- bounds -------------------------------------------------------------------------------------
struct Bounds {
    double area_lo, area_hi;        // quantiles of the area (or -1 / inf)
    double q1_dist, q3_dist, fence; // distance quartiles and outer fence
    unsigned long long outliers;    // rows beyond the fence
    int use_flags;                  // 0: outlier exclusion off (setting, or too many outliers)
    int pad;
};

// numpy's 'linear' percentile: virtual index (n - 1) * q, _lerp between the two neighbours
__device__ double quantile_linear(const unsigned long long *sorted_bits, long long n, double q)
{
    const double vi = (double)(n - 1) * q;
    double lo = floor(vi);
    long long i0 = (long long)lo, i1 = i0 + 1;
    if (i0 < 0) i0 = 0;
    if (i0 > n - 1) i0 = n - 1;
    if (i1 > n - 1) i1 = n - 1;
    const double g = vi - lo;
    const double a = __longlong_as_double((long long)sorted_bits[i0]), b = __longlong_as_double((long long)sorted_bits[i1]);
    const double diff = b - a;
    return g >= 0.5 ? b - diff * (1.0 - g) : a + diff * g;
}

__global__ void k_sel_bounds(const unsigned long long *sorted_area, const unsigned long long *sorted_dist, long long n,
                             double q_lo, double q_hi, int use_area, int use_dist, Bounds *b)
{
    if (threadIdx.x || blockIdx.x) return;
    b->area_lo = use_area ? quantile_linear(sorted_area, n, q_lo) : -1.0;
    b->area_hi = use_area ? quantile_linear(sorted_area, n, q_hi) : INFINITY;
    b->q1_dist = b->q3_dist = b->fence = 0.0;
    if (use_dist) {
        b->q1_dist = quantile_linear(sorted_dist, n, 0.25);
        b->q3_dist = quantile_linear(sorted_dist, n, 0.75);
        b->fence = (b->q3_dist - b->q1_dist) * 3.0 + b->q3_dist;
    }
    b->outliers = 0;
    b->use_flags = 0;
}

__global__ __launch_bounds__(256) void k_sel_flags(const double *__restrict__ dist, long long n, Bounds *b, uint8_t *__restrict__ flag)
{
    const double fence = b->fence;
    unsigned long long mine = 0;
    for (long long i = gtid(); i < n; i += gstride()) {
        const uint8_t f = dist[i] > fence ? 1 : 0;
        flag[i] = f;
        mine += f;
    }
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&b->outliers, mine);
}

__global__ void k_sel_decide(Bounds *b, long long n, double stop_fraction)
{
    if (threadIdx.x || blockIdx.x) return;
    const double percent = (double)b->outliers / (double)n;
    b->use_flags = percent > stop_fraction ? 0 : 1;
}

// ---- tracks -------------------------------------------------------------------------------------
struct TrackParams {
    uint32_t min_len;          // minimal_length_frames
    int max_holes;             // maximal consecutive holes
    double max_empty;          // duration / size must stay below this
    double ratio_min, ratio_max;
    double y_lo, y_hi, x_lo, x_hi;   // edge * H, (1 - edge) * H, edge * W, (1 - edge) * W
    int check_frame;           // edge != 0: positions must stay inside the frame
    double frame_w, frame_h;
    int max_recursion;
    uint32_t limit_frames;     // 0 = no limit
    int limit_exact;
    int stack_cap;
};

struct TrackScratch {
    int4 *stack;               // [waves][stack_cap]: start, stop, depth
    int2 *leaf;                // [n / 64 + tracks + 2]: offset, length of a leaf block
    double *leaf_sum;          // [4][same]
    long long leaf_slots;
};

struct TrackOut {
    uint8_t *good;             // [n] zero on entry
    unsigned long long *kicks; // [9]
    unsigned long long *n_good;
    int *error;
};

// sum of one leaf block exactly as numpy's pairwise_sum does it (n <= 128)
__device__ double leaf_block_sum(const double *a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

// The four means (area, ratio, y, x) of rows [lo, lo + n) by one wave, in numpy's summation order.
// Lane 0 lists the leaf blocks of the pairwise recursion, the lanes sum them, lanes 0-3 combine one
// column each in the recursion's order.  `slot0` is this wave's region of the leaf scratch.
__device__ void segment_means(const Table &c, uint32_t lo, int n, const TrackScratch &sc, long long slot0, int lane, double out[4])
{
    int2 *leaf = sc.leaf + slot0;
    int n_leaf = 0;
    if (lane == 0) {
        // pre-order walk: pieces of > 128 values split at n/2 rounded down to a multiple of 8
        int off_stack[40], len_stack[40], sp = 0;
        off_stack[0] = 0; len_stack[0] = n; sp = 1;
        while (sp) {
            --sp;
            const int off = off_stack[sp], len = len_stack[sp];
            if (len <= 128) { leaf[n_leaf++] = make_int2(off, len); continue; }
            int half = len / 2;
            half -= half % 8;
            off_stack[sp] = off + half; len_stack[sp] = len - half; ++sp;   // right piece: visited second
            off_stack[sp] = off; len_stack[sp] = half; ++sp;
        }
    }
    n_leaf = __shfl(n_leaf, 0);
    __threadfence_block();
    const double *cols[4] = {c.area + lo, c.ratio + lo, c.y + lo, c.x + lo};
    for (int k = lane; k < n_leaf; k += 64) {
        const int2 l = leaf[k];
#pragma unroll
        for (int col = 0; col < 4; ++col) sc.leaf_sum[col * sc.leaf_slots + slot0 + k] = leaf_block_sum(cols[col] + l.x, l.y);
    }
    __threadfence_block();
    double total = 0.0;
    if (lane < 4) {
        // post-order combination = the recursion's "left + right"; the leaves come in visiting order
        const double *sums = sc.leaf_sum + lane * sc.leaf_slots + slot0;
        int len_stack[40], state[40], sp = 0, next = 0;
        double val[40];
        int vp = 0;
        len_stack[0] = n; state[0] = 0; sp = 1;
        while (sp) {
            const int len = len_stack[sp - 1];
            if (len <= 128) { val[vp++] = sums[next++]; --sp; continue; }
            int half = len / 2;
            half -= half % 8;
            if (state[sp - 1] == 0) { state[sp - 1] = 1; len_stack[sp] = half; state[sp] = 0; ++sp; }
            else if (state[sp - 1] == 1) { state[sp - 1] = 2; len_stack[sp] = len - half; state[sp] = 0; ++sp; }
            else { val[vp - 2] = val[vp - 2] + val[vp - 1]; --vp; --sp; }
        }
        total = val[0];
    }
    const double count = (double)n;
#pragma unroll
    for (int col = 0; col < 4; ++col) out[col] = __shfl(total, col) / count;
}

__global__ __launch_bounds__(256) void k_sel_tracks(Table c, const uint8_t *__restrict__ flag, const uint32_t *__restrict__ first,
                                                    const uint32_t *__restrict__ last, uint32_t n_tracks, TrackParams p,
                                                    const Bounds *__restrict__ bounds, TrackScratch sc, TrackOut out)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int n_waves = gridDim.x * 4;
    int4 *stack = sc.stack + (size_t)wave * p.stack_cap;
    const Bounds b = *bounds;
    const bool use_flags = b.use_flags != 0;
    for (uint32_t k = (uint32_t)wave; k < n_tracks; k += (uint32_t)n_waves) {
        const int t_first = (int)first[k], t_last = (int)last[k];
        int kick = 8;
        int best_start = -1, best_stop = -1, best_len = 0;
        int sp = 0;
        if (lane == 0) stack[0] = make_int4(t_first, t_last, 0, 0);
        sp = 1;
        __threadfence_block();
        while (sp > 0) {
            --sp;
            const int4 node = stack[sp];
            const int start = node.x, stop = node.y, depth = node.z;
            const int size = stop - start + 1;
            int own = 8;
            int sub_a0 = 0, sub_a1 = -1, sub_b0 = 0, sub_b1 = -1;   // sub-parts (empty when x1 < x0)
            bool split = false;
            if ((uint32_t)size >= p.min_len) {
                own = 7;
                // largest step of POSITION_T inside the slice (first row has none) and where it first occurs;
                // number of flagged rows and the first of them
                int hole = -1, hole_at = 0x7FFFFFFF, n_flag = 0, flag_at = 0x7FFFFFFF;
                for (int i = start + lane; i <= stop; i += 64) {
                    if (i > start) {
                        const int d = (int)(c.t[i] - c.t[i - 1]);
                        if (d > hole) { hole = d; hole_at = i; }
                    }
                    if (use_flags && flag[i]) { ++n_flag; if (i < flag_at) flag_at = i; }
                }
                for (int d = 32; d >= 1; d >>= 1) {
                    const int oh = __shfl_xor(hole, d), oa = __shfl_xor(hole_at, d);
                    if (oh > hole || (oh == hole && oa < hole_at)) { hole = oh; hole_at = oa; }
                    n_flag += __shfl_xor(n_flag, d);
                    flag_at = min(flag_at, __shfl_xor(flag_at, d));
                }
                if (size >= 2 && hole <= p.max_holes) {
                    own = 6;
                    if (n_flag == 0) {
                        own = 5;
                        const uint32_t duration = c.t[stop] - c.t[start] + 1u;
                        if ((double)duration / (double)size < p.max_empty) {
                            own = 4;
                            double mean[4];
                            segment_means(c, (uint32_t)start, size, sc, (long long)start / 64 + k, lane, mean);
                            if (b.area_lo <= mean[0] && mean[0] <= b.area_hi) {
                                own = 3;
                                if (p.ratio_min < mean[1] && mean[1] < p.ratio_max) {
                                    own = 2;
                                    if (p.y_lo < mean[2] && mean[2] < p.y_hi && p.x_lo < mean[3] && mean[3] < p.x_hi) {
                                        own = 1;
                                        bool inside = true;
                                        if (p.check_frame) {
                                            bool bad = false;
                                            for (int i = start + lane; i <= stop; i += 64)
                                                bad = bad || c.x[i] < 0.0 || c.x[i] > p.frame_w || c.y[i] < 0.0 || c.y[i] > p.frame_h;
                                            inside = __ballot(bad) == 0ull;
                                        }
                                        if (inside) {
                                            own = 0;
                                            if (size > best_len) { best_len = size; best_start = start; best_stop = stop; }
                                        }
                                    }
                                }
                            }
                        }
                    } else {          // split around the first outlier (it is dropped)
                        split = true;
                        sub_a0 = start; sub_a1 = flag_at - 1; sub_b0 = flag_at + 1; sub_b1 = stop;
                    }
                } else if (size >= 2) {   // split in front of the largest hole
                    split = true;
                    sub_a0 = start; sub_a1 = hole_at - 1; sub_b0 = hole_at; sub_b1 = stop;
                }
            }
            kick = min(kick, own);
            if (split && depth < p.max_recursion) {
                const int need = p.min_len < 3u ? 3 : (int)p.min_len;
                const bool take_a = sub_a1 - sub_a0 + 1 >= need, take_b = sub_b1 - sub_b0 + 1 >= need;
                if (sp + 2 > p.stack_cap) { if (lane == 0) atomicExch(out.error, 1); break; }
                // depth-first, left part first: push the right one below it
                if (lane == 0) {
                    int q = sp;
                    if (take_b) stack[q++] = make_int4(sub_b0, sub_b1, depth + 1, 0);
                    if (take_a) stack[q++] = make_int4(sub_a0, sub_a1, depth + 1, 0);
                }
                sp += (take_a ? 1 : 0) + (take_b ? 1 : 0);
                __threadfence_block();
            }
        }
        if (lane == 0) atomicAdd(&out.kicks[kick], 1ull);
        if (best_len == 0) continue;
        if (p.limit_frames) {
            // POSITION_T <= start time + limit - 1: last such row (times increase); exact: the row AT the limit
            const uint32_t limit = p.limit_frames + c.t[best_start] - 1u;
            int lo = best_start, hi = best_stop;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (c.t[mid] <= limit) lo = mid; else hi = mid - 1;
            }
            if (p.limit_exact && c.t[lo] != limit) continue;
            best_stop = lo;
        }
        for (int i = best_start + lane; i <= best_stop; i += 64) out.good[i] = 1;
        if (lane == 0) atomicAdd(out.n_good, 1ull);
    }
}

__global__ __launch_bounds__(256) void k_sel_good32(const uint8_t *__restrict__ good, long long n, uint32_t *__restrict__ g32)
{
    for (long long i = gtid(); i < n; i += gstride()) g32[i] = good[i];
}

__global__ __launch_bounds__(256) void k_sel_emit(const uint8_t *__restrict__ good, const uint32_t *__restrict__ pos,
                                                  const uint32_t *__restrict__ orig, long long n, int64_t *__restrict__ sel_row,
                                                  int64_t *__restrict__ sel_index)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        if (!good[i]) continue;
        const uint32_t j = pos[i] - 1u;
        sel_row[j] = (int64_t)orig[i];
        sel_index[j] = (int64_t)i;
    }
}

// ---- workspace ------------------------------------------------------------------------------------
struct SelLayout {
    size_t area, start, incl, seg, first, last, bits_a, bits_b, val_a, val_b, key32_a, key32_b, median, keep, pos;
    size_t c_id, c_t, c_orig, c_x, c_y, c_area, c_ratio, c_dist, flag, good, bounds, counters, leaf, leaf_sum, stack, temp;
    size_t temp_bytes, total;
    long long leaf_slots;
    int stack_cap;
};

SelLayout sel_layout(long long n, int max_recursion)
{
    SelLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = ysmr::align_up(off + bytes, 256); return o; };
    const size_t N = (size_t)(n > 0 ? n : 1);
    L.area = take(8 * N); L.start = take(4 * N); L.incl = take(4 * N); L.seg = take(4 * N);
    L.first = take(4 * N); L.last = take(4 * N);
    L.bits_a = take(8 * N); L.bits_b = take(8 * N); L.val_a = take(8 * N); L.val_b = take(8 * N);
    L.key32_a = take(4 * N); L.key32_b = take(4 * N);
    L.median = take(8 * N); L.keep = take(4 * N); L.pos = take(4 * N);
    L.c_id = take(4 * N); L.c_t = take(4 * N); L.c_orig = take(4 * N);
    L.c_x = take(8 * N); L.c_y = take(8 * N); L.c_area = take(8 * N); L.c_ratio = take(8 * N); L.c_dist = take(8 * N);
    L.flag = take(N); L.good = take(N);
    L.bounds = take(sizeof(Bounds)); L.counters = take(16 * 8);
    L.leaf_slots = (long long)(N / 64 + N + 2);   // (tracks <= rows)
    L.leaf = take(sizeof(int2) * (size_t)L.leaf_slots);
    L.leaf_sum = take(sizeof(double) * 4 * (size_t)L.leaf_slots);
    long long cap = (long long)max_recursion + 4;
    if (cap < 16) cap = 16;
    if (cap > 8192) cap = 8192;   // (16 B x 1024 waves per entry; a deeper walk is reported as YSMR_ERR_CAPACITY)
    L.stack_cap = (int)cap;
    L.stack = take(sizeof(int4) * (size_t)L.stack_cap * TRACK_BLOCKS * 4);
    L.temp_bytes = std::max(ysmr::prim::radix_temp_bytes(N), sizeof(uint32_t) * ysmr::prim::scan_temp_words(N));
    L.temp = take(L.temp_bytes);
    L.total = off;
    return L;
}

unsigned grid_for(long long n) { return (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, SEL_BLOCKS)); }

}  // namespace

extern "C" {

size_t ysmr_select_workspace_bytes(long long n_rows, int max_recursion)
{
    if (n_rows < 0 || n_rows > 0x7FFFFFFFll) return 0;
    return sel_layout(n_rows, max_recursion).total;
}

int ysmr_select_tracks(void *stream, long long n_rows, const uint32_t *track_id_dev, const uint32_t *t_dev,
                       const double *x_dev, const double *y_dev, const double *w_dev, const double *h_dev,
                       const ysmr_select_params *prm, void *workspace_dev, size_t workspace_bytes,
                       int64_t *sel_row_dev, int64_t *sel_index_dev, ysmr_select_summary *summary)
{
    if (!prm || !summary) return ysmr::fail(YSMR_ERR_ARG, "params and summary must not be NULL");
    std::memset(summary, 0, sizeof(*summary));
    if (n_rows < 0 || n_rows > 0x7FFFFFFFll) return ysmr::fail(YSMR_ERR_ARG, "n_rows must be in 0..2^31-1, got %lld", n_rows);
    if (prm->min_length_frames < 0 || prm->limit_frames < 0 || prm->max_recursion < 0 || prm->frame_height <= 0 || prm->frame_width <= 0)
        return ysmr::fail(YSMR_ERR_ARG, "negative length/recursion setting or non-positive frame size");
    summary->rows_before = n_rows;
    // "File is empty/of insufficient length before initial clean-up" (track_eval.py:612-619)
    if (n_rows < prm->min_length_frames || n_rows == 0) { summary->status = YSMR_SELECT_TOO_SHORT; return YSMR_OK; }
    if (!track_id_dev || !t_dev || !x_dev || !y_dev || !w_dev || !h_dev || !workspace_dev || !sel_row_dev || !sel_index_dev)
        return ysmr::fail(YSMR_ERR_ARG, "a required device pointer is NULL");
    const SelLayout L = sel_layout(n_rows, prm->max_recursion);
    if (workspace_bytes < L.total)
        return ysmr::fail(YSMR_ERR_CAPACITY, "select workspace too small: %zu < %zu bytes", workspace_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)workspace_dev;
    auto P = [&](size_t off) { return (void *)(w + off); };
    double *area = (double *)P(L.area), *median = (double *)P(L.median);
    uint32_t *start = (uint32_t *)P(L.start), *incl = (uint32_t *)P(L.incl), *seg = (uint32_t *)P(L.seg);
    uint32_t *first = (uint32_t *)P(L.first), *last = (uint32_t *)P(L.last), *keep = (uint32_t *)P(L.keep), *pos = (uint32_t *)P(L.pos);
    auto *bits_a = (unsigned long long *)P(L.bits_a), *bits_b = (unsigned long long *)P(L.bits_b);
    auto *val_a = (unsigned long long *)P(L.val_a), *val_b = (unsigned long long *)P(L.val_b);   // (values of pass 2 / distance keys)
    uint32_t *key32_a = (uint32_t *)P(L.key32_a), *key32_b = (uint32_t *)P(L.key32_b);
    void *temp = P(L.temp);
    const long long n = n_rows;
    const unsigned g = grid_for(n);

    // ---- clean-up
    hipLaunchKernelGGL(k_sel_area, dim3(g), dim3(256), 0, st, track_id_dev, w_dev, h_dev, n, area, start);
    ysmr::prim::inclusive_scan_u32(st, start, incl, (size_t)n, (uint32_t *)temp);
    hipLaunchKernelGGL(k_sel_segments, dim3(g), dim3(256), 0, st, start, incl, n, seg, first, last);
    uint32_t n_tracks = 0;
    YSMR_HIP_CHECK(hipMemcpyAsync(&n_tracks, incl + (n - 1), 4, hipMemcpyDeviceToHost, st));
    YSMR_HIP_CHECK(hipStreamSynchronize(st));
    summary->tracks_before = n_tracks;
    // per-track median of the area: stable sort by area (non-negative doubles order like their bit patterns),
    // then stable sort by track (as many digits as the track count needs)
    hipLaunchKernelGGL(k_sel_bits, dim3(g), dim3(256), 0, st, area, n, bits_a);
    YSMR_HIP_CHECK(hipMemcpyAsync(key32_a, seg, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    int at = ysmr::prim::radix_sort(st, bits_a, bits_b, key32_a, key32_b, (size_t)n, 64, temp);
    int track_bits = 8;
    while (track_bits < 32 && (n_tracks >> track_bits)) track_bits += 8;
    unsigned long long *area_sorted = at ? bits_b : bits_a, *area_other = at ? bits_a : bits_b;
    uint32_t *seg_sorted = at ? key32_b : key32_a, *seg_other = at ? key32_a : key32_b;
    at = ysmr::prim::radix_sort(st, seg_sorted, seg_other, area_sorted, area_other, (size_t)n, track_bits, temp);
    const unsigned long long *area_by_track = at ? area_other : area_sorted;
    hipLaunchKernelGGL(k_sel_median, dim3(grid_for(n_tracks)), dim3(256), 0, st, area_by_track, first, last, n_tracks, median);
    CleanParams cp{prm->area_lo, prm->area_hi, prm->area_factor, prm->area_factor != 0.0 ? 1 : 0, (uint32_t)prm->min_length_frames};
    hipLaunchKernelGGL(k_sel_keep, dim3(g), dim3(256), 0, st, area, seg, median, first, last, t_dev, n, cp, keep);
    ysmr::prim::inclusive_scan_u32(st, keep, pos, (size_t)n, (uint32_t *)temp);
    uint32_t n_kept = 0;
    YSMR_HIP_CHECK(hipMemcpyAsync(&n_kept, pos + (n - 1), 4, hipMemcpyDeviceToHost, st));
    Table c{(uint32_t *)P(L.c_id), (uint32_t *)P(L.c_t), (uint32_t *)P(L.c_orig), (double *)P(L.c_x), (double *)P(L.c_y),
            (double *)P(L.c_area), (double *)P(L.c_ratio), (double *)P(L.c_dist)};
    hipLaunchKernelGGL(k_sel_compact, dim3(g), dim3(256), 0, st, keep, pos, track_id_dev, t_dev, x_dev, y_dev, w_dev, h_dev, area, n, c);
    YSMR_HIP_CHECK(hipStreamSynchronize(st));
    summary->rows_after = n_kept;
    // "File is empty/of insufficient length after initial clean-up" (track_eval.py:676-684)
    if ((long long)n_kept < prm->min_length_frames || n_kept == 0) { summary->status = YSMR_SELECT_TOO_SHORT_CLEANED; return YSMR_OK; }

    // ---- the cleaned table: tracks, distances, bounds
    const long long m = n_kept;
    const unsigned gm = grid_for(m);
    hipLaunchKernelGGL(k_sel_dist, dim3(gm), dim3(256), 0, st, c, m, start);
    ysmr::prim::inclusive_scan_u32(st, start, incl, (size_t)m, (uint32_t *)temp);
    hipLaunchKernelGGL(k_sel_segments, dim3(gm), dim3(256), 0, st, start, incl, m, seg, first, last);
    uint32_t n_tracks2 = 0;
    YSMR_HIP_CHECK(hipMemcpyAsync(&n_tracks2, incl + (m - 1), 4, hipMemcpyDeviceToHost, st));
    const int use_area = prm->q_area > 0.0 ? 1 : 0, use_dist = prm->omit_motility ? 1 : 0;
    const unsigned long long *areas_ordered = bits_a, *dists_ordered = val_a;
    ysmr::prim::NoValue *none = nullptr;
    if (use_area) {
        hipLaunchKernelGGL(k_sel_bits, dim3(gm), dim3(256), 0, st, c.area, m, bits_a);
        areas_ordered = ysmr::prim::radix_sort(st, bits_a, bits_b, none, none, (size_t)m, 64, temp) ? bits_b : bits_a;
    }
    if (use_dist) {
        hipLaunchKernelGGL(k_sel_bits, dim3(gm), dim3(256), 0, st, c.dist, m, val_a);
        dists_ordered = ysmr::prim::radix_sort(st, val_a, val_b, none, none, (size_t)m, 64, temp) ? val_b : val_a;
    }
    Bounds *bounds = (Bounds *)P(L.bounds);
    // pandas hands numpy q * 100 and numpy divides by 100 again
    const double q_lo = prm->q_area * 100.0 / 100.0, q_hi = (1.0 - prm->q_area) * 100.0 / 100.0;
    hipLaunchKernelGGL(k_sel_bounds, dim3(1), dim3(1), 0, st, areas_ordered, dists_ordered, m, q_lo, q_hi, use_area, use_dist, bounds);
    uint8_t *flag = (uint8_t *)P(L.flag), *good = (uint8_t *)P(L.good);
    YSMR_HIP_CHECK(hipMemsetAsync(flag, 0, (size_t)m, st));
    YSMR_HIP_CHECK(hipMemsetAsync(good, 0, (size_t)m, st));
    unsigned long long *counters = (unsigned long long *)P(L.counters);   // [0..8] kick reasons, [9] good tracks, [10] error (int)
    YSMR_HIP_CHECK(hipMemsetAsync(counters, 0, 16 * 8, st));
    if (use_dist) {
        hipLaunchKernelGGL(k_sel_flags, dim3(gm), dim3(256), 0, st, c.dist, m, bounds, flag);
        hipLaunchKernelGGL(k_sel_decide, dim3(1), dim3(1), 0, st, bounds, m, prm->motility_stop_fraction);
    }
    YSMR_HIP_CHECK(hipStreamSynchronize(st));
    summary->tracks_after = n_tracks2;

    // ---- tracks
    TrackParams tp{};
    tp.min_len = (uint32_t)prm->min_length_frames;
    tp.max_holes = prm->max_holes;
    tp.max_empty = prm->max_empty_ratio;
    tp.ratio_min = prm->ratio_min; tp.ratio_max = prm->ratio_max;
    tp.y_lo = prm->edge_fraction * prm->frame_height; tp.y_hi = (1 - prm->edge_fraction) * prm->frame_height;
    tp.x_lo = prm->edge_fraction * prm->frame_width; tp.x_hi = (1 - prm->edge_fraction) * prm->frame_width;
    tp.check_frame = prm->edge_fraction != 0.0 ? 1 : 0;
    tp.frame_w = prm->frame_width; tp.frame_h = prm->frame_height;
    tp.max_recursion = prm->max_recursion;
    tp.limit_frames = (uint32_t)prm->limit_frames;
    tp.limit_exact = prm->limit_exact;
    tp.stack_cap = L.stack_cap;
    TrackScratch sc{(int4 *)P(L.stack), (int2 *)P(L.leaf), (double *)P(L.leaf_sum), L.leaf_slots};
    TrackOut to{good, counters, counters + 9, (int *)(counters + 10)};
    const unsigned tblocks = (unsigned)std::max<long long>(1, std::min<long long>(((long long)n_tracks2 + 3) / 4, TRACK_BLOCKS));
    hipLaunchKernelGGL(k_sel_tracks, dim3(tblocks), dim3(256), 0, st, c, flag, first, last, n_tracks2, tp, bounds, sc, to);
    YSMR_LAUNCH_CHECK();

    // ---- output
    hipLaunchKernelGGL(k_sel_good32, dim3(gm), dim3(256), 0, st, good, m, keep);
    ysmr::prim::inclusive_scan_u32(st, keep, pos, (size_t)m, (uint32_t *)temp);
    hipLaunchKernelGGL(k_sel_emit, dim3(gm), dim3(256), 0, st, good, pos, c.orig, m, sel_row_dev, sel_index_dev);
    uint32_t n_sel = 0;
    unsigned long long h_counters[11];
    Bounds hb;
    YSMR_HIP_CHECK(hipMemcpyAsync(&n_sel, pos + (m - 1), 4, hipMemcpyDeviceToHost, st));
    YSMR_HIP_CHECK(hipMemcpyAsync(h_counters, counters, sizeof(h_counters), hipMemcpyDeviceToHost, st));
    YSMR_HIP_CHECK(hipMemcpyAsync(&hb, bounds, sizeof(hb), hipMemcpyDeviceToHost, st));
    YSMR_HIP_CHECK(hipStreamSynchronize(st));
    if ((int)(h_counters[10] & 0xFFFFFFFFull) != 0)
        return ysmr::fail(YSMR_ERR_CAPACITY, "select_tracks: recursion stack of %d entries exhausted", L.stack_cap);
    summary->area_lo = hb.area_lo; summary->area_hi = hb.area_hi;
    summary->q1_dist = hb.q1_dist; summary->q3_dist = hb.q3_dist; summary->dist_fence = hb.fence;
    summary->dist_outliers = (long long)hb.outliers;
    summary->outliers_used = hb.use_flags;
    for (int i = 0; i < 9; ++i) summary->kick_reasons[i] = (long long)h_counters[i];
    summary->good_tracks = (long long)h_counters[9];
    summary->rows_selected = n_sel;
    summary->status = n_sel ? YSMR_SELECT_OK : YSMR_SELECT_NONE;
    return YSMR_OK;
}

}  // extern "C"
