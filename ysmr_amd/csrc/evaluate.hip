// libysmr_hip -- per-track statistics of the selected tracks: the numerical part of evaluate_tracks
// (ysmr/track_eval.py:846-1318).  Input: the six columns of the (TRACK_ID, POSITION_T)-ordered table
// of selected tracks, resident in HBM.  Output: the per-row columns of <name>_analysed.csv and the per-track table
// of <name>_statistics.csv.  The reference does this with ~60 pandas / NumPy / SciPy calls on one core; their
// arithmetic is reproduced operation for operation where it is defined by IEEE basic operations:
//   * groupby sum / mean = Kahan (compensated) summation in row order (pandas _libs/groupby.pyx group_sum /
//     group_mean), float16 `bac_length` whose group mean pandas takes in float32;
//   * scipy.signal.medfilt of a 0/1 column with zero padding = "more than half of the window is 1", evaluated
//     from a prefix sum (window clipped to the track);
//   * scipy.signal.argrelextrema(greater_equal, order 10, mode 'clip') = not smaller than any of the 10 rows
//     either side, inside the track;
//   * the numbering of the stretches between turning points runs through the whole table, a track start right
//     behind a turning point does not start a new stretch, and the table's last row keeps number 0
//     (track_eval.py:976-989);
//   * df.index // fps is NumPy's float floor_divide (npy_divmod), the median of the per-second sums pandas'
//     median_linear;
//   * scipy pdist / np.sqrt(np.square + np.square): s = dx * dx + dy * dy unfused, one correctly rounded sqrt.
// atan2 is the device library's (not correctly rounded): the heading may differ from glibc's by an ulp, which
// matters only if a change of heading lies within an ulp of an integer number of degrees (it is truncated).
// Compiled with -ffp-contract=off.  One call per video, all grids resident-sized.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "common.h"
#include "prim.h"

namespace {

constexpr int EV_BLOCKS = 512;
__device__ __forceinline__ long long gtid() { return (long long)blockIdx.x * blockDim.x + threadIdx.x; }
__device__ __forceinline__ long long gstride() { return (long long)gridDim.x * blockDim.x; }
__device__ __forceinline__ double nan64() { return __longlong_as_double(0x7FF8000000000000ll); }

struct Kahan {   // pandas group_sum
    double sum = 0.0, comp = 0.0;
    __device__ __forceinline__ void add(double v)
    {
        const double y = v - comp;
        const double t = sum + y;
        comp = t - sum - y;
        sum = t;
    }
};

struct EvParams {
    double px, fps, min_angle;
    int lag_angle, lag_reach, half1, half2;     // medfilt half widths: 1 and (max_kernel - 1) / 2
};

struct EvRows {        // per row, n entries each
    const uint32_t *id, *t;
    const double *x, *y, *w, *h;
    uint32_t *flag, *seg, *u32a, *u32b, *run_incl;   // scratch: start flags, track number, scan inputs / outputs
    double *w_um, *h_um, *travelled, *heading, *tp_of_tracks, *tp_dist, *blen;   // blen: float16 bac_length as double
    int32_t *angle, *cand;
    int8_t *moving, *tp, *phenotype;
    uint32_t *first, *last;                          // per track
    uint32_t *run_pos;                               // per stretch: its first row
    double *bins;                                    // scratch of the per-second sums (one slot per row)
};

__global__ __launch_bounds__(256) void k_ev_flags(EvRows r, long long n)
{
    for (long long i = gtid(); i < n; i += gstride()) r.flag[i] = (i == 0 || r.id[i] != r.id[i - 1]) ? 1u : 0u;
}

// seg = inclusive scan of flag - 1; first / last row of every track
__global__ __launch_bounds__(256) void k_ev_segments(EvRows r, long long n)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const uint32_t s = r.seg[i] - 1u;
        r.seg[i] = s;
    }
}
__global__ __launch_bounds__(256) void k_ev_bounds(EvRows r, long long n)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        if (r.flag[i]) r.first[r.seg[i]] = (uint32_t)i;
        if (i == n - 1 || r.flag[i + 1]) r.last[r.seg[i]] = (uint32_t)i;
    }
}

// steps, lengths in micrometres, float16 body length, path per row, "moving" before the median filters
__global__ __launch_bounds__(256) void k_ev_steps(EvRows r, long long n, EvParams p)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const bool start = r.flag[i] != 0;
        const double xd = start ? 0.0 : r.x[i] - r.x[i - 1], yd = start ? 0.0 : r.y[i] - r.y[i - 1];
        const double td = start ? 1.0 : (double)r.t[i] - (double)r.t[i - 1];
        const double wu = r.w[i] / p.px, hu = r.h[i] / p.px;
        r.w_um[i] = wu;
        r.h_um[i] = hu;
        r.blen[i] = (double)(_Float16)(wu >= hu ? wu : hu);
        const double d = sqrt(xd * xd + yd * yd) / p.px;
        r.travelled[i] = d;
        r.u32a[i] = (d / td > 0.001) ? 1u : 0u;
    }
}

// medfilt of a 0/1 column from its inclusive prefix sum P: ones in [i - half, i + half] clipped to the track
__global__ __launch_bounds__(256) void k_ev_median(EvRows r, long long n, const uint32_t *__restrict__ incl, int half,
                                                   uint32_t *out_u32, int8_t *out_i8)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const long long a = r.first[r.seg[i]], b = r.last[r.seg[i]];
        const long long lo = std::max(a, i - half), hi = std::min(b, i + half);
        const uint32_t ones = incl[hi] - (lo ? incl[lo - 1] : 0u);
        const uint32_t v = (2u * ones > (uint32_t)(2 * half + 1)) ? 1u : 0u;
        if (out_u32) out_u32[i] = v;
        if (out_i8) out_i8[i] = (int8_t)v;
    }
}

__global__ __launch_bounds__(256) void k_ev_heading(EvRows r, long long n, EvParams p)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const long long a = r.first[r.seg[i]];
        double deg = nan64();
        if (i - p.lag_angle >= a)
            deg = atan2(r.x[i] - r.x[i - p.lag_angle], r.y[i] - r.y[i - p.lag_angle]) * 57.29577951308232;   // np.degrees
        r.heading[i] = deg;
    }
}

__global__ __launch_bounds__(256) void k_ev_turn(EvRows r, long long n, EvParams p)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        double turn = 0.0;
        if (!r.flag[i]) {
            const double d = r.heading[i] - r.heading[i - 1];
            turn = d != d ? 0.0 : fabs(d);                    // groupby diff, fillna(0), abs
        }
        const double folded = (360.0 - turn <= turn) ? 360.0 - turn : turn;
        const int32_t deg = (int32_t)folded;                  // astype(int32): towards zero
        r.angle[i] = deg;
        r.cand[i] = ((double)deg > p.min_angle && r.moving[i] == 1) ? deg : 0;
    }
}

__global__ __launch_bounds__(256) void k_ev_peaks(EvRows r, long long n)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const long long a = r.first[r.seg[i]], b = r.last[r.seg[i]];
        const int32_t c = r.cand[i];
        bool peak = c != 0;
        for (long long j = std::max(a, i - 10); peak && j <= std::min(b, i + 10); ++j) peak = c >= r.cand[j];
        const int8_t tp = (peak || r.flag[i]) ? 1 : 0;
        r.tp[i] = tp;
    }
}
// a stretch starts where the column turns from 0 to 1
__global__ __launch_bounds__(256) void k_ev_run_starts(EvRows r, long long n)
{
    for (long long i = gtid(); i < n; i += gstride()) r.u32a[i] = (r.tp[i] == 1 && (i == 0 || r.tp[i - 1] == 0)) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_ev_numbers(EvRows r, long long n)
{
    for (long long i = gtid(); i < n; i += gstride()) {
        const uint32_t number = i == n - 1 ? 0u : r.run_incl[i] - 1u;
        r.tp_of_tracks[i] = r.moving[i] == 0 ? nan64() : (double)number;
        if (r.u32a[i]) r.run_pos[r.run_incl[i] - 1u] = (uint32_t)i;
    }
}
// path per stretch (rows of the stretch that are moving, in row order; the table's last row belongs to stretch 0)
__global__ __launch_bounds__(256) void k_ev_stretches(EvRows r, long long n, uint32_t n_runs)
{
    for (long long q = gtid(); q < n_runs; q += gstride()) {
        const long long a = r.run_pos[q], b = q + 1 < n_runs ? (long long)r.run_pos[q + 1] : n - 1;
        Kahan k;
        for (long long i = a; i < b; ++i)
            if (r.moving[i]) k.add(r.travelled[i]);
        if (q == 0 && r.moving[n - 1]) k.add(r.travelled[n - 1]);
        for (long long i = a; i < b; ++i) r.tp_dist[i] = r.moving[i] ? k.sum : nan64();
        if (q == 0) r.tp_dist[n - 1] = r.moving[n - 1] ? k.sum : nan64();
    }
}

__device__ __forceinline__ double block_max(double v, double *s_red)
{
    // NaN-skipping maximum of the block (all NaN -> NaN)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(v, d);
        v = (v != v) ? o : ((o != o) ? v : (o > v ? o : v));
    }
    __syncthreads();
    if (lane == 0) s_red[w] = v;
    __syncthreads();
    double m = s_red[0];
    for (int k = 1; k < 4; ++k) { const double o = s_red[k]; m = (m != m) ? o : ((o != o) ? m : (o > m ? o : m)); }
    return m;
}
__device__ __forceinline__ long long block_sum(long long v, long long *s_red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if (lane == 0) s_red[w] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// numpy's float floor_divide (npy_divmod): the bin of row i in df.index // fps
__device__ __forceinline__ double np_floor_divide(double a, double b)
{
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0) != (mod < 0))) div -= 1.0;
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, a / b);
}

// One block per track: everything the statistics table holds (track_eval.py:990-1090).
__global__ __launch_bounds__(256) void k_ev_tracks(EvRows r, long long n, uint32_t n_tracks, EvParams p, double *stats)
{
    __shared__ double s_redd[4];
    __shared__ long long s_redl[4];
    __shared__ double s_body, s_path, s_median;
    for (uint32_t k = blockIdx.x; k < n_tracks; k += gridDim.x) {
        const long long a = r.first[k], b = r.last[k], len = b - a + 1;
        // sequential sums (row order matters): mean body length, path, per-second sums and their median
        if (threadIdx.x == 0) {
            Kahan path;
            float bs = 0.f, bc = 0.f;       // pandas takes the group mean of a float16 column in float32 (Kahan, too)
            for (long long i = a; i <= b; ++i) {
                const float y = (float)r.blen[i] - bc;
                const float t = bs + y;
                bc = (t - bs) - y;
                bs = t;
                path.add(r.travelled[i]);
            }
            s_body = (double)(bs / (float)len);
            s_path = path.sum;
            double *bins = r.bins + a;
            long long m = 0;
            Kahan sec;
            double cur = np_floor_divide((double)a, p.fps);
            for (long long i = a; i <= b; ++i) {
                const double bin = np_floor_divide((double)i, p.fps);
                if (bin != cur) { bins[m++] = sec.sum; sec = Kahan(); cur = bin; }
                sec.add(r.travelled[i]);
            }
            bins[m++] = sec.sum;
            for (long long u = 1; u < m; ++u) {                           // insertion sort: a few hundred values at most
                const double v = bins[u];
                long long q = u - 1;
                while (q >= 0 && bins[q] > v) { bins[q + 1] = bins[q]; --q; }
                bins[q + 1] = v;
            }
            s_median = (m & 1) ? bins[m / 2] : (bins[m / 2] + bins[m / 2 - 1]) / 2.0;   // pandas median_linear
        }
        __syncthreads();
        const double body = s_body;
        // displacement over lag_reach rows, longest stretch: maxima over the track
        double reach = nan64(), stretch = nan64();
        long long moving_rows = 0;
        for (long long i = a + threadIdx.x; i <= b; i += 256) {
            if (i - p.lag_reach >= a) {
                // x_norm differences: ((x_i - x_a) / px) - ((x_j - x_a) / px), as pandas computes them
                const double xi = (r.x[i] - r.x[a]) / p.px, xj = (r.x[i - p.lag_reach] - r.x[a]) / p.px;
                const double yi = (r.y[i] - r.y[a]) / p.px, yj = (r.y[i - p.lag_reach] - r.y[a]) / p.px;
                const double dx = xi - xj, dy = yi - yj;
                const double d = sqrt(dx * dx + dy * dy);
                reach = (reach != reach || d > reach) ? d : reach;
            }
            const double td = r.tp_dist[i];
            if (td == td) stretch = (stretch != stretch || td > stretch) ? td : stretch;
            moving_rows += r.moving[i];
        }
        reach = block_max(reach, s_redd) / body;
        stretch = block_max(stretch, s_redd) / body;
        moving_rows = block_sum(moving_rows, s_redl);
        const int phen = (reach > 1.5 && stretch > 5.0) ? 2 : ((reach > 1.5 && stretch <= 5.0) ? 1 : 0);
        // turning points of immotile tracks are dropped (the track start stays one)
        long long turns = 0;
        for (long long i = a + threadIdx.x; i <= b; i += 256) {
            r.phenotype[i] = (int8_t)phen;
            const int8_t tp = (i == a) ? 1 : (phen != 0 ? r.tp[i] : 0);
            r.tp[i] = tp;
            turns += tp;
        }
        turns = block_sum(turns, s_redl);
        // largest distance between any two positions of the track
        double widest2 = -1.0;
        for (long long i = a + threadIdx.x; i <= b; i += 256) {
            const double xi = (r.x[i] - r.x[a]) / p.px, yi = (r.y[i] - r.y[a]) / p.px;
            for (long long j = i + 1; j <= b; ++j) {
                const double dx = xi - (r.x[j] - r.x[a]) / p.px, dy = yi - (r.y[j] - r.y[a]) / p.px;
                const double s = dx * dx + dy * dy;
                widest2 = s > widest2 ? s : widest2;
            }
        }
        widest2 = block_max(widest2, s_redd);
        if (threadIdx.x == 0) {
            const double widest = sqrt(widest2);                          // (len >= 2: selected tracks are long)
            const double frames = (double)((long long)r.t[b] - (long long)r.t[a]) + 1.0;
            const double seconds = frames / p.fps;
            const double xl = (r.x[b] - r.x[a]) / p.px, yl = (r.y[b] - r.y[a]) / p.px;
            const double chord = sqrt(xl * xl + yl * yl);
            double *o = stats + (size_t)k * 12;
            o[0] = moving_rows != 0 ? ((double)(turns - 1) * p.fps) / (double)moving_rows : 0.0;
            o[1] = s_path;
            o[2] = moving_rows != 0 ? s_path / seconds : 0.0;
            o[3] = seconds;
            o[4] = widest;
            o[5] = (double)moving_rows / frames * 100.0;
            o[6] = s_path != 0.0 ? chord / s_path : 0.0;
            o[7] = body;
            o[8] = body != 0.0 ? widest / body : 0.0;
            o[9] = (double)phen;
            o[10] = (double)r.id[a];
            o[11] = s_median;
        }
        __syncthreads();
    }
}

struct EvLayout {
    size_t flag, seg, u32a, u32b, run_incl, heading, tp_dist, blen, cand, first, last, run_pos, bins, temp, total;
};
EvLayout ev_layout(long long n)
{
    EvLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = ysmr::align_up(off + bytes, 256); return o; };
    const size_t N = (size_t)(n > 0 ? n : 1);
    L.flag = take(4 * N); L.seg = take(4 * N); L.u32a = take(4 * N); L.u32b = take(4 * N); L.run_incl = take(4 * N);
    L.heading = take(8 * N); L.tp_dist = take(8 * N); L.blen = take(8 * N); L.cand = take(4 * N);
    L.first = take(4 * N); L.last = take(4 * N); L.run_pos = take(4 * N); L.bins = take(8 * N);
    L.temp = take(sizeof(uint32_t) * ysmr::prim::scan_temp_words(N));
    L.total = off;
    return L;
}
unsigned ev_grid(long long n) { return (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, EV_BLOCKS)); }

}  // namespace

extern "C" {

size_t ysmr_evaluate_workspace_bytes(long long n_rows)
{
    if (n_rows < 0 || n_rows > 0x7FFFFFFFll) return 0;
    return ev_layout(n_rows).total;
}

int ysmr_evaluate_tracks(void *stream, long long n_rows, const uint32_t *track_id_dev, const uint32_t *t_dev,
                         const double *x_dev, const double *y_dev, const double *w_dev, const double *h_dev,
                         const ysmr_evaluate_params *prm, void *workspace_dev, size_t workspace_bytes,
                         double *width_um_dev, double *height_um_dev, int32_t *angle_diff_dev, int8_t *moving_dev,
                         int8_t *turn_points_dev, double *tp_of_tracks_dev, double *travelled_dist_dev,
                         int8_t *motility_phenotype_dev, double *stats_dev, long long *n_tracks_out)
{
    if (!prm || !n_tracks_out) return ysmr::fail(YSMR_ERR_ARG, "params and n_tracks_out must not be NULL");
    *n_tracks_out = 0;
    if (n_rows < 0 || n_rows > 0x7FFFFFFFll) return ysmr::fail(YSMR_ERR_ARG, "n_rows must be in 0..2^31-1, got %lld", n_rows);
    if (n_rows == 0) return YSMR_OK;
    if (!(prm->pixel_per_micrometre > 0) || !(prm->fps > 0) || prm->angle_lag < 1 || prm->reach_lag < 1 || prm->median_kernel < 1 ||
        !(prm->median_kernel & 1))
        return ysmr::fail(YSMR_ERR_ARG, "pixel_per_micrometre, fps must be positive, the lags >= 1, median_kernel odd");
    if (!track_id_dev || !t_dev || !x_dev || !y_dev || !w_dev || !h_dev || !workspace_dev || !width_um_dev || !height_um_dev ||
        !angle_diff_dev || !moving_dev || !turn_points_dev || !tp_of_tracks_dev || !travelled_dist_dev || !motility_phenotype_dev ||
        !stats_dev)
        return ysmr::fail(YSMR_ERR_ARG, "a required device pointer is NULL");
    const EvLayout L = ev_layout(n_rows);
    if (workspace_bytes < L.total)
        return ysmr::fail(YSMR_ERR_CAPACITY, "evaluate workspace too small: %zu < %zu bytes", workspace_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)workspace_dev;
    const long long n = n_rows;
    EvRows r{};
    r.id = track_id_dev; r.t = t_dev; r.x = x_dev; r.y = y_dev; r.w = w_dev; r.h = h_dev;
    r.flag = (uint32_t *)(w + L.flag); r.seg = (uint32_t *)(w + L.seg); r.u32a = (uint32_t *)(w + L.u32a);
    r.u32b = (uint32_t *)(w + L.u32b); r.run_incl = (uint32_t *)(w + L.run_incl);
    r.w_um = width_um_dev; r.h_um = height_um_dev; r.travelled = travelled_dist_dev; r.heading = (double *)(w + L.heading);
    r.tp_of_tracks = tp_of_tracks_dev; r.tp_dist = (double *)(w + L.tp_dist); r.blen = (double *)(w + L.blen);
    r.angle = angle_diff_dev; r.cand = (int32_t *)(w + L.cand);
    r.moving = moving_dev; r.tp = turn_points_dev; r.phenotype = motility_phenotype_dev;
    r.first = (uint32_t *)(w + L.first); r.last = (uint32_t *)(w + L.last); r.run_pos = (uint32_t *)(w + L.run_pos);
    r.bins = (double *)(w + L.bins);
    uint32_t *temp = (uint32_t *)(w + L.temp);
    EvParams p{prm->pixel_per_micrometre, prm->fps, prm->min_turn_angle, prm->angle_lag, prm->reach_lag, 1, (prm->median_kernel - 1) / 2};
    const dim3 g(ev_grid(n)), tb(256);

    hipLaunchKernelGGL(k_ev_flags, g, tb, 0, st, r, n);
    ysmr::prim::inclusive_scan_u32(st, r.flag, r.seg, (size_t)n, temp);
    uint32_t n_tracks = 0;
    YSMR_HIP_CHECK(hipMemcpyAsync(&n_tracks, r.seg + (n - 1), 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_ev_segments, g, tb, 0, st, r, n);
    hipLaunchKernelGGL(k_ev_bounds, g, tb, 0, st, r, n);
    hipLaunchKernelGGL(k_ev_steps, g, tb, 0, st, r, n, p);
    // moving: two median filters (3 rows, then about a second)
    ysmr::prim::inclusive_scan_u32(st, r.u32a, r.u32b, (size_t)n, temp);
    hipLaunchKernelGGL(k_ev_median, g, tb, 0, st, r, n, (const uint32_t *)r.u32b, p.half1, r.u32a, (int8_t *)nullptr);
    ysmr::prim::inclusive_scan_u32(st, r.u32a, r.u32b, (size_t)n, temp);
    hipLaunchKernelGGL(k_ev_median, g, tb, 0, st, r, n, (const uint32_t *)r.u32b, p.half2, (uint32_t *)nullptr, r.moving);
    hipLaunchKernelGGL(k_ev_heading, g, tb, 0, st, r, n, p);
    hipLaunchKernelGGL(k_ev_turn, g, tb, 0, st, r, n, p);
    hipLaunchKernelGGL(k_ev_peaks, g, tb, 0, st, r, n);
    hipLaunchKernelGGL(k_ev_run_starts, g, tb, 0, st, r, n);
    ysmr::prim::inclusive_scan_u32(st, r.u32a, r.run_incl, (size_t)n, temp);
    uint32_t n_runs = 0;
    YSMR_HIP_CHECK(hipMemcpyAsync(&n_runs, r.run_incl + (n - 1), 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_ev_numbers, g, tb, 0, st, r, n);
    YSMR_HIP_CHECK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_ev_stretches, dim3(ev_grid(n_runs)), tb, 0, st, r, n, n_runs);
    hipLaunchKernelGGL(k_ev_tracks, dim3(std::max(1u, std::min(n_tracks, 1024u))), tb, 0, st, r, n, n_tracks, p, stats_dev);
    YSMR_LAUNCH_CHECK();
    YSMR_HIP_CHECK(hipStreamSynchronize(st));
    *n_tracks_out = n_tracks;
    return YSMR_OK;
}

}  // extern "C"
