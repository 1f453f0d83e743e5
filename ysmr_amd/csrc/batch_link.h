// batch_link.h -- the link of a whole BATCH of frames in one launch (included by track.hip, inside its namespace).
//
// CentroidTracker.update (ysmr/tracker.py:93-230) + GaussianSumFIR.correct / predict (ysmr/gsff.py:204-347) + row
// emission (ysmr/track_eval.py:313-316) for `batch` consecutive frames: ONE workgroup of 512 threads on ONE compute
// unit, a track per LANE, the track's whole state -- 32 measurements of history, the filter bank's weights and
// estimates, prediction, box, id, counters -- in the lane's registers from the first frame of the batch to the last.
// Per frame nothing is loaded or stored but the frame's detections (in) and its rows (out):
//
//   detections   k_bgrid (one workgroup per frame, the whole batch at once, before this kernel) bins every frame's
//                detections into a uniform grid of cells (counting sort) and leaves header | cell starts (u16) |
//                centres in cell order | their column numbers (u16) as one contiguous block per frame; this kernel
//                brings the block of frame f+1 into LDS by LDS-DMA while it works on frame f.
//   row minimum  tracker.py:151-163 reads only D.min(1) and D.argmin(1): a lane looks at the cells around its
//                prediction (3 x 3, then wider rings while a nearer detection could hide outside: the same bound and
//                the same tie rule -- lowest column among equal ROUNDED distances -- as rowmin_grid / rowmin_wave).
//   claims       the winner of a detection column is the proposer with the smallest (distance, id): two LDS atomicMin
//                rounds (ids ascend with table rows, so (distance, id) orders like the reference's (distance, row)).
//   lifecycle    ageing / deregistration per lane; a lane that loses its track is simply free; new tracks take free
//                lanes, in CPython set order of the unclaimed columns (cpython_order_lds).  Table ROWS (the order of the
//                reference's OrderedDict = ascending id = the order of a frame's rows) are kept as a per-lane rank:
//                a death lowers the rank of every younger track by one, a birth appends.
//   filter bank  per lane, serial: three FIR estimates per coordinate as fused multiply-add chains over the register
//                history (gains zero-padded to the 32 entries, newest first), three exp(), three divisions.
//   rows         one 40-byte ysmr_row per live lane at rows[base + rank], fire and forget.
//
// Three workgroup barriers per frame (after each atomic round, and at the end of the frame, where the next frame's
// detections must have landed); no global round trip on the frame-to-frame chain except the lane's own claimed box
// (requested after the claims, consumed after the filter bank).
//
// Between launches the state rests in HBM in seat-major arrays (field by field, a track per column, in table order):
// coalesced lane-wise loads at the start of a launch and stores at its end.  k_to_std / k_to_batch convert to and from
// the per-slot layout of k_frame / k_link + k_track (ysmr_tracker_update, tables beyond this kernel's 512 seats).
#pragma once

// (BL_* constants and struct BatchDev: track.hip, next to TrackerDev -- the host handle holds one)

// ---- a frame's detections as this kernel wants them: one block of dwords per frame ------------------------------
//   [0, 16)            header: x0, y0, cell, 1 / cell (f32), cells per side G, m (i32)
//   [16, 16 + SW)      start[G * G + 1] as u16: first item of each cell (row-major), the last entry = m
//   [.., + 2 * MP)     centres (x, y) f32 of the detections in cell order;  MP = m rounded up to 8
//   [.., + MP / 2)     their column numbers as u16
__host__ __device__ inline int bl_grid_n(int m) { return m <= 1024 ? 32 : 64; }
__host__ __device__ inline int bl_start_dwords(int G) { return ((G * G + 2) / 2 + 3) / 4 * 4; }
__host__ __device__ inline int bl_pad8(int m) { return (m + 7) / 8 * 8; }
__host__ __device__ inline int bl_grid_dwords(int m)   // rounded up to whole 1-KiB pieces (one LDS-DMA wave-instruction)
{
    const int raw = 16 + bl_start_dwords(bl_grid_n(m)) + 2 * bl_pad8(m) + bl_pad8(m) / 2;
    return (raw + 255) / 256 * 256;
}
__host__ __device__ inline int bl_grid_dwords_max(int max_det)
{
    const int a = bl_grid_dwords(max_det), b = bl_grid_dwords(max_det < 1024 ? max_det : 1024);
    return a > b ? a : b;
}

// One workgroup per frame: bounding box of the centres, G x G cells over it with one cell of margin, counting sort.
__global__ __launch_bounds__(256) void k_bgrid(const float *__restrict__ det_all, const int32_t *__restrict__ det_count,
                                               int max_det, char *grid, unsigned grid_stride)
{
    __shared__ int s_cnt[64 * 64];
    __shared__ float s_red[4][4];
    __shared__ int s_wave_sum[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *det = det_all + (size_t)f * max_det * 5;
    int m = det_count[f];
    m = m < 0 ? 0 : (m > max_det ? max_det : m);
    const int G = bl_grid_n(m), cells = G * G;
    uint32_t *out = reinterpret_cast<uint32_t *>(grid + (size_t)grid_stride * f);
    uint32_t *start = out + 16;
    float2 *xy = reinterpret_cast<float2 *>(start + bl_start_dwords(G));
    unsigned short *items = reinterpret_cast<unsigned short *>(reinterpret_cast<uint32_t *>(xy) + 2 * bl_pad8(m));
    for (int c = tid; c < cells; c += 256) s_cnt[c] = 0;
    float lo_x = 3.0e38f, lo_y = 3.0e38f, hi_x = -3.0e38f, hi_y = -3.0e38f;
    for (int j = tid; j < m; j += 256) {
        const float x = det[(size_t)j * 5], y = det[(size_t)j * 5 + 1];
        lo_x = fminf(lo_x, x); hi_x = fmaxf(hi_x, x); lo_y = fminf(lo_y, y); hi_y = fmaxf(hi_y, y);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo_x = fminf(lo_x, __shfl_xor(lo_x, d)); hi_x = fmaxf(hi_x, __shfl_xor(hi_x, d));
        lo_y = fminf(lo_y, __shfl_xor(lo_y, d)); hi_y = fmaxf(hi_y, __shfl_xor(hi_y, d));
    }
    if (lane == 0) { s_red[0][w] = lo_x; s_red[1][w] = hi_x; s_red[2][w] = lo_y; s_red[3][w] = hi_y; }
    __syncthreads();
    lo_x = s_red[0][0]; hi_x = s_red[1][0]; lo_y = s_red[2][0]; hi_y = s_red[3][0];
    for (int k = 1; k < 4; ++k) {
        lo_x = fminf(lo_x, s_red[0][k]); hi_x = fmaxf(hi_x, s_red[1][k]);
        lo_y = fminf(lo_y, s_red[2][k]); hi_y = fmaxf(hi_y, s_red[3][k]);
    }
    if (m == 0) { lo_x = lo_y = 0.f; hi_x = hi_y = 1.f; }
    const float extent = fmaxf(fmaxf(hi_x - lo_x, hi_y - lo_y), 1.0f);
    const float cell = extent / (float)(G - 2), inv = 1.0f / cell;
    const float x0 = lo_x - cell, y0 = lo_y - cell;
    if (tid == 0) {
        float *h = reinterpret_cast<float *>(out);
        h[0] = x0; h[1] = y0; h[2] = cell; h[3] = inv;
        out[4] = (uint32_t)G; out[5] = (uint32_t)m;
    }
    auto cell_of = [&](int j) {
        int cx = (int)floorf((det[(size_t)j * 5] - x0) * inv), cy = (int)floorf((det[(size_t)j * 5 + 1] - y0) * inv);
        cx = cx < 0 ? 0 : (cx > G - 1 ? G - 1 : cx);
        cy = cy < 0 ? 0 : (cy > G - 1 ? G - 1 : cy);
        return cy * G + cx;
    };
    for (int j = tid; j < m; j += 256) atomicAdd(&s_cnt[cell_of(j)], 1);
    __syncthreads();
    // exclusive scan of the counts: K consecutive cells per thread (K = 4 or 16), wave scan, wave sums
    const int K = cells / 256;
    int local[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < K) { local[k] = sum; sum += s_cnt[tid * K + k]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) s_wave_sum[w] = incl;
    __syncthreads();
    int before = incl - sum;
    for (int k = 0; k < w; ++k) before += s_wave_sum[k];
#pragma unroll
    for (int k = 0; k < 16; k += 2)
        if (k < K) {
            const int a = before + local[k], b = before + local[k + 1];
            s_cnt[tid * K + k] = a;
            s_cnt[tid * K + k + 1] = b;
            start[(tid * K + k) >> 1] = (uint32_t)a | ((uint32_t)b << 16);
        }
    if (tid == 255) start[cells >> 1] = (uint32_t)m;
    __syncthreads();
    for (int j = tid; j < m; j += 256) {
        const int at = atomicAdd(&s_cnt[cell_of(j)], 1);
        items[at] = (unsigned short)j;
        xy[at] = make_float2(det[(size_t)j * 5], det[(size_t)j * 5 + 1]);
    }
}

// ---- a track in registers -------------------------------------------------------------------------------------------
struct BlSeat {
    double hx[BL_HB], hy[BL_HB];            // measurements, newest first
    double w[BL_NF], xa[BL_NF], xb[BL_NF];  // filter weights, x-hat rows 0 / 1
    double px, py;                          // CentroidTracker.objects[id]: the prediction (tracker.py:225)
    float info[3];
    int id, gone, len, mode, rank;
    bool alive;
};

__device__ __forceinline__ void bl_seat_blank(BlSeat &S)
{
#pragma unroll
    for (int e = 0; e < BL_HB; ++e) { S.hx[e] = 0.0; S.hy[e] = 0.0; }
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) { S.w[f] = 0.0; S.xa[f] = 0.0; S.xb[f] = 0.0; }
    S.px = S.py = 0.0;
    S.info[0] = S.info[1] = S.info[2] = 0.f;
    S.id = S.gone = S.len = S.mode = S.rank = 0;
    S.alive = false;
}

__device__ __forceinline__ void bl_seat_load(BlSeat &S, const BatchDev &bd, int at)
{
    const size_t sc = (size_t)bd.seat_cap;
    const double *p = bd.f64 + at;
#pragma unroll
    for (int e = 0; e < BL_HB; ++e) { S.hx[e] = p[sc * e]; S.hy[e] = p[sc * (BL_HB + e)]; }
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) {
        S.w[f] = p[sc * (2 * BL_HB + f)];
        S.xa[f] = p[sc * (2 * BL_HB + BL_NF + f)];
        S.xb[f] = p[sc * (2 * BL_HB + 2 * BL_NF + f)];
    }
    S.px = p[sc * (2 * BL_HB + 3 * BL_NF)];
    S.py = p[sc * (2 * BL_HB + 3 * BL_NF + 1)];
#pragma unroll
    for (int k = 0; k < 3; ++k) S.info[k] = bd.f32[sc * k + at];
    S.id = bd.i32[at]; S.gone = bd.i32[sc + at]; S.len = bd.i32[2 * sc + at]; S.mode = bd.i32[3 * sc + at];
}

__device__ __forceinline__ void bl_seat_store(const BlSeat &S, const BatchDev &bd, int at)
{
    const size_t sc = (size_t)bd.seat_cap;
    double *p = bd.f64 + at;
#pragma unroll
    for (int e = 0; e < BL_HB; ++e) { p[sc * e] = S.hx[e]; p[sc * (BL_HB + e)] = S.hy[e]; }
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) {
        p[sc * (2 * BL_HB + f)] = S.w[f];
        p[sc * (2 * BL_HB + BL_NF + f)] = S.xa[f];
        p[sc * (2 * BL_HB + 2 * BL_NF + f)] = S.xb[f];
    }
    p[sc * (2 * BL_HB + 3 * BL_NF)] = S.px;
    p[sc * (2 * BL_HB + 3 * BL_NF + 1)] = S.py;
#pragma unroll
    for (int k = 0; k < 3; ++k) bd.f32[sc * k + at] = S.info[k];
    bd.i32[at] = S.id; bd.i32[sc + at] = S.gone; bd.i32[2 * sc + at] = S.len; bd.i32[3 * sc + at] = S.mode;
}

// FIR estimates of the three filters (lsff_calc, gsff.py:156-177: rows 0 / 1 of the gain times the last N measurements)
// from the register history.  The gain of a constant-velocity least-squares filter is AFFINE in the age a of a
// measurement: g_N[a] = alpha_N - beta_N * a  (closed_form_gain: c_j = 1/N + t_j (N+1)/2 / sum t^2, t_j = (N-1)/2 - a), so
//     x-hat_N = alpha_N * S0_N - beta_N * S1_N,   S0_N = sum_{a<N} h[a],   S1_N = sum_{a<N} a h[a],
// and the three horizons share ONE pass over the history: two running sums per coordinate, read off where a horizon
// ends (a uniform bit test per entry).  124 float64 operations instead of 3 x 2 x N multiply-adds against a table of
// gains that no register file holds (192 constants; as scalar operands they spilled, 1200 v_readlane / v_writelane).
// Rounding differs from the table form by ~1e-11 px on positions of ~1e3 px (the sums reach 5e5 before they are scaled).
// (struct BlGains { alpha[filter][x / y row], beta[..][..] }: track.hip, the host handle holds one)
__device__ __forceinline__ void bl_fir(const BlSeat &S, const BlGains &g, int n0, int n1, int n2, unsigned ends,
                                       double *x0, double *x1)
{
    double s0x = 0.0, s1x = 0.0, s0y = 0.0, s1y = 0.0;
    double q0x[BL_NF], q1x[BL_NF], q0y[BL_NF], q1y[BL_NF];
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) { q0x[f] = q1x[f] = q0y[f] = q1y[f] = 0.0; }
#pragma unroll
    for (int a = 0; a < BL_HB; ++a) {
        s0x = s0x + S.hx[a];
        s0y = s0y + S.hy[a];
        if (a > 0) {
            s1x = __builtin_fma((double)a, S.hx[a], s1x);
            s1y = __builtin_fma((double)a, S.hy[a], s1y);
        }
        if (__builtin_amdgcn_readfirstlane((ends >> a) & 1u)) {    // a horizon ends with this entry (uniform)
            if (a + 1 == n0) { q0x[0] = s0x; q1x[0] = s1x; q0y[0] = s0y; q1y[0] = s1y; }
            else if (a + 1 == n1) { q0x[1] = s0x; q1x[1] = s1x; q0y[1] = s0y; q1y[1] = s1y; }
            else { q0x[2] = s0x; q1x[2] = s1x; q0y[2] = s0y; q1y[2] = s1y; }
        }
    }
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) {
        x0[f] = g.alpha[f][0] * q0x[f] - g.beta[f][0] * q1x[f];
        x1[f] = g.alpha[f][1] * q0y[f] - g.beta[f][1] * q1y[f];
    }
}

// GaussianSumFIR.correct + predict of one track by its lane (gsff.py:204-347; the statement order of gsff_wave).
__device__ __forceinline__ void bl_gsff(BlSeat &S, const TrackerDev &t, const BlGains &gt, double z0, double z1,
                                        bool fresh, double &o0, double &o1)
{
    const int nf = t.n_f, L = t.hist_cap;
    const int n0 = t.n_i[0], n1 = nf > 1 ? t.n_i[1] : 0, n2 = nf > 2 ? t.n_i[2] : 0;
    const unsigned ends = (1u << (n0 - 1)) | (nf > 1 ? 1u << (n1 - 1) : 0u) | (nf > 2 ? 1u << (n2 - 1) : 0u);
    int len = fresh ? 0 : S.len, mode = fresh ? 0 : S.mode;
    if (len == 0) {      // history starts as n_i[0] copies of the first measurement (entries beyond are never read)
#pragma unroll
        for (int e = 0; e < BL_HB; ++e) { S.hx[e] = z0; S.hy[e] = z1; }
        len = n0;
    }
    bool grew = false;   // gsff.py:283-289: while len(history) >= n_i[mode]: mode += 1
    if (mode == 0 && 0 < nf && len >= n0) { mode = 1; grew = true; }
    if (mode == 1 && 1 < nf && len >= n1) { mode = 2; grew = true; }
    if (mode == 2 && 2 < nf && len >= n2) { mode = 3; grew = true; }
    if (grew) {
        double x0[BL_NF], x1[BL_NF];
        bl_fir(S, gt, n0, n1, n2, ends, x0, x1);
        const double w0 = mode == 1 ? 1.0 : (mode == 2 ? 0.5 : 1.0 / 3.0);
#pragma unroll
        for (int f = 0; f < BL_NF; ++f)
            if (f < mode) { S.xa[f] = x0[f]; S.xb[f] = x1[f]; S.w[f] = w0; }
    }
    // append the measurement: every entry one frame older
#pragma unroll
    for (int e = BL_HB - 1; e > 0; --e) { S.hx[e] = S.hx[e - 1]; S.hy[e] = S.hy[e - 1]; }
    S.hx[0] = z0; S.hy[0] = z1;
    if (len < L) ++len;
    double nx0[BL_NF], nx1[BL_NF];
    bl_fir(S, gt, n0, n1, n2, ends, nx0, nx1);
    // likelihoods of the measurement under last frame's estimates (gsff.py:179-202)
    double lik[BL_NF], total = 0.0;
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) {
        const double d0 = z0 - S.xa[f], d1 = z1 - S.xb[f];
        double q = d0 * d0;
        q = q + d1 * d1;
        double l = exp(-0.5 * q);
        if (l < t.lik_min) l = t.lik_min;
        lik[f] = l;
        total = (f < mode) ? total + l * S.w[f] : total;
    }
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) {
        const double wn = lik[f] * S.w[f] / total;
        S.w[f] = (f < mode) ? wn : S.w[f];
    }
    // output = np.sum(x_hat * w, axis=1) = a0 + (a1 + a2)
    double f0 = 0.0, f1 = 0.0, r0 = 0.0, r1 = 0.0;
#pragma unroll
    for (int f = 0; f < BL_NF; ++f)
        if (f < mode) {
            const double a = S.xa[f] * S.w[f], b = S.xb[f] * S.w[f];
            if (f == 0) { f0 = a; f1 = b; }
            else if (f == 1) { r0 = a; r1 = b; }
            else { r0 = r0 + a; r1 = r1 + b; }
        }
    o0 = mode > 1 ? f0 + r0 : f0;
    o1 = mode > 1 ? f1 + r1 : f1;
    // predict: the new estimates, weighted
    f0 = f1 = r0 = r1 = 0.0;
#pragma unroll
    for (int f = 0; f < BL_NF; ++f)
        if (f < mode) {
            S.xa[f] = nx0[f]; S.xb[f] = nx1[f];
            const double a = S.xa[f] * S.w[f], b = S.xb[f] * S.w[f];
            if (f == 0) { f0 = a; f1 = b; }
            else if (f == 1) { r0 = a; r1 = b; }
            else { r0 = r0 + a; r1 = r1 + b; }
        }
    S.px = mode > 1 ? f0 + r0 : f0;
    S.py = mode > 1 ? f1 + r1 : f1;
    S.len = len; S.mode = mode;
}

// Nearest detection of a prediction among the frame's detections in LDS: D.min(1) / D.argmin(1) of tracker.py:151-163.
struct BlNear { double s; int q, col; };
__device__ __forceinline__ BlNear bl_search(const uint32_t *buf, double px, double py, int m)
{
    const float *hdr = reinterpret_cast<const float *>(buf);
    const double x0 = (double)hdr[0], y0 = (double)hdr[1], cell = (double)hdr[2], inv = (double)hdr[3];
    const int G = bl_grid_n(m);
    const unsigned short *start = reinterpret_cast<const unsigned short *>(buf + 16);
    const float2 *xy = reinterpret_cast<const float2 *>(buf + 16 + bl_start_dwords(G));
    const unsigned short *items = reinterpret_cast<const unsigned short *>(buf + 16 + bl_start_dwords(G) + 2 * bl_pad8(m));
    int cx = (int)floor((px - x0) * inv), cy = (int)floor((py - y0) * inv);
    cx = cx < 0 ? 0 : (cx > G - 1 ? G - 1 : cx);
    cy = cy < 0 ? 0 : (cy > G - 1 ? G - 1 : cy);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    BlNear r{inf, 0, 0x7FFFFFFF};
    for (int k = 1; k <= BL_KMAX + 1; ++k) {
        const bool whole = k > BL_KMAX;                   // a track far from everything: every detection
        const int xl = whole ? 0 : max(cx - k, 0), xh = whole ? G - 1 : min(cx + k, G - 1);
        const int yl = whole ? 0 : max(cy - k, 0), yh = whole ? G - 1 : min(cy + k, G - 1);
        // smallest squared distance, the smallest one above it, and the lowest column among the equal smallest
        double best = inf, second = inf;
        int bq = 0, bcol = 0x7FFFFFFF;
        for (int iy = yl; iy <= yh; ++iy) {
            const int a = start[iy * G + xl], b = start[iy * G + xh + 1];
            for (int q = a; q < b; ++q) {
                const float2 c = xy[q];
                const double dx = px - (double)c.x;
                const double dy = py - (double)c.y;
                double s = dx * dx;
                s = s + dy * dy;
                if (s < best) { second = best; best = s; bq = q; bcol = items[q]; }
                else if (s == best) { const int col = items[q]; if (col < bcol) { bcol = col; bq = q; } }
                else if (s < second) second = s;
            }
        }
        // distance from the track to the outside of the block of cells; a side of the block on the edge of the grid has
        // nothing beyond it (every detection lies inside the grid).  1e-3 of a cell: cells were assigned in float arithmetic
        double bound = inf;
        if (!whole) {
            if (xl > 0) bound = fmin(bound, px - (x0 + xl * cell));
            if (xh < G - 1) bound = fmin(bound, (x0 + (xh + 1) * cell) - px);
            if (yl > 0) bound = fmin(bound, py - (y0 + yl * cell));
            if (yh < G - 1) bound = fmin(bound, (y0 + (yh + 1) * cell) - py);
            bound -= 1e-3 * cell;
        }
        const bool done = bound == inf || (best < inf && bound > 0.0 && bound * bound > best * (1.0 + 1e-9));
        if (!done) continue;
        // sqrt is monotone but two different s can round to the same root, and then the lower column wins: only when some
        // s lies within 2^-48 (relative) above the smallest are the rounded roots themselves compared
        if (second <= best + best * 0x1p-48) {
            const double d_min = sqrt(best), near_limit = best + best * 0x1p-48;
            for (int iy = yl; iy <= yh; ++iy) {
                const int a = start[iy * G + xl], b = start[iy * G + xh + 1];
                for (int q = a; q < b; ++q) {
                    const float2 c = xy[q];
                    const double dx = px - (double)c.x;
                    const double dy = py - (double)c.y;
                    double s = dx * dx;
                    s = s + dy * dy;
                    if (s <= near_limit && sqrt(s) == d_min) { const int col = items[q]; if (col < bcol) { bcol = col; bq = q; } }
                }
            }
        }
        r.s = best; r.q = bq; r.col = bcol;
        return r;
    }
    return r;   // (not reached: the last round is `whole` and always done)
}

// ---- the kernel ----------------------------------------------------------------------------------------------------
struct BlShared {      // static part of the LDS
    int cnt[BL_MAX_BATCH];           // detections per frame (clamped)
    int used[2], n_dead[2];          // per frame parity: claims made, tracks deregistered
    int dead_id[2][BL_THREADS];      // ids of the tracks deregistered in the frame
    int wave_cnt[2][BL_WAVES];       // registration: per-wave counts of the two ranked lists
    int set_state[2];
};

__host__ __device__ inline int bl_md_padded(int max_det) { return (max_det + 3) / 4 * 4; }
__host__ __device__ inline size_t bl_lds_bytes(int max_det)
{
    return 2 * 4 * (size_t)bl_grid_dwords_max(max_det) + 2 * 12 * (size_t)bl_md_padded(max_det) + 4 * BL_TABLE + 64;
}

__global__ __launch_bounds__(BL_THREADS) void k_batch(TrackerDev t, BatchDev bd, const float *__restrict__ det_all,
                                                      const int32_t *__restrict__ det_count, int batch, int frame0,
                                                      ysmr_row *rows, long long rows_capacity, long long *row_count,
                                                      BlGains gt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_dyn[];
    __shared__ BlShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int md = t.max_det, mdp = bl_md_padded(md);
    const int bufw = bl_grid_dwords_max(md);
    unsigned long long *s_key[2] = {s_dyn, s_dyn + mdp};                          // smallest proposing distance per column
    uint32_t *s_cid[2] = {reinterpret_cast<uint32_t *>(s_dyn + 2 * mdp), reinterpret_cast<uint32_t *>(s_dyn + 2 * mdp) + mdp};
    uint32_t *s_buf[2] = {s_cid[1] + mdp, s_cid[1] + mdp + bufw};                 // the frames' detections (k_bgrid blocks)
    uint32_t *s_tab = s_buf[1] + bufw;                                            // CPython set model
    const int seats = min(t.capacity, BL_THREADS);

    // ---- start of the batch: counters, this lane's track, the first frame's detections
    int n = *t.n_tracks, next_id = *t.next_id;
    long long base = *row_count;
    for (int f = tid; f < batch; f += BL_THREADS) {
        int m = det_count[f];
        if (m > md) { m = md; atomicOr(t.err, ERR_DET_CLAMPED); }
        sh.cnt[f] = m < 0 ? 0 : m;
    }
    if (tid < 2) { sh.used[tid] = 0; sh.n_dead[tid] = 0; }
    BlSeat S;
    bl_seat_blank(S);
    if (tid < n) { bl_seat_load(S, bd, tid); S.alive = true; S.rank = tid; }
    __syncthreads();
    auto dma = [&](int f) {        // frame f's block -> s_buf[f & 1], whole 1-KiB pieces, a wave-instruction each
        const int pieces = bl_grid_dwords(sh.cnt[f]) >> 8;
        const char *src = bd.grid + (size_t)bd.grid_stride * f;
        for (int c = wave; c < pieces; c += BL_WAVES) {
            const uint32_t lds = (uint32_t)(uintptr_t)(s_buf[f & 1] + c * 256);
            const uint32_t off = (uint32_t)c * 1024u + (uint32_t)lane * 16u;
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(lds)), "v"(off), "s"(src) : "memory");
        }
    };
    dma(0);
    for (int c = tid; c < sh.cnt[0]; c += BL_THREADS) { s_key[0][c] = ~0ull; s_cid[0][c] = 0xFFFFFFFFu; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int f = 0; f < batch; ++f) {
        const int par = f & 1;
        const int m = sh.cnt[f], m_next = f + 1 < batch ? sh.cnt[f + 1] : 0;
        const uint32_t *buf = s_buf[par];
        if (f + 1 < batch) dma(f + 1);
        // ---- each track proposes its nearest detection (tracker.py:151-163)
        const bool propose = S.alive && m > 0;
        BlNear nr{0.0, 0, 0};
        unsigned long long key = 0;
        if (propose) {
            nr = bl_search(buf, S.px, S.py, m);
            key = (unsigned long long)__double_as_longlong(sqrt(nr.s));
            atomicMin(&s_key[par][nr.col], key);
        }
        block_sync<true>();
        // (the other parity's tables and counters were last read before the end of the previous frame)
        for (int c = tid; c < m_next; c += BL_THREADS) { s_key[par ^ 1][c] = ~0ull; s_cid[par ^ 1][c] = 0xFFFFFFFFu; }
        if (tid == 0) { sh.used[par ^ 1] = 0; sh.n_dead[par ^ 1] = 0; }
        if (propose && key == s_key[par][nr.col]) atomicMin(&s_cid[par][nr.col], (uint32_t)S.id);
        block_sync<true>();
        // ---- claims (tracker.py:171-189), ageing and deregistration (:95-107, 198-211)
        const bool age = (m == 0) || (n > 0 && n >= m);
        const bool mine = propose && s_cid[par][nr.col] == (uint32_t)S.id;
        double z0 = S.px, z1 = S.py;
        float box[3] = {S.info[0], S.info[1], S.info[2]};
        bool fresh = false, died = false;
        if (mine) {
            const float2 c = reinterpret_cast<const float2 *>(buf + 16 + bl_start_dwords(bl_grid_n(m)))[nr.q];
            z0 = (double)c.x; z1 = (double)c.y;
            const float *d = det_all + ((size_t)f * md + nr.col) * 5;
            box[0] = d[2]; box[1] = d[3]; box[2] = d[4];
            S.gone = 0;
        } else if (S.alive && age) {
            ++S.gone;
            box[0] = box[1] = box[2] = 0.f;
            if ((double)S.gone > t.max_gone) { S.alive = false; died = true; }
        }
        {
            const unsigned long long bm = __ballot(mine), bx = __ballot(died);
            if (lane == 0 && bm) atomicAdd(&sh.used[par], (int)__popcll(bm));
            if (bx) {      // the ids of the deregistered tracks: every younger track moves up one table row
                int at = 0;
                if (lane == 0) at = atomicAdd(&sh.n_dead[par], (int)__popcll(bx));
                at = __builtin_amdgcn_readfirstlane(at);
                if (died) sh.dead_id[par][at + __popcll(bx & below)] = S.id;
            }
        }
        // ---- registration (tracker.py:135-137, 212-217): unclaimed columns become tracks, in CPython set order
        int n_new = 0, n_new_all = 0;
        if (m > 0 && (n == 0 || n < m)) {        // (uniform; nobody was aged in such a frame)
            int *unused = reinterpret_cast<int *>(s_key[par ^ 1]), *newcols = unused + mdp;
            uint32_t *list = s_cid[par ^ 1];
            __syncthreads();
            if (n == 0) {
                for (int c = tid; c < m; c += BL_THREADS) newcols[c] = c;
                n_new_all = m;
            } else {
                // the unclaimed columns in ascending order: K consecutive columns per thread, ranked by a wave scan
                const int K = (m + BL_THREADS - 1) / BL_THREADS;
                const int c0 = tid * K, c1 = min(c0 + K, m);
                int cnt = 0;
                for (int c = c0; c < c1; ++c) cnt += s_cid[par][c] == 0xFFFFFFFFu;
                int incl = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                if (lane == 63) sh.wave_cnt[0][wave] = incl;
                __syncthreads();
                int at = incl - cnt, total = 0;
#pragma unroll
                for (int k = 0; k < BL_WAVES; ++k) { const int v = sh.wave_cnt[0][k]; at += k < wave ? v : 0; total += v; }
                for (int c = c0; c < c1; ++c)
                    if (s_cid[par][c] == 0xFFFFFFFFu) unused[at++] = c;
                __syncthreads();
                int cnt_set = cpython_order_lds<BL_THREADS, BL_TABLE>(unused, total, m, sh.used[par], newcols, s_tab, list, sh.set_state);
                if (cnt_set < 0) { if (tid == 0) atomicOr(t.err, ERR_TRACK_CAPACITY); cnt_set = 0; }
                n_new_all = cnt_set;
            }
            n_new = n_new_all;
            if (n + n_new > seats) {
                if (tid == 0) atomicOr(t.err, ERR_TRACK_CAPACITY);
                n_new = seats - n;
            }
            // free lanes take the new tracks, in lane order
            const bool free_lane = !S.alive && tid < seats;
            const unsigned long long bf = __ballot(free_lane);
            if (lane == 0) sh.wave_cnt[1][wave] = (int)__popcll(bf);
            __syncthreads();
            int fr = (int)__popcll(bf & below);
#pragma unroll
            for (int k = 0; k < BL_WAVES; ++k) fr += k < wave ? sh.wave_cnt[1][k] : 0;
            if (free_lane && fr < n_new) {
                const int c = newcols[fr];
                const float *d = det_all + ((size_t)f * md + c) * 5;
                z0 = (double)d[0]; z1 = (double)d[1];
                box[0] = d[2]; box[1] = d[3]; box[2] = d[4];
                S.id = next_id + fr; S.rank = n + fr; S.gone = 0;
                S.alive = true; fresh = true;
            }
            __syncthreads();     // (the lists lived in the next frame's tables)
            for (int c = tid; c < m_next; c += BL_THREADS) { s_key[par ^ 1][c] = ~0ull; s_cid[par ^ 1][c] = 0xFFFFFFFFu; }
        }
        // ---- the filter bank (tracker.py:219-227)
        double o0 = z0, o1 = z1;
        if (S.alive) {
            S.info[0] = box[0]; S.info[1] = box[1]; S.info[2] = box[2];
            if (t.use_gsff) bl_gsff(S, t, gt, z0, z1, fresh, o0, o1);
            else { S.px = z0; S.py = z1; }
        }
        // ---- end of the frame: the next frame's detections have landed, the frame's counts are complete
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int n_dead = sh.n_dead[par];
        if (n_dead && S.alive)
            for (int k = 0; k < n_dead; ++k) S.rank -= sh.dead_id[par][k] < S.id;
        const int n_live = n - n_dead + n_new;
        if (S.alive && base + S.rank < rows_capacity) {      // track_eval.py:313-316
            ysmr_row rr;
            rr.frame = frame0 + f;
            rr.track_id = S.id;
            rr.x = o0; rr.y = o1;
            rr.w = S.info[0]; rr.h = S.info[1]; rr.angle = S.info[2];
            rr.disappeared = S.gone;
            rows[base + S.rank] = rr;
        }
        if (tid == 0 && base + n_live > rows_capacity) atomicOr(t.err, ERR_ROWS_CAPACITY);
        base += n_live;
        n = n_live;
        next_id += n_new_all;
    }
    // ---- end of the batch: the table goes back to HBM in row order
    if (S.alive) bl_seat_store(S, bd, S.rank);
    if (tid == 0) { *t.n_tracks = n; *t.next_id = next_id; *row_count = base; }
}

// ---- conversions between the seat-major rest format and the per-slot layout of k_frame / k_link + k_track ----------
// (a: the CURRENT parity view of the per-slot state)
__global__ void k_to_batch(TrackerDev a, BatchDev bd)
{
    const int n = *a.n_tracks, cap = a.capacity, L = a.hist_cap, nf = a.n_f;
    const size_t sc = (size_t)bd.seat_cap;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const int slot = a.order[r];
        double *p = bd.f64 + r;
        const double *hist = a.hist + (size_t)slot * 2 * L;
        for (int e = 0; e < BL_HB; ++e) {
            p[sc * e] = e < L ? hist[2 * e] : 0.0;
            p[sc * (BL_HB + e)] = e < L ? hist[2 * e + 1] : 0.0;
        }
        const double *rec = a.rec + (size_t)slot * a.rec_stride;
        for (int f = 0; f < BL_NF; ++f) {
            p[sc * (2 * BL_HB + f)] = f < nf ? rec[2 + f] : 0.0;
            p[sc * (2 * BL_HB + BL_NF + f)] = f < nf ? rec[2 + nf + f] : 0.0;
            p[sc * (2 * BL_HB + 2 * BL_NF + f)] = f < nf ? rec[2 + 2 * nf + f] : 0.0;
        }
        p[sc * (2 * BL_HB + 3 * BL_NF)] = a.pos[slot];
        p[sc * (2 * BL_HB + 3 * BL_NF + 1)] = a.pos[cap + slot];
        for (int k = 0; k < 3; ++k) bd.f32[sc * k + r] = a.info[k * cap + slot];
        bd.i32[r] = a.id[slot];
        bd.i32[sc + r] = a.gone[a.gone_by_row ? r : slot];
        bd.i32[2 * sc + r] = (int)(__double_as_longlong(rec[0]) & 0xFFFFFFFFll);
        bd.i32[3 * sc + r] = (int)(__double_as_longlong(rec[1]) & 0xFFFFFFFFll);
    }
}

// (d: the parity-0 view; row r takes slot r, the free stack hands out slot n next)
__global__ void k_to_std(TrackerDev d, BatchDev bd)
{
    const int n = *d.n_tracks, cap = d.capacity, L = d.hist_cap, nf = d.n_f;
    const size_t sc = (size_t)bd.seat_cap;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < cap; r += gridDim.x * blockDim.x) {
        if (r < cap - n) d.free_slots[r] = cap - 1 - r;
        if (r == 0) *d.n_free = cap - n;
        if (r >= n) continue;
        const double *p = bd.f64 + r;
        double *hist = d.hist + (size_t)r * 2 * L;
        for (int e = 0; e < L && e < BL_HB; ++e) { hist[2 * e] = p[sc * e]; hist[2 * e + 1] = p[sc * (BL_HB + e)]; }
        double *rec = d.rec + (size_t)r * d.rec_stride;
        rec[0] = __longlong_as_double((long long)(unsigned int)bd.i32[2 * sc + r]);
        rec[1] = __longlong_as_double((long long)(unsigned int)bd.i32[3 * sc + r]);
        for (int f = 0; f < nf && f < BL_NF; ++f) {
            rec[2 + f] = p[sc * (2 * BL_HB + f)];
            rec[2 + nf + f] = p[sc * (2 * BL_HB + BL_NF + f)];
            rec[2 + 2 * nf + f] = p[sc * (2 * BL_HB + 2 * BL_NF + f)];
        }
        d.pos[r] = p[sc * (2 * BL_HB + 3 * BL_NF)];
        d.pos[cap + r] = p[sc * (2 * BL_HB + 3 * BL_NF + 1)];
        for (int k = 0; k < 3; ++k) d.info[k * cap + r] = bd.f32[sc * k + r];
        d.id[r] = bd.i32[r];
        d.order[r] = r;
        d.gone[r] = bd.i32[sc + r];       // (row r = slot r: right for either indexing)
        d.row_gone[r] = bd.i32[sc + r];
    }
}

__global__ void k_peek_batch(TrackerDev t, BatchDev bd, int32_t *ids, double *xy, int32_t *gone, int32_t *n_out)
{
    const int n = *t.n_tracks;
    const size_t sc = (size_t)bd.seat_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && n_out) *n_out = n;
    if (i >= n) return;
    if (ids) ids[i] = bd.i32[i];
    if (xy) { xy[2 * i] = bd.f64[sc * (2 * BL_HB + 3 * BL_NF) + i]; xy[2 * i + 1] = bd.f64[sc * (2 * BL_HB + 3 * BL_NF + 1) + i]; }
    if (gone) gone[i] = bd.i32[sc + i];
}
