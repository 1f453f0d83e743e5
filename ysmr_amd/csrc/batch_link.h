// batch_link.h -- the link of a whole BATCH of frames in one launch (included by track.hip, inside its namespace).
//
// CentroidTracker.update (ysmr/tracker.py:93-230) + GaussianSumFIR.correct / predict (ysmr/gsff.py:204-347) + row
// emission (ysmr/track_eval.py:313-316) for `batch` consecutive frames: ONE workgroup of BL_THREADS = 768 threads on ONE
// compute unit, a track per LANE, everything a frame needs of a track -- the filter bank's window sums, weights and
// estimates, prediction, box, id, counters: 55 registers -- in the lane from the first frame of the batch to the last.
//
//   detections   k_bgrid (one workgroup per frame, the whole batch at once, before this kernel) bins every frame's
//                detections into a uniform grid of cells (counting sort) and leaves header | cell starts (u16) |
//                centres in cell order | their column numbers (u16) as one contiguous block per frame; this kernel
//                brings the block of frame f+1 into LDS by LDS-DMA while it works on frame f.
//   row minimum  tracker.py:151-163 reads only D.min(1) and D.argmin(1): a lane looks at the 3 x 3 cells around its
//                prediction, in float, and evaluates the winner once in float64 (bl_search); what float cannot decide
//                -- a candidate a hair from the best, a nearer detection possibly outside the block -- the wave settles
//                exactly over all detections (bl_search_wave: rowmin_wave's tie rule, lowest column among equal ROUNDED
//                distances).
//   claims       the winner of a detection column is the proposer with the smallest (distance, id): two LDS atomicMin
//                rounds (ids ascend with table rows, so (distance, id) orders like the reference's (distance, row)).
//   lifecycle    ageing / deregistration per lane; a lane that loses its track is simply free; new tracks take free
//                lanes, in CPython set order of the unclaimed columns (cpython_order_lds).  Table ROWS (the order of the
//                reference's OrderedDict = ascending id = the order of a frame's rows) are kept as a per-lane rank:
//                a death lowers the rank of every younger track by one, a birth appends.
//   filter bank  The gain of a constant-velocity least-squares filter is AFFINE in the age a of a measurement:
//                g_N[a] = alpha_N - beta_N a (closed_form_gain), so x-hat_N = alpha_N S0_N - beta_N S1_N with the window
//                sums S0_N = sum_{a<N} h[a], S1_N = sum_{a<N} a h[a].  A new measurement z turns them into
//                S1' = S1 + S0 - N h[N-1], S0' = S0 + z - h[N-1]: six sums per coordinate in registers, updated with four
//                operations each, and of the 31 measurements of history a frame touches FOUR: the one it appends and the
//                three that leave the windows.  The history therefore lives in HBM, a ring of 32 frames x 768 seats
//                (every live track appends exactly one measurement per frame, so one head serves all seats and a ring
//                position is one coalesced line per wave); the leaving entries are requested before the claims and used
//                after them.  The sums are recomputed from the ring, exactly, in every frame whose NUMBER is a multiple of
//                BL_REFRESH = 64 (not at the start of a launch: the rows must not depend on how the frames were batched),
//                so rounding drift is bounded by 64 updates: ~1e-11 px, the size of a from-scratch sum's own rounding.
//   rows         one 40-byte ysmr_row per live lane at rows[base + rank], fire and forget.
//
// Three workgroup barriers per frame (after each atomic round, and at the end of the frame, where the next frame's
// detections must have landed).  Why not the history in registers (the first build of this kernel: 158 registers per
// track): 512 tracks fill a compute unit's register file, BASELINE configs[2] holds up to 535; and a frame then costs the
// 186 float64 operations of re-summing and shifting 62 values per track, on ONE unit's float64 pipe (7.4 us per frame).
//
// Between launches the small state rests in HBM seat by seat (a seat keeps its track for the track's whole life; free
// seats are flagged), next to the ring.  k_to_std / k_to_batch convert to and from the per-slot layout of k_frame /
// k_link + k_track (ysmr_tracker_update, tables beyond this kernel's 768 seats).
#pragma once

// (BL_* constants and struct BatchDev: track.hip, next to TrackerDev -- the host handle holds one)

// ---- a frame's detections as this kernel wants them: one block of dwords per frame ------------------------------
//   [0, 16)            header: x0, y0, cell, 1 / cell (f32), cells per side G, m (i32)
//   [16, 16 + SW)      start[G * G + 1] as u16: first item of each cell (row-major), the last entry = m
//   [.., + 2 * MP)     centres (x, y) f32 of the detections in cell order;  MP = m rounded up to 8
//   [.., + MP / 2)     their column numbers as u16
// (cells per side so that a cell holds ~0.45 detections: the three cells of a row of the 3 x 3 block around a prediction
// then hold more than five candidates once in 200 rows, and the block reaches one cell -- ~1.7 mean nearest-neighbour
// distances -- beyond the prediction's own cell)
__host__ __device__ inline int bl_grid_n(int m) { return m <= 128 ? 16 : (m <= 600 ? 32 : (m <= 1300 ? 48 : 64)); }
__host__ __device__ inline int bl_start_dwords(int G) { return ((G * G + 2) / 2 + 3) / 4 * 4; }
__host__ __device__ inline int bl_pad8(int m) { return (m + 7) / 8 * 8; }
__host__ __device__ inline int bl_grid_dwords(int m)   // rounded up to whole 1-KiB pieces (one LDS-DMA wave-instruction)
{
    const int raw = 16 + bl_start_dwords(bl_grid_n(m)) + 2 * bl_pad8(m) + bl_pad8(m) / 2;
    return (raw + 255) / 256 * 256;
}
__host__ __device__ inline int bl_grid_dwords_max(int max_det)
{
    int most = bl_grid_dwords(max_det);
    const int steps[4] = {128, 600, 1300, max_det};      // (the size is monotone in m between two changes of the grid)
    for (int k = 0; k < 4; ++k)
        if (steps[k] <= max_det && bl_grid_dwords(steps[k]) > most) most = bl_grid_dwords(steps[k]);
    return most;
}

// One workgroup per frame: bounding box of the centres, G x G cells over it with one cell of margin, counting sort.  The
// detections are read ONCE (up to BG_PER per thread, kept in registers through the three passes), the block is assembled
// in LDS and leaves with 16-byte stores: one round of loads and one of stores instead of the five dependent round trips of
// the first version (35 us per batch beside the next batch's detection kernels).
constexpr int BG_THREADS = 256, BG_PER = 10;      // 2560 >= the 2456 detections a one-launch link serves
__global__ __launch_bounds__(BG_THREADS) void k_bgrid(const float *__restrict__ det_all, const int32_t *__restrict__ det_count,
                                                      int max_det, char *grid, unsigned grid_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_out[];      // the block as it will lie in HBM
    __shared__ int s_cnt[64 * 64];
    __shared__ float s_red[4][4];
    __shared__ int s_wave_sum[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *det = det_all + (size_t)f * max_det * 5;
    int m = det_count[f];
    m = m < 0 ? 0 : (m > max_det ? max_det : m);
    const int G = bl_grid_n(m), cells = G * G;
    unsigned short *start16 = reinterpret_cast<unsigned short *>(s_out + 16);
    float2 *xy = reinterpret_cast<float2 *>(s_out + 16 + bl_start_dwords(G));
    unsigned short *items = reinterpret_cast<unsigned short *>(s_out + 16 + bl_start_dwords(G) + 2 * bl_pad8(m));
    float x[BG_PER], y[BG_PER];
#pragma unroll
    for (int k = 0; k < BG_PER; ++k) {
        const int j = tid + k * BG_THREADS;
        x[k] = j < m ? det[(size_t)j * 5] : 0.f;
        y[k] = j < m ? det[(size_t)j * 5 + 1] : 0.f;
    }
    for (int c = tid; c < cells; c += BG_THREADS) s_cnt[c] = 0;
    float lo_x = 3.0e38f, lo_y = 3.0e38f, hi_x = -3.0e38f, hi_y = -3.0e38f;
#pragma unroll
    for (int k = 0; k < BG_PER; ++k)
        if (tid + k * BG_THREADS < m) { lo_x = fminf(lo_x, x[k]); hi_x = fmaxf(hi_x, x[k]); lo_y = fminf(lo_y, y[k]); hi_y = fmaxf(hi_y, y[k]); }
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
    if (tid < 16) s_out[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        float *h = reinterpret_cast<float *>(s_out);
        h[0] = x0; h[1] = y0; h[2] = cell; h[3] = inv;
        s_out[4] = (uint32_t)G; s_out[5] = (uint32_t)m;
    }
    int cell_of[BG_PER];
#pragma unroll
    for (int k = 0; k < BG_PER; ++k) {
        int cx = (int)floorf((x[k] - x0) * inv), cy = (int)floorf((y[k] - y0) * inv);
        cx = cx < 0 ? 0 : (cx > G - 1 ? G - 1 : cx);
        cy = cy < 0 ? 0 : (cy > G - 1 ? G - 1 : cy);
        cell_of[k] = cy * G + cx;
        if (tid + k * BG_THREADS < m) atomicAdd(&s_cnt[cell_of[k]], 1);
    }
    __syncthreads();
    // exclusive scan of the counts: K consecutive cells per thread (K = 1, 4, 9 or 16), wave scan, wave sums
    const int K = cells / BG_THREADS;
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
    for (int k = 0; k < 16; ++k)
        if (k < K) {
            const int at = before + local[k];
            s_cnt[tid * K + k] = at;
            start16[tid * K + k] = (unsigned short)at;
        }
    if (tid == BG_THREADS - 1) { start16[cells] = (unsigned short)m; start16[cells + 1] = 0; }
    // (the padding of the lists reads as far away, should a lane ever look at it)
    for (int j = m + tid; j < bl_pad8(m); j += BG_THREADS) { xy[j] = make_float2(1.0e30f, 1.0e30f); items[j] = 0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BG_PER; ++k) {
        const int j = tid + k * BG_THREADS;
        if (j < m) {
            const int at = atomicAdd(&s_cnt[cell_of[k]], 1);
            items[at] = (unsigned short)j;
            xy[at] = make_float2(x[k], y[k]);
        }
    }
    __syncthreads();
    uint4 *out = reinterpret_cast<uint4 *>(grid + (size_t)grid_stride * f);
    const uint4 *img = reinterpret_cast<const uint4 *>(s_out);
    for (int i = tid; i < bl_grid_dwords(m) / 4; i += BG_THREADS) out[i] = img[i];
}

// ---- a track in registers -------------------------------------------------------------------------------------------
struct BlSeat {
    double s0x[BL_NF], s1x[BL_NF], s0y[BL_NF], s1y[BL_NF];   // window sums of the three horizons (x, y)
    double w[BL_NF], xa[BL_NF], xb[BL_NF];                   // filter weights, x-hat rows 0 / 1
    double px, py;                                           // CentroidTracker.objects[id]: the prediction (tracker.py:225)
    float info[3];
    int id, gone, len, mode, rank;
    bool alive;
};
// rest format, seat-major: f64 [BL_SF64][seat_cap] = w, xa, xb, px, py, the twelve window sums; f32 [3][seat_cap];
// i32 [6][seat_cap] = id, gone, history length, mode, rank, alive; ring double2 [BL_HB][seat_cap]

__device__ __forceinline__ void bl_seat_blank(BlSeat &S)
{
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) { S.s0x[f] = S.s1x[f] = S.s0y[f] = S.s1y[f] = 0.0; S.w[f] = S.xa[f] = S.xb[f] = 0.0; }
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
    for (int f = 0; f < BL_NF; ++f) { S.w[f] = p[sc * f]; S.xa[f] = p[sc * (BL_NF + f)]; S.xb[f] = p[sc * (2 * BL_NF + f)]; }
    S.px = p[sc * (3 * BL_NF)];
    S.py = p[sc * (3 * BL_NF + 1)];
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) {
        const double *q = p + sc * (3 * BL_NF + 2 + 4 * f);
        S.s0x[f] = q[0]; S.s1x[f] = q[sc]; S.s0y[f] = q[2 * sc]; S.s1y[f] = q[3 * sc];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) S.info[k] = bd.f32[sc * k + at];
    S.id = bd.i32[at]; S.gone = bd.i32[sc + at]; S.len = bd.i32[2 * sc + at]; S.mode = bd.i32[3 * sc + at];
    S.rank = bd.i32[4 * sc + at];
}

__device__ __forceinline__ void bl_seat_store(const BlSeat &S, const BatchDev &bd, int at)
{
    const size_t sc = (size_t)bd.seat_cap;
    double *p = bd.f64 + at;
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) { p[sc * f] = S.w[f]; p[sc * (BL_NF + f)] = S.xa[f]; p[sc * (2 * BL_NF + f)] = S.xb[f]; }
    p[sc * (3 * BL_NF)] = S.px;
    p[sc * (3 * BL_NF + 1)] = S.py;
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) {
        double *q = p + sc * (3 * BL_NF + 2 + 4 * f);
        q[0] = S.s0x[f]; q[sc] = S.s1x[f]; q[2 * sc] = S.s0y[f]; q[3 * sc] = S.s1y[f];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) bd.f32[sc * k + at] = S.info[k];
    bd.i32[at] = S.id; bd.i32[sc + at] = S.gone; bd.i32[2 * sc + at] = S.len; bd.i32[3 * sc + at] = S.mode;
    bd.i32[4 * sc + at] = S.rank;
}

// A ring entry as this compute unit's L2 holds it: the ring is written and read back by the same launch, frames apart;
// sc1 loads pass the unit's L1 (it is not refreshed by stores, the unit's own included, in any way the ISA promises).
__device__ __forceinline__ double2 bl_ring_load(const BatchDev &bd, int pos, int seat)
{
    const double *p = reinterpret_cast<const double *>(bd.ring + (size_t)pos * bd.seat_cap + seat);
    double2 v;
    v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
__device__ __forceinline__ void bl_ring_store(const BatchDev &bd, int pos, int seat, double x, double y)
{
    bd.ring[(size_t)pos * bd.seat_cap + seat] = make_double2(x, y);
}

// The window sums of a track from its history, exactly: the measurement of age a (0 = newest) is at ring position
// head - 1 - a; a filter's window holds the newest min(N, len) of them.  Done for every track at the frames whose index
// is a multiple of BL_REFRESH (a schedule in FRAME numbers: the rows of a video must not depend on how its frames were cut
// into batches), and by k_to_batch.
__device__ __forceinline__ void bl_sums_from_ring(BlSeat &S, const BatchDev &bd, int seat, int head, int n0, int n1, int n2)
{
    const int lim[BL_NF] = {min(n0, S.len), min(n1, S.len), min(n2, S.len)};
    double s0x = 0.0, s1x = 0.0, s0y = 0.0, s1y = 0.0;
    for (int a = 0; a < BL_HB - 1; ++a) {
        const double2 e = bl_ring_load(bd, (head - 1 - a) & (BL_HB - 1), seat);
        if (a < S.len) {
            s0x = s0x + e.x;
            s0y = s0y + e.y;
            s1x = __builtin_fma((double)a, e.x, s1x);
            s1y = __builtin_fma((double)a, e.y, s1y);
        }
#pragma unroll
        for (int f = 0; f < BL_NF; ++f)
            if (a + 1 == lim[f]) { S.s0x[f] = s0x; S.s1x[f] = s1x; S.s0y[f] = s0y; S.s1y[f] = s1y; }
    }
}

// GaussianSumFIR.correct + predict of one track by its lane (gsff.py:204-347; the statement order of gsff_wave).
// leave[f]: the measurement of age N_f - 1 (requested by the caller before the claims); head: the ring position this
// frame's measurement takes.
// (struct BlGains { alpha[filter][x / y row], beta[..][..] }: track.hip, the host handle holds one)
__device__ __forceinline__ void bl_gsff(BlSeat &S, const TrackerDev &t, const BatchDev &bd, const BlGains &g, int seat, int head,
                                        const double2 (&leave)[BL_NF], double z0, double z1, bool fresh, double &o0, double &o1)
{
    const int nf = t.n_f, L = t.hist_cap;
    const int n_i[BL_NF] = {t.n_i[0], nf > 1 ? t.n_i[1] : 0x7FFFFFFF, nf > 2 ? t.n_i[2] : 0x7FFFFFFF};
    int len = fresh ? 0 : S.len, mode = fresh ? 0 : S.mode;
    if (len == 0) {      // history starts as n_i[0] copies of the first measurement (gsff.py:281)
        const int n0 = n_i[0];
#pragma unroll
        for (int f = 0; f < BL_NF; ++f) { S.w[f] = 0.0; S.xa[f] = 0.0; S.xb[f] = 0.0; }
        const double c1 = (double)(n0 * (n0 - 1) / 2);
#pragma unroll
        for (int f = 0; f < BL_NF; ++f) {
            S.s0x[f] = (double)n0 * z0; S.s1x[f] = c1 * z0;
            S.s0y[f] = (double)n0 * z1; S.s1y[f] = c1 * z1;
        }
        for (int a = 0; a < n0; ++a) bl_ring_store(bd, (head - 1 - a) & (BL_HB - 1), seat, z0, z1);
        len = n0;
    }
    bool grew = false;   // gsff.py:283-289: while len(history) >= n_i[mode]: mode += 1
    if (mode == 0 && 0 < nf && len >= n_i[0]) { mode = 1; grew = true; }
    if (mode == 1 && 1 < nf && len >= n_i[1]) { mode = 2; grew = true; }
    if (mode == 2 && 2 < nf && len >= n_i[2]) { mode = 3; grew = true; }
    // Filters that are not switched on yet carry weight 0 (and finite estimates): every sum below may then run over all
    // three without a select -- adding a zero term is exact, so a0 + (a1 + a2) is the reference's a0, or a0 + a1
    if (grew) {          // the estimates of every active filter from the history as it stands, uniform weights
        const double w0 = mode == 1 ? 1.0 : (mode == 2 ? 0.5 : 1.0 / 3.0);
#pragma unroll
        for (int f = 0; f < BL_NF; ++f)
            if (f < mode) {
                S.xa[f] = __builtin_fma(-g.beta[f][0], S.s1x[f], g.alpha[f][0] * S.s0x[f]);
                S.xb[f] = __builtin_fma(-g.beta[f][1], S.s1y[f], g.alpha[f][1] * S.s0y[f]);
                S.w[f] = w0;
            }
    }
    // append the measurement: every entry one frame older, the entry of age N - 1 leaves a full window
    double nx0[BL_NF], nx1[BL_NF];
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) {
        const bool full = len >= n_i[f];
        // (a track born in this frame: its history is copies of z, written above -- after `leave` was requested)
        const double lx = full ? (fresh ? z0 : leave[f].x) : 0.0, ly = full ? (fresh ? z1 : leave[f].y) : 0.0;
        const double nn = -(double)(f < nf ? n_i[f] : 0);
        S.s1x[f] = __builtin_fma(nn, lx, S.s1x[f] + S.s0x[f]);
        S.s1y[f] = __builtin_fma(nn, ly, S.s1y[f] + S.s0y[f]);
        S.s0x[f] = (S.s0x[f] + z0) - lx;
        S.s0y[f] = (S.s0y[f] + z1) - ly;
        nx0[f] = __builtin_fma(-g.beta[f][0], S.s1x[f], g.alpha[f][0] * S.s0x[f]);
        nx1[f] = __builtin_fma(-g.beta[f][1], S.s1y[f], g.alpha[f][1] * S.s0y[f]);
    }
    bl_ring_store(bd, head, seat, z0, z1);
    if (len < L) ++len;
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
        total = total + l * S.w[f];
    }
    // w_i <- lik_i w_i / total (gsff.py:333-336) as lik_i w_i * (1 / total): one division for the three weights; a weight
    // may differ from the quotient in its last bit (the per-frame kernels divide three times)
    const double r_total = 1.0 / total;
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) S.w[f] = (lik[f] * S.w[f]) * r_total;
    // output = np.sum(x_hat * w, axis=1) = a0 + (a1 + a2)
    o0 = S.xa[0] * S.w[0] + (S.xa[1] * S.w[1] + S.xa[2] * S.w[2]);
    o1 = S.xb[0] * S.w[0] + (S.xb[1] * S.w[1] + S.xb[2] * S.w[2]);
    // predict: the new estimates, weighted
#pragma unroll
    for (int f = 0; f < BL_NF; ++f) { S.xa[f] = nx0[f]; S.xb[f] = nx1[f]; }
    S.px = S.xa[0] * S.w[0] + (S.xa[1] * S.w[1] + S.xa[2] * S.w[2]);
    S.py = S.xb[0] * S.w[0] + (S.xb[1] * S.w[1] + S.xb[2] * S.w[2]);
    S.len = len; S.mode = mode;
}

// Nearest detection of a prediction among the frame's detections in LDS: D.min(1) / D.argmin(1) of tracker.py:151-163.
//
// Fast path, per lane, straight-line and in FLOAT: the 3 x 3 cells around the prediction are three runs of the cell-ordered
// list (one per row of cells); their six bounds come with one round of LDS reads, the first five candidates of each run
// with a second.  A candidate's float squared distance and its slot become ONE 32-bit key (a non-negative float orders
// like its bits; the slot number replaces the four lowest mantissa bits), and the two smallest keys are kept with a
// minimum and a median-of-three per candidate -- no second pass, no index bookkeeping.  A run with more than five
// candidates is finished in a loop on the same two registers.  Float decides only what float can decide: the result
// stands when the second smallest key lies outside the float error band of the smallest (then the smallest is the
// float64 argmin among the block's candidates, and no tie rule is needed) and nothing outside the block can be as near
// (the block's boundary is farther than the best distance, with a margin far above rounding; cells were assigned in
// float arithmetic too).  Its float64 squared distance is then evaluated once, for the claim key.  Everything else -- a
// lost track far from every detection, two candidates a hair apart, an exact tie -- is left to bl_search_wave.
// Error band: |fx - px| <= 2^-24 * 4096 = 2.4e-4 for coordinates below 4096, the candidates are float already, so a float
// squared distance of a candidate within ~100 px is off by less than 2 * 200 * 2.4e-4 + 3 * 2^-24 * s < 0.1 + 2e-7 s, and the
// key drops 2^-19 s more.
struct BlNear { double s; float zx, zy; int col; bool done; };
// The kernel's dynamic LDS, at namespace scope and addressed by OFFSET everywhere: a pointer that went through an array
// or a struct of pointers comes back generic, and the compiler then reads LDS with flat_load -- twice the latency and a
// 64-bit address per access (97 of them in the first build of this kernel).
extern __shared__ __attribute__((aligned(16))) unsigned long long bl_lds[];
__device__ __forceinline__ uint32_t *bl_u32(int dword_off) { return reinterpret_cast<uint32_t *>(bl_lds) + dword_off; }
struct BlGridView {
    int start, xy, items, G;       // dword offsets into bl_lds; cells per side
    float x0, y0, cell, inv;
    __device__ __forceinline__ int start_at(int i) const { return reinterpret_cast<const unsigned short *>(bl_u32(start))[i]; }
    __device__ __forceinline__ int item_at(int i) const { return reinterpret_cast<const unsigned short *>(bl_u32(items))[i]; }
    __device__ __forceinline__ float2 xy_at(int i) const { return reinterpret_cast<const float2 *>(bl_u32(xy))[i]; }
};
__device__ __forceinline__ BlGridView bl_grid_view(int buf, int m)     // buf: dword offset of the frame's block
{
    const float *hdr = reinterpret_cast<const float *>(bl_u32(buf));
    BlGridView g;
    g.G = bl_grid_n(m);
    g.start = buf + 16;
    g.xy = buf + 16 + bl_start_dwords(g.G);
    g.items = buf + 16 + bl_start_dwords(g.G) + 2 * bl_pad8(m);
    g.x0 = hdr[0]; g.y0 = hdr[1]; g.cell = hdr[2]; g.inv = hdr[3];
    return g;
}
__device__ __forceinline__ double bl_dist2(double px, double py, float2 c)
{
    const double dx = px - (double)c.x;
    const double dy = py - (double)c.y;
    double s = dx * dx;
    s = s + dy * dy;
    return s;
}
__device__ __forceinline__ float bl_dist2f(float fx, float fy, float2 c)
{
    const float dx = fx - c.x, dy = fy - c.y;
    return __builtin_fmaf(dy, dy, dx * dx);
}
__device__ __forceinline__ uint32_t bl_med3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;                                          // (the compiler spells the median out as three min / max)
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ BlNear bl_search(const BlGridView &g, double px, double py, int m)
{
    constexpr int R = 3, C = 5;
    const int G = g.G;
    const float fx = (float)px, fy = (float)py;
    int cx = (int)floorf((fx - g.x0) * g.inv), cy = (int)floorf((fy - g.y0) * g.inv);
    const bool in_grid = cx >= 0 && cx < G && cy >= 0 && cy < G;
    cx = cx < 0 ? 0 : (cx > G - 1 ? G - 1 : cx);
    cy = cy < 0 ? 0 : (cy > G - 1 ? G - 1 : cy);
    const int xl = max(cx - 1, 0), xh = min(cx + 1, G - 1), yl = max(cy - 1, 0), yh = min(cy + 1, G - 1);
    int a[R], b[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int iy = min(yl + r, G - 1);
        a[r] = g.start_at(iy * G + xl);
        b[r] = g.start_at(iy * G + xh + 1);
        if (yl + r > yh) b[r] = a[r];                   // (a block cut by the edge of the grid has fewer rows)
    }
    // every candidate's read is issued before the first is used (left to itself the compiler, short of registers, waits
    // for each read before it issues the next: fifteen LDS round trips in a row); a slot beyond the end of its run reads
    // whatever follows in LDS and is masked afterwards
    float2 c[R * C];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float2 *run = reinterpret_cast<const float2 *>(bl_u32(g.xy)) + a[r];
#pragma unroll
        for (int j = 0; j < C; ++j) c[C * r + j] = run[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;        // the two smallest keys
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float d2 = bl_dist2f(fx, fy, c[C * r + j]);
            uint32_t key = (__float_as_uint(d2) & 0xFFFFFFF0u) | (uint32_t)(C * r + j);
            key = a[r] + j < b[r] ? key : 0xFFFFFFFFu;
            hi = bl_med3(lo, hi, key);
            lo = min(lo, key);
        }
    int q_extra = 0;
    bool more = false;
#pragma unroll
    for (int r = 0; r < R; ++r) more = more || (b[r] - a[r] > C);
    if (more) {                                        // the runs' sixth and later candidates: slot 15, position kept aside
#pragma unroll
        for (int r = 0; r < R; ++r)
            for (int qq = a[r] + C; qq < b[r]; ++qq) {
                const uint32_t key = (__float_as_uint(bl_dist2f(fx, fy, g.xy_at(qq))) & 0xFFFFFFF0u) | 15u;
                q_extra = key < lo ? qq : q_extra;
                hi = bl_med3(lo, hi, key);
                lo = min(lo, key);
            }
    }
    const float best = __uint_as_float(lo & 0xFFFFFFF0u), second = __uint_as_float(hi & 0xFFFFFFF0u);
    const float band = best + (0.2f + 5e-6f * best);   // (twice the error bound: both distances are off)
    const int slot = (int)(lo & 15u);
    const int run = slot >= 2 * C ? 2 : (slot >= C ? 1 : 0);
    const int bq = slot == 15 ? q_extra : (run == 2 ? a[2] : (run == 1 ? a[1] : a[0])) + slot - C * run;
    // Nothing outside the block is as near: a prediction inside the grid lies in its own cell, a whole cell away from the
    // block's boundary -- one comparison for nearly every lane; the others measure the distance to the outside of the
    // block (a side on the edge of the grid has nothing beyond it)
    const bool found = lo != 0xFFFFFFFFu;
    const float reach = 0.998f * g.cell;
    bool inside = in_grid && band < reach * reach;
    if (!inside) {
        const float big = 3.0e38f;
        float bound = big;
        if (xl > 0) bound = fminf(bound, fx - (g.x0 + (float)xl * g.cell));
        if (xh < G - 1) bound = fminf(bound, (g.x0 + (float)(xh + 1) * g.cell) - fx);
        if (yl > 0) bound = fminf(bound, fy - (g.y0 + (float)yl * g.cell));
        if (yh < G - 1) bound = fminf(bound, (g.y0 + (float)(yh + 1) * g.cell) - fy);
        bound -= 1e-3f * g.cell;
        inside = bound >= big * 0.5f || (bound > 0.f && bound * bound > band);
    }
    const float2 cw = g.xy_at(found ? bq : 0);
    BlNear r;
    r.s = bl_dist2(px, py, cw);
    r.zx = cw.x; r.zy = cw.y;
    r.col = g.item_at(found ? bq : 0);
    r.done = found && inside && !(second <= band);     // (hi = 0xFFFFFFFF reads as a NaN: no second candidate)
    return r;
}

// The same for ONE track by the whole wave over EVERY detection of the frame, 512 at a time (eight per lane, their reads
// in flight together).  First in float, like bl_search: when exactly one detection lies within the float error band of
// the smallest float distance it is the argmin, and its float64 distance is evaluated once (a lost track far from
// everything: most calls end here).  Otherwise exactly, rowmin_wave's three passes: the smallest squared distance; the
// lowest column within 2^-48 of it; the rounded roots themselves only if some s differs from the smallest at all.
// px, py, the result: wave-uniform.
__device__ __forceinline__ BlNear bl_search_wave(const BlGridView &g, double px, double py, int m, int lane)
{
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    BlNear r;
    r.done = true;
    if (m <= 512) {
        const float fx = (float)px, fy = (float)py;
        float sf[8];
        int lo = 0x7F800000;                              // (a non-negative float orders like its bits, as an int too)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = u * 64 + lane;
            sf[u] = j < m ? bl_dist2f(fx, fy, g.xy_at(min(j, m - 1))) : 3.0e38f;
            lo = min(lo, (int)__float_as_uint(sf[u]));
        }
        const float best = __uint_as_float((uint32_t)wave_min(lo));
        const float band = best + (0.2f + 5e-6f * best);
        int n_near = 0, jn = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long b = __ballot(sf[u] <= band);
            n_near += (int)__popcll(b);
            if (b) jn = u * 64 + __builtin_ctzll(b);
        }
        if (n_near == 1) {
            const float2 c = g.xy_at(jn);
            r.s = bl_dist2(px, py, c); r.zx = c.x; r.zy = c.y; r.col = g.item_at(jn);
            return r;
        }
    }
    double lane_min = inf;
    for (int j0 = 0; j0 < m; j0 += 512) {
        float2 c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = g.xy_at(min(j0 + u * 64 + lane, m - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double s = bl_dist2(px, py, c[u]);
            lane_min = (j0 + u * 64 + lane < m) ? __builtin_fmin(lane_min, s) : lane_min;
        }
    }
    const double s_min = wave_min(lane_min);
    const double near_limit = s_min + s_min * 0x1p-48;
    int cand = 0x7FFFFFFF;
    bool inexact = false;
    for (int j0 = 0; j0 < m; j0 += 512) {
        float2 c[8];
        int it[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = min(j0 + u * 64 + lane, m - 1); c[u] = g.xy_at(j); it[u] = g.item_at(j); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * 64 + lane;
            const double s = bl_dist2(px, py, c[u]);
            const bool near = j < m && s <= near_limit;
            cand = near ? min(cand, (it[u] << 15) | j) : cand;          // (column, then position: both below 32768)
            inexact = inexact || (near && s != s_min);
        }
    }
    if (__any(inexact)) {
        const double d_min = sqrt(s_min);
        cand = 0x7FFFFFFF;
        for (int j = lane; j < m; j += 64) {
            const double s = bl_dist2(px, py, g.xy_at(j));
            if (s <= near_limit && sqrt(s) == d_min) cand = min(cand, (g.item_at(j) << 15) | j);
        }
    }
    const int win = wave_min(cand);
    const float2 c = g.xy_at(win & 0x7FFF);
    r.s = s_min; r.zx = c.x; r.zy = c.y; r.col = win >> 15;
    return r;
}

// ---- the kernel ----------------------------------------------------------------------------------------------------
#ifdef YSMR_STAMPS
__device__ unsigned long long g_bstamps[BL_WAVES][16];
#ifndef YSMR_BL_FRAME
#define YSMR_BL_FRAME 40
#endif
#define BLSTAMP(k) do { if (f == YSMR_BL_FRAME && lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_bstamps[wave][k] = t_; } } while (0)
#else
#define BLSTAMP(k) do {} while (0)
#endif
struct BlShared {      // static part of the LDS
    int cnt[BL_MAX_BATCH];           // detections per frame (clamped)
    int used[2], n_dead[2];          // per frame parity: claims made, tracks deregistered
    int dead_id[2][BL_THREADS];      // ids of the tracks deregistered in the frame
    int wave_cnt[2][BL_WAVES];       // registration: per-wave counts of the two ranked lists
    int set_state[2];
    int top;                         // one past the highest seat ever taken
};

__host__ __device__ inline int bl_md_padded(int max_det) { return (max_det + 3) / 4 * 4; }
__host__ __device__ inline size_t bl_lds_bytes(int max_det)
{
    return 2 * 4 * (size_t)bl_grid_dwords_max(max_det) + 2 * 12 * (size_t)bl_md_padded(max_det) + 4 * BL_TABLE + 64;
}

// The kernel's arguments as ONE structure: the kernel argument segment then IS this structure, and an argument that one
// phase of a frame needs (the row buffer, the detections' boxes, the error word, the state arrays at the end) is read
// from it where it is needed -- one scalar load out of the scalar cache -- instead of sitting in scalar registers for the
// whole launch: with everything passed and kept the usual way this kernel had 113 of them spilled into vector lanes.
struct BlKernArgs {
    TrackerDev t;
    BatchDev bd;
    const float *det_all;
    const int32_t *det_count;
    int batch, frame0;
    ysmr_row *rows;
    long long rows_capacity;
    long long *row_count;
    const BlGains *gains;
};
typedef const __attribute__((address_space(4))) BlKernArgs *BlKernArgsPtr;
__device__ __forceinline__ BlKernArgsPtr bl_kernargs()
{
    BlKernArgsPtr p = (BlKernArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));          // (opaque: a load through it is not merged with the preloaded arguments)
    return p;
}

__global__ __launch_bounds__(BL_THREADS) void k_batch(BlKernArgs ka)
{
    const TrackerDev &t = ka.t;
    const BatchDev &bd = ka.bd;
    const int32_t *__restrict__ det_count = ka.det_count;
    const int batch = ka.batch, frame0 = ka.frame0;
    long long *row_count = ka.row_count;
    const BlGains *gains = ka.gains;
    __shared__ BlShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int md = t.max_det, mdp = bl_md_padded(md);
    const int bufw = bl_grid_dwords_max(md);
    // LDS by offset (bl_lds): per frame parity p the smallest proposing distance per column (u64), the winning id per
    // column (u32) and the frame's detections (k_bgrid's block); the CPython set model's table
    auto key_at = [&](int p) { return bl_lds + p * mdp; };
    auto cid_at = [&](int p) { return bl_u32(4 * mdp + p * mdp); };
    auto buf_off = [&](int p) { return 6 * mdp + p * bufw; };
    const int tab_off = 6 * mdp + 2 * bufw;
    const int seats = min(t.capacity, BL_THREADS);
    const int nf = t.n_f;
    const int gone_max = (int)floor(t.max_gone);        // tracker.py:104, 208: disappeared > maxDisappeared, a float
    // (max_disappeared < 32000 for a one-launch handle, so the conversion is exact)
    const int hn[BL_NF] = {t.n_i[0], nf > 1 ? t.n_i[1] : 1, nf > 2 ? t.n_i[2] : 1};    // horizons (ring offsets of the leavers)

    // ---- start of the batch: counters, this lane's track, the first frame's detections
    int n = *t.n_tracks, next_id = *t.next_id, head = *bd.head & (BL_HB - 1);
    long long base = *row_count;
    for (int f = tid; f < batch; f += BL_THREADS) {
        int m = det_count[f];
        if (m > md) { m = md; atomicOr(t.err, ERR_DET_CLAMPED); }
        sh.cnt[f] = m < 0 ? 0 : m;
    }
    if (tid < 2) { sh.used[tid] = 0; sh.n_dead[tid] = 0; }
    if (tid == 0) sh.top = 0;
    BlSeat S;
    bl_seat_blank(S);
    if (n > 0 && tid < seats && bd.i32[5 * (size_t)bd.seat_cap + tid]) {     // (an empty table: whatever the flags say)
        bl_seat_load(S, bd, tid);
        S.alive = true;
    }
    __syncthreads();
    {   // (the ballot outside the branch: under `lane == 0 &&` only lane 0 would vote)
        const unsigned long long ba = __ballot(S.alive);
        if (lane == 0 && ba) atomicMax(&sh.top, 64 * (wave + 1));
    }
    __syncthreads();
    // The chores of a frame -- the next frame's LDS-DMA, clearing its tables -- go to the waves that hold no track, when
    // there are any: with 9 of 12 waves in use, three of them share one SIMD and set the frame's pace, and the idle waves
    // sit on the other SIMDs.  (Seats are handed out lowest first, so the waves in use are the first ones.)
    int helpers_from = (sh.top + 63) >> 6;               // first wave without a track; BL_WAVES: none
    auto chore_first = [&]() { return helpers_from < BL_WAVES ? tid - 64 * helpers_from : tid; };
    auto chore_stride = [&]() { return helpers_from < BL_WAVES ? 64 * (BL_WAVES - helpers_from) : BL_THREADS; };
    auto dma = [&](int f) {        // frame f's block -> s_buf[f & 1], whole 1-KiB pieces, a wave-instruction each
        const int pieces = bl_grid_dwords(sh.cnt[f]) >> 8;
        const char *src = bd.grid + (size_t)bd.grid_stride * f;
        const int w0 = helpers_from < BL_WAVES ? helpers_from : 0, nw = BL_WAVES - w0;
        for (int c = wave - w0; c >= 0 && c < pieces; c += nw) {
            const uint32_t lds = (uint32_t)(uintptr_t)(bl_u32(buf_off(f & 1) + c * 256));
            const uint32_t off = (uint32_t)c * 1024u + (uint32_t)lane * 16u;
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(lds)), "v"(off), "s"(src) : "memory");
        }
    };
    dma(0);
    for (int c = tid; c < sh.cnt[0]; c += BL_THREADS) { key_at(0)[c] = ~0ull; cid_at(0)[c] = 0xFFFFFFFFu; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int f = 0; f < batch; ++f) {
        const int par = f & 1;
        const int m = sh.cnt[f], m_next = f + 1 < batch ? sh.cnt[f + 1] : 0;
        BLSTAMP(0);
        if (((frame0 + f) & (BL_REFRESH - 1)) == 0 && S.alive && t.use_gsff)      // (uniform but for `alive`: see bl_sums_from_ring)
            bl_sums_from_ring(S, bd, tid, head, t.n_i[0], nf > 1 ? t.n_i[1] : 0, nf > 2 ? t.n_i[2] : 0);
        if (f + 1 < batch) dma(f + 1);
        BLSTAMP(1);
        // ---- each track proposes its nearest detection (tracker.py:151-163)
        const bool propose = S.alive && m > 0;
        const BlGridView gv = bl_grid_view(buf_off(par), m);
        BlNear nr{0.0, 0.f, 0.f, 0, true};
        unsigned long long key = 0;
        if (propose) nr = bl_search(gv, S.px, S.py, m);
        BLSTAMP(11);
        {   // the lanes the 5 x 5 cells did not settle, one after the other, by the whole wave
            unsigned long long todo = __ballot(propose && !nr.done);
#ifdef YSMR_STAMPS
            if (f == YSMR_BL_FRAME && lane == 0) g_bstamps[wave][12] = __popcll(todo);
            // (the shader clock over frames 8 .. 56: s_memrealtime counts at a constant 100 MHz; column 15 of rows 0 .. 3)
            if (wave == 0 && lane == 0 && (f == 8 || f == 56)) {
                unsigned long long rt_;
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory");
                g_bstamps[f == 8 ? 0 : 1][15] = rt_;
                g_bstamps[f == 8 ? 2 : 3][15] = __builtin_amdgcn_s_memtime();
            }
            if (lane == 0) { if (f == 0) { g_bstamps[wave][13] = 0; g_bstamps[wave][14] = 0; } g_bstamps[wave][13] += __popcll(todo); g_bstamps[wave][14] += todo ? 1 : 0; }
#endif
            while (todo) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const BlNear w = bl_search_wave(gv, lane_value(S.px, l), lane_value(S.py, l), m, lane);
                if (lane == l) nr = w;
            }
        }
        // the measurements that leave the filters' windows with this frame (used behind the claims)
        double2 leave[BL_NF];
#pragma unroll
        for (int k = 0; k < BL_NF; ++k) leave[k] = make_double2(0.0, 0.0);
        if (S.alive && t.use_gsff) {
#pragma unroll
            for (int k = 0; k < BL_NF; ++k)
                if (k < nf) leave[k] = bl_ring_load(bd, (head - hn[k]) & (BL_HB - 1), tid);
        }
        if (propose) {      // (round 1 on the SQUARED distance: a non-negative double orders like its bits)
            key = (unsigned long long)__double_as_longlong(nr.s);
            atomicMin(&key_at(par)[nr.col], key);
        }
        BLSTAMP(2);
        block_sync<true>();
        BLSTAMP(3);
        // (the other parity's tables and counters were last read before the end of the previous frame)
        for (int c = chore_first(); c >= 0 && c < m_next; c += chore_stride()) { key_at(par ^ 1)[c] = ~0ull; cid_at(par ^ 1)[c] = 0xFFFFFFFFu; }
        if (tid == BL_THREADS - 1) { sh.used[par ^ 1] = 0; sh.n_dead[par ^ 1] = 0; }
        if (propose) {
            // the proposers at the column's smallest DISTANCE contend by id (tracker.py:158: ascending row minimum, then
            // row).  sqrt is monotone, so that is the smallest s -- and, once in a blue moon, an s a few ulps above it
            // that rounds to the same root: only those take the square roots
            const unsigned long long kmin = key_at(par)[nr.col];
            bool tie = key == kmin;
            if (!tie) {
                const double smin = __longlong_as_double((long long)kmin);
                if (nr.s <= smin + smin * 0x1p-48) tie = sqrt(nr.s) == sqrt(smin);
            }
            if (tie) atomicMin(&cid_at(par)[nr.col], (uint32_t)S.id);
        }
        BLSTAMP(4);
        block_sync<true>();
        BLSTAMP(5);
        // ---- claims (tracker.py:171-189), ageing and deregistration (:95-107, 198-211)
        const bool age = (m == 0) || (n > 0 && n >= m);
        const bool mine = propose && cid_at(par)[nr.col] == (uint32_t)S.id;
        double z0 = S.px, z1 = S.py;
        float box[3] = {S.info[0], S.info[1], S.info[2]};
        bool fresh = false, died = false;
        if (mine) {
            z0 = (double)nr.zx; z1 = (double)nr.zy;
            const float *d = bl_kernargs()->det_all + ((size_t)f * md + nr.col) * 5;
            box[0] = d[2]; box[1] = d[3]; box[2] = d[4];
            S.gone = 0;
        } else if (S.alive && age) {
            ++S.gone;
            box[0] = box[1] = box[2] = 0.f;
            if (S.gone > gone_max) { S.alive = false; died = true; }
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
            int *unused = reinterpret_cast<int *>(key_at(par ^ 1)), *newcols = unused + mdp;
            uint32_t *list = cid_at(par ^ 1);
            __syncthreads();
            if (n == 0) {
                for (int c = tid; c < m; c += BL_THREADS) newcols[c] = c;
                n_new_all = m;
            } else {
                // the unclaimed columns in ascending order: K consecutive columns per thread, ranked by a wave scan
                const int K = (m + BL_THREADS - 1) / BL_THREADS;
                const int c0 = tid * K, c1 = min(c0 + K, m);
                int cnt = 0;
                for (int c = c0; c < c1; ++c) cnt += cid_at(par)[c] == 0xFFFFFFFFu;
                int incl = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                if (lane == 63) sh.wave_cnt[0][wave] = incl;
                __syncthreads();
                int at = incl - cnt, total = 0;
#pragma unroll
                for (int k = 0; k < BL_WAVES; ++k) { const int v = sh.wave_cnt[0][k]; at += k < wave ? v : 0; total += v; }
                for (int c = c0; c < c1; ++c)
                    if (cid_at(par)[c] == 0xFFFFFFFFu) unused[at++] = c;
                __syncthreads();
                int cnt_set = cpython_order_lds<BL_THREADS, BL_TABLE>(unused, total, m, sh.used[par], newcols, bl_u32(tab_off), list, sh.set_state);
                if (cnt_set < 0) { if (tid == 0) atomicOr(bl_kernargs()->t.err, ERR_TRACK_CAPACITY); cnt_set = 0; }
                n_new_all = cnt_set;
            }
            n_new = n_new_all;
            if (n + n_new > seats) {
                if (tid == 0) atomicOr(bl_kernargs()->t.err, ERR_TRACK_CAPACITY);
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
                const float *d = bl_kernargs()->det_all + ((size_t)f * md + c) * 5;
                z0 = (double)d[0]; z1 = (double)d[1];
                box[0] = d[2]; box[1] = d[3]; box[2] = d[4];
                S.id = next_id + fr; S.rank = n + fr; S.gone = 0;
                S.alive = true; fresh = true;
                atomicMax(&sh.top, tid + 1);
            }
            __syncthreads();     // (the lists lived in the next frame's tables)
            for (int c = tid; c < m_next; c += BL_THREADS) { key_at(par ^ 1)[c] = ~0ull; cid_at(par ^ 1)[c] = 0xFFFFFFFFu; }
            helpers_from = (sh.top + 63) >> 6;
        }
        // ---- the filter bank (tracker.py:219-227)
        BLSTAMP(6);
        double o0 = z0, o1 = z1;
        if (S.alive) {
            if (t.use_gsff) {
                // (the twelve gain constants come out of the scalar cache in every frame: as kernel arguments they sat in 24
                // scalar registers for the whole launch, and this kernel spilled 113 of them into vector lanes -- a tenth of a frame's
                // vector instructions were v_readlane / v_writelane)
                const BlGains *gq = gains;
                asm volatile("" : "+s"(gq));
                bl_gsff(S, t, bd, *gq, tid, head, leave, z0, z1, fresh, o0, o1);
            }
            else { S.px = z0; S.py = z1; }
        }
        // ---- end of the frame: the next frame's detections have landed, the frame's counts are complete
        BLSTAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BLSTAMP(8);
        __syncthreads();
        BLSTAMP(9);
        const int n_dead = sh.n_dead[par];
        if (n_dead && S.alive)
            for (int k = 0; k < n_dead; ++k) S.rank -= sh.dead_id[par][k] < S.id;
        if (S.alive) { S.info[0] = box[0]; S.info[1] = box[1]; S.info[2] = box[2]; }   // (the claimed box: requested before the filter bank)
        const int n_live = n - n_dead + n_new;
        const BlKernArgsPtr kr = bl_kernargs();
        ysmr_row *rows = kr->rows;
        const long long rows_capacity = kr->rows_capacity;
        if (S.alive && base + S.rank < rows_capacity) {      // track_eval.py:313-316
            ysmr_row rr;
            rr.frame = frame0 + f;
            rr.track_id = S.id;
            rr.x = o0; rr.y = o1;
            rr.w = S.info[0]; rr.h = S.info[1]; rr.angle = S.info[2];
            rr.disappeared = S.gone;
            rows[base + S.rank] = rr;
        }
        if (tid == 0 && base + n_live > rows_capacity) atomicOr(kr->t.err, ERR_ROWS_CAPACITY);
        base += n_live;
        n = n_live;
        next_id += n_new_all;
        head = (head + 1) & (BL_HB - 1);
        BLSTAMP(10);
    }
    // ---- end of the batch: the small state goes back to HBM, seat by seat
    {
        const BlKernArgsPtr ke = bl_kernargs();
        BatchDev be;                       // (field by field: the structure lies in the constant address space)
        be.ring = ke->bd.ring; be.f64 = ke->bd.f64; be.f32 = ke->bd.f32; be.i32 = ke->bd.i32; be.head = ke->bd.head;
        be.grid = ke->bd.grid; be.grid_stride = ke->bd.grid_stride; be.seat_cap = ke->bd.seat_cap;
        if (tid < seats) {
            be.i32[5 * (size_t)be.seat_cap + tid] = S.alive ? 1 : 0;
            if (S.alive) bl_seat_store(S, be, tid);
        }
        if (tid == 0) { *ke->t.n_tracks = n; *ke->t.next_id = next_id; *ke->row_count = base; *be.head = head; }
    }
}

// ---- the split link's second kernel with a track per LANE (round 5) ------------------------------------------------
// Tables beyond k_batch's 768 seats (4K: ~5 000 tracks) link with two launches per frame: k_link DECIDES (one workgroup), then
// every track applies its claim, runs its filter bank, writes its row and finds its nearest detection of the next frame.  That
// second kernel was k_track, a WAVE per track: 5 000 waves every frame, whose tail waits for wave slots that the detection
// kernels of the next batch hold (12 us alone, 19 beside them: profiles/r04_timeline_4k.txt).  k_track_lanes does the same per
// LANE with the batch link's arithmetic: the filter bank as window sums over an HBM ring (bl_gsff: four operations per filter
// and coordinate, four ring entries per frame), the state seat-major by SLOT (a track keeps its slot for life), the row
// minimum from the 3 x 3 cells of the next frame's grid around the prediction in exact float64, then from the 5 x 5 for the
// lanes that leaves open -- 20 workgroups of 256 lanes instead of 1 250 of four waves.  What neither settles (a prediction
// farther from every detection than the block reaches, or more candidates than slots) goes to the wave, one lane at a time,
// through k_track's own rowmin_grid / rowmin_wave.
// Measured (profiles/r05_link_4k_lanes.log): as fast as k_track, not faster -- 21 us per launch beside detection,
// 14.7 with no lane falling through.  The kernel is a chain of dependent rounds of loads at 1.5-2 us each (data k_link wrote on
// another XCD, HBM busy with the next batch's detection), six of them; arithmetic and gathers are ~2 us.  Hence the order of
// the loads below: whatever hangs on the row or on the slot alone is requested in the first round that can, and before stores.
// A handle that can (three filters of <= 31 frames, affine gains) keeps its filter state in this layout from the start:
// ysmr_tracker_update, ysmr_tracker_run and the peeks need no conversion; rec / hist of the per-slot layout stay unused.
// bd.i32 row 0 holds the id the slot's filter state belongs to (-1: none): a slot whose track id differs is a new track.

// nearest detection of (px, py) among the (2 REACH + 1)^2 cells around it, exactly (rowmin_grid's rules: smallest squared
// distance; the lowest column among the equal ROUNDED distances); false: the block does not settle it (nothing in it, something
// outside it could be nearer, or it holds more than SLOTS candidates).  Two rounds of loads -- the bounds of the block's row-runs,
// then every candidate's centre and column at once into registers, the runs laid end to end over the slots: a loop that fetched
// a candidate per iteration was a chain of round trips (the first build of k_track_lanes: 26-36 us per launch).
template <int REACH, int SLOTS>
__device__ __forceinline__ bool bl_rowmin_lane(const DetGrid &g, const GridHdr &gh, double px, double py, int m, double &d_min, int &arg)
{
    constexpr int ROWS = 2 * REACH + 1;
    const double x0 = (double)gh.x0, y0 = (double)gh.y0, cell = (double)gh.cell, inv = (double)gh.inv;
    int cx = (int)floor((px - x0) * inv), cy = (int)floor((py - y0) * inv);
    cx = cx < 0 ? 0 : (cx > GRID_N - 1 ? GRID_N - 1 : cx);
    cy = cy < 0 ? 0 : (cy > GRID_N - 1 ? GRID_N - 1 : cy);
    const int xl = max(cx - REACH, 0), xh = min(cx + REACH, GRID_N - 1), yl = max(cy - REACH, 0), yh = min(cy + REACH, GRID_N - 1);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    int a[ROWS], e[ROWS + 1];        // first entry of each row's run; the runs laid end to end: run r takes slots e[r] .. e[r + 1] - 1
    e[0] = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int iy = min(yl + r, GRID_N - 1);
        a[r] = g.start[iy * GRID_N + xl];
        const int b = g.start[iy * GRID_N + xh + 1];
        e[r + 1] = e[r] + (yl + r > yh ? 0 : b - a[r]);
    }
    const int n_all = e[ROWS];
    float2 c[SLOTS];
    int col[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        int q = a[0] + k;
#pragma unroll
        for (int r = 1; r < ROWS; ++r) q = k >= e[r] ? a[r] + (k - e[r]) : q;
        q = min(q, m - 1);                              // (a slot behind the runs reads some entry and is masked below)
        c[k] = g.xy[q];
        col[k] = g.items[q];
    }
    double sq[SLOTS], s_min = inf;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const double dx = px - (double)c[k].x;
        const double dy = py - (double)c[k].y;
        double v = dx * dx;
        v = v + dy * dy;
        sq[k] = k < n_all ? v : inf;
        s_min = __builtin_fmin(sq[k], s_min);
    }
    const bool fits = n_all <= SLOTS;
    // distance from the prediction to the outside of the block; a side on the edge of the grid has nothing beyond it
    double bound = inf;
    if (cx - REACH > 0) bound = fmin(bound, px - (x0 + (cx - REACH) * cell));
    if (cx + REACH < GRID_N - 1) bound = fmin(bound, (x0 + (cx + REACH + 1) * cell) - px);
    if (cy - REACH > 0) bound = fmin(bound, py - (y0 + (cy - REACH) * cell));
    if (cy + REACH < GRID_N - 1) bound = fmin(bound, (y0 + (cy + REACH + 1) * cell) - py);
    bound -= 1e-3 * cell;       // (cells were assigned in float arithmetic)
    if (!fits || !(s_min < inf)) return false;
    if (!(bound == inf || (bound > 0.0 && bound * bound > s_min * (1.0 + 1e-9)))) return false;
    const double near_limit = s_min + s_min * 0x1p-48;
    d_min = sqrt(s_min);
    int cand = 0x7FFFFFFF;
    bool inexact = false;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const bool near = sq[k] <= near_limit;
        cand = near ? min(cand, col[k]) : cand;
        inexact = inexact || (near && sq[k] != s_min);
    }
    if (inexact) {              // some s differs from min s by a few ulps: compare the rounded roots
        cand = 0x7FFFFFFF;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (sq[k] <= near_limit && sqrt(sq[k]) == d_min) cand = min(cand, col[k]);
    }
    arg = cand;
    return true;
}

constexpr int TL_THREADS = 256;
template <typename DetT>
__global__ __launch_bounds__(TL_THREADS) void k_track_lanes(TrackerDev t, BatchDev bd, const BlGains *gains, int head, int frame,
                                                            ysmr_row *rows, long long rows_capacity, const DetT *__restrict__ det,
                                                            const DetT *__restrict__ next_det, const int32_t *next_m_dev,
                                                            DetGrid next_grid)
{
    const int cap = t.capacity, nf = t.n_f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = blockIdx.x * TL_THREADS + tid;                       // table row
    const int n_live = *t.n_tracks, claims_by_row = t.n_tracks[4];
    const int m_next = next_det ? det_count(-1, next_m_dev, t.max_det, nullptr) : 0;
    const GridHdr gh = grid_hdr(next_grid);
    // (the row's slot and claim are requested beside the counts that say whether the row is live, not after them: every round of
    // dependent loads is 1-2 us here, the data having been written by k_link on another XCD a moment ago)
    const int i_safe = min(i, cap - 1);
    const int slot = t.order[i_safe];
    const int c_row = t.claim_row[i_safe];
    const bool live = i < n_live;
    double p0 = 0.0, p1 = 0.0;
    bool to_wave = false;
    if (live) {
        const int c = claims_by_row ? c_row : t.claim_slot[slot];
        // (everything the row needs that hangs on the slot alone, requested now: a load issued after the seat's stores would wait for them)
        const int id = t.id[slot], gone = t.gone[slot];
        const long long base = t.row_base[0];
        // the measurement: the detection k_link let this track claim (position and box stored here), or the position it has
        double z0, z1;
        float bw, bh, ba;
        if (c >= 0) {
            const DetT *d = det + (size_t)c * 5;
            z0 = (double)d[0]; z1 = (double)d[1];
            bw = (float)d[2]; bh = (float)d[3]; ba = (float)d[4];
            t.info[slot] = bw; t.info[cap + slot] = bh; t.info[2 * cap + slot] = ba;
        } else {
            z0 = t.pos[slot]; z1 = t.pos[cap + slot];
            bw = t.info[slot]; bh = t.info[cap + slot]; ba = t.info[2 * cap + slot];
        }
        double o0 = z0, o1 = z1;
        p0 = z0; p1 = z1;
        if (t.use_gsff) {
            // (the seat and the ring entries are requested whatever the slot holds -- in the same round of loads as the id that says
            // whether they are this track's; a new track's are blanked afterwards)
            BlSeat S;
            bl_seat_load(S, bd, slot);
            const int hn[BL_NF] = {t.n_i[0], nf > 1 ? t.n_i[1] : 1, nf > 2 ? t.n_i[2] : 1};
            double2 leave[BL_NF];
#pragma unroll
            for (int k = 0; k < BL_NF; ++k) leave[k] = k < nf ? bl_ring_load(bd, (head - hn[k]) & (BL_HB - 1), slot) : make_double2(0.0, 0.0);
            const bool fresh = S.id != id;
            if (fresh) bl_seat_blank(S);
            else if ((frame & (BL_REFRESH - 1)) == 0)          // (on the frame NUMBER, as k_batch: rows must not depend on the batching)
                bl_sums_from_ring(S, bd, slot, head, t.n_i[0], nf > 1 ? t.n_i[1] : 0, nf > 2 ? t.n_i[2] : 0);
            bl_gsff(S, t, bd, *gains, slot, head, leave, z0, z1, fresh, o0, o1);
            p0 = S.px; p1 = S.py;
            S.id = id;
            bl_seat_store(S, bd, slot);
        }
        t.pos[slot] = p0; t.pos[cap + slot] = p1;      // CentroidTracker.objects[id]: the prediction (tracker.py:225), or the raw centroid
        {
            t.row_gone[i] = gone;
            if (rows && base + i < rows_capacity) {
                ysmr_row r;
                r.frame = frame;
                r.track_id = id;
                r.x = o0; r.y = o1;
                r.w = bw; r.h = bh; r.angle = ba;
                r.disappeared = gone;
                rows[base + i] = r;
            }
        }
        to_wave = m_next > 0;
    }
    // nearest detection of the NEXT frame (tracker.py:151-163): the 3 x 3 cells around the prediction; for the lanes they do not
    // settle (lost tracks, mostly: nothing near the prediction) the 5 x 5; the whole wave for what is left
    if (next_grid.start) {
        double d_min = 0.0;
        int arg = 0;
        if (to_wave && bl_rowmin_lane<1, 16>(next_grid, gh, p0, p1, m_next, d_min, arg)) { t.row_min[i] = d_min; t.row_arg[i] = arg; to_wave = false; }
#ifndef TL_ONE_PASS
        if (__ballot(to_wave) != 0ull) {
            if (to_wave && bl_rowmin_lane<2, 32>(next_grid, gh, p0, p1, m_next, d_min, arg)) { t.row_min[i] = d_min; t.row_arg[i] = arg; to_wave = false; }
        }
#endif
    }
    // the lanes whose 3 x 3 cells did not settle it, one after the other, by the whole wave (k_track's own search)
    unsigned long long todo = __ballot(to_wave);
#ifdef TL_NOFALLBACK
    todo = 0ull;       // (tuning build: what the lanes alone cost; results are wrong)
#endif
    while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int row_l = blockIdx.x * TL_THREADS + (tid & ~63) + l;
        const double qx = lane_value(p0, l), qy = lane_value(p1, l);
        if (next_grid.start && rowmin_grid(t, row_l, qx, qy, next_det, next_grid, lane, gh)) continue;
        DetChunk<DetT> first;
        load_chunk(first, next_det, m_next, 0, lane);
        rowmin_wave(t, row_l, qx, qy, next_det, m_next, lane, first);
    }
}

// ---- conversions between the seat-major rest format and the per-slot layout of k_frame / k_link + k_track ----------
// (a: the CURRENT parity view of the per-slot state; table row r takes seat r)
__global__ void k_to_batch(TrackerDev a, BatchDev bd)
{
    const int n = *a.n_tracks, cap = a.capacity, L = a.hist_cap, nf = a.n_f;
    const int head = *bd.head & (BL_HB - 1);
    const size_t sc = (size_t)bd.seat_cap;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < bd.seat_cap; r += gridDim.x * blockDim.x) {
        bd.i32[5 * sc + r] = r < n ? 1 : 0;
        if (r >= n) continue;
        const int slot = a.order[r];
        const double *hist = a.hist + (size_t)slot * 2 * L;
        for (int e = 0; e < L && e < BL_HB - 1; ++e)           // entry e: the measurement of e frames ago
            bd.ring[(size_t)((head - 1 - e) & (BL_HB - 1)) * sc + r] = make_double2(hist[2 * e], hist[2 * e + 1]);
        double *p = bd.f64 + r;
        const double *rec = a.rec + (size_t)slot * a.rec_stride;
        const int mode_r = (int)(__double_as_longlong(rec[1]) & 0xFFFFFFFFll);
        for (int f = 0; f < BL_NF; ++f) {      // (a filter that is not switched on: weight 0, as k_batch keeps it)
            const bool on = f < nf && f < mode_r;
            p[sc * f] = on ? rec[2 + f] : 0.0;
            p[sc * (BL_NF + f)] = on ? rec[2 + nf + f] : 0.0;
            p[sc * (2 * BL_NF + f)] = on ? rec[2 + 2 * nf + f] : 0.0;
        }
        p[sc * (3 * BL_NF)] = a.pos[slot];
        p[sc * (3 * BL_NF + 1)] = a.pos[cap + slot];
        {   // the window sums, exactly as bl_sums_from_ring forms them
            const int len = (int)(__double_as_longlong(rec[0]) & 0xFFFFFFFFll);
            double s0x = 0.0, s1x = 0.0, s0y = 0.0, s1y = 0.0;
            for (int e = 0; e < BL_HB - 1; ++e) {
                if (e < len && e < L) {
                    s0x = s0x + hist[2 * e];
                    s0y = s0y + hist[2 * e + 1];
                    s1x = __builtin_fma((double)e, hist[2 * e], s1x);
                    s1y = __builtin_fma((double)e, hist[2 * e + 1], s1y);
                }
                for (int f = 0; f < BL_NF; ++f)
                    if (f < nf && e + 1 == min(a.n_i[f], len)) {
                        double *q = p + sc * (3 * BL_NF + 2 + 4 * f);
                        q[0] = s0x; q[sc] = s1x; q[2 * sc] = s0y; q[3 * sc] = s1y;
                    }
            }
        }
        for (int k = 0; k < 3; ++k) bd.f32[sc * k + r] = a.info[k * cap + slot];
        bd.i32[r] = a.id[slot];
        bd.i32[sc + r] = a.gone[a.gone_by_row ? r : slot];
        bd.i32[2 * sc + r] = (int)(__double_as_longlong(rec[0]) & 0xFFFFFFFFll);
        bd.i32[3 * sc + r] = (int)(__double_as_longlong(rec[1]) & 0xFFFFFFFFll);
        bd.i32[4 * sc + r] = r;
    }
}

// (d: the parity-0 view; the track in seat s goes to table row rank[s], which takes slot rank[s]; the free stack hands
// out slot n next)
__global__ void k_to_std(TrackerDev d, BatchDev bd)
{
    const int n = *d.n_tracks, cap = d.capacity, L = d.hist_cap, nf = d.n_f;
    const int head = *bd.head & (BL_HB - 1);
    const size_t sc = (size_t)bd.seat_cap;
    const int span = cap > bd.seat_cap ? cap : bd.seat_cap;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < span; s += gridDim.x * blockDim.x) {
        if (s < cap - n) d.free_slots[s] = cap - 1 - s;
        if (s == 0) *d.n_free = cap - n;
        if (s >= bd.seat_cap || n == 0 || !bd.i32[5 * sc + s]) continue;
        const int r = bd.i32[4 * sc + s];
        const double *p = bd.f64 + s;
        double *hist = d.hist + (size_t)r * 2 * L;
        for (int e = 0; e < L && e < BL_HB - 1; ++e) {
            const double2 v = bd.ring[(size_t)((head - 1 - e) & (BL_HB - 1)) * sc + s];
            hist[2 * e] = v.x; hist[2 * e + 1] = v.y;
        }
        double *rec = d.rec + (size_t)r * d.rec_stride;
        rec[0] = __longlong_as_double((long long)(unsigned int)bd.i32[2 * sc + s]);
        rec[1] = __longlong_as_double((long long)(unsigned int)bd.i32[3 * sc + s]);
        for (int f = 0; f < nf && f < BL_NF; ++f) {
            rec[2 + f] = p[sc * f];
            rec[2 + nf + f] = p[sc * (BL_NF + f)];
            rec[2 + 2 * nf + f] = p[sc * (2 * BL_NF + f)];
        }
        d.pos[r] = p[sc * (3 * BL_NF)];
        d.pos[cap + r] = p[sc * (3 * BL_NF + 1)];
        for (int k = 0; k < 3; ++k) d.info[k * cap + r] = bd.f32[sc * k + s];
        d.id[r] = bd.i32[s];
        d.order[r] = r;
        d.gone[r] = bd.i32[sc + s];       // (row r = slot r: right for either indexing)
        d.row_gone[r] = bd.i32[sc + s];
    }
}

__global__ void k_peek_batch(TrackerDev t, BatchDev bd, int32_t *ids, double *xy, int32_t *gone, int32_t *n_out)
{
    const size_t sc = (size_t)bd.seat_cap;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *t.n_tracks;
    if (s == 0 && n_out) *n_out = n;
    if (s >= bd.seat_cap || n == 0 || !bd.i32[5 * sc + s]) return;
    const int i = bd.i32[4 * sc + s];
    if (ids) ids[i] = bd.i32[s];
    if (xy) { xy[2 * i] = bd.f64[sc * (3 * BL_NF) + s]; xy[2 * i + 1] = bd.f64[sc * (3 * BL_NF + 1) + s]; }
    if (gone) gone[i] = bd.i32[sc + s];
}
