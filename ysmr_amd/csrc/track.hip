// track.hip -- a7-a19 of the YSMR hot path on gfx950: CentroidTracker.update (ysmr/tracker.py:93-230)
// with the Gaussian-sum FIR filter (ysmr/gsff.py:204-347) and row emission
// (ysmr/track_eval.py:313-316), device-resident across frames.
//
// Per frame ONE launch, no host round trip (k_frame, the fused path): every workgroup recomputes the
// frame's bookkeeping in LDS (claim resolution, ageing / deregistration, stable compaction of the
// id-ordered table, registration in CPython set order), then each WAVE handles one track: GSFF
// correct + predict, the frame's output row, and the nearest detection of the NEXT frame -- the fused
// row-min / arg-min of the N x M float64 distance matrix, which is never stored (tracker.py:151-163
// only ever reads D.min(1), D.argmin(1)).  k_rowmin does that last step stand-alone for the first
// frame of a batch.  Configurations beyond the LDS budget of k_frame (max_det > 2456, very large
// capacity) run the same device functions as two launches: k_link (one workgroup) + k_track (one
// wave per track).
//
// Claim rule (tracker.py:158-189): proposals are visited in ascending (row minimum, row); a
// proposal is accepted iff its column is still free.  Each row occurs once, so the winner of a
// column is simply the proposer with the smallest (distance, row) -- resolved with two atomicMin
// rounds instead of a sort and a sequential loop.
//
// All arithmetic is float64 like the reference; -ffp-contract=off keeps mul/add unfused.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#define YSMR_MAX_FILTERS 8

struct TrackerDev {
    // what the link kernels' first round of loads is addressed with, together at the head of the structure: one scalar
    // load of the kernel's arguments fetches them all
    int *n_tracks, *next_id, *err, *n_free;
    int *order, *gone;
    int *row_arg;                 // [cap]
    double *row_min;              // [cap]
    long long *row_base;          // first output row of the current frame (k_link -> k_gsff)
    int capacity, max_det, n_f, use_gsff, hist_cap, table_cap, gain_total, gone_by_row;
    int gains_decoupled;             // x-hat reads only x columns, y-hat only y columns (every closed-form gain)
    double max_gone, lik_min;
    int n_i[YSMR_MAX_FILTERS];
    int gain_off[YSMR_MAX_FILTERS];  // filter i: row0 at gains[gain_off[i]], row1 at +2*n_i[i]
    const double *gains;
    // persistent state
    int *free_slots;
    int *id;
    double *pos;      // [2][cap]
    float *info;      // [3][cap]
    double *hist;     // [cap][hist_cap][2]  (one contiguous ring per track slot)
    // the filter bank's scalars of a slot as ONE record of rec_stride doubles: [0] history length | head << 32,
    // [1] mode, [2 + f] weight, [2 + n_f + f] / [2 + 2 n_f + f] x-hat of filter f (the likelihoods are recomputed in
    // every step and not kept).  A wave fetches and stores it with one instruction, lane = field: 15 VGPRs fewer than
    // separately addressed scalars in k_frame and 10 in k_track (more of its 5000 waves resident at 4K: +2.5 %)
    double *rec;      // [cap][rec_stride]
    int rec_stride;
    // per-frame scratch
    int *unused;                  // [max_det] unclaimed detection columns, ascending
    int *row_gone;                // [cap] split path: `gone` of the track in a row, written with the row minimum (k_link then
                                  // has everything a row's claim and ageing need from ONE round of loads by row)
    int *dead;                    // [cap]
    int *new_cols;                // [max_det]
    int *set_table;               // [2][table_cap] CPython set model
    // split path (tables too large for LDS): k_link's per-column winners and per-row claims
    unsigned long long *link_key; // [max_det]
    int *link_row;                // [max_det]
    int *link_claim;              // [capacity]
    // split path: the detection column the track in a slot claimed in the current frame (-1: none).  k_link only decides;
    // the winner's detection is fetched, and position / box are stored, by the track's own wave in k_track (5000 waves
    // instead of one workgroup's two rounds of dependent gathers: 5 us of k_link's 12.6 at 4K)
    int *claim_slot;              // [capacity]
    // ... and by ROW of the table k_track sees (after compaction and registration), valid when n_tracks[4] != 0 (k_link's
    // register-resident path, the 4K configuration): the wave then knows its detection one round of loads earlier -- with
    // its slot instead of after it
    int *claim_row;               // [capacity]
};

// ---- the batch link (batch_link.h): one workgroup links a whole batch of frames per launch, a track per lane
#ifndef YSMR_BL_THREADS
#define YSMR_BL_THREADS 768
#endif
constexpr int BL_THREADS = YSMR_BL_THREADS;      // seats: one track per lane
constexpr int BL_WAVES = BL_THREADS / 64;
constexpr int BL_HB = 32;             // ring of measurements per seat (hist_cap <= 32; a power of two)
constexpr int BL_NF = 3;              // filters (n_f <= 3)
#ifndef YSMR_BL_MAX_BATCH
#define YSMR_BL_MAX_BATCH 256
#endif
constexpr int BL_MAX_BATCH = YSMR_BL_MAX_BATCH;      // frames per launch (the grid block is sized for it; longer batches are cut).
                                                    // 256 since the end of round 4: a launch's fixed costs (state in and out, the gap to the next
                                                    // launch) and the detection kernels' per-item costs are paid per BATCH (bench.py: 161.5 k frames/s
                                                    // at 64 frames per batch, 163.5 k at 128, 165 k at 256)
constexpr int BL_TABLE = 4096;        // CPython set model table, 32-bit slots (as FRAME_TABLE)
constexpr int BL_SF64 = 3 * BL_NF + 2 + 4 * BL_NF;   // w, xa, xb, px, py, the window sums
constexpr int BL_REFRESH = 64;        // frames between two exact recomputations of the window sums (a power of two)

struct BatchDev {
    double2 *ring;      // [BL_HB][seat_cap]  the measurements of the last BL_HB frames, seat by seat
    double *f64;        // [BL_SF64][seat_cap]
    float *f32;         // [3][seat_cap]      box of the last claimed detection (0 while lost)
    int *i32;           // [6][seat_cap]      id, gone, history length, mode, table row, seat taken
    int *head;          // ring position of the next frame's measurements
    char *grid;         // [BL_MAX_BATCH][grid_stride] bytes, written by k_bgrid
    unsigned grid_stride;   // bytes per frame (a multiple of 1024)
    int seat_cap;
};

struct BlGains { double alpha[BL_NF][2], beta[BL_NF][2]; };    // [filter][x / y row]; batch_link.h: bl_fir
// Uniform grid over one frame's detections (split path): the nearest detection of a track is looked for in the
// cells around it instead of among all of them (25 M distances per frame at 5000 x 5000).
constexpr int GRID_N = 128;                         // cells per side
constexpr int GRID_CELLS = GRID_N * GRID_N;
struct DetGrid {
    const int *start;     // [GRID_CELLS + 1] first item of each cell (cells row-major: cy * GRID_N + cx)
    const int *items;     // [max_det] detection columns, cell by cell (any order inside a cell)
    const float *hdr;     // x0, y0, cell size, 1 / cell size
    const float2 *xy;     // [max_det] the centres of `items`, in the same order: a cell's candidates arrive with one round
                          // of loads instead of column numbers first and the detections they name after them
};

struct ysmr_tracker {
    TrackerDev d;      // view of parity 0
    TrackerDev d1;     // view of parity 1 (fused path: small per-frame arrays are double-buffered)
    int par;           // parity of the current state
    bool fused;        // one k_frame launch per frame instead of k_link + k_track
    size_t frame_lds;
    void *block;       // the single device allocation
    size_t block_bytes;
    std::vector<double> gains_host;
    bool set_base;                 // fused path: the next k_rowmin also sets row_base from base_ptr
    const long long *base_ptr;
    const void *rowmin_for = nullptr;   // fused path: the frame (its detections' address) whose row minima the last launch of
                                        // the previous ysmr_tracker_run_chained call has already left in the state
    void *grid_block = nullptr;    // split path: DetGrid arrays for grid_frames frames; batch link: k_bgrid's blocks
    int grid_frames = 0;           //   (allocated by ysmr_tracker_create: no allocation on the call path)
    // batch link (batch_link.h): the handle can link a whole batch per launch; where the state rests right now
    BatchDev bd;
    bool batchable = false;        // configuration served by k_batch (three filters of <= 31 frames, <= 1024 tracks, ...)
    bool in_batch = false;         // the state rests in the seat-major arrays of `bd` (else: the per-slot layout)
    int link_mode = 0;             // ysmr_tracker_link_mode: 0 = the library's choice, 1 = one (or two) launches per frame
    size_t batch_lds = 0;
    BlGains bgains;
    const BlGains *bgains_dev = nullptr;     // the same in device memory (behind the grid blocks): k_batch reads it frame by frame
    // ysmr_tracker_prepare: which batch each of the two caller-named grid blocks was binned for (block 2 is run's own)
    struct Prepared { const void *det = nullptr; const void *count = nullptr; int batch = 0; } prepared[2];
    // split link with a track per lane (batch_link.h: k_track_lanes): the filter state of a non-fused handle lives seat-major by
    // slot in `bd` from the start; the ring's head is kept here (every live track appends one measurement per frame)
    bool lanes = false;
    int lanes_head = 0;
    bool use_batch() const { return batchable && link_mode == 0; }
    DetGrid grid(int f) const
    {
        char *b = (char *)grid_block;
        const size_t per = grid_bytes_per_frame(d.max_det);
        const size_t o_items = sizeof(int) * (GRID_CELLS + 64), o_hdr = o_items + sizeof(int) * (size_t)d.max_det;
        return DetGrid{(const int *)(b + per * f), (const int *)(b + per * f + o_items), (const float *)(b + per * f + o_hdr),
                       (const float2 *)(b + per * f + o_hdr + 64)};
    }
    static size_t grid_bytes_per_frame(int max_det)
    {
        return ysmr::align_up(sizeof(int) * (GRID_CELLS + 64) + sizeof(int) * (size_t)max_det + 64 + sizeof(float2) * (size_t)max_det, 256);
    }
    const TrackerDev &cur() const { return par ? d1 : d; }
    const TrackerDev &nxt() const { return par ? d : d1; }
};

#ifdef YSMR_STAMPS
__device__ unsigned long long g_stamps[32];
#define GSTAMP(k) do { if (blockIdx.x == 1 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_stamps[k] = t_; } } while (0)
__device__ unsigned long long g_block_stamps[2 * 2048 * 8];
#ifndef YSMR_BS_A
#define YSMR_BS_A 62   // frames (mod 64) whose per-block stamps are kept: slot 0 / slot 1
#define YSMR_BS_B 63
#endif
#define BSTAMP(k) do { const int fs_ = frame; if (threadIdx.x == 0 && (fs_ == YSMR_BS_A || fs_ == YSMR_BS_B)) { const int sl_ = fs_ == YSMR_BS_B; unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_block_stamps[(sl_ * 2048 + blockIdx.x) * 8 + (k)] = t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_block_stamps[(sl_ * 2048 + blockIdx.x) * 8 + 4 + (k)] = t_; } } while (0)
extern "C" int ysmr_debug_read_block_stamps(unsigned long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_stamps), sizeof(unsigned long long) * n); }
__device__ unsigned long long g_ring[4096 * 2];
__device__ unsigned int g_ring_n;
#define RING(tag) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); unsigned int i_ = atomicAdd(&g_ring_n, 1u) & 4095u; g_ring[2 * i_] = (unsigned long long)(tag); g_ring[2 * i_ + 1] = t_; } while (0)
extern "C" int ysmr_debug_read_ring(unsigned long long *out, unsigned int *n) { hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ring), sizeof(unsigned long long) * 8192); if (e == hipSuccess) e = hipMemcpyFromSymbol(n, HIP_SYMBOL(g_ring_n), 4); return (int)e; }
extern "C" int ysmr_debug_read_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32); }
#define LRING(k) do { if (threadIdx.x == 0) RING(((10ull + (k)) << 40) | (unsigned)frame); } while (0)   // k_link phases
// when every wave of a frame's k_track finished (plain stores, one slot per wave; the host takes the maximum)
__device__ unsigned long long g_wave_end[64 * 8192];
#define TRACK_END() do { if ((threadIdx.x & 63) == 0 && i < 8192) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_wave_end[(frame & 63) * 8192 + i] = t_; } } while (0)
#define TRACK_END_TO_RING() do {} while (0)
extern "C" int ysmr_debug_read_wave_end(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_end), sizeof(unsigned long long) * 64 * 8192); }
#else
#define GSTAMP(k) do {} while (0)
#define BSTAMP(k) do {} while (0)
#define RING(tag) do {} while (0)
#define LRING(k) do {} while (0)
#define TRACK_END() do {} while (0)
#define TRACK_END_TO_RING() do {} while (0)
#endif

namespace {

constexpr int ERR_TRACK_CAPACITY = 1;  // more live tracks than `capacity`
constexpr int ERR_ROWS_CAPACITY = 2;   // row buffer full
constexpr int ERR_DET_CLAMPED = 4;     // detection count exceeded max_det

__device__ __forceinline__ int det_count(int m_host, const int32_t *m_dev, int max_det, int *err)
{
    int m = m_host >= 0 ? m_host : *m_dev;
    if (m > max_det) { m = max_det; if (err) atomicOr(err, ERR_DET_CLAMPED); }
    return m < 0 ? 0 : m;
}

// ------------------------------------------------------------------------------------------
// Nearest detection of one track, computed by one wave (64 lanes stride over the detections).
// scipy cdist('euclidean') on 2-D points is sqrt(dx*dx + dy*dy) in float64; argmin
// takes the lowest column among equal distances.  sqrt is monotone, so the row minimum is
// sqrt(min s) -- but two different s can round to the same sqrt, and then the lower column wins.
// Pass 1 finds min s; pass 2 takes the lowest column whose s is within 2^-48 (relative) of it.
// Only if one of those differs from min s at all (practically never) a third pass compares the
// float64 square roots themselves; that pass sits behind a wave-uniform branch.
// ------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_min(double v)
{
    long long b = __double_as_longlong(v);
    // lanes outside ROW_MASK / without a source read +inf (old = +inf bits, bound_ctrl off keeps old)
    int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0x7FF00000, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    const double o = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    return __builtin_fmin(o, v);      // (one v_min_f64; squared distances are never NaN)
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min(int v)
{
    const int o = __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, CTRL, ROW_MASK, 0xF, false);
    return min(o, v);
}
template <typename T>
__device__ __forceinline__ T wave_min_lane63(T v)
{
    v = dpp_min<0x111, 0xF>(v);   // row_shr:1
    v = dpp_min<0x112, 0xF>(v);   // row_shr:2
    v = dpp_min<0x114, 0xF>(v);   // row_shr:4
    v = dpp_min<0x118, 0xF>(v);   // row_shr:8  -> lane 15 of each row holds the row minimum
    v = dpp_min<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_min<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the minimum
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
    long long b = __double_as_longlong(wave_min_lane63(v));
    int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), 63);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int wave_min(int v) { return __builtin_amdgcn_readlane(wave_min_lane63(v), 63); }

constexpr int ROWMIN_U = 8;   // detections per lane per chunk (all loads of a chunk in flight together)

template <typename DetT>
struct DetChunk { DetT qx[ROWMIN_U], qy[ROWMIN_U]; };   // raw: converting here would wait for the loads here

template <typename DetT>
__device__ __forceinline__ void load_chunk(DetChunk<DetT> &c, const DetT *__restrict__ det, int m, int j0, int lane)
{
#pragma unroll
    for (int u = 0; u < ROWMIN_U; ++u) {
        int j = min(j0 + u * 64 + lane, m - 1);
        c.qx[u] = det[(size_t)j * 5 + 0];
        c.qy[u] = det[(size_t)j * 5 + 1];
    }
}

// `first` holds detections [0, 64*ROWMIN_U) already loaded by the caller (issued before the GSFF so
// that their latency hides behind it).  With m <= 64*ROWMIN_U every pass runs out of registers.
template <typename DetT>
__device__ __forceinline__ void rowmin_wave(const TrackerDev &t, int row, double px, double py,
                                            const DetT *__restrict__ det, int m, int lane, DetChunk<DetT> &first)
{
    constexpr int CHUNK = 64 * ROWMIN_U;
    const bool multi = m > CHUNK;
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double lane_min = inf;
    // entries past m (stale rows of the frame slice / the clamped tail of the last chunk) are moved out of the way ONCE,
    // in their 32-bit x: 1e30 squared is finite, larger than any real squared distance and never "near" the minimum --
    // no select on the 64-bit distance in either pass
    auto mask_tail = [&](int j0) {
#pragma unroll
        for (int u = 0; u < ROWMIN_U; ++u)
            first.qx[u] = (j0 + u * 64 + lane < m) ? first.qx[u] : (DetT)1e30f;
    };
    for (int j0 = 0; j0 < m; j0 += CHUNK) {
        if (j0 > 0) load_chunk(first, det, m, j0, lane);
        mask_tail(j0);
#pragma unroll
        for (int u = 0; u < ROWMIN_U; ++u) {
            const double dx = px - (double)first.qx[u];
            const double dy = py - (double)first.qy[u];
            double s = dx * dx;
            s = s + dy * dy;
            lane_min = __builtin_fmin(s, lane_min);
        }
    }
    const double s_min = wave_min(lane_min);
    const double near_limit = s_min + s_min * 0x1p-48;
    int cand = 0x7FFFFFFF;
    bool inexact = false;
    for (int j0 = 0; j0 < m; j0 += CHUNK) {
        if (multi) { load_chunk(first, det, m, j0, lane); mask_tail(j0); }
#pragma unroll
        for (int u = 0; u < ROWMIN_U; ++u) {
            const int j = j0 + u * 64 + lane;
            const double dx = px - (double)first.qx[u];
            const double dy = py - (double)first.qy[u];
            double s = dx * dx;
            s = s + dy * dy;
            const bool near = s <= near_limit;
            cand = near ? min(cand, j) : cand;
            inexact = inexact || (near && s != s_min);
        }
    }
    const double d_min = sqrt(s_min);
    if (__any(inexact)) {   // some s differs from min s by a few ulps: compare the rounded roots
        cand = 0x7FFFFFFF;
        for (int j = lane; j < m; j += 64) {
            const double dx = px - (double)det[(size_t)j * 5 + 0];
            const double dy = py - (double)det[(size_t)j * 5 + 1];
            double s = dx * dx;
            s = s + dy * dy;
            if (s <= near_limit && sqrt(s) == d_min) cand = min(cand, j);
        }
    }
    const int best = wave_min(cand);
    if (lane == 0) {
        t.row_min[row] = d_min;
        t.row_arg[row] = best;
    }
}

// One block per frame: bounding box of the detection centres, 128 x 128 cells over it (one cell of margin), a
// counting sort of the detection columns by cell.  Runs once per batch, off the frame-to-frame chain.
template <typename DetT>
__global__ __launch_bounds__(1024) void k_grid_build(const DetT *__restrict__ det_all, const int32_t *__restrict__ det_count,
                                                     int max_det, char *grid_block, size_t per_frame)
{
    __shared__ int s_cnt[GRID_CELLS];          // counts, then cursors
    __shared__ float s_red[4][16];
    __shared__ int s_wave_sum[16];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const DetT *det = det_all + (size_t)f * max_det * 5;
    int m = det_count[f];
    m = m < 0 ? 0 : (m > max_det ? max_det : m);
    int *start = (int *)(grid_block + per_frame * f);
    int *items = start + GRID_CELLS + 64;
    float *hdr = (float *)(items + max_det);
    float2 *xy = (float2 *)(hdr + 16);
    for (int c = tid; c < GRID_CELLS; c += 1024) s_cnt[c] = 0;
    float lo_x = 3.0e38f, lo_y = 3.0e38f, hi_x = -3.0e38f, hi_y = -3.0e38f;
    for (int j = tid; j < m; j += 1024) {
        const float x = (float)det[(size_t)j * 5], y = (float)det[(size_t)j * 5 + 1];
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
    for (int k = 1; k < 16; ++k) {
        lo_x = fminf(lo_x, s_red[0][k]); hi_x = fmaxf(hi_x, s_red[1][k]);
        lo_y = fminf(lo_y, s_red[2][k]); hi_y = fmaxf(hi_y, s_red[3][k]);
    }
    float extent = fmaxf(fmaxf(hi_x - lo_x, hi_y - lo_y), 1.0f);
    const float cell = extent / (float)(GRID_N - 2), inv = 1.0f / cell;
    const float x0 = lo_x - cell, y0 = lo_y - cell;
    if (tid == 0) { hdr[0] = x0; hdr[1] = y0; hdr[2] = cell; hdr[3] = inv; }
    auto cell_of = [&](int j) {
        int cx = (int)floorf(((float)det[(size_t)j * 5] - x0) * inv), cy = (int)floorf(((float)det[(size_t)j * 5 + 1] - y0) * inv);
        cx = cx < 0 ? 0 : (cx > GRID_N - 1 ? GRID_N - 1 : cx);
        cy = cy < 0 ? 0 : (cy > GRID_N - 1 ? GRID_N - 1 : cy);
        return cy * GRID_N + cx;
    };
    for (int j = tid; j < m; j += 1024) atomicAdd(&s_cnt[cell_of(j)], 1);
    __syncthreads();
    // exclusive scan of the 16384 counts: 16 consecutive cells per thread, wave scan of the thread sums, wave sums
    int local[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { local[k] = sum; sum += s_cnt[tid * 16 + k]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) s_wave_sum[w] = incl;
    __syncthreads();
    int before = incl - sum;
    for (int k = 0; k < w; ++k) before += s_wave_sum[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int at = before + local[k];
        start[tid * 16 + k] = at;
        s_cnt[tid * 16 + k] = at;
    }
    if (tid == 1023) start[GRID_CELLS] = before + sum;
    __syncthreads();
    for (int j = tid; j < m; j += 1024) {
        const int at = atomicAdd(&s_cnt[cell_of(j)], 1);
        items[at] = j;
        xy[at] = make_float2((float)det[(size_t)j * 5], (float)det[(size_t)j * 5 + 1]);
    }
}

// Row minimum through the grid: rings of cells around the track until no unseen detection can be as near as the
// best one found (the block of cells' boundary is farther, with a margin far above rounding), then the same tie
// rule as rowmin_wave on the detections of that block -- lowest column among the equal rounded distances.
// Returns false when four rings were not enough (a track far from every detection): the caller falls back to
// all pairs.
struct GridHdr { float x0, y0, cell, inv; };
__device__ __forceinline__ GridHdr grid_hdr(const DetGrid &g)   // (fetched by the caller ahead of everything the search waits for)
{
    return g.start ? GridHdr{g.hdr[0], g.hdr[1], g.hdr[2], g.hdr[3]} : GridHdr{0.f, 0.f, 1.f, 1.f};
}
template <typename DetT>
__device__ __forceinline__ bool rowmin_grid(const TrackerDev &t, int row, double px, double py, const DetT *__restrict__ det,
                                            const DetGrid &g, int lane, const GridHdr &gh)
{
    const double x0 = (double)gh.x0, y0 = (double)gh.y0, cell = (double)gh.cell, inv = (double)gh.inv;
    int cx = (int)floor((px - x0) * inv), cy = (int)floor((py - y0) * inv);
    cx = cx < 0 ? 0 : (cx > GRID_N - 1 ? GRID_N - 1 : cx);
    cy = cy < 0 ? 0 : (cy > GRID_N - 1 ? GRID_N - 1 : cy);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    for (int k = 0; k <= 3; ++k) {
        const int side = 2 * k + 1;
        const int ix = cx - k + lane % side, iy = cy - k + lane / side;
        const bool have = lane < side * side && ix >= 0 && ix < GRID_N && iy >= 0 && iy < GRID_N;
        int a = 0, b = 0;
        if (have) { a = g.start[iy * GRID_N + ix]; b = g.start[iy * GRID_N + ix + 1]; }
        double lane_min = inf;
        for (int q = a; q < b; ++q) {
            const float2 c = g.xy[q];
            const double dx = px - (double)c.x;
            const double dy = py - (double)c.y;
            double s = dx * dx;
            s = s + dy * dy;
            lane_min = __builtin_fmin(s, lane_min);
        }
        const double s_min = wave_min(lane_min);
        // distance from the track to the outside of the block [cx-k, cx+k] x [cy-k, cy+k]; a side of the block on
        // the edge of the grid has nothing beyond it (every detection lies inside the grid)
        double bound = inf;
        if (cx - k > 0) bound = fmin(bound, px - (x0 + (cx - k) * cell));
        if (cx + k < GRID_N - 1) bound = fmin(bound, (x0 + (cx + k + 1) * cell) - px);
        if (cy - k > 0) bound = fmin(bound, py - (y0 + (cy - k) * cell));
        if (cy + k < GRID_N - 1) bound = fmin(bound, (y0 + (cy + k + 1) * cell) - py);
        bound -= 1e-3 * cell;       // (cells were assigned in float arithmetic)
        const bool done = bound == inf || (s_min < inf && bound > 0.0 && bound * bound > s_min * (1.0 + 1e-9));
        if (!done) continue;
        if (!(s_min < inf)) return false;   // no detection at all in a block that covers the grid: m == 0 is the caller's
        const double near_limit = s_min + s_min * 0x1p-48;
        const double d_min = sqrt(s_min);
        int cand = 0x7FFFFFFF;
        bool inexact = false;
        for (int q = a; q < b; ++q) {
            const int j = g.items[q];
            const float2 c = g.xy[q];
            const double dx = px - (double)c.x;
            const double dy = py - (double)c.y;
            double s = dx * dx;
            s = s + dy * dy;
            const bool near = s <= near_limit;
            cand = near ? min(cand, j) : cand;
            inexact = inexact || (near && s != s_min);
        }
        if (__any(inexact)) {   // some s differs from min s by a few ulps: compare the rounded roots
            cand = 0x7FFFFFFF;
            for (int q = a; q < b; ++q) {
                const int j = g.items[q];
                const float2 c = g.xy[q];
                const double dx = px - (double)c.x;
                const double dy = py - (double)c.y;
                double s = dx * dx;
                s = s + dy * dy;
                if (s <= near_limit && sqrt(s) == d_min) cand = min(cand, j);
            }
        }
        const int best = wave_min(cand);
        if (lane == 0) {
            t.row_min[row] = d_min;
            t.row_arg[row] = best;
        }
        return true;
    }
    return false;
}

// Stand-alone row-min pass (first frame of a batch / single-frame updates); inside a batch the
// row-min of frame f+1 rides on k_track of frame f.
template <typename DetT>
__global__ __launch_bounds__(256) void k_rowmin(TrackerDev t, const DetT *__restrict__ det, int m_host,
                                                const int32_t *m_dev, int set_row_base, const long long *row_count_ext,
                                                DetGrid grid)
{
    // (fused path, start of ysmr_tracker_run / _update: row_base of the current state := the caller's
    // running row count; folded in here to save a launch per batch)
    if (set_row_base && blockIdx.x == 0 && threadIdx.x == 0) t.row_base[0] = row_count_ext ? *row_count_ext : 0;
    const int n = *t.n_tracks;
    const int m = det_count(m_host, m_dev, t.max_det, nullptr);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n || m == 0) return;
    const int slot = t.order[row];
    const int lane = threadIdx.x & 63;
    if (lane == 0 && !t.gone_by_row) t.row_gone[row] = t.gone[slot];
    const GridHdr gh = grid_hdr(grid);
    if (grid.start && rowmin_grid(t, row, t.pos[slot], t.pos[t.capacity + slot], det, grid, lane, gh)) return;
    DetChunk<DetT> first;
    load_chunk(first, det, m, 0, lane);
    rowmin_wave(t, row, t.pos[slot], t.pos[t.capacity + slot], det, m, lane, first);
}

// ------------------------------------------------------------------------------------------
// GSFF (gsff.py), one WAVE per track.  The track's history ring (hist_cap entries of (x, y),
// contiguous in HBM) is spread over the lanes -- value k = 2*entry + component lives in lane
// k % 64 -- so a FIR estimate sum_k G[k] * y[k] is one multiply per lane plus a butterfly
// reduction instead of a 2N-long chain of dependent float64 adds, and the whole state arrives with
// one coalesced load.
// ------------------------------------------------------------------------------------------
constexpr int TRACK_VALS = 2;  // values per lane: supports hist_cap <= 64 entries

struct TrackRegs {
    double v[TRACK_VALS];   // history values held by this lane
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    // lanes outside ROW_MASK / without a source get +0.0 (old = 0, bound_ctrl off writes old)
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// sum over the 64 lanes, returned to every lane; a fixed (deterministic) association order
__device__ __forceinline__ double wave_total(double v)
{
    v = dpp_add<0x111, 0xF>(v);   // row_shr:1
    v = dpp_add<0x112, 0xF>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8  -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), 63);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// History layout inside the wave: lane l, register q holds coordinate (l >> 5) of history entry
// (l & 31) + 32 q -- x in lanes 0..31, y in lanes 32..63 -- so that for a decoupled gain (x-hat
// from x only, y-hat from y only: every constant-velocity LSF gain) one 5-step half-wave reduction
// yields both sums.  In memory the history stays [entry][coordinate].
// Entry e is the measurement of e frames ago (NEWEST FIRST, shifted by one lane every frame and written back whole: one
// 512-byte store, as a ring's single new entry was one store instruction too).  The gain that multiplies a lane's entry
// is then a property of the lane: it is fetched with the very first loads of the kernel (FirGains), where a ring needed
// its head -- part of the state -- before it could even address the gain table (1150 of the filter bank's 6000 cycles
// were that gather and its index arithmetic).
__device__ __forceinline__ int hist_entry(int lane, int q) { return (lane & 31) + 32 * q; }
__device__ __forceinline__ int hist_comp(int lane) { return lane >> 5; }

__device__ __forceinline__ double wave_shr1_f64(double v)   // the value of lane - 1
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_value(double v, int src_lane)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), src_lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// lanes 0..31 and lanes 32..63 summed separately: lane 31 and lane 63 hold the two totals
__device__ __forceinline__ double half_wave_totals(double v)
{
    v = dpp_add<0x111, 0xF>(v);   // row_shr:1
    v = dpp_add<0x112, 0xF>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8  -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    return v;
}

// This lane's gains: for each filter the coefficient of history entry (lane & 31) in the x-hat row (lanes 0..31) or the
// y-hat row (lanes 32..63) of a DECOUPLED gain; 0 where the entry lies beyond the filter's horizon.  Depends on nothing
// but the lane, so it is requested before the track's state is known.  (Entries 32.. -- horizons beyond 32 frames --
// and coupled gains fetch theirs where they are used.)
template <int NF>
struct FirGains { double g[NF]; };
template <int NF>
__device__ __forceinline__ void fir_gains_fetch(const TrackerDev &t, const double *gains, int lane, FirGains<NF> &G)
{
    const int comp = hist_comp(lane), e = hist_entry(lane, 0);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int N = t.n_i[f < t.n_f ? f : 0];
        const bool in = f < t.n_f && e < t.hist_cap && e < N && t.gains_decoupled;
        const double g = gains[in ? t.gain_off[f] + comp * (2 * N + 1) + 2 * (N - 1 - e) : 0];
        G.g[f] = in ? g : 0.0;
    }
}

// FIR estimates of ALL active filters at once (rows 0 and 1 of gain f times the last n_f[f]
// measurements: lsff_calc, gsff.py:156-177).  Branch-free over the filter bank so that the
// independent reductions interleave; filters >= mode are computed on masked zeros and discarded.
template <int NF>
__device__ __forceinline__ void fir_wave_all(const TrackerDev &t, const double *gains, const TrackRegs &h, int lane,
                                             int mode, const FirGains<NF> &G, double *x0, double *x1)
{
    const int L = t.hist_cap;
    const int comp = hist_comp(lane);
    GSTAMP(8);
    if (t.gains_decoupled) {
        double p[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) p[f] = (f < mode) ? G.g[f] * h.v[0] : 0.0;      // (G.g is 0 beyond the horizon)
        if (L > 32) {                      // (uniform) horizons beyond 32 frames: the second register of the history
            const int e = hist_entry(lane, 1);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int N = t.n_i[f < t.n_f ? f : 0];
                const bool in = (f < mode) && (e < L) && (e < N);
                const double g = gains[in ? t.gain_off[f] + comp * (2 * N + 1) + 2 * (N - 1 - e) : 0];
                p[f] = in ? p[f] + g * h.v[1] : p[f];
            }
        }
        GSTAMP(9);
#pragma unroll
        for (int f = 0; f < NF; ++f) p[f] = half_wave_totals(p[f]);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const double sx = lane_value(p[f], 31), sy = lane_value(p[f], 63);
            if (f < mode) { x0[f] = sx; x1[f] = sy; }
        }
        GSTAMP(10);
        return;
    }
    double p0[NF], p1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { p0[f] = 0.0; p1[f] = 0.0; }
#pragma unroll
    for (int q = 0; q < TRACK_VALS; ++q) {
        const int e = hist_entry(lane, q);          // = age: 0 is the newest measurement
        double ga[NF], gb[NF];
        bool in[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int N = t.n_i[f < t.n_f ? f : 0];
            in[f] = (f < mode) && (e < L) && (e < N);
            const int col = in[f] ? 2 * (N - 1 - e) + comp : 0;
            ga[f] = gains[t.gain_off[f < t.n_f ? f : 0] + col];
            gb[f] = gains[t.gain_off[f < t.n_f ? f : 0] + 2 * N + col];
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            p0[f] = in[f] ? p0[f] + ga[f] * h.v[q] : p0[f];
            p1[f] = in[f] ? p1[f] + gb[f] * h.v[q] : p1[f];
        }
    }
    // wave sums by DPP (no LDS round trips): row_shr 1,2,4,8 leave each 16-lane row's sum in its
    // last lane, row_bcast15 / row_bcast31 carry it across rows, lane 63 ends with the total
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const double sx = wave_total(p0[f]), sy = wave_total(p1[f]);
        if (f < mode) { x0[f] = sx; x1[f] = sy; }
    }
}

// GSFF correct + predict of one track by one wave (gsff.py:204-347).  z = measurement; returns the
// filtered position (o) and the prediction (p), and leaves the prediction in t.pos[slot].
// fresh_track: the slot was just (re)assigned -- ignore whatever filter state it still holds.
// filter state of one track, spread over the wave (history) or replicated (the rest)
template <int NF>
struct GsffState {
    TrackRegs h;
    double w[NF], xh0[NF], xh1[NF];
    int len, mode;
    double raw;          // this lane's field of the slot's record, as fetched (decoded by gsff_decode)
    // where the slot's state lives, worked out when it is fetched and kept for the write-back (re-reading the three base
    // pointers and the record stride from the kernel's arguments there cost four scalar loads, each with its own wait)
    double *hist_p, *rec_p, *pos_p;
    int rec_lanes;       // rec_stride
};
template <int NF>
__device__ __forceinline__ void gsff_point(const TrackerDev &t, int slot, GsffState<NF> &s)
{
    s.hist_p = t.hist + (size_t)slot * 2 * t.hist_cap;
    s.rec_p = t.rec + (size_t)slot * t.rec_stride;
    s.pos_p = t.pos + slot;
    s.rec_lanes = t.rec_stride;
}
// Issue every load of a track's filter state; no load depends on another, so one round trip.
template <int NF>
__device__ __forceinline__ void gsff_fetch(const TrackerDev &t, int slot, int lane, GsffState<NF> &s)
{
    const int L = t.hist_cap;
    gsff_point(t, slot, s);
    const double *hist = s.hist_p;
    s.raw = lane < s.rec_lanes ? s.rec_p[lane] : 0.0;
#pragma unroll
    for (int q = 0; q < TRACK_VALS; ++q) {
        const int e = hist_entry(lane, q);
        s.h.v[q] = (e < L) ? hist[2 * e + hist_comp(lane)] : 0.0;
    }
}
// the fields of the fetched record, broadcast to the wave (v_readlane: the wait for the load lands here)
template <int NF>
__device__ __forceinline__ void gsff_decode(const TrackerDev &t, GsffState<NF> &s)
{
    const int nf = t.n_f;
    const long long b = __double_as_longlong(s.raw);
    const int lo = (int)(b & 0xFFFFFFFFll), hi = (int)(b >> 32);
    auto field = [&](int k) {
        return __longlong_as_double(((long long)__builtin_amdgcn_readlane(hi, k) << 32) | (unsigned int)__builtin_amdgcn_readlane(lo, k));
    };
    s.len = __builtin_amdgcn_readlane(lo, 0);
    s.mode = __builtin_amdgcn_readlane(lo, 1);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const bool on = f < nf;   // entries >= mode hold stale values that nothing reads
        s.w[f] = on ? field(2 + f) : 0.0;
        s.xh0[f] = on ? field(2 + nf + f) : 0.0;
        s.xh1[f] = on ? field(2 + 2 * nf + f) : 0.0;
    }
}
template <int NF>
__device__ __forceinline__ void gsff_blank(GsffState<NF> &s)
{
    s.len = s.mode = 0;
    s.raw = 0.0;         // (decodes to an empty bank)
#pragma unroll
    for (int q = 0; q < TRACK_VALS; ++q) s.h.v[q] = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) s.w[f] = s.xh0[f] = s.xh1[f] = 0.0;
}
template <int NF>
__device__ __forceinline__ void gsff_wave(const TrackerDev &t, const double *gains, int slot, int lane, double z0,
                                          double z1, bool fresh_track, GsffState<NF> &st, const FirGains<NF> &G,
                                          double &o0, double &o1, double &p0, double &p1)
{
    const int cap = t.capacity, nf = t.n_f, L = t.hist_cap;
    GSTAMP(0);
    double *hist = st.hist_p;
    gsff_decode(t, st);
    int len = st.len, mode = st.mode;
    TrackRegs &h = st.h;
    double (&w)[NF] = st.w;
    double (&xh0)[NF] = st.xh0;
    double (&xh1)[NF] = st.xh1;
    GSTAMP(1);
    // ---- correct (gsff.py:251-347)
    bool fresh = (len == 0) || fresh_track;
    if (fresh_track) { len = 0; mode = 0; }
    if (fresh) {   // history starts as n_i[0] copies of the first measurement
#pragma unroll
        for (int q = 0; q < TRACK_VALS; ++q)
            if (hist_entry(lane, q) < t.n_i[0]) h.v[q] = hist_comp(lane) ? z1 : z0;
        len = t.n_i[0];
    }
    bool grew = false;
    if (mode < nf) {
        while (len >= t.n_i[mode]) {
            ++mode;
            grew = true;
            if (mode >= nf) break;
        }
    }
    if (grew) {
        const double w0 = 1.0 / (double)mode;
        fir_wave_all<NF>(t, gains, h, lane, mode, G, xh0, xh1);
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if (f < mode) w[f] = w0;
    }
    GSTAMP(2);
    // append the measurement and start the predict-step FIR (gsff.py:204-249) right away: it reads
    // only the history, so its reductions overlap the likelihood arithmetic below
    // (every entry one frame older: one lane up; the measurement becomes entry 0 of each half)
    if (L > 32) {                          // (uniform) the second register takes over what leaves the first
        const double cx = lane_value(h.v[0], 31), cy = lane_value(h.v[0], 63);
        h.v[1] = wave_shr1_f64(h.v[1]);
        h.v[1] = (lane & 31) == 0 ? (hist_comp(lane) ? cy : cx) : h.v[1];
    }
    h.v[0] = wave_shr1_f64(h.v[0]);
    h.v[0] = (lane & 31) == 0 ? (hist_comp(lane) ? z1 : z0) : h.v[0];
    if (len < L) ++len;
    double nx0[NF], nx1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { nx0[f] = 0.0; nx1[f] = 0.0; }
    fir_wave_all<NF>(t, gains, h, lane, mode, G, nx0, nx1);
    GSTAMP(3);
    // likelihoods (gsff.py:179-202): lane f evaluates filter f's exp(), the results are broadcast
    // (the float64 exp is ~150 instructions; doing the n_f of them one after the other on every
    // lane was a third of this function)
    double lik[NF];
    double total = 0.0;
    {
        double xa = 0.0, xb = 0.0;
#pragma unroll
        for (int f = 0; f < NF; ++f) { xa = (lane == f) ? xh0[f] : xa; xb = (lane == f) ? xh1[f] : xb; }
        double d0 = z0 - xa, d1 = z1 - xb;
        double q = d0 * d0;
        q = q + d1 * d1;
        double l = exp(-0.5 * q);
        if (l < t.lik_min) l = t.lik_min;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            lik[f] = lane_value(l, f);
            total = (f < mode) ? total + lik[f] * w[f] : total;
        }
    }
    // new weights; output = np.sum(x_hat * w, axis=1) = a0 + ((a1 + a2) + ...)
    double f0 = 0.0, f1 = 0.0, r0 = 0.0, r1 = 0.0;
    {   // new weights w_i = lik_i * w_i / total: one division per lane instead of n_f per lane
        double lw = 1.0;
#pragma unroll
        for (int f = 0; f < NF; ++f) lw = (lane == f) ? lik[f] * w[f] : lw;
        const double wn = lw / total;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const double v = lane_value(wn, f);
            w[f] = (f < mode) ? v : w[f];
        }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
        if (f < mode) {
            double a = xh0[f] * w[f], b = xh1[f] * w[f];
            if (f == 0) { f0 = a; f1 = b; }
            else if (f == 1) { r0 = a; r1 = b; }
            else { r0 = r0 + a; r1 = r1 + b; }
        }
    o0 = mode > 1 ? f0 + r0 : f0;
    o1 = mode > 1 ? f1 + r1 : f1;
    GSTAMP(4);
    // ---- predict: the new estimates, weighted
    f0 = f1 = r0 = r1 = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f)
        if (f < mode) {
            xh0[f] = nx0[f]; xh1[f] = nx1[f];
            double a = xh0[f] * w[f], b = xh1[f] * w[f];
            if (f == 0) { f0 = a; f1 = b; }
            else if (f == 1) { r0 = a; r1 = b; }
            else { r0 = r0 + a; r1 = r1 + b; }
        }
    p0 = mode > 1 ? f0 + r0 : f0;
    p1 = mode > 1 ? f1 + r1 : f1;
    GSTAMP(5);
    // ---- write back
#pragma unroll
    for (int q = 0; q < TRACK_VALS; ++q) {
        const int e = hist_entry(lane, q);
        if (e < L && (q == 0 || L > 32)) hist[2 * e + hist_comp(lane)] = h.v[q];     // the whole history, shifted
    }
    if (lane == 0) {
        st.pos_p[0] = p0;
        st.pos_p[cap] = p1;
    }
    {   // the record, one store: every lane picks its field
        double val = __longlong_as_double(lane == 0 ? (long long)(unsigned int)len : (long long)(unsigned int)mode);
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if (f < nf) {
                val = lane == 2 + f ? w[f] : val;
                val = lane == 2 + nf + f ? xh0[f] : val;
                val = lane == 2 + 2 * nf + f ? xh1[f] : val;
            }
        if (lane < st.rec_lanes) st.rec_p[lane] = val;
    }
    GSTAMP(6);
}

template <typename DetT, int NF>
__global__ __launch_bounds__(256) void k_track(TrackerDev t, int frame, ysmr_row *rows, long long rows_capacity,
                                               const DetT *__restrict__ det, const DetT *__restrict__ next_det,
                                               int next_m_host, const int32_t *next_m_dev, DetGrid next_grid)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) RING((4ull << 40) | (unsigned)frame);   // k_track entry
    BSTAMP(0);
    // one round trip: the table size, this wave's slot (row i of a table of `capacity` rows: valid memory whatever n is)
    // and the next frame's grid header are requested together
    // (one straight-line block, the counters first -- see k_frame: a branch in here costs a scalar load of the arguments
    // behind it and a wait of its own)
    const int cap = t.capacity;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_live = *t.n_tracks, claims_by_row = t.n_tracks[4];
    const int slot_spec = t.order[min(i, cap - 1)];
    const int claim_spec = t.claim_row[min(i, cap - 1)];
    const float *hdr = next_grid.start ? next_grid.hdr : reinterpret_cast<const float *>(t.n_tracks);   // (always readable)
    const GridHdr gh{hdr[0], hdr[1], hdr[2], hdr[3]};
    FirGains<NF> G;
    fir_gains_fetch(t, t.gains, threadIdx.x & 63, G);
    if (i >= n_live) return;
    const int lane = threadIdx.x & 63;
    // the next frame's detections do not depend on this frame's state: fetch them first
    const int m_next = next_det ? det_count(next_m_host, next_m_dev, t.max_det, nullptr) : 0;
    DetChunk<DetT> first;
    if (m_next > 0 && !next_grid.start) load_chunk(first, next_det, m_next, 0, lane);
    const int slot = __builtin_amdgcn_readfirstlane(slot_spec);
    if (blockIdx.x == 0 && threadIdx.x == 0) RING((5ull << 40) | (unsigned)frame);   // slot known
    BSTAMP(1);
    // the measurement: the detection k_link let this track claim (tracker.py:186-196 -- position and box are
    // stored here, by the track's own wave), or the position it already has
    GsffState<NF> st;
    if (t.use_gsff) gsff_fetch(t, slot, lane, st);
    const int c = __builtin_amdgcn_readfirstlane(claims_by_row ? claim_spec : t.claim_slot[slot]);
    double z0, z1;
    float bw, bh, ba;
    if (c >= 0) {
        const DetT *d = det + (size_t)c * 5;
        z0 = (double)d[0]; z1 = (double)d[1];
        bw = (float)d[2]; bh = (float)d[3]; ba = (float)d[4];
        if (lane == 0) {
            t.pos[slot] = z0; t.pos[cap + slot] = z1;
            t.info[slot] = bw; t.info[cap + slot] = bh; t.info[2 * cap + slot] = ba;
        }
    } else {
        z0 = t.pos[slot]; z1 = t.pos[cap + slot];
        bw = t.info[slot]; bh = t.info[cap + slot]; ba = t.info[2 * cap + slot];
    }
    double o0 = z0, o1 = z1, p0 = z0, p1 = z1;
    if (t.use_gsff) gsff_wave(t, t.gains, slot, lane, z0, z1, false, st, G, o0, o1, p0, p1);
    if (blockIdx.x == 0 && threadIdx.x == 0) RING((6ull << 40) | (unsigned)frame);   // filter bank done
    BSTAMP(2);
    if (lane == 0) {
        const long long base = t.row_base[0];
        const int gone = t.gone[slot];
        t.row_gone[i] = gone;
        if (rows && base + i < rows_capacity) {
            ysmr_row r;
            r.frame = frame;
            r.track_id = t.id[slot];
            r.x = o0; r.y = o1;
            r.w = bw; r.h = bh; r.angle = ba;
            r.disappeared = gone;
            rows[base + i] = r;
        }
    }
    // ---- nearest detection of the NEXT frame for this track (tracker.py:151-163)
    if (m_next > 0) {
        if (next_grid.start) {
            if (rowmin_grid(t, i, p0, p1, next_det, next_grid, lane, gh)) {
                if (blockIdx.x == 0 && threadIdx.x == 0) RING((7ull << 40) | (unsigned)frame);   // next row minimum known
                BSTAMP(3);
                TRACK_END();
                return;
            }
            load_chunk(first, next_det, m_next, 0, lane);
        }
        rowmin_wave(t, i, p0, p1, next_det, m_next, lane, first);
    }
}

// ------------------------------------------------------------------------------------------
// CPython set model: iteration order of set(range(m)).difference(used_cols) (tracker.py:193,216).
// Ints hash to themselves; table sizes, linear probing (LINEAR_PROBES = 9) and perturbation
// (PERTURB_SHIFT = 5) as in Objects/setobject.c of CPython 3.7-3.12.
// The insertions are order-dependent and run on one thread; the number of unused columns per frame
// is small (k_frame does everything else of the model with the whole block).
// ------------------------------------------------------------------------------------------
__device__ void set_insert_clean(int *table, unsigned mask, int key)
{
    unsigned long long perturb = (unsigned long long)key;
    unsigned i = (unsigned)key & mask;
    while (true) {
        int probes = (i + 9u <= mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j)
            if (table[i + j] < 0) { table[i + j] = key; return; }
        perturb >>= 5;
        i = (unsigned)(((unsigned long long)i * 5ull + 1ull + perturb) & mask);
    }
}

__device__ int cpython_unused_order(const TrackerDev &t, int m, int n_used, int n_unused, int *out)
{
    const int *unused = t.unused;  // ascending
    if ((m >> 2) > n_used) {       // set_copy_and_difference: a copy of set(range(m)) iterates ascending
        for (int k = 0; k < n_unused; ++k) out[k] = unused[k];
        return n_unused;
    }
    int *table = t.set_table, *other = t.set_table + t.table_cap;
    unsigned mask = 7;
    int fill = 0;
    for (int i = 0; i < 8; ++i) table[i] = -1;
    for (int k = 0; k < n_unused; ++k) {
        set_insert_clean(table, mask, unused[k]);  // no equal keys, no dummies: add == insert_clean
        ++fill;
        if ((unsigned long long)fill * 5ull >= (unsigned long long)mask * 3ull) {
            unsigned minused = fill > 50000 ? (unsigned)fill * 2u : (unsigned)fill * 4u;
            unsigned newsize = 8;
            while (newsize <= minused) newsize <<= 1;
            if ((int)newsize > t.table_cap) return -1;
            for (unsigned i = 0; i < newsize; ++i) other[i] = -1;
            for (unsigned i = 0; i <= mask; ++i)
                if (table[i] >= 0) set_insert_clean(other, newsize - 1, table[i]);
            int *tmp = table; table = other; other = tmp;
            mask = newsize - 1;
        }
    }
    int count = 0;
    for (unsigned i = 0; i <= mask; ++i)
        if (table[i] >= 0) out[count++] = table[i];
    return count;
}

// ------------------------------------------------------------------------------------------
// k_link: one workgroup
// ------------------------------------------------------------------------------------------
#ifndef YSMR_LINK_THREADS
#define YSMR_LINK_THREADS 1024
#endif
constexpr int LINK_THREADS = YSMR_LINK_THREADS;
constexpr int LINK_ROWS = 8192 / LINK_THREADS;       // rows per thread k_link keeps in registers for tables of more than 1024 rows

// LDS_ONLY: the two barriers wait for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier) -- __syncthreads also waits
// for every global load and store the wave has in flight, a round trip to HBM each time; only for callers that order
// nothing in global memory through these barriers.
template <bool LDS_ONLY>
__device__ __forceinline__ void block_sync()
{
    if constexpr (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}

template <bool LDS_ONLY = false>
__device__ int block_exclusive_scan(int v, int *s_scan, int *total)
{
    // exclusive rank of a 0/1 flag among the block's 1024 flags: a ballot per wave, the sixteen wave
    // counts through LDS, two barriers (the Hillis-Steele scan this replaces needed twenty-one)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long bal = __ballot(v != 0);
    if (lane == 0) s_scan[w] = __popcll(bal);
    block_sync<LDS_ONLY>();
    int before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < LINK_THREADS / 64; ++k) {
        const int c = s_scan[k];
        before += k < w ? c : 0;
        all += c;
    }
    *total = all;
    block_sync<LDS_ONLY>();      // (s_scan is reused by the caller's next chunk)
    return before + __popcll(bal & ((1ull << lane) - 1ull));
}

// Up to 64 insertions at once, with the result of doing them one after the other in lane order.  A slot holds
// (sequence number << 16 | key); every lane walks its key's probe sequence and takes a slot with an atomic minimum, so
// a slot ends up with the EARLIEST key that ever asked for it; a lane whose entry was displaced by an earlier key (or
// that found an earlier key there) moves on along its sequence.  The fixed point -- every key in the first slot of its
// sequence that no earlier key holds -- is exactly what sequential insertion produces (an earlier key's place never
// depends on a later one), and it is reached in as many rounds as the longest displacement chain instead of one LDS
// round trip per probe per key (36 keys, with the re-insertions of two resizes: 14 us one key at a time).
// Entries of earlier batches carry smaller sequence numbers and never move.  seq + 63 < 65535, key < 65536.
constexpr uint32_t SET_EMPTY32 = 0xFFFFFFFFu;
__device__ __forceinline__ void set_insert_batch(uint32_t *table, unsigned mask, bool active, unsigned key, unsigned seq)
{
    const uint32_t mine = (seq << 16) | key;
    unsigned long long perturb = (unsigned long long)key;
    unsigned i = key & mask;
    int j = 0;                       // offset inside the linear run that starts at i
    unsigned at = 0;
    bool need = active;              // has to propose its next candidate
    auto advance = [&]() {           // setobject.c set_insert_clean: i .. i + 9, then the perturbed jump
        const int probes = (i + 9u <= mask) ? 9 : 0;
        if (j < probes) ++j;
        else { perturb >>= 5; i = (unsigned)(((unsigned long long)i * 5ull + 1ull + perturb) & mask); j = 0; }
    };
    while (true) {
        if (need) {
            at = i + (unsigned)j;
            const uint32_t old = atomicMin(&table[at], mine);
            if (old > mine) need = false;      // ours for now (a displaced later entry's lane notices below)
            else advance();                    // an earlier key holds it
        }
        if (active && !need && table[at] != mine) { need = true; advance(); }   // displaced by an earlier key
        if (__ballot(need) == 0ull) return;
    }
}

// The model for k_link, in the LDS its claim tables no longer need (all LINK_THREADS threads call it): one table of
// 32-bit slots (set_insert_batch), a list for the resizes and the staged list of unclaimed columns.  Wave 0 inserts, 64
// keys at a time; clearing the table, collecting it in slot order at a resize and the final walk are done by the whole
// block.  (One thread over tables in HBM, cpython_unused_order, spent 575 us on 300 unregistered columns: 700 dependent
// insertions and a 2048-slot walk.)  Returns the number of keys, or -1 when the table would outgrow `cap_table` slots
// (the caller falls back to cpython_unused_order).
__device__ int cpython_order_block(const int *unused, int n_unused, int m, int n_used, int *out, uint32_t *table,
                                   int cap_table, uint32_t *list, unsigned short *stage, int *s_scan, int *s_state)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if ((m >> 2) > n_used) {       // set_copy_and_difference: a copy of set(range(m)) iterates ascending
        for (int k = tid; k < n_unused; k += LINK_THREADS) out[k] = unused[k];
        return n_unused;
    }
    for (int k = tid; k < n_unused; k += LINK_THREADS) stage[k] = (unsigned short)unused[k];
    if (tid < 8) table[tid] = SET_EMPTY32;
    __syncthreads();
    unsigned mask = 7, seq = 0;
    int k = 0, fill = 0;
    // occupied slots of the table in slot order: keys -> dst (LDS list or the global result)
    auto collect = [&](auto *dst) {
        int base = 0;
        for (unsigned i0 = 0; i0 <= mask; i0 += LINK_THREADS) {
            const unsigned i = i0 + tid;
            const uint32_t e = i <= mask ? table[i] : SET_EMPTY32;
            int total;
            const int ex = block_exclusive_scan<true>(e != SET_EMPTY32 ? 1 : 0, s_scan, &total);
            if (e != SET_EMPTY32) dst[base + ex] = e & 0xFFFFu;
            base += total;
        }
        return base;
    };
    while (true) {
        // wave 0: keys go in until the table wants to grow (setobject.c: fill*5 >= mask*3 after an add)
        const int thresh = (int)((3u * mask + 4u) / 5u);
        const int upto = min(n_unused, k + (thresh - fill));
        if (tid < 64) {
            for (int k0 = k; k0 < upto; k0 += 64) {
                const int b = min(64, upto - k0);
                set_insert_batch(table, mask, lane < b, lane < b ? (unsigned)stage[k0 + lane] : 0u, seq + (unsigned)(k0 - k) + lane);
            }
        }
        seq += (unsigned)(upto - k); fill += upto - k; k = upto;
        __syncthreads();
        if (fill < thresh) break;
        const unsigned minused = (unsigned)fill > 50000u ? (unsigned)fill * 2u : (unsigned)fill * 4u;
        unsigned newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        if ((int)newsize > cap_table) return -1;                      // (uniform)
        const int cnt = collect(list);
        __syncthreads();
        for (unsigned i = tid; i < newsize; i += LINK_THREADS) table[i] = SET_EMPTY32;
        mask = newsize - 1;
        __syncthreads();
        if (tid < 64) {
            for (int j0 = 0; j0 < cnt; j0 += 64) {
                const int b = min(64, cnt - j0);
                set_insert_batch(table, mask, lane < b, lane < b ? list[j0 + lane] : 0u, seq + (unsigned)j0 + lane);
            }
        }
        seq += (unsigned)cnt;
        __syncthreads();
    }
    return collect(out);
}

// LDS_TABLES: the per-column winner tables fit in LDS (12 B per detection column <= 140 KiB, e.g. 96 KiB at 8192; the
// per-row claims are in HBM): the claim rounds cost LDS atomics.  Otherwise the tables live in HBM (three rounds of
// device-scope atomics, ~6 us more at 5000 rows) and capacity / max_det are only bounded by 65536.
template <typename DetT, bool LDS_TABLES>
__global__ __launch_bounds__(LINK_THREADS) void k_link(TrackerDev t, const DetT *__restrict__ det, int m_host,
                                                       const int32_t *m_dev, int frame, ysmr_row *rows,
                                                       long long rows_capacity, long long *row_count,
                                                       int32_t *n_rows_out, int32_t *claim_out, int32_t *n_before_out,
                                                       int32_t *new_cols_out, int32_t *n_new_out)
{
    extern __shared__ unsigned long long s_dyn[];
    unsigned long long *s_col_key = LDS_TABLES ? s_dyn : t.link_key;                                        // [max_det]
    int *s_col_row = LDS_TABLES ? reinterpret_cast<int *>(s_dyn + t.max_det) : t.link_row;                  // [max_det]
    // (a row's claim is written and read back by one thread only: it lives in HBM, and the workgroup asks a compute unit
    // for 12 bytes of LDS per column, not 16 -- 96 KB at 8192, which fits beside four of k_windows' workgroups)
    int *s_claim = t.link_claim;                                                                            // [capacity]
    // (in HBM, a value another wave has just changed with an atomic is read past this CU's L1)
    auto key_of = [&](int c) {
        if constexpr (LDS_TABLES) return s_col_key[c];
        else return __hip_atomic_load(&s_col_key[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto row_of = [&](int c) {
        if constexpr (LDS_TABLES) return s_col_row[c];
        else return __hip_atomic_load(&s_col_row[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    __shared__ int s_scan[LINK_THREADS];
    __shared__ int s_n_used, s_n_new, s_any_dead, s_set_state[2];
    __shared__ __attribute__((aligned(16))) int s_keep[LINK_ROWS][LINK_THREADS / 64];
    const int tid = threadIdx.x;
    const int cap = t.capacity;
    if (tid == 0) RING((9ull << 40) | (unsigned)frame);    // entry (before the first load)
    TRACK_END_TO_RING();
    const int n = *t.n_tracks, nfree0 = *t.n_free;
    int nfree_now = nfree0;        // (the register-resident path keeps the height of the free stack here)
    const int m = det_count(m_host, m_dev, t.max_det, t.err);
    LRING(0);
    if (tid == 0) { s_n_used = 0; s_n_new = 0; s_any_dead = 0; }
    for (int c = tid; c < m; c += LINK_THREADS) {
        if constexpr (LDS_TABLES) { s_col_key[c] = ~0ull; s_col_row[c] = 0x7FFFFFFF; }
        else {
            __hip_atomic_store(&s_col_key[c], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&s_col_row[c], 0x7FFFFFFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // (with the tables in LDS the barriers of the claim rounds order LDS traffic only: the rows' loads and the stores of
    // the decisions stay in flight across them)
    block_sync<LDS_TABLES>();
    LRING(1);
    // rows of a large table held per thread (see the `big` branch below)
    const bool big = n > LINK_THREADS && n <= LINK_ROWS * LINK_THREADS && m > 0;
    int pa[LINK_ROWS], po[LINK_ROWS], pg[LINK_ROWS], pd[LINK_ROWS];
    unsigned long long pk[LINK_ROWS];
#pragma unroll
    for (int k = 0; k < LINK_ROWS; ++k) { pa[k] = 0; po[k] = 0; pg[k] = 0; pd[k] = 0; pk[k] = ~0ull; }

    // ---- claims: the winner of a column is the proposer with the smallest (distance, row)
    if (n > 0 && m > 0 && n <= LINK_THREADS) {
        // common case, one row per thread: everything a claim will need is fetched up front so the
        // LDS rounds below overlap the loads instead of waiting behind them
        const int r = tid;
        int c = 0, slot = 0;
        unsigned long long key = ~0ull;
        if (r < n) {
            c = t.row_arg[r];
            key = (unsigned long long)__double_as_longlong(t.row_min[r]);
            slot = t.order[r];
            atomicMin(&s_col_key[c], key);
        }
        __syncthreads();
        if (r < n && key == key_of(c)) atomicMin(&s_col_row[c], r);
        __syncthreads();
        bool won = false;
        if (r < n) {
            const bool mine = (row_of(c) == r);
            s_claim[r] = mine ? c : -1;
            t.claim_slot[slot] = mine ? c : -1;
            if (mine) t.gone[slot] = 0;
            won = mine;
        }
        const int used = __popcll(__ballot(won));          // one LDS atomic per wave, not per thread
        if ((tid & 63) == 0 && used) atomicAdd(&s_n_used, used);
    } else if (big) {
        // large tables (the 4K configuration: ~5000 rows): thread tid owns rows tid + LINK_THREADS * k.  Everything
        // the LDS rounds, the ageing and the compaction below need is requested up front and kept in
        // registers, so that a pass costs one round of loads or atomics instead of a global round trip per row chunk
        // (requesting the rows before n is known, 8192 instead of ~5000, was slower: one unit's load path is the limit)
#pragma unroll
        for (int k = 0; k < LINK_ROWS; ++k) {
            const int r = tid + k * LINK_THREADS;
            if (r < n) {
                pa[k] = t.row_arg[r];
                pk[k] = (unsigned long long)__double_as_longlong(t.row_min[r]);
                po[k] = t.order[r];
                pg[k] = t.row_gone[r];
            }
        }
#ifdef YSMR_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LRING(7);          // the rows' loads have landed
#endif
#pragma unroll
        for (int k = 0; k < LINK_ROWS; ++k)
            if (tid + k * LINK_THREADS < n) atomicMin(&s_col_key[pa[k]], pk[k]);
        block_sync<LDS_TABLES>();
        LRING(8);
#pragma unroll
        for (int k = 0; k < LINK_ROWS; ++k) {
            const int r = tid + k * LINK_THREADS;
            if (r < n && pk[k] == key_of(pa[k])) atomicMin(&s_col_row[pa[k]], r);
        }
        block_sync<LDS_TABLES>();
        LRING(9);
        int used = 0;
#pragma unroll
        for (int k = 0; k < LINK_ROWS; ++k) {
            const int r = tid + k * LINK_THREADS;
            const bool mine = r < n && row_of(pa[k]) == r;
            used += __popcll(__ballot(mine));
            if (r >= n) continue;
            if (claim_out) s_claim[r] = mine ? pa[k] : -1;
            if (mine) {
                t.gone[po[k]] = 0;
                pg[k] = -1;          // claimed (ageing below skips it; claim_row below)
            }
        }
        if ((tid & 63) == 0 && used) atomicAdd(&s_n_used, used);   // one LDS atomic per wave (1024 on one address: 3 us)
    } else if (n > 0 && m > 0) {
        for (int r = tid; r < n; r += LINK_THREADS)
            atomicMin(&s_col_key[t.row_arg[r]], (unsigned long long)__double_as_longlong(t.row_min[r]));
        __syncthreads();
        for (int r = tid; r < n; r += LINK_THREADS) {
            int c = t.row_arg[r];
            if ((unsigned long long)__double_as_longlong(t.row_min[r]) == key_of(c)) atomicMin(&s_col_row[c], r);
        }
        __syncthreads();
        int used = 0;
        for (int r = tid; r < n; r += LINK_THREADS) {
            int c = t.row_arg[r];
            int mine = (row_of(c) == r) ? c : -1;
            const int slot = t.order[r];
            s_claim[r] = mine;
            t.claim_slot[slot] = mine;
            if (mine >= 0) {
                t.gone[slot] = 0;
                ++used;
            }
        }
        if (used) atomicAdd(&s_n_used, used);
    } else {
        for (int r = tid; r < n; r += LINK_THREADS) { s_claim[r] = -1; t.claim_slot[t.order[r]] = -1; }
    }
    block_sync<LDS_TABLES>();      // (s_n_used; what the paths above stored in global memory is read back by no other thread)

    LRING(2);
    // ---- ageing (tracker.py:95-107, 198-211): only when there are no detections or N >= M
    const bool age = (m == 0) || (n > 0 && n >= m);
    if (age && big) {
#pragma unroll
        for (int k = 0; k < LINK_ROWS; ++k) {
            const int r = tid + k * LINK_THREADS;
            if (r >= n) continue;
            int d = 0;
            if (pg[k] >= 0) {            // not claimed this frame
                const int slot = po[k], g = pg[k] + 1;
                t.gone[slot] = g;
                t.info[slot] = 0.f; t.info[cap + slot] = 0.f; t.info[2 * cap + slot] = 0.f;
                if ((double)g > t.max_gone) { d = 1; s_any_dead = 1; }
            }
            pd[k] = d;
        }
    } else if (age) {
        for (int r = tid; r < n; r += LINK_THREADS) {
            int d = 0;
            if (s_claim[r] < 0) {
                int slot = t.order[r];
                int g = t.gone[slot] + 1;
                t.gone[slot] = g;
                t.info[slot] = 0.f; t.info[cap + slot] = 0.f; t.info[2 * cap + slot] = 0.f;
                if ((double)g > t.max_gone) { d = 1; s_any_dead = 1; }
            }
            t.dead[r] = d;
        }
    }
    block_sync<true>();            // (s_any_dead; a row's `dead` flag is read back by the thread that wrote it)

    LRING(3);
    // ---- stable compaction of the ordered track table
    int n_live = n;
    if (s_any_dead && big) {
        // one pass: every wave counts its survivors chunk by chunk (a ballot each), ONE barrier, every thread adds up the
        // counts in front of its rows (the chunk-by-chunk scan took two barriers per 1024 rows).  In place: every row's
        // slot has been in registers since the start
        // The slots of the dead go on the free stack in row order, ranked the same way (a returning global atomic per
        // dead row was a round trip to HBM inside this phase).  Nothing written here is read again in this launch: ageing
        // happens when n >= m, registration -- the reader of the free stack -- when n < m.
        const int lane = tid & 63, w = tid >> 6;
        const unsigned long long below = (1ull << lane) - 1ull;
        unsigned long long bal[LINK_ROWS], dbal[LINK_ROWS];
#pragma unroll
        for (int k = 0; k < LINK_ROWS; ++k) {
            const int r = tid + k * LINK_THREADS;
            bal[k] = __ballot(r < n && !pd[k]);
            dbal[k] = __ballot(r < n && pd[k]);
            if (lane == 0) s_keep[k][w] = __popcll(bal[k]) | (__popcll(dbal[k]) << 16);
        }
        block_sync<true>();
        int base = 0, dbase = 0;
#pragma unroll
        for (int k = 0; k < LINK_ROWS; ++k) {
            const int r = tid + k * LINK_THREADS;
            int before = 0, all = 0;
            static_assert((LINK_THREADS / 64) % 4 == 0, "16-byte reads of the waves' counts");
#pragma unroll
            for (int q4 = 0; q4 < LINK_THREADS / 64 / 4; ++q4) {
                const int4 c = reinterpret_cast<const int4 *>(s_keep[k])[q4];
                const int cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    before += 4 * q4 + e < w ? cs[e] : 0;
                    all += cs[e];
                }
            }
            if (r < n) {
                if (!pd[k]) {
                    const int at = base + (before & 0xFFFF) + __popcll(bal[k] & below);
                    t.order[at] = po[k];
                    t.claim_row[at] = pg[k] == -1 ? pa[k] : -1;
                } else t.free_slots[nfree0 + dbase + (before >> 16) + __popcll(dbal[k] & below)] = po[k];
            }
            base += all & 0xFFFF;
            dbase += all >> 16;
        }
        n_live = base;
        nfree_now = nfree0 + dbase;
        if (tid == 0) *t.n_free = nfree_now;
    } else if (big) {              // nobody died: the rows stay where they are
#pragma unroll
        for (int k = 0; k < LINK_ROWS; ++k) {
            const int r = tid + k * LINK_THREADS;
            if (r < n) t.claim_row[r] = pg[k] == -1 ? pa[k] : -1;
        }
    } else if (s_any_dead) {
        int base = 0;
        for (int r0 = 0; r0 < n; r0 += LINK_THREADS) {
            int r = r0 + tid;
            int keep = (r < n && !t.dead[r]) ? 1 : 0;
            const int slot_r = r < n ? t.order[r] : 0;   // read before the scan's barriers: the writes below go to rows
            int total;                                   // at or before this chunk's, all of them read by then
            int ex = block_exclusive_scan(keep, s_scan, &total);
            if (keep) t.order[base + ex] = slot_r;
            if (r < n && !keep) {
                int k = atomicAdd(t.n_free, 1);
                t.free_slots[k] = slot_r;
            }
            base += total;
        }
        n_live = base;
        __threadfence_block();
        __syncthreads();
    }

    LRING(4);
    // ---- registration (tracker.py:135-137, 212-217)
    if (m > 0 && (n == 0 || n < m)) {
        if (n == 0) {
            for (int c = tid; c < m; c += LINK_THREADS) t.new_cols[c] = c;
            if (tid == 0) s_n_new = m;
        } else {
            // unused columns in ascending order (parallel), then CPython's set order (one thread)
            int base = 0;
            for (int c0 = 0; c0 < m; c0 += LINK_THREADS) {
                int c = c0 + tid;
                int un = (c < m && row_of(c) == 0x7FFFFFFF) ? 1 : 0;
                int total;
                int ex = block_exclusive_scan(un, s_scan, &total);
                if (un) t.unused[base + ex] = c;
                base += total;
            }
            __threadfence_block();
            __syncthreads();
            int cnt = -1;
            if constexpr (LDS_TABLES) {
                // (the claim tables are dead from here on: of the 8 B per column of the key table, 4 hold the set model's
                // table of max_det slots and 2 the staged list; the 4 B per column of the row table hold its resize list)
                uint32_t *tab = reinterpret_cast<uint32_t *>(s_dyn);
                unsigned short *stage = reinterpret_cast<unsigned short *>(tab + t.max_det);
                uint32_t *list = reinterpret_cast<uint32_t *>(s_dyn + t.max_det);
                if (t.max_det < 65535)
                    cnt = cpython_order_block(t.unused, base, m, s_n_used, t.new_cols, tab, t.max_det, list, stage, s_scan,
                                              s_set_state);
            }
            if (cnt >= 0) {
                if (tid == 0) s_n_new = cnt;
            } else if (tid == 0) {
                cnt = cpython_unused_order(t, m, s_n_used, base, t.new_cols);
                if (cnt < 0) { atomicOr(t.err, ERR_TRACK_CAPACITY); cnt = 0; }
                s_n_new = cnt;
            }
        }
        __threadfence_block();
        __syncthreads();
        int n_new = s_n_new;
        if (n_live + n_new > cap) {
            if (tid == 0) atomicOr(t.err, ERR_TRACK_CAPACITY);
            n_new = cap - n_live;
        }
        const int nfree = big ? nfree_now : *t.n_free;     // (the other paths push the dead with global atomics)
        const int id0 = *t.next_id;
        for (int j = tid; j < n_new; j += LINK_THREADS) {
            int c = t.new_cols[j];
            int slot = t.free_slots[nfree - 1 - j];
            t.order[n_live + j] = slot;
            t.id[slot] = id0 + j;
            t.pos[slot] = (double)det[(size_t)c * 5 + 0];
            t.pos[cap + slot] = (double)det[(size_t)c * 5 + 1];
            t.info[slot] = (float)det[(size_t)c * 5 + 2];
            t.info[cap + slot] = (float)det[(size_t)c * 5 + 3];
            t.info[2 * cap + slot] = (float)det[(size_t)c * 5 + 4];
            t.gone[slot] = 0;
            t.claim_slot[slot] = -1;                        // (k_track reads the position stored here)
            t.claim_row[n_live + j] = -1;
            t.rec[(size_t)slot * t.rec_stride + 0] = 0.0;   // history length, head
            t.rec[(size_t)slot * t.rec_stride + 1] = 0.0;   // mode
            if (new_cols_out) new_cols_out[j] = c;
        }
        __syncthreads();
        if (tid == 0) {
            *t.n_free = nfree - n_new;
            *t.next_id = id0 + s_n_new;  // ids are consumed even for registrations dropped on overflow
            s_n_new = n_new;
        }
        __syncthreads();
        n_live += n_new;
    }

    LRING(5);
    // ---- bookkeeping for k_track
    if (claim_out)
        for (int r = tid; r < n; r += LINK_THREADS) claim_out[r] = s_claim[r];
    if (tid == 0) {
        long long base = row_count ? *row_count : 0;
        *t.n_tracks = n_live;
        t.n_tracks[4] = big ? 1 : 0;      // claim_row is valid for this frame
        t.row_base[0] = base;
        if (rows && base + n_live > rows_capacity) atomicOr(t.err, ERR_ROWS_CAPACITY);
        if (row_count) *row_count = base + n_live;
        if (n_rows_out) *n_rows_out = n_live;
        if (n_before_out) *n_before_out = n;
        if (n_new_out) *n_new_out = s_n_new;
    }
    LRING(6);
}

// ------------------------------------------------------------------------------------------
// k_frame: the whole CentroidTracker.update of one frame in ONE launch (fused path).
//
// Every block redundantly resolves the frame's bookkeeping in LDS from the same inputs -- claims,
// ageing, which tracks die, the compacted id-ordered table, which detections become new tracks and
// in which (CPython set) order -- so no inter-block synchronisation is needed; then each WAVE
// processes one track of the NEW table: applies its claim, runs the GSFF, writes its row, and
// computes its nearest detection of the next frame.  Small per-frame arrays that one block would
// write while another still reads them (order, gone, row_min/row_arg, the counters) are
// double-buffered by frame parity: `a` is the state before this frame, `b` the state after it.
// ------------------------------------------------------------------------------------------
constexpr int FRAME_THREADS = 256;
constexpr int FRAME_TABLE = 4096;   // CPython set model table (entries) in LDS: up to 2457 unused columns

// LDS of k_frame.  Kept small on purpose (59 KB at capacity = max_det = 2048): a block that wants
// most of a CU's 160 KB cannot be placed while kernels of another stream hold LDS there, and the
// link waited tens of microseconds per frame for that.  Per-row inputs (slot, gone, row_arg,
// row_min) stay in the registers of the thread that owns the row instead.
struct FrameLds {
    double *gains;                 // [gain_total] LDS copy of the gain rows
    unsigned long long *col_key;   // [max_det] smallest proposing distance per column ...
    int *unused, *newcols;         //   ... reused after the claims: unused columns, registration order
    int *col_row;                  // [max_det] winning row per column
    int *cg;                       // [capacity] (claimed column + 1) | new `gone` << 16, per old row
    int *inv;                      // [capacity] new row -> old row
    int *scan;                     // [16]
    short *table;                  // 4 * FRAME_TABLE bytes: the set model's table of 32-bit slots
};

__host__ __device__ inline size_t frame_lds_bytes(int cap, int max_det, int gain_total)
{
    return 8 * ((size_t)gain_total + max_det) + 4 * ((size_t)max_det + 2 * (size_t)cap + 16) + 2 * 2 * FRAME_TABLE + 64;
}

// Exclusive rank of this thread's flag among the block's flags (thread order) and their count:
// a ballot per wave, four wave counts through LDS, ONE barrier.  Consecutive calls must alternate
// between two 4-int buffers (a wave may still be reading the previous call's counts).
__device__ __forceinline__ int block_flag_rank(bool f, int *s_cnt, int *total)
{
    const unsigned long long bal = __ballot(f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_cnt[w] = __popcll(bal);
    __syncthreads();
    const int c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
    *total = c0 + c1 + c2 + c3;
    const int before = (w > 0 ? c0 : 0) + (w > 1 ? c1 : 0) + (w > 2 ? c2 : 0);
    return before + __popcll(bal & ((1ull << lane) - 1ull));
}

// CPython order of the ascending list `unused` (see cpython_unused_order), 16-bit tables in LDS.
// All FRAME_THREADS threads call it; the model itself runs on wave 0 alone, without a block barrier inside.  One table
// of 32-bit slots (set_insert_batch) and a list: the keys of a stage go in 64 at a time; at a resize the wave writes the
// occupied slots in slot order to `list` (ballot prefix), clears the table at its new size and re-inserts the list, 64
// at a time; the final walk writes the iteration order the same way.  LDS operations of one wave execute in order, so
// nothing but program order is needed between these steps.
// Returns the number of keys, or -1 if the model would need a table larger than FRAME_TABLE.
template <int THREADS, int TABLE>
__device__ int cpython_order_lds(const int *unused, int n_unused, int m, int n_used, int *out, uint32_t *table,
                                 uint32_t *list, int *s_state)
{
    const int tid = threadIdx.x;
    if ((m >> 2) > n_used) {
        for (int k = tid; k < n_unused; k += THREADS) out[k] = unused[k];
        __syncthreads();
        return n_unused;
    }
    if (tid < 64) {
        const int lane = tid;
        unsigned mask = 7, seq = 0;
        int k = 0, fill = 0, result = 0;
        if (lane < 8) table[lane] = SET_EMPTY32;
        while (true) {
            // keys go in until the table wants to grow (setobject.c: fill*5 >= mask*3 after an add)
            const int thresh = (int)((3u * mask + 4u) / 5u);
            while (k < n_unused && fill < thresh) {
                const int b = min(64, min(n_unused - k, thresh - fill));
                set_insert_batch(table, mask, lane < b, lane < b ? (unsigned)unused[k + lane] : 0u, seq + lane);
                k += b; fill += b; seq += b;
            }
            if (fill < thresh) break;
            const unsigned minused = (unsigned)fill > 50000u ? (unsigned)fill * 2u : (unsigned)fill * 4u;
            unsigned newsize = 8;
            while (newsize <= minused) newsize <<= 1;
            if ((int)newsize > TABLE) { result = -1; break; }
            int cnt = 0;
            for (unsigned i0 = 0; i0 <= mask; i0 += 64) {   // the old table in slot order -> list
                const uint32_t e = i0 + lane <= mask ? table[i0 + lane] : SET_EMPTY32;
                const unsigned long long bal = __ballot(e != SET_EMPTY32);
                if (e != SET_EMPTY32) list[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = e & 0xFFFFu;
                cnt += __popcll(bal);
            }
            for (unsigned i = lane; i < newsize; i += 64) table[i] = SET_EMPTY32;
            mask = newsize - 1;
            for (int j0 = 0; j0 < cnt; j0 += 64) {
                const int b = min(64, cnt - j0);
                set_insert_batch(table, mask, lane < b, lane < b ? list[j0 + lane] : 0u, seq + lane);
                seq += b;
            }
        }
        if (result == 0) {
            for (unsigned i0 = 0; i0 <= mask; i0 += 64) {
                const uint32_t e = i0 + lane <= mask ? table[i0 + lane] : SET_EMPTY32;
                const unsigned long long bal = __ballot(e != SET_EMPTY32);
                if (e != SET_EMPTY32) out[result + __popcll(bal & ((1ull << lane) - 1ull))] = (int)(e & 0xFFFFu);
                result += __popcll(bal);
            }
        }
        if (lane == 0) s_state[0] = result;
    }
    __syncthreads();
    const int r = s_state[0];
    __syncthreads();          // (s_state may be written again by the caller's next use)
    return r;
}

#ifdef YSMR_STAMPS
#ifndef YSMR_ST_FRAME
#define YSMR_ST_FRAME 40
#endif
#define STAMP(k) do { if (blockIdx.x == 1 && threadIdx.x == 0 && (YSMR_ST_FRAME < 64 ? (frame & 63) == YSMR_ST_FRAME : frame == YSMR_ST_FRAME)) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ((unsigned long long *)(rows + rows_capacity - 4))[k] = t_; } } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

template <typename DetT, int NF>
__global__ __launch_bounds__(FRAME_THREADS) void k_frame(TrackerDev a, TrackerDev b, const DetT *__restrict__ det,
                                                         int m_host, const int32_t *m_dev, int frame, ysmr_row *rows,
                                                         long long rows_capacity, long long *row_count_ext,
                                                         int32_t *n_rows_out, int32_t *claim_out, int32_t *n_before_out,
                                                         int32_t *new_cols_out, int32_t *n_new_out,
                                                         const DetT *__restrict__ next_det, const int32_t *next_m_dev,
                                                         int base_from_ext)
{
    extern __shared__ unsigned long long s_raw[];
    __shared__ int s_n_used, s_n_dead, s_set_state[2];
    // the link is a chain of short latency-bound kernels sharing SIMDs with throughput kernels of the
    // detection stream: let its waves win the issue arbitration
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cap = a.capacity, md = a.max_det;
    STAMP(0); BSTAMP(0);
#ifdef YSMR_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) RING((2ull << 40) | (unsigned)frame);
#endif
    // ---- round trip 1: everything that can be addressed without knowing n or m is requested
    // together with the counters (entries past n / m are stale and never used).  ONE straight-line block, the
    // counters first: every branch in here made the compiler fetch the kernel arguments it needs behind it in a
    // separate scalar load with its own wait, and the counters -- which the first decision waits for -- were requested
    // at the end of that chain of waits (4 k cycles from entry to the exit test).  (Passing what this block is addressed
    // with as eight leading pointer arguments, so that one 64-byte scalar load precedes the requests, was 6 % SLOWER end
    // to end, with or without kernel-argument preload.)
    // All counters in the same round trip: the two detection counts are read through pointers that
    // are always valid (a dummy when the count comes from the host / there is no next frame), so
    // that no load hides behind a branch and a wait
    const int32_t *m_ptr = m_host < 0 ? m_dev : a.n_tracks;
    const int32_t *mn_ptr = next_det ? next_m_dev : a.n_tracks;
    const int n_raw = *a.n_tracks, nfree_raw = *a.n_free, id0_raw = *a.next_id;
    const int m_raw = *m_ptr, mn_raw = *mn_ptr;
    // (first frame of a run whose k_rowmin -- which also takes the caller's row count as the base -- was not needed)
    const long long *base_ptr = (base_from_ext && row_count_ext) ? row_count_ext : a.row_base;
    const long long row_base_raw = *base_ptr;
    constexpr int SPEC_ROWS = 3;       // table rows tid + 256k, k < 3, are fetched before n is known
    const int i = blockIdx.x * 4 + wave;
    int so[SPEC_ROWS], sg[SPEC_ROWS], sa[SPEC_ROWS];
    unsigned long long sk[SPEC_ROWS];
#pragma unroll
    for (int k = 0; k < SPEC_ROWS; ++k) {
        const int r = min(tid + k * FRAME_THREADS, cap - 1);
        so[k] = a.order[r];
        sg[k] = a.gone[r];          // fused path: `gone` is indexed by table row, not by slot
        sa[k] = a.row_arg[r];
        sk[k] = (unsigned long long)__double_as_longlong(a.row_min[r]);
    }
    // Speculation for phase B: unless a track ahead of row i is deregistered this frame, row i of
    // the new table is row i of the old one, keeps its slot and can only claim row_arg[i].
    int slot_s = a.order[min(i, cap - 1)], c_s = a.row_arg[min(i, cap - 1)];
    FirGains<NF> G;                    // (this lane's filter gains, straight from the table: no copy of it in LDS;
    fir_gains_fetch(a, a.gains, lane, G);   //  without the filter bank the table has one readable entry and every gain is 0)
    // the next frame's detections do not depend on this frame's state either (rows past m_next
    // of the [max_det][5] frame slice are stale; rowmin_wave masks them; without a next frame this frame's are read)
    DetChunk<DetT> first;
    load_chunk(first, next_det ? next_det : det, md, 0, lane);
    const long long row_base = (base_from_ext && !row_count_ext) ? 0ll : row_base_raw;
    const int n = __builtin_amdgcn_readfirstlane(n_raw);
    const int nfree = __builtin_amdgcn_readfirstlane(nfree_raw), id0 = __builtin_amdgcn_readfirstlane(id0_raw);
    int m = __builtin_amdgcn_readfirstlane(m_host < 0 ? m_raw : m_host);
    if (m > md) { m = md; if (blockIdx.x == 0) atomicOr(a.err, ERR_DET_CLAMPED); }
    m = max(m, 0);
    const int m_next = next_det ? max(min(__builtin_amdgcn_readfirstlane(mn_raw), md), 0) : 0;
    if ((long long)blockIdx.x * 4 >= (long long)n + m && blockIdx.x != 0) return;   // cannot own a live track
    STAMP(1);

    FrameLds L;
    L.gains = reinterpret_cast<double *>(s_raw);
    L.col_key = s_raw + a.gain_total;
    L.unused = reinterpret_cast<int *>(L.col_key);
    L.newcols = L.unused + md;
    L.col_row = reinterpret_cast<int *>(L.col_key + md);
    L.cg = L.col_row + md;
    L.inv = L.cg + cap;
    L.scan = L.inv + cap;
    L.table = reinterpret_cast<short *>(L.scan + 16);

    // ---- phase A (redundant in every block): the frame's bookkeeping
    if (tid == 0) { s_n_used = 0; s_n_dead = 0; }
    for (int c = tid; c < m; c += FRAME_THREADS) { L.col_key[c] = ~0ull; L.col_row[c] = 0x7FFFFFFF; }
    slot_s = __builtin_amdgcn_readfirstlane(i < n ? slot_s : -1);
    c_s = __builtin_amdgcn_readfirstlane((i < n && m > 0) ? c_s : -1);
    GsffState<NF> S;
    double zs0 = 0.0, zs1 = 0.0;
    int id_s = 0;
    float in_s[3] = {0.f, 0.f, 0.f};
    DetT dd[5] = {0, 0, 0, 0, 0};
    if (slot_s >= 0) {
        if (a.use_gsff) gsff_fetch(a, slot_s, lane, S);
        zs0 = a.pos[slot_s]; zs1 = a.pos[cap + slot_s];
        id_s = a.id[slot_s];
#pragma unroll
        for (int k = 0; k < 3; ++k) in_s[k] = a.info[k * cap + slot_s];
        if (c_s >= 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) dd[k] = det[(size_t)c_s * 5 + k];
        }
    }
    // row r = tid + 256 k of the old table: from the registers above for k < SPEC_ROWS (the passes over them are
    // unrolled: picking so[k] with a run-time k was eight selects per row and pass on a wave that pays for every
    // instruction), from HBM beyond
    auto row_from_hbm = [&](int r, int &o_, int &g_, int &a_, unsigned long long &key_) {
        if (r < n) {
            o_ = a.order[r]; g_ = a.gone[r]; a_ = a.row_arg[r];
            key_ = (unsigned long long)__double_as_longlong(a.row_min[r]);
        }
    };
    __syncthreads();
    STAMP(2);
    if (n > 0 && m > 0) {   // winner of a column = proposer with the smallest (distance, row)
#pragma unroll
        for (int k = 0; k < SPEC_ROWS; ++k) {
            const int r = k * FRAME_THREADS + tid;
            if (r < n) atomicMin(&L.col_key[sa[k]], sk[k]);
        }
        for (int r0 = SPEC_ROWS * FRAME_THREADS; r0 < n; r0 += FRAME_THREADS) {
            const int r = r0 + tid;
            int o_ = 0, g_ = 0, a_ = 0; unsigned long long key_ = 0;
            row_from_hbm(r, o_, g_, a_, key_);
            if (r < n) atomicMin(&L.col_key[a_], key_);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPEC_ROWS; ++k) {
            const int r = k * FRAME_THREADS + tid;
            if (r < n && sk[k] == L.col_key[sa[k]]) atomicMin(&L.col_row[sa[k]], r);
        }
        for (int r0 = SPEC_ROWS * FRAME_THREADS; r0 < n; r0 += FRAME_THREADS) {
            const int r = r0 + tid;
            int o_ = 0, g_ = 0, a_ = 0; unsigned long long key_ = 0;
            row_from_hbm(r, o_, g_, a_, key_);
            if (r < n && key_ == L.col_key[a_]) atomicMin(&L.col_row[a_], r);
        }
        __syncthreads();
    }
    STAMP(3);
    const bool age = (m == 0) || (n > 0 && n >= m);   // tracker.py:95-107, 198-211
    // claims, ageing and the stable compaction of the id-ordered table in one sweep: a row's fate is
    // known to the thread that ranks it, so nothing goes through LDS in between
    int used = 0;   // claims made by this wave's rows (wave-uniform)
    int n_keep = 0;
    auto sweep_chunk = [&](int chunk, int r, int o_, int g, int c) {
        bool mine = false, keep = false;
        if (r < n) {
            mine = (m > 0) && (L.col_row[c] == r);
            if (mine) g = 0;
            else if (age) ++g;
            keep = mine || !age || !((double)g > a.max_gone);
            L.cg[r] = (mine ? c + 1 : 0) | (g << 16);
        }
        used += __popcll(__ballot(mine));
        int total;
        const int ex = block_flag_rank(keep, L.scan + (chunk & 1) * 4, &total);
        if (keep) L.inv[n_keep + ex] = r;
        if (r < n && !keep && blockIdx.x == 0) {
            const int k = atomicAdd(&s_n_dead, 1);
            a.free_slots[nfree + k] = o_;   // entries above n_free are read by nobody this frame
        }
        n_keep += total;
    };
#pragma unroll
    for (int k = 0; k < SPEC_ROWS; ++k)
        if (k * FRAME_THREADS < n) sweep_chunk(k, k * FRAME_THREADS + tid, so[k], sg[k], sa[k]);     // (uniform)
    for (int r0 = SPEC_ROWS * FRAME_THREADS; r0 < n; r0 += FRAME_THREADS) {
        const int r = r0 + tid;
        int o_ = 0, g = 0, c = 0; unsigned long long key_ = 0;
        row_from_hbm(r, o_, g, c, key_);
        sweep_chunk(r0 / FRAME_THREADS, r, o_, g, c);
    }
    if (lane == 0 && used) atomicAdd(&s_n_used, used);   // one LDS atomic per wave, not per thread
    __syncthreads();
    STAMP(10);
    STAMP(11);
    int n_new = 0;
    if (m > 0 && (n == 0 || n < m)) {   // registration (tracker.py:135-137, 212-217)
        // (col_key is dead from here on: its space holds `unused` and `newcols`)
        if (n == 0) {
            for (int c = tid; c < m; c += FRAME_THREADS) L.newcols[c] = c;
            n_new = m;
        } else {
            int base = 0;
            for (int c0 = 0; c0 < m; c0 += FRAME_THREADS) {
                const int c = c0 + tid;
                const bool un = (c < m && L.col_row[c] == 0x7FFFFFFF);
                int total;
                const int ex = block_flag_rank(un, L.scan + 8 + ((c0 / FRAME_THREADS) & 1) * 4, &total);
                if (un) L.unused[base + ex] = c;
                base += total;
            }
            __syncthreads();
            STAMP(12);
            {
                // (col_row is dead once the unclaimed columns are listed: it holds the model's list)
                int cnt = cpython_order_lds<FRAME_THREADS, FRAME_TABLE>(L.unused, base, m, s_n_used, L.newcols, reinterpret_cast<uint32_t *>(L.table),
                                            reinterpret_cast<uint32_t *>(L.col_row), s_set_state);
                if (cnt < 0) { if (blockIdx.x == 0 && tid == 0) atomicOr(a.err, ERR_TRACK_CAPACITY); cnt = 0; }
                n_new = cnt;
            }
            __syncthreads();
            STAMP(13);
        }
        __syncthreads();
    }
    const int n_new_all = n_new;
    if (n_keep + n_new > cap) {
        if (blockIdx.x == 0 && tid == 0) atomicOr(a.err, ERR_TRACK_CAPACITY);
        n_new = cap - n_keep;
    }
    const int n_live = n_keep + n_new;

    STAMP(4); BSTAMP(1);
    // ---- phase B: one wave per track of the new table
    if (i < n_live) {
        int slot, gone, c;
        bool fresh_track = false;
        if (i < n_keep) {
            const int r = __builtin_amdgcn_readfirstlane(L.inv[i]);
            const int cg = __builtin_amdgcn_readfirstlane(L.cg[r]);
            gone = cg >> 16;
            c = (cg & 0xFFFF) - 1;
            slot = slot_s;
            if (r != i || slot_s < 0) {   // a deregistration ahead of this row moved it: fetch for real
                slot = __builtin_amdgcn_readfirstlane(a.order[r]);
                if (a.use_gsff) gsff_fetch(a, slot, lane, S);
                zs0 = a.pos[slot]; zs1 = a.pos[cap + slot];
                id_s = a.id[slot];
#pragma unroll
                for (int k = 0; k < 3; ++k) in_s[k] = a.info[k * cap + slot];
                if (c >= 0) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) dd[k] = det[(size_t)c * 5 + k];
                }
            }
            if (c < 0) {
                if (age && lane < 3) a.info[lane * cap + slot] = 0.f;
            }
        } else {
            const int j = i - n_keep;
            c = __builtin_amdgcn_readfirstlane(L.newcols[j]);
            slot = __builtin_amdgcn_readfirstlane(a.free_slots[nfree - 1 - j]);
            gone = 0;
            fresh_track = true;
            id_s = id0 + j;
#pragma unroll
            for (int k = 0; k < 5; ++k) dd[k] = det[(size_t)c * 5 + k];
            gsff_blank(S);    // nothing of the slot's previous owner is read
            gsff_point(a, slot, S);
            if (lane == 0) a.id[slot] = id_s;
        }
        double z0 = zs0, z1 = zs1;
        if (c >= 0) {
            z0 = (double)dd[0];
            z1 = (double)dd[1];
            const DetT v = lane == 0 ? dd[2] : lane == 1 ? dd[3] : dd[4];
            if (lane < 3) a.info[lane * cap + slot] = (float)v;
        }
        double o0 = z0, o1 = z1, p0 = z0, p1 = z1;
        STAMP(5);
        if (a.use_gsff) gsff_wave(a, a.gains, slot, lane, z0, z1, fresh_track, S, G, o0, o1, p0, p1);
        else if (lane == 0) { a.pos[slot] = z0; a.pos[cap + slot] = z1; }
        STAMP(6);
        if (lane == 0) {
            b.order[i] = slot;
            b.gone[i] = gone;
            if (rows && row_base + i < rows_capacity) {
                ysmr_row rr;
                rr.frame = frame;
                rr.track_id = id_s;
                rr.x = o0; rr.y = o1;
                if (c >= 0) {
                    rr.w = (float)dd[2]; rr.h = (float)dd[3]; rr.angle = (float)dd[4];
                } else if (age) {
                    rr.w = rr.h = rr.angle = 0.f;
                } else {
                    rr.w = in_s[0]; rr.h = in_s[1]; rr.angle = in_s[2];
                }
                rr.disappeared = gone;
                rows[row_base + i] = rr;
            }
        }
        STAMP(7);
        if (m_next > 0) rowmin_wave(b, i, p0, p1, next_det, m_next, lane, first);
        STAMP(8);
    }
    BSTAMP(2);
#ifdef YSMR_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) RING((3ull << 40) | (unsigned)frame);
#endif
    // ---- block 0 publishes the counters of the new state
    if (blockIdx.x == 0) {
        if (claim_out)
            for (int r = tid; r < n; r += FRAME_THREADS) claim_out[r] = (L.cg[r] & 0xFFFF) - 1;
        if (new_cols_out)
            for (int j = tid; j < n_new; j += FRAME_THREADS) new_cols_out[j] = L.newcols[j];
        __syncthreads();
        if (tid == 0) {
            *b.n_tracks = n_live;
            *b.next_id = id0 + n_new_all;   // ids are consumed even for registrations dropped on overflow
            *b.n_free = nfree - n_new + s_n_dead;
            b.row_base[0] = row_base + n_live;
            if (rows && row_base + n_live > rows_capacity) atomicOr(a.err, ERR_ROWS_CAPACITY);
            if (row_count_ext) *row_count_ext = row_base + n_live;
            if (n_rows_out) *n_rows_out = n_live;
            if (n_before_out) *n_before_out = n;
            if (n_new_out) *n_new_out = n_new;
        }
    }
}

#include "batch_link.h"

__global__ void k_tracker_reset(TrackerDev t)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < t.capacity) t.free_slots[i] = t.capacity - 1 - i;  // slot 0 is handed out first
    if (i == 0) { *t.n_tracks = 0; *t.next_id = 0; *t.err = 0; *t.n_free = t.capacity; }
}

int horizons(double fps, int n_min, double n_max, int n_f, int *n_i)
{
    // generate_n_i (gsff.py:87-109); CentroidTracker passes n_max = fps when it is None (tracker.py:58-59)
    if (n_f < 1 || n_f > YSMR_MAX_FILTERS)
        return ysmr::fail(YSMR_ERR_ARG, "number of LSFFs must be in 1..%d, got %d", YSMR_MAX_FILTERS, n_f);
    if (!(fps > 0)) return ysmr::fail(YSMR_ERR_ARG, "fps must be positive");
    double top = n_max > 0 ? n_max : fps;
    double p = (top - n_min) / n_f;
    for (int i = 1; i <= n_f; ++i) n_i[i - 1] = (int)(n_min + p * i);
    for (int i = 0; i < n_f; ++i)
        if (n_i[i] < 1 || (i && n_i[i] <= n_i[i - 1]))
            return ysmr::fail(YSMR_ERR_ARG, "filter horizons must be >= 1 and strictly increasing (n_min=%d n_max=%g n_f=%d)",
                              n_min, top, n_f);
    return YSMR_OK;
}

void closed_form_gain(int N, double *g)
{
    // rows 0/1 of (L^T L)^-1 L^T for the constant-velocity model: the one-step-ahead
    // least-squares line fit, c_j = 1/N + t_j * ((N+1)/2) / sum(t^2), t_j = j - (N-1)/2
    double st2 = 0.0;
    for (int j = 0; j < N; ++j) { double tj = j - (N - 1) / 2.0; st2 += tj * tj; }
    std::memset(g, 0, sizeof(double) * 4 * N);
    for (int j = 0; j < N; ++j) {
        double tj = j - (N - 1) / 2.0;
        double c = 1.0 / N + (st2 > 0 ? tj * ((N + 1) / 2.0) / st2 : 0.0);
        g[2 * j] = c;              // row 0, x columns
        g[2 * N + 2 * j + 1] = c;  // row 1, y columns
    }
}

template <typename DetT>
int launch_update_t(ysmr_tracker *t, hipStream_t st, const DetT *det, int m, const int32_t *m_dev, int frame,
                    ysmr_row *rows, long long rows_capacity, long long *row_count, int32_t *n_rows, int32_t *claim,
                    int32_t *n_before, int32_t *new_cols, int32_t *n_new, bool rowmin_done, const DetT *next_det,
                    const int32_t *next_m_dev, DetGrid grid = DetGrid{nullptr, nullptr, nullptr, nullptr},
                    DetGrid next_grid = DetGrid{nullptr, nullptr, nullptr, nullptr}, bool base_from_ext = false)
{
    const DetGrid no_grid{nullptr, nullptr, nullptr, nullptr};
    const dim3 wgrid((t->d.capacity + 3) / 4);
    if (t->fused) {
        const TrackerDev &a = t->cur(), &b = t->nxt();
        if (!rowmin_done) {
            hipLaunchKernelGGL(k_rowmin<DetT>, wgrid, dim3(256), 0, st, a, det, m, m_dev, t->set_base ? 1 : 0, t->base_ptr, no_grid);
            t->set_base = false;
        }
        // the filter bank is unrolled at compile time: 3 covers tracking.ini's default (and 1, 2), 8 the rest
        if (a.n_f <= 3)
            hipLaunchKernelGGL((k_frame<DetT, 3>), wgrid, dim3(FRAME_THREADS), t->frame_lds, st, a, b, det, m, m_dev, frame,
                               rows, rows_capacity, row_count, n_rows, claim, n_before, new_cols, n_new, next_det, next_m_dev,
                               base_from_ext ? 1 : 0);
        else
            hipLaunchKernelGGL((k_frame<DetT, YSMR_MAX_FILTERS>), wgrid, dim3(FRAME_THREADS), t->frame_lds, st, a, b, det, m,
                               m_dev, frame, rows, rows_capacity, row_count, n_rows, claim, n_before, new_cols, n_new,
                               next_det, next_m_dev, base_from_ext ? 1 : 0);
        t->par ^= 1;
    } else {
        const TrackerDev &d = t->d;
        if (!rowmin_done) hipLaunchKernelGGL(k_rowmin<DetT>, wgrid, dim3(256), 0, st, d, det, m, m_dev, 0, nullptr, grid);
        const size_t link_lds = 12 * (size_t)d.max_det;
#ifdef YSMR_TUNING
        static const bool hbm_tables = getenv("YSMR_LINK_TABLES") && !strcmp(getenv("YSMR_LINK_TABLES"), "hbm");
#else
        const bool hbm_tables = false;
#endif
        if (link_lds <= 140 * 1024 && !hbm_tables)
            hipLaunchKernelGGL((k_link<DetT, true>), dim3(1), dim3(LINK_THREADS), link_lds, st, d, det, m, m_dev, frame, rows,
                               rows_capacity, row_count, n_rows, claim, n_before, new_cols, n_new);
        else
            hipLaunchKernelGGL((k_link<DetT, false>), dim3(1), dim3(LINK_THREADS), 0, st, d, det, m, m_dev, frame, rows,
                               rows_capacity, row_count, n_rows, claim, n_before, new_cols, n_new);
        if (t->lanes) {
            hipLaunchKernelGGL(k_track_lanes<DetT>, dim3((d.capacity + TL_THREADS - 1) / TL_THREADS), dim3(TL_THREADS), 0, st, d, t->bd,
                               t->bgains_dev, t->lanes_head, frame, rows, rows_capacity, det, next_det, next_m_dev, next_grid);
            t->lanes_head = (t->lanes_head + 1) & (BL_HB - 1);
        } else if (d.n_f <= 3)
            hipLaunchKernelGGL((k_track<DetT, 3>), wgrid, dim3(256), 0, st, d, frame, rows, rows_capacity, det, next_det, -1,
                               next_m_dev, next_grid);
        else
            hipLaunchKernelGGL((k_track<DetT, YSMR_MAX_FILTERS>), wgrid, dim3(256), 0, st, d, frame, rows, rows_capacity, det,
                               next_det, -1, next_m_dev, next_grid);
    }
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

__global__ void k_peek(TrackerDev t, int32_t *ids, double *xy, int32_t *gone, int32_t *n_out)
{
    const int n = *t.n_tracks;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && n_out) *n_out = n;
    if (i >= n) return;
    int slot = t.order[i];
    if (ids) ids[i] = t.id[slot];
    if (xy) { xy[2 * i] = t.pos[slot]; xy[2 * i + 1] = t.pos[t.capacity + slot]; }
    if (gone) gone[i] = t.gone[t.gone_by_row ? i : slot];
}

}  // namespace

// One wave that waits `ticks` x 10 ns (the 100 MHz realtime counter), in front of every launch of the batch link.  In the pipeline the
// event behind a link launch releases two things at the same instant: the NEXT link launch (same stream) and the detection of the
// batch whose buffers that launch was reading -- the threshold kernel's 248 workgroups, one per free compute unit, 31 per XCD.  A link
// workgroup that takes its seat while those 248 are being placed, on an XCD where the previous launch's unit is not free yet, leaves
// that XCD one unit short: one threshold workgroup then waits for another to finish its frame, and the launch takes 372-379 us instead
// of 203 -- none in most runs, every eighth launch in some (profiles/r05_thr_launch_outliers.log: always at a link boundary; with a
// 3 us wait here EVERY launch, with 6 us one run in eight, with 10 us none).  The wait lets the 248 sit down first; the link's
// workgroup then finds its XCD's spare unit.  12 us per launch of up to 256 frames (0.8 % of the link's time at the bench's size).
#ifndef BL_PAUSE_TICKS_N
#define BL_PAUSE_TICKS_N 1200
#endif
constexpr int BL_PAUSE_TICKS = BL_PAUSE_TICKS_N;
__global__ __launch_bounds__(64) void k_pause(int ticks)
{
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    do {
        __builtin_amdgcn_s_sleep(8);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    } while (t - t0 < (unsigned long long)ticks);
}

extern "C" {

int ysmr_gsff_gains(double fps, int n_min, double n_max, int n_f, int32_t *n_i_out, double *gains_out)
{
    int n_i[YSMR_MAX_FILTERS];
    if (int rc = horizons(fps, n_min, n_max, n_f, n_i)) return rc;
    if (!n_i_out) return ysmr::fail(YSMR_ERR_ARG, "n_i_out must not be NULL");
    size_t off = 0;
    for (int i = 0; i < n_f; ++i) {
        n_i_out[i] = n_i[i];
        if (gains_out) closed_form_gain(n_i[i], gains_out + off);
        off += 4 * (size_t)n_i[i];
    }
    return YSMR_OK;
}

int ysmr_tracker_create(double max_disappeared, double fps, int n_min, double n_max, int n_f, int use_gsff,
                        int capacity, int max_det, const double *gains_host, ysmr_tracker **out)
{
    if (!out) return ysmr::fail(YSMR_ERR_ARG, "out must not be NULL");
    *out = nullptr;
    // (the tables of the two-launch path live in HBM; the bound only keeps 32-bit indices and the state block sane)
    if (capacity <= 0 || max_det <= 0 || capacity > 65536 || max_det > 65536)
        return ysmr::fail(YSMR_ERR_ARG, "capacity/max_det out of range: need 1 .. 65536, got %d/%d", capacity, max_det);
    ysmr_tracker *t = new ysmr_tracker();
    TrackerDev &d = t->d;
    std::memset(&d, 0, sizeof(d));
    d.capacity = capacity;
    d.max_det = max_det;
    d.use_gsff = use_gsff ? 1 : 0;
    d.max_gone = max_disappeared;
    d.lik_min = 1e-20;  // tracker.py:67
    d.n_f = use_gsff ? n_f : 1;
    size_t gain_doubles = 0;
    if (use_gsff) {
        if (int rc = horizons(fps, n_min, n_max, n_f, d.n_i)) { delete t; return rc; }
        for (int i = 0; i < n_f; ++i) { d.gain_off[i] = (int)gain_doubles; gain_doubles += 4 * (size_t)d.n_i[i]; }
        t->gains_host.resize(gain_doubles);
        if (gains_host) std::memcpy(t->gains_host.data(), gains_host, sizeof(double) * gain_doubles);
        else for (int i = 0; i < n_f; ++i) closed_form_gain(d.n_i[i], t->gains_host.data() + d.gain_off[i]);
        d.gains_decoupled = 1;   // row 0 must vanish on the y columns, row 1 on the x columns
        for (int i = 0; i < n_f && d.gains_decoupled; ++i) {
            const double *g = t->gains_host.data() + d.gain_off[i];
            for (int j = 0; j < d.n_i[i]; ++j)
                if (g[2 * j + 1] != 0.0 || g[2 * d.n_i[i] + 2 * j] != 0.0) { d.gains_decoupled = 0; break; }
        }
        d.hist_cap = d.n_i[n_f - 1] + 1;
        if (d.hist_cap > 64) { delete t; return ysmr::fail(YSMR_ERR_ARG, "maximum horizon size %d exceeds the supported 63", d.n_i[n_f - 1]); }
    } else {
        d.n_i[0] = 1;
        d.hist_cap = 1;
    }
    unsigned tc = 8;
    while (tc <= (unsigned)max_det * 4u) tc <<= 1;
    d.table_cap = (int)tc;
    d.gain_total = (int)gain_doubles;

    const size_t cap = capacity, nf = d.n_f;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = ysmr::align_up(off + bytes, 256); return o; };
    const size_t o_scal = take(sizeof(int) * 16);
    const size_t o_order = take(sizeof(int) * cap), o_free = take(sizeof(int) * cap);
    const size_t o_id = take(sizeof(int) * cap), o_gone = take(sizeof(int) * cap);
    const size_t o_pos = take(sizeof(double) * 2 * cap), o_info = take(sizeof(float) * 3 * cap);
    const size_t o_hist = take(sizeof(double) * 2 * cap * d.hist_cap);
    d.rec_stride = (2 + 3 * nf + 7) / 8 * 8;
    const size_t o_rec = take(sizeof(double) * (size_t)d.rec_stride * cap);
    const size_t o_gain = take(sizeof(double) * (gain_doubles ? gain_doubles : 1));
    const size_t o_unused = take(sizeof(int) * max_det);
    const size_t o_rmin = take(sizeof(double) * cap), o_rarg = take(sizeof(int) * cap), o_rgone = take(sizeof(int) * cap);
    const size_t o_dead = take(sizeof(int) * cap);
    const size_t o_new = take(sizeof(int) * max_det), o_table = take(sizeof(int) * 2 * (size_t)d.table_cap);
    const size_t o_lkey = take(sizeof(unsigned long long) * max_det), o_lrow = take(sizeof(int) * max_det);
    const size_t o_lclaim = take(sizeof(int) * cap), o_cslot = take(sizeof(int) * cap), o_crow = take(sizeof(int) * cap);
    // parity-1 copies of the arrays k_frame double-buffers
    const size_t o_scal1 = take(sizeof(int) * 16), o_order1 = take(sizeof(int) * cap), o_gone1 = take(sizeof(int) * cap);
    const size_t o_rmin1 = take(sizeof(double) * cap), o_rarg1 = take(sizeof(int) * cap);
    // the batch link's rest format (seat-major) and its gain table
    const size_t seat_cap = capacity > BL_THREADS ? (size_t)capacity : (size_t)BL_THREADS;     // (k_batch: 768 seats; k_track_lanes: a seat per slot)
    const size_t o_ring = take(sizeof(double2) * BL_HB * seat_cap);
    const size_t o_b64 = take(sizeof(double) * BL_SF64 * seat_cap), o_b32 = take(sizeof(float) * 3 * seat_cap);
    const size_t o_bi = take(sizeof(int) * 6 * seat_cap);
    t->block_bytes = off;
    hipError_t e = hipMalloc(&t->block, off);
    if (e != hipSuccess) {
        delete t;
        return ysmr::fail(YSMR_ERR_HIP, "hipMalloc(%zu) failed: %s", off, hipGetErrorString(e));
    }
    char *b = (char *)t->block;
    int *scal = (int *)(b + o_scal);
    d.n_tracks = scal; d.next_id = scal + 1; d.err = scal + 2; d.n_free = scal + 3;
    d.row_base = (long long *)(scal + 8);
    d.order = (int *)(b + o_order); d.free_slots = (int *)(b + o_free);
    d.id = (int *)(b + o_id); d.gone = (int *)(b + o_gone);
    d.pos = (double *)(b + o_pos); d.info = (float *)(b + o_info); d.hist = (double *)(b + o_hist);
    d.rec = (double *)(b + o_rec);
    d.gains = (const double *)(b + o_gain);
    d.unused = (int *)(b + o_unused);
    d.row_min = (double *)(b + o_rmin); d.row_arg = (int *)(b + o_rarg); d.row_gone = (int *)(b + o_rgone);
    d.dead = (int *)(b + o_dead);
    d.new_cols = (int *)(b + o_new); d.set_table = (int *)(b + o_table);
    d.link_key = (unsigned long long *)(b + o_lkey); d.link_row = (int *)(b + o_lrow); d.link_claim = (int *)(b + o_lclaim);
    d.claim_slot = (int *)(b + o_cslot);
    d.claim_row = (int *)(b + o_crow);
    t->bd.ring = (double2 *)(b + o_ring);
    t->bd.f64 = (double *)(b + o_b64); t->bd.f32 = (float *)(b + o_b32); t->bd.i32 = (int *)(b + o_bi);
    t->bd.head = scal + 12;
    t->bd.seat_cap = (int)seat_cap;
    t->bd.grid = nullptr; t->bd.grid_stride = 0;
    t->d1 = d;
    {
        TrackerDev &q = t->d1;
        int *scal1 = (int *)(b + o_scal1);
        q.n_tracks = scal1; q.next_id = scal1 + 1; q.n_free = scal1 + 3;   // err stays shared (sticky)
        q.row_base = (long long *)(scal1 + 8);
        q.order = (int *)(b + o_order1); q.gone = (int *)(b + o_gone1);
        q.row_min = (double *)(b + o_rmin1); q.row_arg = (int *)(b + o_rarg1);
    }
    t->par = 0;
    t->set_base = false;
    t->base_ptr = nullptr;
    t->frame_lds = frame_lds_bytes(capacity, max_det, (int)gain_doubles);
#ifdef YSMR_TUNING
    const char *mode_env = getenv("YSMR_LINK_MODE");   // "split" forces the two-kernel path (tuning builds only)
#else
    const char *mode_env = nullptr;
#endif
    // (max_det <= 2456: the LDS set model holds that many unregistered columns; the split path keeps its
    // tables in HBM and has no such limit)
    // (and a `gone` counter that fits the 15 bits k_frame packs it into)
    t->fused = t->frame_lds <= 150 * 1024 && max_det <= 2456 && max_disappeared < 32000.0 &&
               !(mode_env && !strcmp(mode_env, "split"));
    if (t->fused) {
        const void *variants[4] = {(const void *)k_frame<float, 3>, (const void *)k_frame<double, 3>,
                                   (const void *)k_frame<float, YSMR_MAX_FILTERS>, (const void *)k_frame<double, YSMR_MAX_FILTERS>};
        for (const void *fn : variants)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->frame_lds) != hipSuccess) t->fused = false;
    }
    t->d.gone_by_row = t->d1.gone_by_row = t->fused ? 1 : 0;
    // The batch link serves what tracking.ini's defaults ask for: up to three filters with horizons of at most 31 frames
    // (hist_cap <= 32 ring entries), gains affine in a measurement's age, tables of at most 1024 tracks -- a track per lane of ONE
    // workgroup -- and detection counts whose tables fit its LDS.  Everything else links with one (or two) launches per
    // frame as before.
    t->batch_lds = bl_lds_bytes(max_det);
    t->batchable = t->fused && d.n_f <= BL_NF && d.hist_cap <= BL_HB && (!use_gsff || d.gains_decoupled) &&
                   capacity <= BL_THREADS && t->batch_lds <= 140 * 1024 && !(mode_env && strcmp(mode_env, "batch"));
    if (t->batchable &&
        hipFuncSetAttribute((const void *)k_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->batch_lds) != hipSuccess)
        t->batchable = false;
    // ... and the two-launch link of larger tables runs its per-track half with a track per LANE (k_track_lanes) under the same
    // conditions on the filter bank (tuning builds: YSMR_LINK_MODE=waves keeps k_track, a wave per track, for A/B runs)
    t->lanes = !t->fused && d.n_f <= BL_NF && d.hist_cap <= BL_HB && (!use_gsff || d.gains_decoupled) &&
               !(mode_env && !strcmp(mode_env, "waves"));
    // the gains as the batch kernel takes them: affine in the age of a measurement (batch_link.h: bl_fir)
    std::memset(&t->bgains, 0, sizeof(t->bgains));
    if (use_gsff && (t->batchable || t->lanes)) {
        bool affine = true;
        for (int f = 0; f < d.n_f && affine; ++f) {
            const int N = d.n_i[f];
            const double *g = t->gains_host.data() + d.gain_off[f];
            for (int c = 0; c < 2 && affine; ++c) {
                auto gain = [&](int a) { return c ? g[2 * N + 2 * (N - 1 - a) + 1] : g[2 * (N - 1 - a)]; };   // age a, row c
                const double alpha = gain(0), beta = N > 1 ? gain(0) - gain(1) : 0.0;
                for (int a = 0; a < N; ++a)
                    if (std::fabs(gain(a) - (alpha - beta * a)) > 1e-14 * std::fabs(alpha)) affine = false;
                t->bgains.alpha[f][c] = alpha;
                t->bgains.beta[f][c] = beta;
            }
        }
        if (!affine) t->batchable = t->lanes = false;
    }
    // the per-frame grids of a batch: sized here, once, for BL_MAX_BATCH frames (longer batches are cut to that)
    {
        const size_t per = t->batchable ? (size_t)bl_grid_dwords_max(max_det) * 4 : ysmr_tracker::grid_bytes_per_frame(max_det);
        const size_t blocks = t->batchable ? 3 : 1;      // (two for ysmr_tracker_prepare's callers, one for ysmr_tracker_run itself)
        if (t->batchable || !t->fused) {
            e = hipMalloc(&t->grid_block, per * BL_MAX_BATCH * blocks + 256);
            if (e == hipSuccess && (t->batchable || t->lanes)) {
                t->bgains_dev = reinterpret_cast<const BlGains *>((char *)t->grid_block + per * BL_MAX_BATCH * blocks);
                e = hipMemcpy((void *)t->bgains_dev, &t->bgains, sizeof(BlGains), hipMemcpyHostToDevice);
            }
            if (e != hipSuccess) {
                (void)hipFree(t->block);
                delete t;
                return ysmr::fail(YSMR_ERR_HIP, "hipMalloc(%zu) failed: %s", per * BL_MAX_BATCH * blocks, hipGetErrorString(e));
            }
            t->grid_frames = BL_MAX_BATCH;
            t->bd.grid = (char *)t->grid_block;
            t->bd.grid_stride = (unsigned)per;
        }
    }
    e = hipMemset(t->block, 0, off);
    if (e == hipSuccess && t->lanes) e = hipMemset(t->bd.i32, 0xFF, sizeof(int) * seat_cap);     // no slot's filter state belongs to a track yet
    if (e == hipSuccess && gain_doubles)
        e = hipMemcpy(b + o_gain, t->gains_host.data(), sizeof(double) * gain_doubles, hipMemcpyHostToDevice);

    if (e != hipSuccess) {
        (void)hipFree(t->block);
        if (t->grid_block) (void)hipFree(t->grid_block);
        delete t;
        return ysmr::fail(YSMR_ERR_HIP, "tracker state initialisation failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return ysmr_tracker_reset(t, nullptr);
}

int ysmr_tracker_reset(ysmr_tracker *t, void *stream)
{
    if (!t) return ysmr::fail(YSMR_ERR_ARG, "tracker handle is NULL");
    int n = t->d.capacity > t->d.max_det ? t->d.capacity : t->d.max_det;
    t->par = 0;
    t->rowmin_for = nullptr;
    t->prepared[0] = t->prepared[1] = ysmr_tracker::Prepared();
    if (t->lanes) {                    // (ids start over: a slot's old filter state must not pass for the new track 0's)
        YSMR_HIP_CHECK(hipMemsetAsync(t->bd.i32, 0xFF, sizeof(int) * (size_t)t->bd.seat_cap, (hipStream_t)stream));
        t->lanes_head = 0;
    }
    t->in_batch = t->use_batch();      // (an empty table is the same in both layouts)
    hipLaunchKernelGGL(k_tracker_reset, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->d);
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

int ysmr_tracker_destroy(ysmr_tracker *t)
{
    if (!t) return YSMR_OK;
    hipError_t e = hipFree(t->block);
    if (t->grid_block) (void)hipFree(t->grid_block);
    delete t;
    if (e != hipSuccess) return ysmr::fail(YSMR_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
    return YSMR_OK;
}

// where the state rests: the seat-major arrays of the batch link, or the per-slot layout of the per-frame kernels
static int state_to_std(ysmr_tracker *t, hipStream_t st)
{
    // a binned block is matched by the addresses of its detections alone: once a frame is linked by the per-frame kernels
    // (this function is in front of every such call) the caller may refill that buffer without preparing it again, and a
    // later batch run must not take the old binning for the new contents
    t->prepared[0] = t->prepared[1] = ysmr_tracker::Prepared();
    if (!t->in_batch) return YSMR_OK;
    hipLaunchKernelGGL(k_to_std, dim3((std::max(t->d.capacity, t->bd.seat_cap) + 255) / 256), dim3(256), 0, st, t->d, t->bd);
    YSMR_LAUNCH_CHECK();
    t->in_batch = false;
    t->par = 0;
    t->rowmin_for = nullptr;
    return YSMR_OK;
}
static int state_to_batch(ysmr_tracker *t, hipStream_t st)
{
    if (t->in_batch) return YSMR_OK;
    hipLaunchKernelGGL(k_to_batch, dim3((t->bd.seat_cap + 255) / 256), dim3(256), 0, st, t->cur(), t->bd);
    YSMR_LAUNCH_CHECK();
    if (t->par) {      // the counters live in the parity-0 words from here on
        YSMR_HIP_CHECK(hipMemcpyAsync(t->d.n_tracks, t->d1.n_tracks, sizeof(int) * 2, hipMemcpyDeviceToDevice, st));
    }
    t->in_batch = true;
    t->par = 0;
    t->rowmin_for = nullptr;
    return YSMR_OK;
}

int ysmr_tracker_link_mode(ysmr_tracker *t, int mode)
{
    if (!t) return ysmr::fail(YSMR_ERR_ARG, "tracker handle is NULL");
    if (mode != 0 && mode != 1) return ysmr::fail(YSMR_ERR_ARG, "link mode must be 0 (the library's choice) or 1 (per-frame launches)");
    if (t->link_mode != mode) t->prepared[0] = t->prepared[1] = ysmr_tracker::Prepared();
    t->link_mode = mode;       // (the state changes its layout at the next call that needs the other one)
    return YSMR_OK;
}

#ifdef YSMR_STAMPS
int ysmr_debug_read_bstamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamps), sizeof(unsigned long long) * BL_WAVES * 16); }
#endif

int ysmr_tracker_batched(ysmr_tracker *t) { return t && t->use_batch() ? 1 : 0; }

int ysmr_tracker_prepare(ysmr_tracker *t, void *stream, const float *det_dev, const int32_t *det_count_dev, int batch, int slot)
{
    if (!t) return ysmr::fail(YSMR_ERR_ARG, "tracker handle is NULL");
    if (!det_dev || !det_count_dev || batch <= 0 || slot < 0 || slot > 1)
        return ysmr::fail(YSMR_ERR_ARG, "det_dev, det_count_dev must be set, batch > 0, slot 0 or 1");
    if (!t->use_batch() || batch > BL_MAX_BATCH) return YSMR_OK;       // (nothing to prepare: ysmr_tracker_run does it all)
    hipLaunchKernelGGL(k_bgrid, dim3(batch), dim3(BG_THREADS), t->bd.grid_stride, (hipStream_t)stream, det_dev, det_count_dev,
                       t->d.max_det, t->bd.grid + (size_t)slot * BL_MAX_BATCH * t->bd.grid_stride, t->bd.grid_stride);
    YSMR_LAUNCH_CHECK();
    t->prepared[slot].det = det_dev; t->prepared[slot].count = det_count_dev; t->prepared[slot].batch = batch;
    return YSMR_OK;
}

int ysmr_tracker_update(ysmr_tracker *t, void *stream, const void *det_dev, int det_is_f64, int m,
                        const int32_t *m_dev, int32_t frame_index, ysmr_row *rows_dev, int32_t *n_rows_dev,
                        int32_t *claim_col_dev, int32_t *n_before_dev, int32_t *new_cols_dev, int32_t *n_new_dev)
{
    if (!t) return ysmr::fail(YSMR_ERR_ARG, "tracker handle is NULL");
    if (m < 0 && !m_dev) return ysmr::fail(YSMR_ERR_ARG, "m < 0 requires m_dev");
    if (m > t->d.max_det) return ysmr::fail(YSMR_ERR_CAPACITY, "m = %d exceeds max_det = %d", m, t->d.max_det);
    if (!det_dev && m != 0) return ysmr::fail(YSMR_ERR_ARG, "det_dev is NULL");
    if (int rc = state_to_std(t, (hipStream_t)stream)) return rc;     // (one frame at a time: the per-frame kernels)
    t->rowmin_for = nullptr;
    if (t->fused) { t->set_base = true; t->base_ptr = nullptr; }
    if (det_is_f64)
        return launch_update_t<double>(t, (hipStream_t)stream, (const double *)det_dev, m, m_dev, frame_index, rows_dev,
                                       t->d.capacity, nullptr, n_rows_dev, claim_col_dev, n_before_dev, new_cols_dev,
                                       n_new_dev, false, nullptr, nullptr);
    return launch_update_t<float>(t, (hipStream_t)stream, (const float *)det_dev, m, m_dev, frame_index, rows_dev,
                                  t->d.capacity, nullptr, n_rows_dev, claim_col_dev, n_before_dev, new_cols_dev,
                                  n_new_dev, false, nullptr, nullptr);
}

int ysmr_tracker_run(ysmr_tracker *t, void *stream, const float *det_dev, const int32_t *det_count_dev, int batch,
                     int32_t first_frame_index, ysmr_row *rows_dev, int64_t rows_capacity, int64_t *row_count_dev)
{
    if (!t) return ysmr::fail(YSMR_ERR_ARG, "tracker handle is NULL");
    if (!det_dev || !det_count_dev || !rows_dev || !row_count_dev || batch <= 0)
        return ysmr::fail(YSMR_ERR_ARG, "det_dev, det_count_dev, rows_dev, row_count_dev must be set and batch > 0");
    // (ysmr_tracker_run_chained, which also took the NEXT call's first frame, left the ABI with version 12: its look-ahead
    // had been ignored since version 10 -- a launch that had the next call's first row minima ready had to take its first
    // output row from *row_count_dev, the word its own first workgroup advances, ADVICE r03 -- and a batch-link handle
    // links a batch with ONE launch, so there is nothing to chain.)
    const float *after_det_dev = nullptr;
    const int32_t *after_count_dev = nullptr;
    if (t->use_batch()) {
        // one launch links the batch (cut to the BL_MAX_BATCH frames the grid block holds); `after` has nothing to save here
        if (int rc = state_to_batch(t, (hipStream_t)stream)) return rc;
        for (int f0 = 0; f0 < batch; f0 += BL_MAX_BATCH) {
            const int nb = batch - f0 < BL_MAX_BATCH ? batch - f0 : BL_MAX_BATCH;
            const float *det = det_dev + (size_t)f0 * t->d.max_det * 5;
            // the batch's detections binned frame by frame: by ysmr_tracker_prepare ahead of this call, or here
            int block = 2;
            for (int s = 0; s < 2; ++s)
                if (f0 == 0 && batch <= BL_MAX_BATCH && t->prepared[s].det == (const void *)det_dev &&
                    t->prepared[s].count == (const void *)det_count_dev && t->prepared[s].batch == batch) {
                    block = s;
                    t->prepared[s] = ysmr_tracker::Prepared();       // (good for one call)
                }
            BatchDev bd = t->bd;
            bd.grid = t->bd.grid + (size_t)block * BL_MAX_BATCH * t->bd.grid_stride;
            if (block == 2)
                hipLaunchKernelGGL(k_bgrid, dim3(nb), dim3(BG_THREADS), t->bd.grid_stride, (hipStream_t)stream, det,
                                   det_count_dev + f0, t->d.max_det, bd.grid, bd.grid_stride);
            BlKernArgs ka;
            ka.t = t->d; ka.bd = bd; ka.det_all = det; ka.det_count = det_count_dev + f0; ka.batch = nb;
            ka.frame0 = first_frame_index + f0; ka.rows = rows_dev; ka.rows_capacity = (long long)rows_capacity;
            ka.row_count = (long long *)row_count_dev; ka.gains = t->bgains_dev;
            if (BL_PAUSE_TICKS > 0) hipLaunchKernelGGL(k_pause, dim3(1), dim3(64), 0, (hipStream_t)stream, BL_PAUSE_TICKS);
            hipLaunchKernelGGL(k_batch, dim3(1), dim3(BL_THREADS), t->batch_lds, (hipStream_t)stream, ka);
            YSMR_LAUNCH_CHECK();
        }
        return YSMR_OK;
    }
    if (int rc = state_to_std(t, (hipStream_t)stream)) return rc;
    // the previous call may have left this call's first row minima behind (it was told this frame comes next)
    const bool have_rowmin = t->fused && t->rowmin_for != nullptr && t->rowmin_for == (const void *)det_dev;
    t->rowmin_for = nullptr;
    if (t->fused) { t->set_base = !have_rowmin; t->base_ptr = (const long long *)row_count_dev; }
    const DetGrid no_grid{nullptr, nullptr, nullptr, nullptr};
    bool grids = false;
    if (!t->fused) {
        // large tables: a uniform grid over every frame's detections, built for the whole batch in one launch
        // (the detections of a batch are all there before the first frame is linked)
        const size_t per = ysmr_tracker::grid_bytes_per_frame(t->d.max_det);
        if (batch > t->grid_frames) {      // (the grid block was sized by ysmr_tracker_create: longer batches go in pieces)
            for (int f0 = 0; f0 < batch; f0 += t->grid_frames) {
                const int nb = batch - f0 < t->grid_frames ? batch - f0 : t->grid_frames;
                if (int rc = ysmr_tracker_run(t, stream, det_dev + (size_t)f0 * t->d.max_det * 5, det_count_dev + f0, nb,
                                              first_frame_index + f0, rows_dev, rows_capacity, row_count_dev))
                    return rc;
            }
            return YSMR_OK;
        }
        hipLaunchKernelGGL(k_grid_build<float>, dim3(batch), dim3(1024), 0, (hipStream_t)stream, det_dev, det_count_dev,
                           t->d.max_det, (char *)t->grid_block, per);
        YSMR_LAUNCH_CHECK();
        grids = true;
    }
    for (int f = 0; f < batch; ++f) {
        const bool inside = f + 1 < batch, has_next = inside || after_det_dev != nullptr;
        const float *next_det = inside ? det_dev + (size_t)(f + 1) * t->d.max_det * 5 : after_det_dev;
        const int32_t *next_m = inside ? det_count_dev + f + 1 : after_count_dev;
        int rc = launch_update_t<float>(t, (hipStream_t)stream, det_dev + (size_t)f * t->d.max_det * 5, -1,
                                        det_count_dev + f, first_frame_index + f, rows_dev, (long long)rows_capacity,
                                        (long long *)row_count_dev, nullptr, nullptr, nullptr, nullptr, nullptr,
                                        f > 0 || have_rowmin, has_next ? next_det : nullptr, has_next ? next_m : nullptr,
                                        grids ? t->grid(f) : no_grid, grids && inside ? t->grid(f + 1) : no_grid,
                                        f == 0 && have_rowmin);
        if (rc) return rc;
    }
    if (after_det_dev) t->rowmin_for = (const void *)after_det_dev;
    return YSMR_OK;
}

int ysmr_tracker_fused(ysmr_tracker *t) { return t && t->fused && !t->use_batch() ? 1 : 0; }

int ysmr_tracker_peek(ysmr_tracker *t, void *stream, int32_t *ids_dev, double *xy_dev, int32_t *disappeared_dev,
                      int32_t *n_dev)
{
    if (!t) return ysmr::fail(YSMR_ERR_ARG, "tracker handle is NULL");
    if (t->in_batch)
        hipLaunchKernelGGL(k_peek_batch, dim3((t->bd.seat_cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->d, t->bd,
                           ids_dev, xy_dev, disappeared_dev, n_dev);
    else
        hipLaunchKernelGGL(k_peek, dim3((t->d.capacity + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->cur(), ids_dev,
                           xy_dev, disappeared_dev, n_dev);
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

int ysmr_tracker_info(ysmr_tracker *t, void *stream, int32_t *n_tracks, int32_t *next_id, int32_t *error_bits)
{
    if (!t) return ysmr::fail(YSMR_ERR_ARG, "tracker handle is NULL");
    int host[4], err = 0;
    YSMR_HIP_CHECK(hipMemcpyAsync(host, t->cur().n_tracks, sizeof(int) * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    YSMR_HIP_CHECK(hipMemcpyAsync(&err, t->d.err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    YSMR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    if (n_tracks) *n_tracks = host[0];
    if (next_id) *next_id = host[1];
    if (error_bits) *error_bits = err;
    return YSMR_OK;
}

}  // extern "C"
