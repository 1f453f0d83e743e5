// libysmr_hip.so -- the mean-gray threshold branch of the detection path (ysmr/track_eval.py:219-253,
// taken when 'adaptive double threshold' < 0): instead of the Gaussian adaptive threshold, every
// frame is compared against ONE level derived from the gray image's mean and standard deviation,
// averaged over the last 5 s of frames.  Three launches per batch:
//   k_gray_sums        sum and sum of squares of the gray frame (exact integers; cv2.meanStdDev)
//   k_mean_levels      mean/stddev -> per-frame level -> moving average in the reference's summation
//                      order -> int(level); carries the list of the last levels across batches
//   k_level_threshold  cvtColor + GaussianBlur(3x3) + cv2.threshold(blurred, level) -> class map
// The class map uses value 3 (thresh and marker bit) for foreground so that ysmr_components_batch
// keeps every component, as the reference does in this branch (no binary_propagation).
// All grids are resident-sized and stride over their work (see detect.hip on why).
#include "common.h"
#include <cmath>

namespace {

constexpr int MG_BLOCKS = 1024;
constexpr int SUM_CHUNK = 8192;    // pixels per (block, step) of k_gray_sums: 256 threads x 8 x 4 pixels
constexpr int LEVEL_SEG = 32;      // output rows per thread item of k_level_threshold

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
__device__ __forceinline__ uint32_t bgr2gray15(uint32_t b, uint32_t g, uint32_t r)
{
    return (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15;   // cv2 COLOR_BGR2GRAY (a1)
}
// gray value of pixel p of a frame
template <int CH>
__device__ __forceinline__ uint32_t gray_at(const uint8_t *frame, uint32_t p)
{
    if constexpr (CH == 1) return frame[p];
    else return bgr2gray15(frame[3 * (size_t)p], frame[3 * (size_t)p + 1], frame[3 * (size_t)p + 2]);
}
// the 4 gray pixels p .. p+3 (all inside the frame) packed into one dword
template <int CH>
__device__ __forceinline__ uint32_t gray4_at(const uint8_t *frame, uint32_t p)
{
    if constexpr (CH == 1) return load_u32(frame + p);
    else {
        const uint8_t *q = frame + 3 * (size_t)p;
        const uint32_t w0 = load_u32(q), w1 = load_u32(q + 4), w2 = load_u32(q + 8);
        const uint32_t g0 = bgr2gray15(w0 & 0xFFu, (w0 >> 8) & 0xFFu, (w0 >> 16) & 0xFFu);
        const uint32_t g1 = bgr2gray15(w0 >> 24, w1 & 0xFFu, (w1 >> 8) & 0xFFu);
        const uint32_t g2 = bgr2gray15((w1 >> 16) & 0xFFu, w1 >> 24, w2 & 0xFFu);
        const uint32_t g3 = bgr2gray15((w2 >> 8) & 0xFFu, (w2 >> 16) & 0xFFu, w2 >> 24);
        return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    }
}

// sums[f][0] += sum of gray, sums[f][1] += sum of gray^2 (u64; the caller zeroes them)
template <int CH>
__global__ __launch_bounds__(256) void k_gray_sums(const uint8_t *__restrict__ frames, uint32_t HW, int batch,
                                                   unsigned long long *__restrict__ sums, int stride)
{
    __shared__ unsigned long long s_part[2][4];
    const uint32_t chunks = (HW + SUM_CHUNK - 1) / SUM_CHUNK;
    const long long items = (long long)batch * chunks;
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int f = (int)(it / chunks);
        const uint32_t base = (uint32_t)(it - (long long)f * chunks) * SUM_CHUNK;
        const uint8_t *frame = frames + (size_t)f * HW * CH;
        uint32_t s = 0, q = 0;   // <= 32 pixels per thread: 255^2 * 32 fits easily
#pragma unroll
        for (int k = 0; k < SUM_CHUNK / 1024; ++k) {
            const uint32_t p = base + 4u * (threadIdx.x + 256u * k);
            if (p + 3 < HW) {
                const uint32_t g = gray4_at<CH>(frame, p);
#pragma unroll
                for (int o = 0; o < 4; ++o) { const uint32_t v = (g >> (8 * o)) & 0xFFu; s += v; q += v * v; }
            } else {
                for (uint32_t e = p; e < HW; ++e) { const uint32_t v = gray_at<CH>(frame, e); s += v; q += v * v; }
            }
        }
        unsigned long long s64 = s, q64 = q;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { s64 += __shfl_xor(s64, d); q64 += __shfl_xor(q64, d); }
        if ((threadIdx.x & 63) == 0) { s_part[0][threadIdx.x >> 6] = s64; s_part[1][threadIdx.x >> 6] = q64; }
        __syncthreads();
        if (threadIdx.x < 2) {
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
__global__ __launch_bounds__(256) void k_mean_levels(double *stats, int batch, double n_px, int inv, double offset,
                                                     int window, LevelState st, int32_t *levels)
{
    const long long seen = *st.seen;
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
    }
    __syncthreads();
    for (int f = threadIdx.x; f < batch; f += 256) {
        // int(sum(threshold_list) / len(threshold_list)) with the list as it is after this frame's append:
        // the last min(g + 1, window) levels, summed oldest first (Python's sum starts from int 0)
        const long long g = seen + f;
        const long long cnt = g + 1 < window ? g + 1 : window;
        double acc = 0.0;
        for (long long j = g - cnt + 1; j <= g; ++j)
            acc += j >= seen ? stats[4 * (size_t)(j - seen) + 2] : st.ring[j % window];
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
                                                         LevelGeo G, const int32_t *__restrict__ levels)
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

        auto row_sums = [&](int r, uint32_t &a, uint32_t &b) {
            const uint32_t base = (uint32_t)r * (uint32_t)W;
            uint32_t g, left, right;
            if (inner) {
                g = gray4_at<CH>(frame, base + c);
                left = gray_at<CH>(frame, base + c - 1);
                right = gray_at<CH>(frame, base + c + 4);
            } else {
                // border groups: columns beyond the image take their BORDER_REFLECT_101 source (columns
                // that only exist as padding of a partial group are never stored)
                g = 0;
                for (int o = 0; o < 4; ++o)
                    g |= gray_at<CH>(frame, base + border_index(c + o, W)) << (8 * o);
                left = gray_at<CH>(frame, base + cl);
                right = gray_at<CH>(frame, base + cr);
            }
            hsum4(g, left, right, a, b);
        };

        uint32_t ua, ub, ca, cb, da, db;
        row_sums(border_index(y0 - 1, H), ua, ub);
        row_sums(y0, ca, cb);
        for (int y = y0; y < y1; ++y) {
            row_sums(border_index(y + 1, H), da, db);
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

}  // namespace

extern "C" {

size_t ysmr_mean_threshold_state_bytes(int window)
{
    return window > 0 ? sizeof(double) * (size_t)window + sizeof(long long) : 0;
}

int ysmr_mean_threshold_batch(void *stream, const uint8_t *frames_dev, int batch, int height, int width, int channels,
                              int inv, double offset, int window, void *state_dev, double *stats_dev,
                              int32_t *levels_dev, uint8_t *cls_dev)
{
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
    const long long sum_items = (long long)batch * ((HW + SUM_CHUNK - 1) / SUM_CHUNK);
    const unsigned sum_blocks = (unsigned)(sum_items < MG_BLOCKS ? sum_items : MG_BLOCKS);
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(stats_dev);
    if (channels == 1) hipLaunchKernelGGL(k_gray_sums<1>, dim3(sum_blocks), dim3(256), 0, st, frames_dev, HW, batch, sums, 4);
    else hipLaunchKernelGGL(k_gray_sums<3>, dim3(sum_blocks), dim3(256), 0, st, frames_dev, HW, batch, sums, 4);
    LevelState ls{reinterpret_cast<double *>(state_dev), reinterpret_cast<long long *>(reinterpret_cast<double *>(state_dev) + window)};
    hipLaunchKernelGGL(k_mean_levels, dim3(1), dim3(256), 0, st, stats_dev, batch, (double)HW, inv, offset, window, ls, levels_dev);
    LevelGeo G{height, width, batch, (width + 3) / 4, (height + LEVEL_SEG - 1) / LEVEL_SEG, inv};
    const long long thr_items = (long long)G.groups_x * G.segs_y * batch;
    const long long want = (thr_items + 255) / 256;
    const unsigned thr_blocks = (unsigned)(want < MG_BLOCKS ? want : MG_BLOCKS);
    if (channels == 1) hipLaunchKernelGGL(k_level_threshold<1>, dim3(thr_blocks), dim3(256), 0, st, frames_dev, cls_dev, G, levels_dev);
    else hipLaunchKernelGGL(k_level_threshold<3>, dim3(thr_blocks), dim3(256), 0, st, frames_dev, cls_dev, G, levels_dev);
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

}  // extern "C"
