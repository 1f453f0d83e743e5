// ingest.hip -- f1 of the YSMR hot path: what cv2.VideoCapture.read() (ysmr/track_eval.py:159) does to the frames
// of an uncompressed AVI after fetching them from the file -- DIB rows are stored bottom-up with a 4-byte padded
// stride, 8-bit frames are palette indices -- done on the device, so that the host only moves file bytes into
// pinned memory (ysmr_amd/frames.py: DeviceFrameFeed) and the frames the detection kernels read are unpacked
// where they are used.  Pure byte shuffling: the result is the frame sequence cap.read() owes (tests/test_gpu_pipeline.py
// builds it from the clip that went into the file).
#include "common.h"
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>
#include <unistd.h>

namespace {

constexpr int UNPACK_BLOCKS = 1024;   // resident grid (see detect.hip: grids larger than the chip starve other streams)

__device__ __forceinline__ uint4 ld16(const uint8_t *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16(uint8_t *p, const uint4 &v) { __builtin_memcpy(p, &v, 16); }

// A WAVE moves one row at a time: the (frame, row) of a work item costs one division per 64 lanes, each lane moves 16
// bytes per step (the frame rows are W * channels bytes apart, so neither end need be aligned; gfx950 takes unaligned
// 16-byte accesses), the last few bytes of a row go one by one.  With a palette a lane expands four stored indices
// (one dword: DIB rows start on 4-byte boundaries) into their twelve B, G, R bytes from a copy of the palette in LDS.
// The kernel runs on the upload stream behind a PCIe copy ~50x its duration.
template <bool PALETTE>
__global__ __launch_bounds__(256) void k_unpack_dib(const uint8_t *__restrict__ raw, size_t raw_frame_bytes, int n, int H, int W,
                                                    int bpp, int row_stride, int bottom_up, const uint8_t *__restrict__ palette,
                                                    uint8_t *__restrict__ out)
{
    __shared__ uint8_t s_pal[PALETTE ? 768 : 4];
    if (PALETTE) {
        for (int i = threadIdx.x; i < 768; i += 256) s_pal[i] = palette[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)n * H, wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const size_t row_bytes = (size_t)W * (PALETTE ? 3 : bpp);
    for (long long item = wave0; item < rows; item += nwaves) {
        const long long f = item / H;
        const int y = (int)(item - f * H);
        const uint8_t *src = raw + (size_t)f * raw_frame_bytes + (size_t)(bottom_up ? H - 1 - y : y) * row_stride;
        uint8_t *dst = out + (size_t)item * row_bytes;
        if (!PALETTE) {
            const size_t whole = row_bytes & ~(size_t)15;
            for (size_t b = (size_t)lane * 16; b < whole; b += 64 * 16) st16(dst + b, ld16(src + b));
            if (whole + lane < row_bytes) dst[whole + lane] = src[whole + lane];
        } else {
            for (int x = 4 * lane; x < W; x += 4 * 64) {
                if (x + 4 <= W) {
                    uint32_t idx;
                    __builtin_memcpy(&idx, src + x, 4);
                    uint8_t o[12];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint8_t *c = s_pal + 3 * ((idx >> (8 * k)) & 0xFFu);
                        o[3 * k] = c[0]; o[3 * k + 1] = c[1]; o[3 * k + 2] = c[2];
                    }
                    __builtin_memcpy(dst + 3 * (size_t)x, o, 12);
                } else {
                    for (int xx = x; xx < W; ++xx) {
                        const uint8_t *c = s_pal + 3 * src[xx];
                        dst[3 * (size_t)xx] = c[0]; dst[3 * (size_t)xx + 1] = c[1]; dst[3 * (size_t)xx + 2] = c[2];
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int ysmr_unpack_dib_batch(void *stream, const uint8_t *raw_dev, int n_frames, size_t raw_frame_bytes, int height,
                                     int width, int bytes_per_pixel, int row_stride, int bottom_up, const uint8_t *palette_dev,
                                     uint8_t *frames_dev)
{
    if (n_frames <= 0 || height <= 0 || width <= 0)
        return ysmr::fail(YSMR_ERR_ARG, "n_frames, height, width must be positive (got %d, %d, %d)", n_frames, height, width);
    if (bytes_per_pixel != 1 && bytes_per_pixel != 3)
        return ysmr::fail(YSMR_ERR_ARG, "bytes_per_pixel must be 1 or 3, got %d", bytes_per_pixel);
    if (palette_dev && bytes_per_pixel != 1) return ysmr::fail(YSMR_ERR_ARG, "a palette needs 1-byte pixels");
    if (row_stride < width * bytes_per_pixel || raw_frame_bytes < (size_t)row_stride * height)
        return ysmr::fail(YSMR_ERR_ARG, "row_stride %d / raw_frame_bytes %zu too small for %d x %d x %d", row_stride, raw_frame_bytes,
                          width, height, bytes_per_pixel);
    if (!raw_dev || !frames_dev) return ysmr::fail(YSMR_ERR_ARG, "raw_dev and frames_dev must not be NULL");
    if ((uintptr_t)frames_dev & 3) return ysmr::fail(YSMR_ERR_ARG, "frames_dev must be 4-byte aligned");
    const size_t total = (size_t)n_frames * height * width * (palette_dev ? 3 : bytes_per_pixel);
    (void)total;
    const size_t blocks = std::min<size_t>(((size_t)n_frames * height + 3) / 4, UNPACK_BLOCKS);   // four rows (waves) per block
    hipStream_t st = (hipStream_t)stream;
    if (palette_dev)
        hipLaunchKernelGGL(k_unpack_dib<true>, dim3((unsigned)blocks), dim3(256), 0, st, raw_dev, raw_frame_bytes, n_frames, height,
                           width, bytes_per_pixel, row_stride, bottom_up, palette_dev, frames_dev);
    else
        hipLaunchKernelGGL(k_unpack_dib<false>, dim3((unsigned)blocks), dim3(256), 0, st, raw_dev, raw_frame_bytes, n_frames, height,
                           width, bytes_per_pixel, row_stride, bottom_up, palette_dev, frames_dev);
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

// HOST function: n bytes of an open file from `offset` on into `dst` (pinned staging memory of a frame feed), by `threads`
// positional reads side by side -- the kernel copies page-cache pages straight into dst.  One native call per batch of
// frames: the same reads issued from a Python thread pool cost the feed's producer thread a handful of GIL hand-overs per
// piece while the caller's thread issues launches (2.0 ms per 72 MB batch instead of the 1.0 ms the reads take).
extern "C" int ysmr_file_read(int fd, void *dst, size_t n, long long offset, int threads)
{
    if (fd < 0 || (!dst && n) || offset < 0) return ysmr::fail(YSMR_ERR_ARG, "fd, dst and offset must be valid");
    int nt = std::max(1, std::min(threads > 0 ? threads : 8, 64));
    if (n < ((size_t)8 << 20)) nt = 1;
    std::atomic<int> bad{0};
    auto piece = [&](size_t lo, size_t hi) {
        while (lo < hi) {
            const ssize_t got = ::pread(fd, (char *)dst + lo, hi - lo, (off_t)(offset + (long long)lo));
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) { bad.store(got == 0 ? -1 : errno ? errno : -1); return; }
            lo += (size_t)got;
        }
    };
    if (nt == 1) piece(0, n);
    else {
        // (a thread that cannot be started -- EAGAIN under a pids cgroup -- leaves its range to the caller's thread)
        std::vector<std::thread> pool;
        auto range = [&](int t, size_t &lo, size_t &hi) {
            lo = (n * (size_t)t / nt) & ~(size_t)4095;
            hi = t + 1 == nt ? n : (n * (size_t)(t + 1) / nt) & ~(size_t)4095;
        };
        int started = 0;
        try {
            for (; started < nt - 1; ++started) {
                size_t lo, hi;
                range(started, lo, hi);
                pool.emplace_back(piece, lo, hi);
            }
        } catch (const std::system_error &) {
        }
        for (int t = started; t < nt; ++t) {
            size_t lo, hi;
            range(t, lo, hi);
            piece(lo, hi);
        }
        for (auto &th : pool) th.join();
    }
    if (bad.load()) return ysmr::fail(YSMR_ERR_ARG, "short read of %zu bytes at offset %lld: %s", n, offset,
                                      bad.load() > 0 ? std::strerror(bad.load()) : "end of file");
    return YSMR_OK;
}
