// ingest.hip -- f1 of the YSMR hot path: what cv2.VideoCapture.read() (ysmr/track_eval.py:159) does to the frames
// of an uncompressed AVI after fetching them from the file -- DIB rows are stored bottom-up with a 4-byte padded
// stride, 8-bit frames are palette indices -- done on the device, so that the host only moves file bytes into
// pinned memory (ysmr_amd/frames.py: DeviceFrameFeed) and the frames the detection kernels read are unpacked
// where they are used.  Pure byte shuffling: the result is identical to the host reader's (AviVideo.read).
#include "common.h"

namespace {

constexpr int UNPACK_BLOCKS = 1024;   // resident grid (see detect.hip: grids larger than the chip starve other streams)

// One thread per aligned dword of the output (an output row need not start on a dword); every byte finds its
// source on its own.  The kernel runs on the upload stream behind a PCIe copy ~50x its duration.
template <bool PALETTE>
__global__ __launch_bounds__(256) void k_unpack_dib(const uint8_t *__restrict__ raw, size_t raw_frame_bytes, int n, int H, int W,
                                                    int bpp, int row_stride, int bottom_up, const uint8_t *__restrict__ palette,
                                                    uint8_t *__restrict__ out)
{
    const int ch = PALETTE ? 3 : bpp;
    const size_t row_bytes = (size_t)W * ch, frame_bytes = row_bytes * H, total = frame_bytes * n;
    const size_t dwords = (total + 3) / 4;
    for (size_t d = (size_t)blockIdx.x * 256 + threadIdx.x; d < dwords; d += (size_t)gridDim.x * 256) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t at = d * 4 + k;
            if (at >= total) break;
            const size_t f = at / frame_bytes, in_frame = at - f * frame_bytes;
            const int y = (int)(in_frame / row_bytes), b = (int)(in_frame - (size_t)y * row_bytes);
            const uint8_t *src_row = raw + f * raw_frame_bytes + (size_t)(bottom_up ? H - 1 - y : y) * row_stride;
            uint32_t byte;
            if (PALETTE) byte = palette[3 * src_row[b / 3] + b % 3];
            else byte = src_row[b];
            v |= byte << (8 * k);
        }
        if (d * 4 + 4 <= total) reinterpret_cast<uint32_t *>(out)[d] = v;
        else
            for (size_t at = d * 4; at < total; ++at) out[at] = (uint8_t)(v >> (8 * (at - d * 4)));
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
    const size_t blocks = std::min<size_t>((total / 4 + 255) / 256 + 1, UNPACK_BLOCKS);
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
