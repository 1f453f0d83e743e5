// Interface between detect.hip's launcher and the matrix-pipe threshold kernel (thr_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ysmr_thr {

struct Params {
    int H, W, batch;
    int panels, panel_w;   // column panels of at most 1232 columns (a multiple of 16 wide, the last one may be narrower)
    int by_xcd;            // frames are dealt to the 8 XCDs (a grid that is a multiple of 8)
    int inv, use_high, t_low, t_high;
    // x = X (v - theta_1), X = +-S: negative where the first level's bit is set, |x| >= 1.875 where it is decided (v = mean - blurred)
    float neg_x_mul, k_first, d2;     // -X (a tap of the accumulator's preset), -X theta_1, X (theta_1 - theta_2)
    float col_scale, row_scale;       // the Gaussian's taps enter as f16(col_scale w) (columns) and f16(row_scale w) (rows: X / col_scale)
    uint32_t sh1, k1, sh2, k2;        // where a level's sign bits go in the class byte: (f >> sh) & k
    int start_rows;        // the cost of starting an item, in rows of the walk (how the rows are cut into ranges)
    float kw[6];           // the Gaussian's distinct weights k[0..5] (k[i] == k[10 - i]), cv2's float32 values
};

// geometry / arguments the kernel serves (everything else stays on k_threshold_strip / k_threshold)
bool supported(int H, int W, int channels, int t_low, int t_high, int use_high);

// variant: 0 = shipped (EPS = 1.875 / S for the largest admissible scale S: 0.048 gray levels with cv2's taps), 1 = half that
// scale (diagnostic: twice the EPS, twice the list of undecided pixels), 2 = every pixel through the exact path (diagnostic)
// ev_start / ev_stop (either may be null): events the dispatch itself updates (hipExtLaunchKernel)
int launch(hipStream_t st, const uint8_t *frames, uint8_t *cls, int batch, int H, int W, int inv, int t_low, int t_high,
           int use_high, const float *gauss11, int blocks_wanted, int variant, hipEvent_t ev_start = nullptr,
           hipEvent_t ev_stop = nullptr);

}  // namespace ysmr_thr
