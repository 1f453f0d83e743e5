/*
 * ysmr_hip.h -- C ABI of libysmr_hip.so: the MI355X (gfx950) implementation of YSMR's per-frame
 * detect-and-link hot path.  Plain pointers and sizes only; every pointer named *_dev is a HIP
 * device pointer owned by the caller (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  No entry point allocates on the per-frame path
 * (ysmr_tracker_create/destroy own the tracker state).  Every function returns 0 (YSMR_OK) or a
 * YSMR_ERR_* code and never throws -- mirroring the reference's "log + return None" convention
 * (ysmr/track_eval.py:50-77, ysmr/main.py:92-95); ysmr_last_error() gives the message.
 *
 * Reference interfaces replaced (files under /root/reference):
 *   ysmr_unpack_dib_batch cv2.VideoCapture.read (uncompressed AVI frames)   ysmr/track_eval.py:159
 *   ysmr_threshold_batch  cv2.cvtColor + cv2.GaussianBlur + 2 x cv2.adaptiveThreshold
 *                         ysmr/track_eval.py:180-208
 *   ysmr_mean_threshold_batch  cv2.cvtColor + cv2.GaussianBlur + cv2.meanStdDev + threshold_list
 *                         moving average + cv2.threshold   ysmr/track_eval.py:180-182, 219-253
 *   ysmr_components_batch scipy binary_propagation + cv2.findContours + cv2.minAreaRect +
 *                         reshape_result   ysmr/track_eval.py:211-303, ysmr/helper_file.py:1336-1347
 *   ysmr_detect_batch     both of the above in one call
 *   ysmr_tracker_*        CentroidTracker.__init__/update   ysmr/tracker.py:37-71, 93-230
 *                         GaussianSumFIR.correct/predict    ysmr/gsff.py:204-347
 *                         row emission                      ysmr/track_eval.py:313-316
 *   ysmr_gsff_gains       GaussianSumFIR.generate_n_i/compute_lsf_gain  ysmr/gsff.py:87-153
 *   ysmr_rows_sort        sort_list (order by TRACK_ID, POSITION_T)  ysmr/helper_file.py:1538-1574
 *   ysmr_select_tracks    select_tracks + find_good_tracks   ysmr/track_eval.py:408-843
 *   ysmr_evaluate_tracks  evaluate_tracks (statistics, not the plots)  ysmr/track_eval.py:846-1318
 *   ysmr_rows_columns,    save_list text + get_data (pandas.read_csv) + save_df_to_csv
 *   ysmr_rows_format_csv  ysmr/helper_file.py:1403-1478, 860-905, 1366-1400  (host functions)
 */
#ifndef YSMR_HIP_H
#define YSMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YSMR_OK            0
#define YSMR_ERR_ARG       1   /* bad argument (null pointer, size, unsupported option) */
#define YSMR_ERR_HIP       2   /* a HIP runtime call failed */
#define YSMR_ERR_CAPACITY  3   /* a fixed-capacity buffer would overflow (tracks, workspace) */
#define YSMR_ERR_STATE     4   /* handle used in the wrong state */

#define YSMR_ABI_VERSION   15

/* per-frame detection status bits (status_dev) */
#define YSMR_DET_OVERFLOW  1   /* more components than max_det: detections truncated */
#define YSMR_DET_ARENA     2   /* geometry scratch arena exhausted: some rectangles missing */
#define YSMR_DET_STALLED   4   /* an internal grid barrier timed out: this call's outputs are undefined; the next call on
                                 the same buffers clears everything and is valid again */

/* cv_flavour: which OpenCV the a1 / a6 arithmetic follows.  opencv-contrib-python is an unpinned third-party
 * dependency of the reference (setup.py:29, "openCV v3 or v4"); two of its results changed between releases:
 *   0                     OpenCV >= 4.5.1: 15-bit BGR2GRAY coefficients; cv::minAreaRect angle in (0, 90] (the
 *                         restated hull order + rotating calipers produce [0, 90]: an axis-aligned box is
 *                         reported with angle 90, or 0, whichever of its equal-area sides is visited last)
 *   YSMR_CV_ANGLE_PRE451  OpenCV < 4.5.1: the same rectangle with its angle in [-90, 0) and width / height named
 *                         the other way round (angle 90 becomes -90 with the sides as they are)
 *   YSMR_CV_GRAY_3X       OpenCV 3.x: 14-bit BGR2GRAY coefficients (1868 / 9617 / 4899); no effect on gray input
 * (upstream-recollection, like the rest of the cv2 restatement: the image half is parity-unpinned) */
#define YSMR_CV_ANGLE_PRE451 1
#define YSMR_CV_GRAY_3X      2
/* A scheduling hint carried in the same argument (results are the same bytes with or without it): the caller runs the
 * one-launch link (ysmr_tracker_run with capacity and max_det <= 2048 or so) on another stream while this call's kernels
 * execute.  ysmr_threshold_batch then takes the float32-chain kernel, whose resident grid leaves registers and LDS on
 * every compute unit, and ysmr_components_batch launches k_windows / k_geometry with the smaller resident grids that
 * leave the link's workgroups their 59 KB of LDS per unit (detection alone is ~8 % slower that way). */
#define YSMR_BESIDE_LINK     4
/* ... and the hint for a handle that links a whole batch with one launch (ysmr_tracker_batched): that launch holds ONE
 * compute unit for the length of the batch.  The matrix-pipe threshold kernel, which gives every compute unit one
 * workgroup, then cuts its rows for 248 workgroups (31 on each of the 8 XCDs) instead of 256 -- whichever XCD the link sits
 * on, none gets a workgroup it cannot place before another has finished (same bytes). */
#define YSMR_BESIDE_BATCH_LINK 8
/* ... and for the two-launch link of large tables (k_link + k_track, 4K: 5000 tracks): its launches want a fat workgroup and
 * thousands of waves placed every ~30 us, so the matrix-pipe threshold kernel keeps to 160 of the 256 compute units (round 5's
 * kernel at 3840 x 2160: 24.2 k frames/s end to end on 128 or 160 workgroups, 23.7 k on 192, 21.9 k on 248 -- and a threshold
 * fraction of 0.13 / 0.16 / 0.19 / 0.23, profiles/r05_sweep_4k_thr_grid.log; same bytes). */
#define YSMR_BESIDE_SPLIT_LINK 16
#define YSMR_CV_FLAVOUR_MASK 31

/* One output row: a live track in one frame (ysmr/track_eval.py:313-316,
 * CSV columns TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE). */
typedef struct ysmr_row {
    int32_t frame;        /* POSITION_T */
    int32_t track_id;     /* TRACK_ID */
    double  x, y;         /* POSITION_X/Y: GSFF-filtered (or raw when GSFF is disabled) */
    float   w, h, angle;  /* WIDTH, HEIGHT, DEGREES_ANGLE; all 0 while the track is "disappeared" */
    int32_t disappeared;  /* consecutive frames without a detection (0 = matched this frame) */
} ysmr_row;

typedef struct ysmr_tracker ysmr_tracker; /* opaque; bound to one device + used from one thread */

int         ysmr_abi_version(void);
const char *ysmr_last_error(void);        /* message of the calling thread's last failing call */

/* ---- frame ingest: f1 ------------------------------------------------------------------- */

/* What cv2.VideoCapture.read() (ysmr/track_eval.py:159) does to a frame of an UNCOMPRESSED AVI once it has its bytes:
 * DIB rows are stored bottom-up (bottom_up = 1) with a stride padded to 4 bytes; 8-bit frames with a palette that
 * is not the gray ramp are expanded to BGR.  raw_dev: u8 [n_frames][raw_frame_bytes], the 'movi' chunk bodies as
 * they are in the file; bytes_per_pixel 1 or 3; palette_dev: NULL, or u8 [256][3] (B, G, R) for 1-byte pixels.
 * frames_dev: u8 [n_frames][height][width][channels], channels = 3 with a palette, else bytes_per_pixel -- the
 * layout ysmr_threshold_batch / ysmr_detect_batch read; 4-byte aligned. */
int ysmr_unpack_dib_batch(void *stream, const uint8_t *raw_dev, int n_frames, size_t raw_frame_bytes, int height,
                          int width, int bytes_per_pixel, int row_stride, int bottom_up,
                          const uint8_t *palette_dev, uint8_t *frames_dev);

/* HOST function: n bytes of the open file `fd`, from `offset` on, into `dst` (e.g. the pinned staging buffer of a frame
 * feed) by `threads` positional reads side by side (<= 0: 8).  What cap.read() does for an uncompressed file, a batch of
 * frames at a time (ysmr/track_eval.py:159). */
int ysmr_file_read(int fd, void *dst, size_t n, long long offset, int threads);

/* ---- detection: a1-a6 ------------------------------------------------------------------- */

/* Bytes of scratch ysmr_detect_batch needs for this geometry. */
size_t ysmr_detect_workspace_bytes(int batch, int height, int width, int max_det);

/* Call once after allocating a workspace, again if the output buffers used with it were written by anyone else,
 * and after a detection call that returned an error.  The workspace keeps a record of where the previous
 * ysmr_components_batch / ysmr_detect_batch call may have written the label map and the mask (a word per 32-pixel
 * row segment, left by the labelling kernel); when the next call gets the same labels_dev / mask_dev, batch,
 * geometry and max_det it zeroes the two maps there -- and only where that call does not write them again -- instead
 * of the whole maps.  A workspace whose first 256 bytes are zero (this call) makes the next call clear everything. */
int ysmr_detect_workspace_init(void *stream, void *workspace_dev, size_t workspace_bytes);

/* a1-a3 fused.  frames_dev: u8 [batch][height][width][channels], channels 1 (gray) or 3 (BGR).
 * cls_dev: u8 [batch][height][width] (allocation rounded up to a multiple of 16 bytes):
 *   bit0 = `thresh`  (1st adaptiveThreshold call), bit1 = `markers` (2nd call).
 * inv = 0: THRESH_BINARY (bit set iff s - mean > t); inv = 1: THRESH_BINARY_INV (s - mean <= t).
 * use_high = 0 reproduces "adaptive double threshold = 0": bit1 mirrors bit0. */
int ysmr_threshold_batch(void *stream, const uint8_t *frames_dev, int batch, int height, int width,
                         int channels, int inv, int t_low, int t_high, int use_high,
                         uint8_t *cls_dev, int cv_flavour);

/* Fault injection for the library's own test of YSMR_DET_STALLED, through the workspace the caller owns (there is no
 * process-wide switch): a workspace whose header word at byte offset YSMR_WS_FAULT_OFFSET holds
 * YSMR_WS_FAULT_RESIDUE_STALL makes the NEXT ysmr_components_batch / ysmr_detect_batch call on THAT workspace behave as if
 * one workgroup of its barrier kernel never became resident -- the barrier times out (quickly), every frame's status gets
 * YSMR_DET_STALLED and the call returns; the call clears the word.  ysmr_detect_workspace_init clears it too. */
#define YSMR_WS_FAULT_OFFSET        32
#define YSMR_WS_FAULT_RESIDUE_STALL 0xFA17057Au

/* Measurement aid: the NEXT ysmr_threshold_batch / _variant call of this host thread hands the two HIP events (hipEvent_t
 * created with timing enabled; either may be NULL) to its kernel dispatch (hipExtLaunchKernel), which sets them to the
 * kernel's own start and end on the device -- hipEventElapsedTime between them is the duration a kernel trace reports.
 * Events recorded on the stream before and after the call also count the dispatch gaps on both sides (~10 us of a
 * 100 us kernel).  Not used by the mean-gray call. */
int ysmr_threshold_timing(void *start_event, void *stop_event);
/* How many workgroups the matrix-pipe threshold kernel cuts a batch's rows for under these hints (256: one per compute
 * unit; 248 beside the one-launch batch link; 160 beside the two-launch link) -- what a caller needs to size its batches:
 * with as many frames per batch as workgroups (of one panel each: frames up to 1232 columns wide) every workgroup takes ONE
 * whole frame and starts one item, where 256 frames on 248 workgroups start two (the kernel is 5 % faster per frame that way:
 * 0.344 instead of 0.327 of the roofline, profiles/r05_batch_248.log).  ABI 14.  No reference counterpart (the reference
 * thresholds frame by frame, track_eval.py:180-208). */
int ysmr_threshold_workgroups(int cv_flavour);

/* The same call with the kernel named (test and measurement aid; the results are the same bytes whichever is taken).
 * Gray frames of at least 18 rows and 64 columns (width a multiple of 4) are served by a kernel that evaluates the
 * Gaussian mean on the matrix pipe to within 1/512, decides every pixel that is farther than that from both levels
 * and re-evaluates the rest with cv2's float32 arithmetic (csrc/thr_mfma.hip); everything else, and BGR input, by the
 * kernels that run that arithmetic for every pixel (csrc/detect.hip: k_threshold_strip, k_threshold).
 * variant 0: as ysmr_threshold_batch; 1: the float32-chain kernels only; 2: the matrix-pipe kernel deciding every
 * pixel farther than 1/2048 from the levels (a quarter of the shipped margin: any wrong byte says that the margin is
 * being used up; diagnostic); 3: the matrix-pipe kernel with every pixel sent through its exact path (diagnostic).  2 and 3 fail for geometries that kernel does
 * not serve. */
int ysmr_threshold_batch_variant(void *stream, const uint8_t *frames_dev, int batch, int height, int width,
                                 int channels, int inv, int t_low, int t_high, int use_high,
                                 uint8_t *cls_dev, int cv_flavour, int variant);

/* The mean-gray threshold branch, taken by the reference when 'adaptive double threshold' < 0
 * (ysmr/track_eval.py:219-253): replaces cv2.meanStdDev(gray), the 5 s moving average of
 * mean +- stddev +- offset kept in `threshold_list`, and cv2.threshold(blurred, int(average)).
 *   inv      0: white bacteria on dark background (THRESH_BINARY,  level = mean + stddev + offset)
 *            1: dark on bright              (THRESH_BINARY_INV, level = mean - stddev - offset)
 *   offset   settings['threshold offset for detection'] as it stands at track_eval.py:222-229,
 *            i.e. already negated for inv = 1 (track_eval.py:132)
 *   window   longest list the average is taken over = floor(5 * fps) + 1 (the list is trimmed
 *            AFTER the average, when it is longer than 5 * fps: track_eval.py:239-242)
 *   state_dev  ysmr_mean_threshold_state_bytes(window) bytes, 8-byte aligned; all zero = empty
 *            list (start of a video); carries the list from one call to the next
 *   stats_dev  f64 [batch][4] out: mean, stddev, this frame's level, averaged integer level
 *   levels_dev i32 [batch] out: the level each frame was compared with (clamped to [-1, 256])
 *   cls_dev    u8 [batch][H][W] out: 3 where the blurred pixel is foreground, else 0 -- the class
 *            map ysmr_components_batch expects (every foreground pixel is its own marker: this
 *            branch has no binary_propagation step) */
size_t ysmr_mean_threshold_state_bytes(int window);
int ysmr_mean_threshold_batch(void *stream, const uint8_t *frames_dev, int batch, int height, int width,
                              int channels, int inv, double offset, int window, void *state_dev,
                              double *stats_dev, int32_t *levels_dev, uint8_t *cls_dev, int cv_flavour);

/* a4-a6 from a class map already in HBM (written by ysmr_threshold_batch or by the caller):
 * hysteresis + labelling + RETR_EXTERNAL ordering + minAreaRect.  Same outputs as
 * ysmr_detect_batch; cls_dev is read and its bit2 is written. */
int ysmr_components_batch(void *stream, int batch, int height, int width, void *workspace_dev,
                          size_t workspace_bytes, uint8_t *cls_dev, uint8_t *mask_dev,
                          int32_t *labels_dev, int32_t *det_count_dev, float *det_dev,
                          int32_t *anchors_dev, int max_det, int32_t *status_dev, int cv_flavour);

/* a1-a6 = ysmr_threshold_batch followed by ysmr_components_batch.  Outputs (all device):
 *   cls_dev     u8  [batch][H][W]  class map as above (bit2 is used internally as a flag)
 *   mask_dev    u8  [batch][H][W]  final mask {0,255} == binary_propagation(markers, mask=thresh)
 *                                  (may be NULL)
 *   labels_dev  i32 [batch][H][W]  0 = background, else 1 + raster index of the first pixel of
 *                                  the 8-connected component of the final mask (required: it is
 *                                  also the union-find array)
 *   det_count_dev i32 [batch]      detections per frame (<= max_det written)
 *   det_dev     f32 [batch][max_det][5]  cx, cy, w, h, angle_deg of cv2.minAreaRect, in
 *                                  cv2.findContours(RETR_EXTERNAL) order (reverse raster order
 *                                  of first pixels; components nested in a hole are skipped)
 *   anchors_dev i32 [batch][max_det]  raster index of each detection's first pixel (may be NULL)
 *   status_dev  i32 [batch]        YSMR_DET_* bits
 */
int ysmr_detect_batch(void *stream, const uint8_t *frames_dev, int batch, int height, int width,
                      int channels, int inv, int t_low, int t_high, int use_high,
                      void *workspace_dev, size_t workspace_bytes, uint8_t *cls_dev,
                      uint8_t *mask_dev, int32_t *labels_dev, int32_t *det_count_dev,
                      float *det_dev, int32_t *anchors_dev, int max_det, int32_t *status_dev,
                      int cv_flavour);

/* ---- linking: a7-a19 -------------------------------------------------------------------- */

/* Horizon sizes n_i (gsff.py:87-109) and rows 0/1 of each least-squares gain (gsff.py:111-153,
 * closed form of the default constant-velocity model).  n_max <= 0 means "None -> fps".
 * n_i_out: n_f ints.  gains_out (may be NULL): for filter i, 2 rows x 2*n_i[i] doubles, filters
 * concatenated.  Returns YSMR_ERR_ARG for horizons that are not strictly increasing and >= 1. */
int ysmr_gsff_gains(double fps, int n_min, double n_max, int n_f, int32_t *n_i_out,
                    double *gains_out);

/* CentroidTracker(max_disappeared, fps, n_min, n_max, n_f, use_gsff).  capacity = maximum number
 * of simultaneously live tracks, max_det = maximum detections per frame.  gains_host may be NULL
 * (closed form) or point to host memory laid out as ysmr_gsff_gains() writes it. */
int ysmr_tracker_create(double max_disappeared, double fps, int n_min, double n_max, int n_f,
                        int use_gsff, int capacity, int max_det, const double *gains_host,
                        ysmr_tracker **out);
int ysmr_tracker_destroy(ysmr_tracker *t);
int ysmr_tracker_reset(ysmr_tracker *t, void *stream);

/* One CentroidTracker.update(rects).  det_dev: [m][5] = cx, cy, w, h, angle; f32 as written by
 * ysmr_detect_batch (det_is_f64 = 0) or f64 (det_is_f64 = 1; for callers holding float64
 * centroids, like the reference's input_centroids, tracker.py:111).  m >= 0: detection count known
 * on the host; m < 0: read it from *m_dev (device i32, clamped to max_det).
 * Outputs (device, each may be NULL):
 *   rows_dev      ysmr_row [capacity]  one row per live track after the update, ascending id
 *   n_rows_dev    i32                  number of rows
 *   claim_col_dev i32 [capacity]       for each track row BEFORE the update (ascending id):
 *                                      claimed detection column or -1
 *   n_before_dev  i32                  number of tracks before the update
 *   new_cols_dev  i32 [max_det]        detection columns registered as new tracks, in id order
 *   n_new_dev     i32
 */
int ysmr_tracker_update(ysmr_tracker *t, void *stream, const void *det_dev, int det_is_f64, int m,
                        const int32_t *m_dev, int32_t frame_index, ysmr_row *rows_dev,
                        int32_t *n_rows_dev, int32_t *claim_col_dev, int32_t *n_before_dev,
                        int32_t *new_cols_dev, int32_t *n_new_dev);

/* The frame loop of track_bacteria for `batch` consecutive frames whose detections are already
 * on the device: det_dev f32 [batch][max_det][5], det_count_dev i32 [batch].  Rows are appended
 * to rows_dev (capacity rows_capacity) starting at *row_count_dev, which is advanced; rows of a
 * frame are contiguous and in ascending id order.  Overflow sets *row_count_dev past capacity
 * (rows beyond capacity are dropped); the host checks after synchronising. */
/* 1 when the handle links with one launch per frame (k_frame), else 0 */
int ysmr_tracker_fused(ysmr_tracker *t);

/* How ysmr_tracker_run links a batch.  A handle whose configuration allows it -- tracking.ini's defaults do: up to three
 * filters with horizons of at most 31 frames, capacity <= 768 (a track per lane of one 768-thread workgroup), max_det <= 2456
 * -- links a whole batch with ONE launch
 * (one workgroup, a track per lane, the filter state in registers from the first frame to the last, the measurement
 * history in a ring in HBM); every other handle,
 * and ysmr_tracker_update, run one launch (or two, for large tables) per frame.  The two paths emit the same frames, ids,
 * counters, boxes and row order; the filtered positions agree within the parity tolerance (1e-9 px on well-conditioned rows),
 * not bit for bit: the batch launch keeps running window sums and one reciprocal for the three weights where the per-frame
 * kernels evaluate the FIR and three divisions.
 *   ysmr_tracker_batched    1 when ysmr_tracker_run takes the one-launch-per-batch path, else 0
 *   ysmr_tracker_link_mode  mode 0: the library's choice (default); 1: one launch per frame even where a batch launch
 *                           would serve (measurement and tests).  Takes effect with the next call; the track table is
 *                           carried over. */
int ysmr_tracker_batched(ysmr_tracker *t);
int ysmr_tracker_link_mode(ysmr_tracker *t, int mode);

/* Optional, for a handle that links a batch with one launch (a no-op for every other): that launch reads each frame's
 * detections binned into a uniform grid of cells, which ysmr_tracker_run works out in a launch of its own in front of the
 * link.  A caller that detects batch b+1 on one stream while batch b is linked on another can take that launch off the
 * link's chain: call this on the DETECTION stream behind the call that writes det_dev / det_count_dev (up to 256 frames),
 * then ysmr_tracker_run with the same det_dev, det_count_dev and batch on the link stream once it has waited for the
 * detection stream as it must anyway.  slot (0 or 1) names which of two internal blocks to fill: a block must not be
 * prepared again before the ysmr_tracker_run that uses it has executed (two detectors taking turns use their own number).
 * The binning holds for the bytes det_dev held when it executed and is used by ONE ysmr_tracker_run; rewrite the
 * detections and you call it again. */
int ysmr_tracker_prepare(ysmr_tracker *t, void *stream, const float *det_dev, const int32_t *det_count_dev, int batch,
                         int slot);

int ysmr_tracker_run(ysmr_tracker *t, void *stream, const float *det_dev,
                     const int32_t *det_count_dev, int batch, int32_t first_frame_index,
                     ysmr_row *rows_dev, int64_t rows_capacity, int64_t *row_count_dev);

/* Current track table in id order: ids i32 [capacity], positions f64 [capacity][2] (the
 * CentroidTracker.objects values: GSFF predictions, or raw centroids without GSFF), disappeared
 * counters, count.  Device outputs, each may be NULL. */
int ysmr_tracker_peek(ysmr_tracker *t, void *stream, int32_t *ids_dev, double *xy_dev,
                      int32_t *disappeared_dev, int32_t *n_dev);

/* Host-visible counters (synchronises the stream): live tracks, next id, sticky error bits. */
int ysmr_tracker_info(ysmr_tracker *t, void *stream, int32_t *n_tracks, int32_t *next_id,
                      int32_t *error_bits);

/* ---- output: a19, f2 --------------------------------------------------------------------- */

/* Order rows by (TRACK_ID, POSITION_T): what sort_list does to the csv after tracking
 * (helper_file.py:1538-1574, called at track_eval.py:393).  rows_dev and sorted_dev are distinct
 * device arrays of n_rows rows; (track_id, frame) pairs are unique, so the order is total. */
size_t ysmr_rows_sort_workspace_bytes(long long n_rows);
int    ysmr_rows_sort(void *stream, const ysmr_row *rows_dev, long long n_rows, void *workspace_dev,
                      size_t workspace_bytes, ysmr_row *sorted_dev);

/* HOST function (no GPU involved): the csv text of `rows_host` exactly as the reference's final
 * file has it -- header of save_list (helper_file.py:1451), one line per row, numbers printed the
 * way DataFrame.to_csv prints uint32 / float64 columns (shortest repr; WIDTH, HEIGHT and
 * DEGREES_ANGLE are the float32 values widened to float64, 0.0 for disappeared tracks;
 * helper_file.py:1366-1400, 881-889).  out_capacity >= ysmr_rows_csv_bound(n_rows, with_header);
 * threads <= 0: one per hardware thread. */
size_t ysmr_rows_csv_bound(long long n_rows, int with_header);
int    ysmr_rows_format_csv(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas,
                            int threads, char *out, size_t out_capacity, size_t *out_length);

/* HOST function: the same text written to `path` (created or truncated), every formatting thread writing its own
 * piece at its place in the file; *out_length (may be NULL) receives the number of bytes. */
int    ysmr_rows_write_csv(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas,
                           int threads, const char *path, size_t *out_length);
/* ... and, in the same pass, the seven DataFrame columns of ysmr_rows_columns below (ABI 11): the values the text is
 * printed from are the values pandas would read back from it, worked out once for both consumers. */
int    ysmr_rows_write_csv_columns(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas,
                                   int threads, const char *path, size_t *out_length, uint32_t *track_id, uint32_t *t,
                                   double *x, double *y, double *w, double *h, double *angle);

/* DEVICE function (ABI 15): the same csv text and the same seven columns worked out ON THE DEVICE from rows that are already there
 * in order (ysmr_rows_sort's output) -- a thread prints a row (shortest round-trip digits by Burger & Dybvig's free-format
 * algorithm in 128-bit fixed point, CPython's layout, pandas' float converter, csrc/fmt.h), a scan places the rows, a second
 * launch packs them behind the header.  Asynchronous on `stream`.  csv_capacity >= ysmr_rows_csv_bound(n_rows, with_header);
 * *csv_length_dev receives the number of bytes; *unserved_dev counts the rows that hold a value the device form does not print
 * (NaN, infinities, |v| outside 2^-20 .. 2^24 other than zero: not something a track produces) -- if it is not 0 the text and
 * columns are incomplete and the caller takes ysmr_rows_write_csv_columns for the table.  Byte for byte and bit for bit what
 * that host function gives otherwise (track_eval.py:393, helper_file.py:1403-1478, 860-905, 1366-1400). */
size_t ysmr_rows_format_device_workspace_bytes(long long n_rows);
int    ysmr_rows_format_device(void *stream, const ysmr_row *rows_dev, long long n_rows, int with_header, int via_pandas,
                               void *workspace_dev, size_t workspace_bytes, char *csv_dev, size_t csv_capacity,
                               unsigned long long *csv_length_dev, uint32_t *track_id_dev, uint32_t *t_dev, double *x_dev,
                               double *y_dev, double *w_dev, double *h_dev, double *angle_dev, uint32_t *unserved_dev);
/* HOST function: the device form's arithmetic (csrc/fmt.h) run on the host a row at a time -- what the CPU tests compare with
 * ysmr_rows_format_csv (std::to_chars) on millions of values.  columns5: five arrays of n_rows doubles, one behind the other
 * (may be NULL); *unserved as above (such rows are left out of the text). */
int    ysmr_rows_format_csv_devicelike(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas, char *out,
                                       size_t out_capacity, size_t *out_length, double *columns5, long long *unserved);

/* HOST functions (ABI 13): the same csv and columns worked out WHILE the video runs.  Rows go in as the link emits them -- any
 * number of ysmr_rows_stream_push calls with host rows in any order; a call copies its rows and returns, the stream's threads
 * format them meanwhile -- and ysmr_rows_stream_finish orders what has been pushed by (TRACK_ID, POSITION_T) (a row's place is
 * offset[id] + frame - first_frame[id] for the tracker's tables, whose ids are never reused and whose tracks have a row in every
 * frame of their life; any other table is sorted), writes the csv to `path` (NULL: no file) and fills the seven columns (all
 * NULL: none): byte for byte what ysmr_rows_sort + ysmr_rows_write_csv_columns give for the same rows (track_eval.py:393,
 * helper_file.py:1538-1574).  One handle per video, used from one thread at a time. */
typedef struct ysmr_rows_stream ysmr_rows_stream;
int    ysmr_rows_stream_create(int threads, int via_pandas, ysmr_rows_stream **out);
int    ysmr_rows_stream_push(ysmr_rows_stream *s, const ysmr_row *rows_host, long long n_rows);
long long ysmr_rows_stream_count(ysmr_rows_stream *s);   /* rows pushed so far (-1: NULL handle) */
int    ysmr_rows_stream_finish(ysmr_rows_stream *s, int with_header, const char *path, size_t *out_length,
                               uint32_t *track_id, uint32_t *t, double *x, double *y, double *w, double *h, double *angle);
int    ysmr_rows_stream_destroy(ysmr_rows_stream *s);

/* HOST function: the seven DataFrame columns (dtypes of helper_file.py:881-889).
 * via_pandas (here and above): the reference does not keep the tracker's float64 values, it prints
 * them and reads the text back with pandas.read_csv (helper_file.py:860-905), whose default float
 * converter returns about one value in five 1 ulp off.  via_pandas = 1 applies that same text ->
 * double conversion (restated from pandas' precise_xstrtod), so that the DataFrame and the csv equal
 * the reference's; via_pandas = 0 keeps the exact values. */
int    ysmr_rows_columns(const ysmr_row *rows_host, long long n_rows, int via_pandas, uint32_t *track_id,
                         uint32_t *t, double *x, double *y, double *w, double *h, double *angle);

/* ---- selection: select_tracks / find_good_tracks (ysmr/track_eval.py:408-843) ------------------ */

/* Settings of the selection, as get_configs() stores them (ysmr/helper_file.py:776-788) and as
 * select_tracks() derives them (track_eval.py:574-583). */
typedef struct ysmr_select_params {
    double  area_lo, area_hi;        /* 'extreme area outliers lower / upper end in px*px' */
    double  area_factor;             /* 'exclude measurement when above x times average area' (0 = off) */
    double  q_area;                  /* 'percent quantiles excluded area' / 100 (<= 0: no area bounds) */
    double  motility_stop_fraction;  /* 'stop excluding motility outliers if total count above percent' / 100 */
    double  max_empty_ratio;         /* 'maximal empty frames in %' / 100 + 1 */
    double  ratio_min, ratio_max;    /* 'average width/height ratio min. / max.' */
    double  edge_fraction;           /* 'percent of screen edges to exclude' / 100 */
    int32_t min_length_frames;       /* int(round(fps) * 'minimal length in seconds') */
    int32_t limit_frames;            /* int(round(fps) * 'limit track length to x seconds'); 0 = off */
    int32_t limit_exact;             /* 'limit track length exactly' */
    int32_t omit_motility;           /* 'try to omit motility outliers' */
    int32_t max_holes;               /* 'maximal consecutive holes' */
    int32_t max_recursion;           /* 'maximal recursion depth' */
    int32_t frame_height, frame_width;
} ysmr_select_params;

#define YSMR_SELECT_OK                0   /* rows_selected rows were written */
#define YSMR_SELECT_TOO_SHORT         1   /* fewer rows than min_length_frames (track_eval.py:612-619) */
#define YSMR_SELECT_TOO_SHORT_CLEANED 2   /* ... after the clean-up (track_eval.py:676-684) */
#define YSMR_SELECT_NONE              3   /* no acceptable track (track_eval.py:817-820) */

/* What the reference logs along the way (track_eval.py:686-691, 707-708, 721-723, 799-815). */
typedef struct ysmr_select_summary {
    int32_t   status;                /* YSMR_SELECT_* */
    int32_t   outliers_used;         /* 0: distance-outlier exclusion off (setting, or too many outliers) */
    long long rows_before, tracks_before, rows_after, tracks_after;
    double    area_lo, area_hi;      /* area quantiles (-1 / inf when q_area <= 0) */
    double    q1_dist, q3_dist, dist_fence;
    long long dist_outliers;
    long long kick_reasons[9];       /* tracks per lowest reached rejection stage; [0] = passed */
    long long good_tracks, rows_selected;
} ysmr_select_summary;

/* The table must be ordered by (TRACK_ID, POSITION_T) -- what ysmr_rows_sort / sort_list produce --
 * and hold the values the reference's DataFrame holds (ysmr_rows_columns with via_pandas = 1).
 * Outputs: sel_row_dev[i] = row of the input table, sel_index_dev[i] = its index in the cleaned table
 * (the 'index' column of the reference's result), i < summary->rows_selected, in table order; both
 * need room for n_rows entries.  Synchronous: returns when the result is complete.
 * summary is host memory.  Parity note: means follow numpy's pairwise summation, the median pandas'
 * median_linear, quantiles numpy's 'linear' method (what pandas 2.x / numpy 2.x execute). */
size_t ysmr_select_workspace_bytes(long long n_rows, int max_recursion);
int    ysmr_select_tracks(void *stream, long long n_rows, const uint32_t *track_id_dev, const uint32_t *t_dev,
                          const double *x_dev, const double *y_dev, const double *w_dev, const double *h_dev,
                          const ysmr_select_params *params, void *workspace_dev, size_t workspace_bytes,
                          int64_t *sel_row_dev, int64_t *sel_index_dev, ysmr_select_summary *summary);

/* ---- per-track statistics of the selected tracks: evaluate_tracks, ysmr/track_eval.py:846-1318 ---- */

typedef struct ysmr_evaluate_params {
    double pixel_per_micrometre;   /* settings['pixel per micrometre'] */
    double fps;
    double min_turn_angle;         /* settings['minimal angle in degrees for turning point'] */
    int32_t angle_lag;             /* settings['compare angle between n frames'] */
    int32_t reach_lag;             /* int(round(fps * min(10, min length / 2, length limit / 2))), track_eval.py:990-1000 */
    int32_t median_kernel;         /* round(fps), made odd: the second medfilt of `moving` (track_eval.py:931-939) */
    int32_t reserved;
} ysmr_evaluate_params;

/* Input: the six columns of the table evaluate_tracks receives (rows ordered by TRACK_ID, POSITION_T; row i has
 * DataFrame index i), device pointers.  Outputs, device, n_rows entries each: WIDTH and HEIGHT in micrometres,
 * angle_diff (int32), moving, turn_points, motility_phenotype (int8), tp_of_tracks (f64, NaN where not moving),
 * travelled_dist -- the columns of <name>_analysed.csv -- and stats_dev f64 [tracks][12], the columns of
 * <name>_statistics.csv in the reference's order (Turn Points (TP/s), Distance, Speed, Time, Displacement, Perc.
 * Motile, Arc-Chord Ratio, Bacteria Length (a float32 value), Displacement divided by length, Motility Phenotype,
 * TRACK_ID, Median Speed); room for n_rows tracks.  *n_tracks_out: the number of tracks.  Synchronous. */
size_t ysmr_evaluate_workspace_bytes(long long n_rows);
int ysmr_evaluate_tracks(void *stream, long long n_rows, const uint32_t *track_id_dev, const uint32_t *t_dev,
                         const double *x_dev, const double *y_dev, const double *w_dev, const double *h_dev,
                         const ysmr_evaluate_params *params, void *workspace_dev, size_t workspace_bytes,
                         double *width_um_dev, double *height_um_dev, int32_t *angle_diff_dev, int8_t *moving_dev,
                         int8_t *turn_points_dev, double *tp_of_tracks_dev, double *travelled_dist_dev,
                         int8_t *motility_phenotype_dev, double *stats_dev, long long *n_tracks_out);

#ifdef __cplusplus
}
#endif
#endif /* YSMR_HIP_H */
