/*
 * ysmr_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code) of the image half of
 * YSMR's per-frame detect path:
 *
 *   a1 cv2.cvtColor(BGR2GRAY)            reference call site ysmr/track_eval.py:180
 *   a2 cv2.GaussianBlur(gray,(3,3),0)    ysmr/track_eval.py:182
 *   a3 cv2.adaptiveThreshold(...) x2     ysmr/track_eval.py:189-208
 *   a4 scipy binary_propagation          ysmr/track_eval.py:211-214
 *   a5 cv2.findContours(RETR_EXTERNAL)   ysmr/track_eval.py:273-283
 *   a6 cv2.minAreaRect + reshape_result  ysmr/track_eval.py:287, ysmr/helper_file.py:1336-1347
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product path (ysmr_amd/) never does.
 *
 * PARITY STATUS
 *   a4 is pinned: tests compare oracle_propagate() with scipy.ndimage.binary_propagation itself.
 *   a1, a2, a3, a5, a6 are "parity unpinned": the arithmetic lives in opencv-contrib-python
 *   (setup.py:29 of the reference, ">=3.4.1", no version pinned), which is neither under
 *   /root/reference nor installed here, and the reference ships no tests/golden vectors.  The
 *   functions below restate OpenCV's published algorithms (imgproc: color_yuv/smooth/thresh/
 *   contours/convhull/rotcalipers) as documented in SURVEY.md 8.1-8.5:
 *     - BGR2GRAY: 15-bit fixed point (4.x coefficients 3735/19235/9798, +16384 >> 15)
 *     - 3x3 blur: exact integer (sum w*p + 8) >> 4, BORDER_REFLECT_101
 *     - adaptive mean: u8->f32, separable 11-tap Gaussian (sigma = 0.3*((11-1)*0.5-1)+0.8 = 2.0),
 *       BORDER_REPLICATE; row pass = ascending FMA chain from 0, column pass = symmetric form
 *       fma(c,k0,0) then fma(r[+j]+r[-j],kj,s) (the AVX2/FMA3 code path of filter.simd.hpp);
 *       mean = saturate_u8(round-half-even(f32))
 *     - contours: one detection per 8-connected component that is not enclosed by another
 *       component, emitted in reverse raster order of the component's first pixel
 *     - minAreaRect: strict convex hull (order: rightmost -> max-y side -> leftmost -> min-y side),
 *       f32 rotating calipers, "area <= minarea" (last minimum wins).  With this hull order the
 *       angles come out in [0, 90] (an axis-aligned box: 90, or 0), the range OpenCV documents from 4.5.1
 *       on; the [-90, 0) convention of earlier releases is a relabelling (ysmr_oracle.py: rect_convention).
 *       (Round 1's header called this the pre-4.5.1 convention; the values never were.)
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf() is used explicitly where the
 * restated algorithm fuses).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define YO_KSIZE 11
#define YO_KHALF 5

/* ------------------------------------------------------------------------------------------ */
/* a1: BGR -> gray, OpenCV 4.x 15-bit fixed point.  Identity when B == G == R.                 */
void yo_bgr2gray(const uint8_t *bgr, int h, int w, uint8_t *gray)
{
    for (long i = 0; i < (long)h * w; ++i) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15);
    }
}

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

static inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

/* a2: 3x3 binomial blur, exact integer arithmetic, BORDER_REFLECT_101. */
void yo_blur3(const uint8_t *gray, int h, int w, uint8_t *out)
{
    static const int wgt[3] = {1, 2, 1};
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int dy = -1; dy <= 1; ++dy) {
                int yy = reflect101(y + dy, h);
                for (int dx = -1; dx <= 1; ++dx) {
                    int xx = reflect101(x + dx, w);
                    s += wgt[dy + 1] * wgt[dx + 1] * gray[(long)yy * w + xx];
                }
            }
            out[(long)y * w + x] = (uint8_t)((s + 8) >> 4);
        }
}

/* The 11-tap Gaussian of adaptiveThreshold(ADAPTIVE_THRESH_GAUSSIAN_C, blockSize=11):
 * weights computed in double, normalised, then cast to f32 (cv::getGaussianKernel, CV_32F). */
void yo_gauss11(float *k)
{
    const double sigma = 0.3 * ((YO_KSIZE - 1) * 0.5 - 1.0) + 0.8; /* = 2.0 */
    const double scale2x = -0.5 / (sigma * sigma);
    double t[YO_KSIZE], sum = 0.0;
    for (int i = 0; i < YO_KSIZE; ++i) {
        double x = i - (YO_KSIZE - 1) * 0.5;
        t[i] = exp(scale2x * x * x);
        sum += t[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < YO_KSIZE; ++i) k[i] = (float)(t[i] * sum);
}

/* a3 (first half): local mean image, u8.  `blurred` is the output of yo_blur3. */
void yo_adaptive_mean(const uint8_t *blurred, int h, int w, uint8_t *mean)
{
    float k[YO_KSIZE];
    yo_gauss11(k);
    float *rowf = (float *)malloc(sizeof(float) * (size_t)h * w);
    /* row pass: s = 0; s = fma(p[x-5+i], k[i], s) for ascending i */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = 0.0f;
            for (int i = 0; i < YO_KSIZE; ++i) {
                int xx = clampi(x - YO_KHALF + i, 0, w - 1);
                s = fmaf((float)blurred[(long)y * w + xx], k[i], s);
            }
            rowf[(long)y * w + x] = s;
        }
    /* column pass: symmetric form */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = fmaf(rowf[(long)y * w + x], k[YO_KHALF], 0.0f);
            for (int j = 1; j <= YO_KHALF; ++j) {
                int yp = clampi(y + j, 0, h - 1), ym = clampi(y - j, 0, h - 1);
                float pr = rowf[(long)yp * w + x] + rowf[(long)ym * w + x];
                s = fmaf(pr, k[YO_KHALF + j], s);
            }
            float r = rintf(s); /* round-half-even in the default rounding mode (cvRound) */
            int v = (int)r;
            mean[(long)y * w + x] = (uint8_t)clampi(v, 0, 255);
        }
    free(rowf);
}

/* a3 (second half): the two adaptiveThreshold comparisons, as a class map.
 *   bit0 = `thresh` (first call), bit1 = `markers` (second call).
 * inv == 0 (THRESH_BINARY):     bit set iff (s - m) >  t
 * inv != 0 (THRESH_BINARY_INV): bit set iff (s - m) <= t
 * t_low/t_high are the integer thresholds the host derives from the reference's C arguments
 * (ysmr_amd/track_eval.py:threshold_params).  use_high == 0 reproduces adaptive double
 * threshold == 0 (no marker call): bit1 mirrors bit0. */
void yo_classify(const uint8_t *blurred, const uint8_t *mean, long n, int inv, int t_low,
                 int t_high, int use_high, uint8_t *cls)
{
    for (long i = 0; i < n; ++i) {
        int d = (int)blurred[i] - (int)mean[i];
        int lo = inv ? (d <= t_low) : (d > t_low);
        int hi = use_high ? (inv ? (d <= t_high) : (d > t_high)) : lo;
        cls[i] = (uint8_t)(lo | (hi << 1));
    }
}

/* a4: scipy.ndimage.binary_propagation(markers, mask=thresh), default (4-connected) structure.
 * Iterated masked dilation: pixels outside the mask keep their marker value and do seed
 * their 4-neighbours inside the mask.  out = 0/1. */
void yo_propagate(const uint8_t *cls, int h, int w, uint8_t *out)
{
    long n = (long)h * w;
    long *stack = (long *)malloc(sizeof(long) * (size_t)n);
    long sp = 0;
    for (long i = 0; i < n; ++i) {
        out[i] = (cls[i] & 2) ? 1 : 0;
        if (out[i]) stack[sp++] = i;
    }
    static const int dx4[4] = {1, -1, 0, 0}, dy4[4] = {0, 0, 1, -1};
    while (sp > 0) {
        long p = stack[--sp];
        int y = (int)(p / w), x = (int)(p % w);
        for (int d = 0; d < 4; ++d) {
            int xx = x + dx4[d], yy = y + dy4[d];
            if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
            long q = (long)yy * w + xx;
            if (!out[q] && (cls[q] & 1)) {
                out[q] = 1;
                stack[sp++] = q;
            }
        }
    }
    free(stack);
}

/* a5 (first half): 8-connected labelling with canonical labels:
 * label = 1 + raster index of the component's first (top-most, then left-most) pixel. */
void yo_label8(const uint8_t *fg, int h, int w, int32_t *labels)
{
    long n = (long)h * w;
    long *stack = (long *)malloc(sizeof(long) * (size_t)n);
    memset(labels, 0, sizeof(int32_t) * (size_t)n);
    for (long s = 0; s < n; ++s) {
        if (!fg[s] || labels[s]) continue;
        long sp = 0;
        stack[sp++] = s;
        labels[s] = (int32_t)(s + 1);
        while (sp > 0) {
            long p = stack[--sp];
            int y = (int)(p / w), x = (int)(p % w);
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int xx = x + dx, yy = y + dy;
                    if ((dx == 0 && dy == 0) || xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
                    long q = (long)yy * w + xx;
                    if (fg[q] && !labels[q]) {
                        labels[q] = (int32_t)(s + 1);
                        stack[sp++] = q;
                    }
                }
        }
    }
    free(stack);
}

/* a5 (second half): RETR_EXTERNAL.  outside[p] = 1 for background pixels 4-connected to the
 * (virtual, zero) frame around the image.  A component is external iff the pixel to the west of
 * its first pixel is outside the image or is such an `outside` background pixel. */
static void flood_outside(const uint8_t *fg, int h, int w, uint8_t *outside)
{
    long n = (long)h * w;
    long *stack = (long *)malloc(sizeof(long) * (size_t)n);
    long sp = 0;
    memset(outside, 0, (size_t)n);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            if (y != 0 && x != 0 && y != h - 1 && x != w - 1) continue;
            long p = (long)y * w + x;
            if (!fg[p] && !outside[p]) {
                outside[p] = 1;
                stack[sp++] = p;
            }
        }
    static const int dx4[4] = {1, -1, 0, 0}, dy4[4] = {0, 0, 1, -1};
    while (sp > 0) {
        long p = stack[--sp];
        int y = (int)(p / w), x = (int)(p % w);
        for (int d = 0; d < 4; ++d) {
            int xx = x + dx4[d], yy = y + dy4[d];
            if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
            long q = (long)yy * w + xx;
            if (!fg[q] && !outside[q]) {
                outside[q] = 1;
                stack[sp++] = q;
            }
        }
    }
    free(stack);
}

/* ------------------------------------------------------------------------------------------ */
/* a6: minAreaRect of a set of integer pixel centres.                                          */
typedef struct { float x, y; } yo_pt;

static int cmp_pt(const void *a, const void *b)
{
    const yo_pt *p = (const yo_pt *)a, *q = (const yo_pt *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}

static double cross3(yo_pt o, yo_pt a, yo_pt b)
{
    return ((double)a.x - o.x) * ((double)b.y - o.y) - ((double)a.y - o.y) * ((double)b.x - o.x);
}

/* Strict convex hull (collinear points dropped) in cv::convexHull(clockwise=false) order:
 * start at the lexicographically last point (max x, then max y), walk the max-y side to the
 * lexicographically first point (min x, then min y), return along the min-y side.
 * pts is sorted in place.  Returns the hull size. */
static int hull_cv_order(yo_pt *pts, int n, yo_pt *hull)
{
    qsort(pts, (size_t)n, sizeof(yo_pt), cmp_pt);
    /* unique */
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (m == 0 || pts[i].x != pts[m - 1].x || pts[i].y != pts[m - 1].y) pts[m++] = pts[i];
    n = m;
    if (n == 1) { hull[0] = pts[0]; return 1; }
    yo_pt *lo = (yo_pt *)malloc(sizeof(yo_pt) * (size_t)n * 2);
    yo_pt *up = lo + n;
    int nl = 0, nu = 0;
    /* min-y chain, left -> right: keep strict turns only */
    for (int i = 0; i < n; ++i) {
        while (nl >= 2 && cross3(lo[nl - 2], lo[nl - 1], pts[i]) <= 0) --nl;
        lo[nl++] = pts[i];
    }
    /* max-y chain, right -> left */
    for (int i = n - 1; i >= 0; --i) {
        while (nu >= 2 && cross3(up[nu - 2], up[nu - 1], pts[i]) <= 0) --nu;
        up[nu++] = pts[i];
    }
    /* up[0] = rightmost ... up[nu-1] = leftmost ; lo[0] = leftmost ... lo[nl-1] = rightmost */
    int k = 0;
    for (int i = 0; i < nu - 1; ++i) hull[k++] = up[i];
    for (int i = 0; i < nl - 1; ++i) hull[k++] = lo[i];
    free(lo);
    return k; /* all-collinear input gives k == 2: [rightmost, leftmost] */
}

/* f32 rotating calipers, CALIPERS_MINAREARECT.  out = {px,py, v1x,v1y, v2x,v2y}. */
static void rotating_calipers(const yo_pt *points, int n, float *out)
{
    float minarea = FLT_MAX;
    int best_left = 0, best_bottom = 0;
    float best_a = 0, best_b = 0, best_w = 0, best_h = 0;
    float *inv_len = (float *)malloc(sizeof(float) * (size_t)n * 3);
    yo_pt *vect = (yo_pt *)(inv_len + n);
    int left = 0, bottom = 0, right = 0, top = 0;
    int seq[4];
    float orientation = 0, base_a, base_b = 0;
    yo_pt pt0 = points[0];
    float left_x = pt0.x, right_x = pt0.x, top_y = pt0.y, bottom_y = pt0.y;

    for (int i = 0; i < n; ++i) {
        if (pt0.x < left_x) { left_x = pt0.x; left = i; }
        if (pt0.x > right_x) { right_x = pt0.x; right = i; }
        if (pt0.y > top_y) { top_y = pt0.y; top = i; }
        if (pt0.y < bottom_y) { bottom_y = pt0.y; bottom = i; }
        yo_pt pt = points[(i + 1 < n) ? i + 1 : 0];
        double dx = (double)pt.x - (double)pt0.x;
        double dy = (double)pt.y - (double)pt0.y;
        vect[i].x = (float)dx;
        vect[i].y = (float)dy;
        inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
        pt0 = pt;
    }
    {
        double ax = vect[n - 1].x, ay = vect[n - 1].y;
        for (int i = 0; i < n; ++i) {
            double bx = vect[i].x, by = vect[i].y;
            double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = (convexity > 0) ? 1.f : -1.f; break; }
            ax = bx; ay = by;
        }
    }
    base_a = orientation;
    seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;

    for (int k = 0; k < n; ++k) {
        float dp[4];
        dp[0] = +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y;
        dp[1] = -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y;
        dp[2] = -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y;
        dp[3] = +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y;
        float maxcos = dp[0] * inv_len[seq[0]];
        int main_element = 0;
        for (int i = 1; i < 4; ++i) {
            float cosalpha = dp[i] * inv_len[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            int pindex = seq[main_element];
            float lead_x = vect[pindex].x * inv_len[pindex];
            float lead_y = vect[pindex].y * inv_len[pindex];
            switch (main_element) {
            case 0: base_a = lead_x;  base_b = lead_y;  break;
            case 1: base_a = lead_y;  base_b = -lead_x; break;
            case 2: base_a = -lead_x; base_b = -lead_y; break;
            default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        seq[main_element] += 1;
        if (seq[main_element] == n) seq[main_element] = 0;
        {
            float dx = points[seq[1]].x - points[seq[3]].x;
            float dy = points[seq[1]].y - points[seq[3]].y;
            float width = dx * base_a + dy * base_b;
            dx = points[seq[2]].x - points[seq[0]].x;
            dy = points[seq[2]].y - points[seq[0]].y;
            float height = -dx * base_b + dy * base_a;
            float area = width * height;
            if (area <= minarea) {
                minarea = area;
                best_left = seq[3]; best_a = base_a; best_w = width;
                best_b = base_b; best_h = height; best_bottom = seq[0];
            }
        }
    }
    {
        float A1 = best_a, B1 = best_b, A2 = -best_b, B2 = best_a;
        float C1 = A1 * points[best_left].x + points[best_left].y * B1;
        float C2 = A2 * points[best_bottom].x + points[best_bottom].y * B2;
        float idet = 1.f / (A1 * B2 - A2 * B1);
        out[0] = (C1 * B2 - C2 * B1) * idet;
        out[1] = (A1 * C2 - A2 * C1) * idet;
        out[2] = A1 * best_w; out[3] = B1 * best_w;
        out[4] = A2 * best_h; out[5] = B2 * best_h;
    }
    free(inv_len);
}

#define YO_PI 3.1415926535897932384626433832795

/* rect = {cx, cy, w, h, angle_deg}; pts may be reordered. */
void yo_min_area_rect(yo_pt *pts, int n, float *rect)
{
    yo_pt *hull = (yo_pt *)malloc(sizeof(yo_pt) * (size_t)(n > 0 ? n : 1));
    int hn = n > 0 ? hull_cv_order(pts, n, hull) : 0;
    float cx = 0, cy = 0, bw = 0, bh = 0, ang = 0;
    if (hn > 2) {
        float o[6];
        rotating_calipers(hull, hn, o);
        cx = o[0] + (o[2] + o[4]) * 0.5f;
        cy = o[1] + (o[3] + o[5]) * 0.5f;
        bw = (float)sqrt((double)o[2] * o[2] + (double)o[3] * o[3]);
        bh = (float)sqrt((double)o[4] * o[4] + (double)o[5] * o[5]);
        ang = (float)atan2((double)o[3], (double)o[2]);
    } else if (hn == 2) {
        cx = (hull[0].x + hull[1].x) * 0.5f;
        cy = (hull[0].y + hull[1].y) * 0.5f;
        double dx = hull[1].x - hull[0].x, dy = hull[1].y - hull[0].y;
        bw = (float)sqrt(dx * dx + dy * dy);
        bh = 0;
        ang = (float)atan2(dy, dx);
    } else if (hn == 1) {
        cx = hull[0].x; cy = hull[0].y;
    }
    ang = (float)((double)(ang * 180.f) / YO_PI);
    rect[0] = cx; rect[1] = cy; rect[2] = bw; rect[3] = bh; rect[4] = ang;
    free(hull);
}

/* Python-facing helper: rect of n (x,y) int32 pairs. */
void yo_min_area_rect_xy(const int32_t *xy, int n, float *rect)
{
    yo_pt *p = (yo_pt *)malloc(sizeof(yo_pt) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { p[i].x = (float)xy[2 * i]; p[i].y = (float)xy[2 * i + 1]; }
    yo_min_area_rect(p, n, rect);
    free(p);
}

/* ------------------------------------------------------------------------------------------ */
/* a5 + a6 from a final foreground mask `fg` (0 / non-zero): labels (canonical), detections in
 * reverse raster order of the component's first pixel, nested components skipped.
 * det = max_det x {cx, cy, w, h, angle}.  Returns the number of detections (may exceed max_det,
 * in which case only the first max_det are written).  anchors (optional) receives the 0-based
 * raster index of each emitted detection's first pixel. */
int yo_components(const uint8_t *fg, int h, int w, int32_t *labels, float *det, int32_t *anchors,
                  int max_det)
{
    long n = (long)h * w;
    yo_label8(fg, h, w, labels);
    uint8_t *outside = (uint8_t *)malloc((size_t)n);
    flood_outside(fg, h, w, outside);
    /* per-component pixel counts -> offsets, in raster order of anchors */
    int32_t *count = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    for (long p = 0; p < n; ++p)
        if (labels[p]) count[labels[p] - 1]++;
    int ndet = 0;
    yo_pt *pts = NULL;
    long cap = 0;
    for (long a = n - 1; a >= 0; --a) {
        if (count[a] == 0) continue;
        int ax = (int)(a % w);
        if (ax > 0 && !outside[a - 1]) continue; /* nested in a hole of another component */
        int c = count[a];
        if (c > cap) { cap = c; pts = (yo_pt *)realloc(pts, sizeof(yo_pt) * (size_t)cap); }
        /* the component lies at raster indices >= a; stop once all c pixels are collected */
        int got = 0;
        for (long p = a; p < n && got < c; ++p)
            if (labels[p] == (int32_t)(a + 1)) {
                pts[got].x = (float)(p % w);
                pts[got].y = (float)(p / w);
                ++got;
            }
        if (ndet < max_det) {
            yo_min_area_rect(pts, c, det + 5 * (long)ndet);
            if (anchors) anchors[ndet] = (int32_t)a;
        }
        ++ndet;
    }
    free(pts);
    free(count);
    free(outside);
    return ndet;
}

/* Whole image half for one frame (the body of the reference loop, track_eval.py:180-303).
 * frame: h x w x channels (1 or 3) u8.  Optional outputs may be NULL. */
int yo_detect_frame(const uint8_t *frame, int h, int w, int channels, int inv, int t_low,
                    int t_high, int use_high, uint8_t *cls_out, uint8_t *mask_out,
                    int32_t *labels_out, float *det, int32_t *anchors, int max_det)
{
    long n = (long)h * w;
    uint8_t *gray = (uint8_t *)malloc((size_t)n);
    uint8_t *blur = (uint8_t *)malloc((size_t)n);
    uint8_t *mean = (uint8_t *)malloc((size_t)n);
    uint8_t *cls = (uint8_t *)malloc((size_t)n);
    uint8_t *fg = (uint8_t *)malloc((size_t)n);
    int32_t *labels = labels_out ? labels_out : (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (channels == 3) yo_bgr2gray(frame, h, w, gray);
    else memcpy(gray, frame, (size_t)n);
    yo_blur3(gray, h, w, blur);
    yo_adaptive_mean(blur, h, w, mean);
    yo_classify(blur, mean, n, inv, t_low, t_high, use_high, cls);
    if (use_high) yo_propagate(cls, h, w, fg);
    else for (long i = 0; i < n; ++i) fg[i] = cls[i] & 1;
    int nd = yo_components(fg, h, w, labels, det, anchors, max_det);
    if (cls_out) memcpy(cls_out, cls, (size_t)n);
    if (mask_out) for (long i = 0; i < n; ++i) mask_out[i] = fg[i] ? 255 : 0;
    if (!labels_out) free(labels);
    free(gray); free(blur); free(mean); free(cls); free(fg);
    return nd;
}
