// libysmr_hip -- the output side of the path: the rows the link emits frame by frame
// are ordered by (TRACK_ID, POSITION_T) on the device and written as the csv the reference ends up
// with.  The reference appends Python-formatted text per frame (track_eval.py:313-316, 340-346,
// helper_file.py:1403-1478), then re-reads the file with pandas, sorts it and rewrites it with
// DataFrame.to_csv (helper_file.py:1538-1574, 1366-1400).  The final file and DataFrame are a pure
// function of the row values -- shortest-repr text, pandas' float parser (not correctly rounded, see
// pandas_parse below), shortest-repr text again -- so they are produced here directly, without the
// Python string formatting (0.5 s per 250 k rows) and the two pandas passes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include <vector>

#include "common.h"
#include "prim.h"
#include "fmt.h"

namespace {

// ---- ordering by (TRACK_ID, POSITION_T) ------------------------------------------------------------------
// The link emits rows frame by frame, ids ascending within a frame; an id is never reused and a live track has
// a row in EVERY frame from its first to its last (disappeared tracks included, track_eval.py:313-316).  So the
// place of row (id, f) in the ordered table is known without sorting anything:
//     offset[id] + (f - first_frame[id]),   offset = exclusive prefix sum of the rows per track.
// One pass gathers rows per track and first/last frame (atomics on 12 bytes per track), a prefix sum turns the
// counts into offsets, one pass checks that every track's frames are gapless and scatters.  A table that does
// not have this shape (ids beyond the row count, gaps, duplicates -- e.g. rows assembled by a caller) is
// reported by the checks (per track: span = rows; per place: taken exactly once) and goes through the stable radix
// sort of prim.h instead.
struct TrackSpan { uint32_t rows, first, last; };

__global__ __launch_bounds__(256) void k_span_clear(TrackSpan *span, uint32_t *counts, uint32_t *written, long long n, uint32_t *bad)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        span[i] = TrackSpan{0u, 0xFFFFFFFFu, 0u};
        counts[i] = 0u;
        written[i] = 0u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *bad = 0u;
}

__global__ __launch_bounds__(256) void k_span_gather(const ysmr_row *__restrict__ rows, long long n, TrackSpan *span,
                                                     uint32_t *counts, uint32_t *bad)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const uint32_t id = (uint32_t)rows[i].track_id, f = (uint32_t)rows[i].frame;
        if ((long long)id >= n) { *bad = 1u; continue; }       // (every track has at least one row, ids start at 0)
        atomicAdd(&counts[id], 1u);
        atomicMin(&span[id].first, f);
        atomicMax(&span[id].last, f);
    }
}

// counts_incl: inclusive prefix sum of the rows per track
__global__ __launch_bounds__(256) void k_span_scatter(const ysmr_row *__restrict__ rows, long long n, const TrackSpan *span,
                                                      const uint32_t *__restrict__ counts_incl, uint32_t *bad,
                                                      uint32_t *written, ysmr_row *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const ysmr_row r = rows[i];
        const uint32_t id = (uint32_t)r.track_id;
        if ((long long)id >= n) continue;
        const uint32_t end = counts_incl[id], begin = id ? counts_incl[id - 1] : 0u;
        const TrackSpan s = span[id];
        if (s.last - s.first + 1u != end - begin) { *bad = 1u; continue; }   // a gap or a duplicate (frame, id) pair
        // (a duplicate AND a gap in one track cancel in that count -- frames 0, 1, 1, 3 -- but then two rows want one
        // place and another stays empty: every place must be taken exactly once)
        const uint32_t place = begin + ((uint32_t)r.frame - s.first);
        if (atomicAdd(&written[place], 1u)) { *bad = 1u; continue; }
        out[place] = r;
    }
}

__global__ __launch_bounds__(256) void k_row_keys(const ysmr_row *__restrict__ rows, long long n,
                                                  unsigned long long *__restrict__ keys, uint32_t *__restrict__ idx)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        // TRACK_ID and POSITION_T are uint32 columns in the reference (helper_file.py:881-889)
        keys[i] = ((unsigned long long)(uint32_t)rows[i].track_id << 32) | (uint32_t)rows[i].frame;
        idx[i] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_row_gather(const ysmr_row *__restrict__ rows, const uint32_t *__restrict__ idx,
                                                    long long n, ysmr_row *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = rows[idx[i]];
}

struct SortLayout {
    size_t span, counts, bad, keys_a, keys_b, idx_a, idx_b, temp, total;
};

SortLayout sort_layout(long long n)
{
    SortLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = ysmr::align_up(off + bytes, 256); return o; };
    L.bad = take(256);
    // the two orderings never run at the same time: their buffers share the workspace
    L.span = take(sizeof(TrackSpan) * (size_t)n);
    L.counts = take(sizeof(uint32_t) * (size_t)n);
    const size_t after_span = off;
    off = L.span;
    L.keys_a = take(sizeof(unsigned long long) * (size_t)n);
    L.keys_b = take(sizeof(unsigned long long) * (size_t)n);
    L.idx_a = take(sizeof(uint32_t) * (size_t)n);
    L.idx_b = take(sizeof(uint32_t) * (size_t)n);
    off = std::max(off, after_span);
    L.temp = take(std::max(ysmr::prim::radix_temp_bytes((size_t)n), sizeof(uint32_t) * ysmr::prim::scan_temp_words((size_t)n)));
    L.total = off;
    return L;
}

// ---- float formatting: what str(numpy.float64) / repr(float) print --------------------------------------
// shortest digits that round-trip (std::to_chars), laid out by CPython's rules (format_float_short, 'r'):
// exponent form iff the decimal exponent is < -4 or >= 16, otherwise positional with at least ".0".
char *put_float(char *p, double v)
{
    if (std::isnan(v)) return p;                                   // pandas writes NaN as an empty field
    if (std::isinf(v)) { const char *s = v < 0 ? "-inf" : "inf"; size_t k = std::strlen(s); std::memcpy(p, s, k); return p + k; }
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (v == 0.0) { std::memcpy(p, "0.0", 3); return p + 3; }
    char buf[40];
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);   // d[.ddd]e[+-]XX
    char *e = buf;
    while (*e != 'e') ++e;
    char digits[24];
    int nd = 0;
    for (char *q = buf; q < e; ++q)
        if (*q != '.') digits[nd++] = *q;
    int exp10 = 0;
    {
        const char *q = e + 1;
        const bool neg = (*q == '-');
        ++q;
        for (; q < r.ptr; ++q) exp10 = exp10 * 10 + (*q - '0');
        if (neg) exp10 = -exp10;
    }
    if (exp10 < -4 || exp10 >= 16) {
        *p++ = digits[0];
        if (nd > 1) { *p++ = '.'; std::memcpy(p, digits + 1, (size_t)nd - 1); p += nd - 1; }
        *p++ = 'e';
        *p++ = exp10 < 0 ? '-' : '+';
        int a = exp10 < 0 ? -exp10 : exp10;
        if (a >= 100) { *p++ = (char)('0' + a / 100); a %= 100; *p++ = (char)('0' + a / 10); *p++ = (char)('0' + a % 10); }
        else { *p++ = (char)('0' + a / 10); *p++ = (char)('0' + a % 10); }
        return p;
    }
    const int decpt = exp10 + 1;   // digits before the decimal point
    if (decpt <= 0) {
        *p++ = '0'; *p++ = '.';
        for (int k = 0; k < -decpt; ++k) *p++ = '0';
        std::memcpy(p, digits, (size_t)nd); p += nd;
    } else if (decpt >= nd) {
        std::memcpy(p, digits, (size_t)nd); p += nd;
        for (int k = nd; k < decpt; ++k) *p++ = '0';
        *p++ = '.'; *p++ = '0';
    } else {
        std::memcpy(p, digits, (size_t)decpt); p += decpt;
        *p++ = '.';
        std::memcpy(p, digits + decpt, (size_t)(nd - decpt)); p += nd - decpt;
    }
    return p;
}

// The reference never keeps the tracker's float64 values: it writes them as text and reads the file
// back with pandas.read_csv (helper_file.py:860-905, called from sort_list), whose default float
// converter (pandas >= 1.2: "high" = precise_xstrtod in pandas/_libs/src/parser/tokenizer.c) is not
// correctly rounded -- about one value in five comes back 1 ulp off.  To hand over the SAME DataFrame
// and the same final csv, that converter is restated here: at most 17 digits (leading zeros included)
// are accumulated in a double, the rest only shifts the exponent, and the result is scaled by one
// multiplication or division with a table power of ten.  Checked against the installed pandas in
// tests/test_host.py on several hundred thousand values.
double pandas_parse(const char *p, const char *end)
{
    static const double e[] = {
        1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21,
        1e22, 1e23, 1e24, 1e25, 1e26, 1e27, 1e28, 1e29, 1e30, 1e31, 1e32, 1e33, 1e34, 1e35, 1e36, 1e37, 1e38, 1e39, 1e40, 1e41,
        1e42, 1e43, 1e44, 1e45, 1e46, 1e47, 1e48, 1e49, 1e50, 1e51, 1e52, 1e53, 1e54, 1e55, 1e56, 1e57, 1e58, 1e59, 1e60, 1e61,
        1e62, 1e63, 1e64, 1e65, 1e66, 1e67, 1e68, 1e69, 1e70, 1e71, 1e72, 1e73, 1e74, 1e75, 1e76, 1e77, 1e78, 1e79, 1e80, 1e81,
        1e82, 1e83, 1e84, 1e85, 1e86, 1e87, 1e88, 1e89, 1e90, 1e91, 1e92, 1e93, 1e94, 1e95, 1e96, 1e97, 1e98, 1e99, 1e100, 1e101,
        1e102, 1e103, 1e104, 1e105, 1e106, 1e107, 1e108, 1e109, 1e110, 1e111, 1e112, 1e113, 1e114, 1e115, 1e116, 1e117, 1e118,
        1e119, 1e120, 1e121, 1e122, 1e123, 1e124, 1e125, 1e126, 1e127, 1e128, 1e129, 1e130, 1e131, 1e132, 1e133, 1e134, 1e135,
        1e136, 1e137, 1e138, 1e139, 1e140, 1e141, 1e142, 1e143, 1e144, 1e145, 1e146, 1e147, 1e148, 1e149, 1e150, 1e151, 1e152,
        1e153, 1e154, 1e155, 1e156, 1e157, 1e158, 1e159, 1e160, 1e161, 1e162, 1e163, 1e164, 1e165, 1e166, 1e167, 1e168, 1e169,
        1e170, 1e171, 1e172, 1e173, 1e174, 1e175, 1e176, 1e177, 1e178, 1e179, 1e180, 1e181, 1e182, 1e183, 1e184, 1e185, 1e186,
        1e187, 1e188, 1e189, 1e190, 1e191, 1e192, 1e193, 1e194, 1e195, 1e196, 1e197, 1e198, 1e199, 1e200, 1e201, 1e202, 1e203,
        1e204, 1e205, 1e206, 1e207, 1e208, 1e209, 1e210, 1e211, 1e212, 1e213, 1e214, 1e215, 1e216, 1e217, 1e218, 1e219, 1e220,
        1e221, 1e222, 1e223, 1e224, 1e225, 1e226, 1e227, 1e228, 1e229, 1e230, 1e231, 1e232, 1e233, 1e234, 1e235, 1e236, 1e237,
        1e238, 1e239, 1e240, 1e241, 1e242, 1e243, 1e244, 1e245, 1e246, 1e247, 1e248, 1e249, 1e250, 1e251, 1e252, 1e253, 1e254,
        1e255, 1e256, 1e257, 1e258, 1e259, 1e260, 1e261, 1e262, 1e263, 1e264, 1e265, 1e266, 1e267, 1e268, 1e269, 1e270, 1e271,
        1e272, 1e273, 1e274, 1e275, 1e276, 1e277, 1e278, 1e279, 1e280, 1e281, 1e282, 1e283, 1e284, 1e285, 1e286, 1e287, 1e288,
        1e289, 1e290, 1e291, 1e292, 1e293, 1e294, 1e295, 1e296, 1e297, 1e298, 1e299, 1e300, 1e301, 1e302, 1e303, 1e304, 1e305,
        1e306, 1e307, 1e308};
    bool neg = false;
    if (p < end && *p == '-') { neg = true; ++p; }
    double number = 0.0;
    int exponent = 0, nd = 0, ndec = 0;
    const int max_digits = 17;
    auto digit = [&](const char *q) { return q < end && *q >= '0' && *q <= '9'; };
    while (digit(p)) {
        if (nd < max_digits) { number = number * 10.0 + (*p - '0'); ++nd; }
        else ++exponent;
        ++p;
    }
    if (p < end && *p == '.') {
        ++p;
        while (nd < max_digits && digit(p)) { number = number * 10.0 + (*p - '0'); ++p; ++nd; ++ndec; }
        if (nd >= max_digits)
            while (digit(p)) ++p;
        exponent -= ndec;
    }
    if (neg) number = -number;
    if (p < end && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < end && *p == '-') { eneg = true; ++p; }
        else if (p < end && *p == '+') ++p;
        int n = 0;
        while (digit(p)) { n = n * 10 + (*p - '0'); ++p; }
        exponent += eneg ? -n : n;
    }
    if (exponent > 308) return neg ? -HUGE_VAL : HUGE_VAL;
    if (exponent > 0) number *= e[exponent];
    else if (exponent < -308) {
        if (exponent < -616) number = 0.0;
        else { number /= e[-308 - exponent]; number /= e[308]; }
    } else number /= e[-exponent];
    return number;
}

// value -> text -> pandas -> value
double pandas_roundtrip(double v)
{
    if (!std::isfinite(v)) return v;
    char buf[48];
    char *end = put_float(buf, v);
    return pandas_parse(buf, end);
}

// text of the value pandas reads back from v's text, written at p; *v becomes that value.  Four values in five come back
// unchanged: their first text is the final one, and printing a float is the expensive half of a row (round 5: a third of
// the formatting time)
char *put_float_via_pandas(char *p, double *v)
{
    if (!std::isfinite(*v)) return put_float(p, *v);
    char *end = put_float(p, *v);
    const double back = pandas_parse(p, end);
    if (back == *v && !(back == 0.0 && std::signbit(back) != std::signbit(*v))) return end;
    *v = back;
    return put_float(p, back);
}

char *put_u32(char *p, uint32_t v)
{
    auto r = std::to_chars(p, p + 12, v);
    return r.ptr;
}

constexpr size_t ROW_TEXT_MAX = 2 * 11 + 5 * 26 + 8;   // two uint32, five floats, separators
const char CSV_HEADER[] = "TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE\n";   // helper_file.py:1451

// (cols: the DataFrame's columns, filled in the same pass when asked for -- the values the csv is printed from ARE the
//  values pandas would read back, and working them out once instead of once per consumer is a third of the tail's CPU time)
struct ColumnSinks { uint32_t *track_id, *t; double *v[5]; };
size_t format_range(const ysmr_row *rows, long long lo, long long hi, bool via_pandas, char *out, const ColumnSinks *cols = nullptr)
{
    char *p = out;
    for (long long i = lo; i < hi; ++i) {
        const ysmr_row &r = rows[i];
        double v[5] = {r.x, r.y, (double)r.w, (double)r.h, (double)r.angle};
        p = put_u32(p, (uint32_t)r.track_id); *p++ = ',';
        p = put_u32(p, (uint32_t)r.frame); *p++ = ',';
        for (int k = 0; k < 5; ++k) { p = via_pandas ? put_float_via_pandas(p, &v[k]) : put_float(p, v[k]); *p++ = k == 4 ? '\n' : ','; }
        if (cols) {
            cols->track_id[i] = (uint32_t)r.track_id; cols->t[i] = (uint32_t)r.frame;
            for (int k = 0; k < 5; ++k) cols->v[k][i] = v[k];
        }
    }
    return (size_t)(p - out);
}


// ---- the same text and columns on the DEVICE (round 5: file -> rows; fmt.h) ------------------------------------------
// A thread per row prints it into a slot of ROW_TEXT_MAX bytes (and the DataFrame's columns: the values pandas reads back),
// an inclusive scan of the lengths places the rows, a second kernel packs them behind the header.  A row that holds a value
// fmt.h does not serve (NaN, infinities, |v| outside 2^-20 .. 2^24 other than zero) is counted in *unserved: the caller then
// takes the host path for the whole table (tracks do not produce such values; tests do).
__device__ const char FMT_HEADER[] = "TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE\n";
struct DevColumns { uint32_t *track_id, *t; double *v[5]; };

__device__ __forceinline__ uint32_t fmt_row(const ysmr_row &r, bool via_pandas, char *out, double (&v)[5], bool &ok)
{
    char *p = out;
    p = ysmr_fmt::put_u32(p, (uint32_t)r.track_id); *p++ = ',';
    p = ysmr_fmt::put_u32(p, (uint32_t)r.frame); *p++ = ',';
    v[0] = r.x; v[1] = r.y; v[2] = (double)r.w; v[3] = (double)r.h; v[4] = (double)r.angle;
#pragma unroll 1
    for (int k = 0; k < 5; ++k) { p = ysmr_fmt::put_value(p, &v[k], via_pandas, ok); *p++ = k == 4 ? '\n' : ','; }
    return (uint32_t)(p - out);
}

__global__ __launch_bounds__(256) void k_fmt_rows(const ysmr_row *__restrict__ rows, long long n, int via_pandas, char *slots,
                                                  uint32_t *len, DevColumns cols, uint32_t *unserved)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const ysmr_row r = rows[i];
        double v[5];
        bool ok = true;
        const uint32_t l = fmt_row(r, via_pandas != 0, slots + (size_t)i * ROW_TEXT_MAX, v, ok);
        len[i] = ok ? l : 0u;
        if (!ok) atomicAdd(unserved, 1u);
        cols.track_id[i] = (uint32_t)r.track_id; cols.t[i] = (uint32_t)r.frame;
#pragma unroll
        for (int k = 0; k < 5; ++k) cols.v[k][i] = v[k];
    }
}

// (a wave per 64 rows: lane b of a row's turn copies byte b, b + 64, ... -- coalesced writes into the packed text)
__global__ __launch_bounds__(256) void k_fmt_pack(const char *__restrict__ slots, const uint32_t *__restrict__ len,
                                                  const uint32_t *__restrict__ incl, long long n, int with_header, char *csv,
                                                  unsigned long long *csv_length)
{
    const int head = with_header ? (int)sizeof(FMT_HEADER) - 1 : 0;
    if (blockIdx.x == 0 && threadIdx.x < head) csv[threadIdx.x] = FMT_HEADER[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) *csv_length = (unsigned long long)head + (n ? incl[n - 1] : 0u);
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * 4, wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (long long i0 = wave * 64; i0 < n; i0 += waves * 64) {
        const long long mine = i0 + lane;
        const uint32_t l_mine = mine < n ? len[mine] : 0u, e_mine = mine < n ? incl[mine] : 0u;
        for (int k = 0; k < 64 && i0 + k < n; ++k) {
            const uint32_t l = (uint32_t)__shfl((int)l_mine, k), at = (uint32_t)__shfl((int)e_mine, k) - l;
            const char *src = slots + (size_t)(i0 + k) * ROW_TEXT_MAX;
            for (uint32_t b = (uint32_t)lane; b < l; b += 64u) csv[(size_t)head + at + b] = src[b];
        }
    }
}

struct FmtLayout { size_t slots, len, incl, temp, total; };
FmtLayout fmt_layout(long long n)
{
    FmtLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off = ysmr::align_up(off + bytes, 256); return at; };
    L.slots = take((size_t)n * ROW_TEXT_MAX);
    L.len = take(sizeof(uint32_t) * (size_t)n);
    L.incl = take(sizeof(uint32_t) * (size_t)n);
    L.temp = take(sizeof(uint32_t) * ysmr::prim::scan_temp_words((size_t)n));
    L.total = off;
    return L;
}

}  // namespace

extern "C" {

size_t ysmr_rows_sort_workspace_bytes(long long n_rows)
{
    if (n_rows <= 0 || n_rows > 0x7FFFFFFFll) return 0;
    return sort_layout(n_rows).total;
}

int ysmr_rows_sort(void *stream, const ysmr_row *rows_dev, long long n_rows, void *workspace_dev, size_t workspace_bytes,
                   ysmr_row *sorted_dev)
{
    if (n_rows == 0) return YSMR_OK;
    if (n_rows < 0 || n_rows > 0x7FFFFFFFll) return ysmr::fail(YSMR_ERR_ARG, "n_rows must be in 0..2^31-1, got %lld", n_rows);
    if (!rows_dev || !sorted_dev || !workspace_dev || rows_dev == sorted_dev)
        return ysmr::fail(YSMR_ERR_ARG, "rows_dev, sorted_dev (distinct) and workspace_dev must be set");
    const SortLayout L = sort_layout(n_rows);
    if (workspace_bytes < L.total)
        return ysmr::fail(YSMR_ERR_CAPACITY, "sort workspace too small: %zu < %zu bytes", workspace_bytes, L.total);
    char *w = (char *)workspace_dev;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)std::min<long long>((n_rows + 255) / 256, 1024);   // resident grid (see detect.hip)
    // the tracker's table: no sort needed
    TrackSpan *span = (TrackSpan *)(w + L.span);
    uint32_t *counts = (uint32_t *)(w + L.counts), *bad = (uint32_t *)(w + L.bad);
    uint32_t *written = (uint32_t *)(w + L.idx_a);   // (the fallback's buffer: not in use yet)
    hipLaunchKernelGGL(k_span_clear, dim3(grid), dim3(256), 0, st, span, counts, written, n_rows, bad);
    hipLaunchKernelGGL(k_span_gather, dim3(grid), dim3(256), 0, st, rows_dev, n_rows, span, counts, bad);
    ysmr::prim::inclusive_scan_u32(st, counts, counts, (size_t)n_rows, (uint32_t *)(w + L.temp));
    hipLaunchKernelGGL(k_span_scatter, dim3(grid), dim3(256), 0, st, rows_dev, n_rows, span, counts, bad, written, sorted_dev);
    YSMR_LAUNCH_CHECK();
    uint32_t irregular = 0;
    YSMR_HIP_CHECK(hipMemcpyAsync(&irregular, bad, sizeof(irregular), hipMemcpyDeviceToHost, st));
    YSMR_HIP_CHECK(hipStreamSynchronize(st));
    if (!irregular) return YSMR_OK;
    // any other table: stable radix sort of (TRACK_ID << 32 | POSITION_T) with the row number as payload
    auto *keys_a = (unsigned long long *)(w + L.keys_a), *keys_b = (unsigned long long *)(w + L.keys_b);
    auto *idx_a = (uint32_t *)(w + L.idx_a), *idx_b = (uint32_t *)(w + L.idx_b);
    hipLaunchKernelGGL(k_row_keys, dim3(grid), dim3(256), 0, st, rows_dev, n_rows, keys_a, idx_a);
    const int where = ysmr::prim::radix_sort(st, keys_a, keys_b, idx_a, idx_b, (size_t)n_rows, 64, w + L.temp);
    hipLaunchKernelGGL(k_row_gather, dim3(grid), dim3(256), 0, st, rows_dev, where ? idx_b : idx_a, n_rows, sorted_dev);
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

size_t ysmr_rows_csv_bound(long long n_rows, int with_header)
{
    if (n_rows < 0) return 0;
    return (size_t)n_rows * ROW_TEXT_MAX + (with_header ? sizeof(CSV_HEADER) - 1 : 0) + 1;
}

int ysmr_rows_format_csv(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas, int threads,
                         char *out, size_t out_capacity, size_t *out_length)
{
    if (n_rows < 0 || (!rows_host && n_rows) || !out || !out_length)
        return ysmr::fail(YSMR_ERR_ARG, "rows_host, out and out_length must be set");
    if (out_capacity < ysmr_rows_csv_bound(n_rows, with_header))
        return ysmr::fail(YSMR_ERR_CAPACITY, "csv buffer too small: %zu < %zu bytes", out_capacity,
                          ysmr_rows_csv_bound(n_rows, with_header));
    size_t len = 0;
    if (with_header) { std::memcpy(out, CSV_HEADER, sizeof(CSV_HEADER) - 1); len = sizeof(CSV_HEADER) - 1; }
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, 64));
    if (n_rows < 4096) nt = 1;
    if (nt == 1) {
        len += format_range(rows_host, 0, n_rows, via_pandas != 0, out + len);
    } else {
        // each thread formats a contiguous range into the slot reserved for it, then the pieces are packed
        std::vector<size_t> used((size_t)nt, 0);
        std::vector<std::thread> pool;
        const long long per = (n_rows + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            const long long lo = std::min<long long>((long long)t * per, n_rows), hi = std::min<long long>(lo + per, n_rows);
            char *dst = out + len + (size_t)lo * ROW_TEXT_MAX;
            pool.emplace_back([=, &used] { used[(size_t)t] = format_range(rows_host, lo, hi, via_pandas != 0, dst); });
        }
        for (auto &th : pool) th.join();
        char *w = out + len;
        for (int t = 0; t < nt; ++t) {
            const long long lo = std::min<long long>((long long)t * per, n_rows);
            const char *src = out + len + (size_t)lo * ROW_TEXT_MAX;
            if (w != src) std::memmove(w, src, used[(size_t)t]);
            w += used[(size_t)t];
        }
        len = (size_t)(w - out);
    }
    *out_length = len;
    return YSMR_OK;
}

// The same text straight into a file: every thread formats its range of rows into a buffer of its own, learns where its
// piece goes once all pieces are sized, and writes it there itself (pwrite).  The one-buffer form above, written out by
// the caller, spent more time packing the pieces (one memmove over 80 MB) and in ONE thread's write() than formatting.
static int write_csv_impl(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas, int threads,
                          const char *path, size_t *out_length, const ColumnSinks *cols);

int ysmr_rows_write_csv(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas, int threads,
                        const char *path, size_t *out_length)
{
    return write_csv_impl(rows_host, n_rows, with_header, via_pandas, threads, path, out_length, nullptr);
}

int ysmr_rows_write_csv_columns(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas, int threads,
                                const char *path, size_t *out_length, uint32_t *track_id, uint32_t *t, double *x, double *y,
                                double *w, double *h, double *angle)
{
    if (n_rows && (!track_id || !t || !x || !y || !w || !h || !angle))
        return ysmr::fail(YSMR_ERR_ARG, "all seven column pointers must be set");
    const ColumnSinks cols{track_id, t, {x, y, w, h, angle}};
    return write_csv_impl(rows_host, n_rows, with_header, via_pandas, threads, path, out_length, &cols);
}

static int write_csv_impl(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas, int threads,
                          const char *path, size_t *out_length, const ColumnSinks *cols)
{
    if (n_rows < 0 || (!rows_host && n_rows) || !path)
        return ysmr::fail(YSMR_ERR_ARG, "rows_host and path must be set");
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return ysmr::fail(YSMR_ERR_ARG, "cannot open %s for writing: %s", path, std::strerror(errno));
    std::atomic<int> write_errno{0};
    auto write_all = [fd, &write_errno](const char *p, size_t n, off_t at) {
        while (n) {
            const ssize_t w = ::pwrite(fd, p, n, at);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) { write_errno.store(w < 0 ? errno : ENOSPC); return false; }     // (errno where it was set)
            p += w; n -= (size_t)w; at += w;
        }
        return true;
    };
    const size_t head = with_header ? sizeof(CSV_HEADER) - 1 : 0;
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, 64));
    if (n_rows < 4096) nt = 1;
    const long long per = (n_rows + nt - 1) / nt;
    std::vector<size_t> used((size_t)nt, 0);
    std::vector<std::unique_ptr<char[]>> piece((size_t)nt);
    std::atomic<int> sized{0}, failed{0};
    std::mutex mu;
    std::condition_variable cv;
    // a piece is formatted, then -- once every piece knows its length -- written at its own offset by whoever formatted it
    auto format_piece = [&](int t) {
        const long long lo = std::min<long long>((long long)t * per, n_rows), hi = std::min<long long>(lo + per, n_rows);
        piece[(size_t)t].reset(new (std::nothrow) char[(size_t)(hi - lo) * ROW_TEXT_MAX + 1]);
        if (piece[(size_t)t]) used[(size_t)t] = format_range(rows_host, lo, hi, via_pandas != 0, piece[(size_t)t].get(), cols);
        else failed.store(1);
        std::unique_lock<std::mutex> lk(mu);
        if (sized.fetch_add(1) + 1 == nt) cv.notify_all();
    };
    auto all_sized = [&]() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return sized.load() == nt; });
    };
    auto write_piece = [&](int t) {
        if (failed.load()) return;
        size_t at = head;
        for (int k = 0; k < t; ++k) at += used[(size_t)k];
        if (t == 0 && head && !write_all(CSV_HEADER, head, 0)) failed.store(2);
        if (!write_all(piece[(size_t)t].get(), used[(size_t)t], (off_t)at)) failed.store(2);
        piece[(size_t)t].reset();
    };
    {
        // Threads that cannot be started (EAGAIN under a pids cgroup) leave their pieces to the caller's thread: every
        // piece is formatted by somebody, so the wait above always ends (a throw out of emplace_back with
        // workers already waiting ended in std::terminate inside a ctypes call).
        std::vector<std::thread> pool;
        int started = 0;
        if (nt > 1) {
            try {
                for (; started < nt - 1; ++started)
                    pool.emplace_back([&, started] { format_piece(started); all_sized(); write_piece(started); });
            } catch (const std::system_error &) {
            }
        }
        for (int t = started; t < nt; ++t) format_piece(t);
        all_sized();
        for (int t = started; t < nt; ++t) write_piece(t);
        for (auto &th : pool) th.join();
    }
    size_t len = head;
    for (size_t u : used) len += u;
    const int rc_close = ::close(fd);
    const int close_errno = rc_close != 0 ? errno : 0;
    if (failed.load() || rc_close != 0) ::unlink(path);            // (no truncated csv under the list's name)
    if (failed.load() == 1) return ysmr::fail(YSMR_ERR_CAPACITY, "out of memory formatting %lld rows", n_rows);
    if (failed.load() == 2 || rc_close != 0)
        return ysmr::fail(YSMR_ERR_ARG, "writing %s failed: %s", path, std::strerror(failed.load() == 2 ? write_errno.load() : close_errno));
    if (out_length) *out_length = len;
    return YSMR_OK;
}

// ---- the same csv and columns, worked out WHILE the video runs (ABI 13) -----------------------------------------------
// track_bacteria's tail -- rows to the host, 27 ms of formatting, the csv -- used to start when the last frame was linked
// (39 of a 1920-frame file's 97 ms).  A row's text and the value pandas reads back from it do not depend on any
// other row, only its PLACE in the file does: the stream takes each batch's rows as the link emits them (frame-major),
// formats them on its own threads while later batches run, and at the end only orders what it has -- a row's place is
// offset[id] + frame - first_frame[id], the tracker's invariant that ysmr_rows_sort uses on the device; tables without it go
// through a stable sort -- and gathers lines and columns into the file and the caller's arrays.
// One cache line per row: what the ordered table needs of it -- its key, the five values the text was printed from (read back
// through pandas' parser), and where its line lies.  (A first version kept the rows, the values as five arrays and the line
// ends per task: ordering 971 k rows then cost 57 ms, seven random cache lines per row.)
struct alignas(64) StreamRec {
    uint32_t id, frame;
    double v[5];
    const char *text;
    uint16_t len;
};
static_assert(sizeof(StreamRec) == 64, "one line per row");
struct StreamTask {
    long long lo = 0, hi = 0;                   // records [lo, hi) of the chunk
    std::unique_ptr<char[]> text;               // their lines, back to back
};
struct StreamChunk {
    std::unique_ptr<StreamRec[]> rec;
    long long n = 0;
    std::vector<StreamTask> tasks;
    const ysmr_row *src = nullptr;              // the caller's rows while push() is running
};
}  // extern "C"

struct ysmr_rows_stream {
    int via_pandas = 1;
    std::vector<std::thread> pool;
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::vector<std::pair<StreamChunk *, int>> queue;      // (chunk, task index) waiting for a thread
    size_t queue_head = 0;
    int busy = 0;
    bool closing = false, failed = false;
    std::vector<std::unique_ptr<StreamChunk>> chunks;
    long long n_rows = 0;
    std::vector<uint32_t> count, first;                    // per id: rows, first frame (kept up to date by push)

    void work()
    {
        for (;;) {
            std::pair<StreamChunk *, int> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return closing || queue_head < queue.size(); });
                if (queue_head >= queue.size()) return;          // closing, nothing left
                job = queue[queue_head++];
                ++busy;
            }
            StreamChunk &c = *job.first;
            StreamTask &t = c.tasks[(size_t)job.second];
            bool ok = true;
            try {
                std::unique_ptr<char[]> big(new char[(size_t)(t.hi - t.lo) * ROW_TEXT_MAX + 1]);
                std::vector<uint32_t> start((size_t)(t.hi - t.lo) + 1);
                char *p = big.get();
                for (long long i = t.lo; i < t.hi; ++i) {
                    StreamRec &r = c.rec[(size_t)i];
                    start[(size_t)(i - t.lo)] = (uint32_t)(p - big.get());
                    p = put_u32(p, r.id); *p++ = ',';
                    p = put_u32(p, r.frame); *p++ = ',';
                    for (int k = 0; k < 5; ++k) { p = via_pandas ? put_float_via_pandas(p, &r.v[k]) : put_float(p, r.v[k]); *p++ = k == 4 ? '\n' : ','; }
                }
                const size_t used = (size_t)(p - big.get());
                start[(size_t)(t.hi - t.lo)] = (uint32_t)used;
                t.text.reset(new char[used ? used : 1]);                 // (the worst-case buffer is 4 x what the lines take)
                std::memcpy(t.text.get(), big.get(), used);
                for (long long i = t.lo; i < t.hi; ++i) {
                    StreamRec &r = c.rec[(size_t)i];
                    r.text = t.text.get() + start[(size_t)(i - t.lo)];
                    r.len = (uint16_t)(start[(size_t)(i - t.lo) + 1] - start[(size_t)(i - t.lo)]);
                }
            } catch (const std::bad_alloc &) {
                ok = false;
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                if (!ok) failed = true;
                --busy;
                if (busy == 0 && queue_head >= queue.size()) cv_idle.notify_all();
            }
        }
    }
};

extern "C" {

int ysmr_rows_stream_create(int threads, int via_pandas, ysmr_rows_stream **out)
{
    if (!out) return ysmr::fail(YSMR_ERR_ARG, "out must not be NULL");
    *out = nullptr;
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, 64));
    auto *s = new (std::nothrow) ysmr_rows_stream();
    if (!s) return ysmr::fail(YSMR_ERR_CAPACITY, "out of memory");
    s->via_pandas = via_pandas != 0;
    try {
        for (int t = 0; t < nt; ++t) s->pool.emplace_back([s] { s->work(); });
    } catch (const std::system_error &) {
        // (fewer threads than asked for is fine; none at all is not)
    }
    if (s->pool.empty()) { delete s; return ysmr::fail(YSMR_ERR_HIP, "cannot start a formatting thread"); }
    *out = s;
    return YSMR_OK;
}

int ysmr_rows_stream_push(ysmr_rows_stream *s, const ysmr_row *rows_host, long long n)
{
    if (!s) return ysmr::fail(YSMR_ERR_ARG, "stream handle is NULL");
    if (n < 0 || (n && !rows_host)) return ysmr::fail(YSMR_ERR_ARG, "rows_host must be set");
    if (n == 0) return YSMR_OK;
    std::unique_ptr<StreamChunk> c;
    try {
        c.reset(new StreamChunk());
        c->rec.reset(new StreamRec[(size_t)n]);
        c->n = n;
        const long long per = std::max<long long>(2048, (n + (long long)s->pool.size() - 1) / (long long)s->pool.size());
        for (long long lo = 0; lo < n; lo += per) {
            c->tasks.emplace_back();
            c->tasks.back().lo = lo; c->tasks.back().hi = std::min(n, lo + per);
        }
        // the keys and the values leave the caller's buffer here (it is free when this call returns); rows per id and first
        // frames are kept as the rows come in, so that ordering them at the end starts from finished tables
        for (long long i = 0; i < n; ++i) {
            const ysmr_row &r = rows_host[i];
            StreamRec &q = c->rec[(size_t)i];
            q.id = (uint32_t)r.track_id; q.frame = (uint32_t)r.frame;
            q.v[0] = r.x; q.v[1] = r.y; q.v[2] = (double)r.w; q.v[3] = (double)r.h; q.v[4] = (double)r.angle;
            q.text = nullptr; q.len = 0;
            if (q.id >= s->count.size()) {
                const size_t want = std::max<size_t>((size_t)q.id + 1, s->count.size() * 2);
                if ((size_t)q.id > (size_t)0x0FFFFFFF) return ysmr::fail(YSMR_ERR_ARG, "track id %u out of range", q.id);
                s->count.resize(want, 0u); s->first.resize(want, 0xFFFFFFFFu);
            }
            ++s->count[q.id];
            if (q.frame < s->first[q.id]) s->first[q.id] = q.frame;
        }
    } catch (const std::bad_alloc &) {
        return ysmr::fail(YSMR_ERR_CAPACITY, "out of memory taking %lld rows", n);
    }
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->n_rows += n;
        StreamChunk *raw = c.get();
        s->chunks.push_back(std::move(c));
        for (int t = 0; t < (int)raw->tasks.size(); ++t) s->queue.emplace_back(raw, t);
    }
    s->cv_work.notify_all();
    return YSMR_OK;
}

long long ysmr_rows_stream_count(ysmr_rows_stream *s)
{
    if (!s) return -1;
    std::unique_lock<std::mutex> lk(s->mu);
    return s->n_rows;
}

int ysmr_rows_stream_finish(ysmr_rows_stream *s, int with_header, const char *path, size_t *out_length, uint32_t *track_id,
                            uint32_t *t, double *x, double *y, double *w, double *h, double *angle)
{
    if (!s) return ysmr::fail(YSMR_ERR_ARG, "stream handle is NULL");
#ifdef YSMR_TUNING
    const bool dbg = getenv("YSMR_STREAM_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) { if (dbg) { const double t = now(); fprintf(stderr, "[rows stream] %s: %.2f ms\n", what, t - t_prev); t_prev = t; } };
#else
    auto lap = [](const char *) {};
#endif
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_idle.wait(lk, [&] { return s->busy == 0 && s->queue_head >= s->queue.size(); });
        if (s->failed) return ysmr::fail(YSMR_ERR_CAPACITY, "out of memory formatting rows");
    }
    lap("waited for the formatting threads");
    const long long N = s->n_rows;
    const bool want_cols = track_id || t || x || y || w || h || angle;
    if (want_cols && N && !(track_id && t && x && y && w && h && angle))
        return ysmr::fail(YSMR_ERR_ARG, "all seven column pointers must be set, or none");
    if (N > 0x7FFFFFFFll) return ysmr::fail(YSMR_ERR_ARG, "more than 2^31 - 1 rows");
    const int nt = (int)std::max<size_t>(1, std::min<size_t>(s->pool.size(), (size_t)(N / 4096 + 1)));
    auto in_parallel = [&](int parts, auto &&fn) {        // fn(k) for k in 0..parts-1, on up to nt threads (the caller's included)
        std::atomic<int> next{0};
        auto loop = [&] { for (int k; (k = next.fetch_add(1)) < parts;) fn(k); };
        std::vector<std::thread> th;
        try {
            for (int q = 0; q < std::min(nt, parts) - 1; ++q) th.emplace_back(loop);
        } catch (const std::system_error &) {
        }
        loop();
        for (auto &q : th) q.join();
    };
    // where every row goes: offset[id] + frame - first_frame[id].  N rows into N places: the map is a bijection exactly when no
    // place stays empty, so the threads fill `src` without looking at each other and the gaps are counted afterwards.
    std::vector<const StreamRec *> src;
    std::vector<uint16_t> len_at;
    try {
        std::vector<uint32_t> offset(s->count.size() + 1, 0u);
        for (size_t i = 0; i < s->count.size(); ++i) offset[i + 1] = offset[i] + s->count[i];
        src.assign((size_t)N, nullptr);
        len_at.assign((size_t)N, 0);
        std::atomic<int> irregular{0};
        std::vector<std::pair<const StreamChunk *, const StreamTask *>> all_tasks;
        for (auto &c : s->chunks)
            for (auto &tk : c->tasks) all_tasks.emplace_back(c.get(), &tk);
        in_parallel((int)all_tasks.size(), [&](int k) {
            const StreamChunk &c = *all_tasks[(size_t)k].first;
            const StreamTask &tk = *all_tasks[(size_t)k].second;
            for (long long i = tk.lo; i < tk.hi; ++i) {
                const StreamRec &r = c.rec[(size_t)i];
                const uint32_t d = r.frame - s->first[r.id];
                if (d >= s->count[r.id]) { irregular.store(1); continue; }
                src[offset[r.id] + d] = &r;
                len_at[offset[r.id] + d] = r.len;
            }
        });
        if (!irregular.load())
            for (long long d = 0; d < N; ++d)
                if (!src[(size_t)d]) { irregular.store(1); break; }
        if (irregular.load()) {      // gaps or duplicates (rows a caller assembled): a stable sort by (TRACK_ID, POSITION_T)
            size_t at = 0;
            for (auto &c : s->chunks)
                for (long long i = 0; i < c->n; ++i) src[at++] = &c->rec[(size_t)i];
            std::stable_sort(src.begin(), src.end(), [](const StreamRec *a, const StreamRec *b) {
                return a->id != b->id ? a->id < b->id : a->frame < b->frame;
            });
            for (long long d = 0; d < N; ++d) len_at[(size_t)d] = src[(size_t)d]->len;
        }
    } catch (const std::bad_alloc &) {
        return ysmr::fail(YSMR_ERR_CAPACITY, "out of memory ordering %lld rows", N);
    }
    lap("places");
    const size_t head = with_header ? sizeof(CSV_HEADER) - 1 : 0;
    int fd = -1;
    if (path) {
        fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) return ysmr::fail(YSMR_ERR_ARG, "cannot open %s for writing: %s", path, std::strerror(errno));
    }
    lap("file opened (truncated)");
    // pieces of the ordered table: size them (the lengths lie in table order), then every thread gathers the lines and columns of
    // its pieces -- one cache line of record and the line's text per row -- and writes each piece at its offset
    const int parts = (int)std::max<long long>(1, std::min<long long>(4LL * nt, N / 2048 + 1));
    const long long per = (N + parts - 1) / parts;
    std::vector<size_t> bytes((size_t)parts, 0), at((size_t)parts + 1, head);
    for (int k = 0; k < parts; ++k) {
        const long long lo = std::min<long long>((long long)k * per, N), hi = std::min<long long>(lo + per, N);
        size_t sum = 0;
        for (long long d = lo; d < hi; ++d) sum += len_at[(size_t)d];
        bytes[(size_t)k] = sum;
        at[(size_t)k + 1] = at[(size_t)k] + sum;
    }
    lap("pieces sized");
    std::atomic<int> bad{0}, write_errno{0};
    auto write_all = [&](const char *p, size_t n, off_t off) {
        while (n) {
            const ssize_t wr = ::pwrite(fd, p, n, off);
            if (wr < 0 && errno == EINTR) continue;
            if (wr <= 0) { write_errno.store(wr < 0 ? errno : ENOSPC); bad.store(2); return; }
            p += wr; n -= (size_t)wr; off += wr;
        }
    };
    in_parallel(parts, [&](int k) {
        const long long lo = std::min<long long>((long long)k * per, N), hi = std::min<long long>(lo + per, N);
        std::unique_ptr<char[]> buf(path ? new (std::nothrow) char[bytes[(size_t)k] + 1] : nullptr);
        if (path && !buf) { bad.store(1); return; }
        char *o = buf.get();
        for (long long d = lo; d < hi; ++d) {
            const StreamRec &r = *src[(size_t)d];
            if (d + 8 < hi) { __builtin_prefetch(src[(size_t)d + 8]); __builtin_prefetch(src[(size_t)d + 4]->text); }
            if (path) { std::memcpy(o, r.text, r.len); o += r.len; }
            if (want_cols) {
                track_id[d] = r.id; t[d] = r.frame;
                x[d] = r.v[0]; y[d] = r.v[1]; w[d] = r.v[2]; h[d] = r.v[3]; angle[d] = r.v[4];
            }
        }
        if (!path) return;
        if (k == 0 && head) write_all(CSV_HEADER, head, 0);
        write_all(buf.get(), bytes[(size_t)k], (off_t)at[(size_t)k]);
    });
    lap("gathered and written");
    if (fd >= 0) {
        const int rc_close = ::close(fd);
        if (rc_close != 0 && !bad.load()) { write_errno.store(errno); bad.store(2); }
        if (bad.load()) ::unlink(path);
    }
    lap("closed");
    if (bad.load() == 1) return ysmr::fail(YSMR_ERR_CAPACITY, "out of memory assembling %lld rows", N);
    if (bad.load() == 2) return ysmr::fail(YSMR_ERR_ARG, "writing %s failed: %s", path, std::strerror(write_errno.load()));
    if (out_length) *out_length = at[(size_t)parts];
    return YSMR_OK;
}

int ysmr_rows_stream_destroy(ysmr_rows_stream *s)
{
    if (!s) return YSMR_OK;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->closing = true;
        s->queue_head = s->queue.size();         // (whatever is still queued is dropped)
    }
    s->cv_work.notify_all();
    for (auto &th : s->pool) th.join();
    delete s;
    return YSMR_OK;
}

int ysmr_rows_columns(const ysmr_row *rows_host, long long n_rows, int via_pandas, uint32_t *track_id, uint32_t *t,
                      double *x, double *y, double *w, double *h, double *angle)
{
    if (n_rows < 0 || (n_rows && (!rows_host || !track_id || !t || !x || !y || !w || !h || !angle)))
        return ysmr::fail(YSMR_ERR_ARG, "rows_host and all seven column pointers must be set");
    int nt = (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, 64));
    if (n_rows < 4096 || !via_pandas) nt = 1;
    auto work = [=](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i) {
            const ysmr_row &r = rows_host[i];
            track_id[i] = (uint32_t)r.track_id;
            t[i] = (uint32_t)r.frame;
            double v[5] = {r.x, r.y, (double)r.w, (double)r.h, (double)r.angle};
            if (via_pandas)
                for (double &q : v) q = pandas_roundtrip(q);
            x[i] = v[0]; y[i] = v[1]; w[i] = v[2]; h[i] = v[3]; angle[i] = v[4];
        }
    };
    if (nt == 1) { work(0, n_rows); return YSMR_OK; }
    std::vector<std::thread> pool;
    const long long per = (n_rows + nt - 1) / nt;
    for (int k = 0; k < nt; ++k) {
        const long long lo = std::min<long long>((long long)k * per, n_rows), hi = std::min<long long>(lo + per, n_rows);
        pool.emplace_back(work, lo, hi);
    }
    for (auto &th : pool) th.join();
    return YSMR_OK;
}


// ---- the device formatter ----------------------------------------------------------------------------------------------
size_t ysmr_rows_format_device_workspace_bytes(long long n_rows)
{
    if (n_rows <= 0 || n_rows > 0x7FFFFFFFll / (long long)ROW_TEXT_MAX) return 0;
    return fmt_layout(n_rows).total;
}

int ysmr_rows_format_device(void *stream, const ysmr_row *rows_dev, long long n_rows, int with_header, int via_pandas,
                            void *workspace_dev, size_t workspace_bytes, char *csv_dev, size_t csv_capacity,
                            unsigned long long *csv_length_dev, uint32_t *track_id_dev, uint32_t *t_dev, double *x_dev,
                            double *y_dev, double *w_dev, double *h_dev, double *angle_dev, uint32_t *unserved_dev)
{
    if (n_rows < 0 || n_rows > 0x7FFFFFFFll / (long long)ROW_TEXT_MAX)
        return ysmr::fail(YSMR_ERR_ARG, "n_rows must be in 0..%lld, got %lld", 0x7FFFFFFFll / (long long)ROW_TEXT_MAX, n_rows);
    if (!csv_dev || !csv_length_dev || !unserved_dev || (n_rows && (!rows_dev || !workspace_dev || !track_id_dev || !t_dev || !x_dev ||
                                                                     !y_dev || !w_dev || !h_dev || !angle_dev)))
        return ysmr::fail(YSMR_ERR_ARG, "rows_dev, workspace_dev, csv_dev, csv_length_dev, the seven columns and unserved_dev must be set");
    if (csv_capacity < ysmr_rows_csv_bound(n_rows, with_header))
        return ysmr::fail(YSMR_ERR_CAPACITY, "csv buffer too small: %zu < %zu bytes", csv_capacity, ysmr_rows_csv_bound(n_rows, with_header));
    hipStream_t st = (hipStream_t)stream;
    YSMR_HIP_CHECK(hipMemsetAsync(unserved_dev, 0, sizeof(uint32_t), st));
    FmtLayout L{};
    if (n_rows) {
        L = fmt_layout(n_rows);
        if (workspace_bytes < L.total)
            return ysmr::fail(YSMR_ERR_CAPACITY, "format workspace too small: %zu < %zu bytes", workspace_bytes, L.total);
    }
    char *w = (char *)workspace_dev;
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((n_rows + 255) / 256, 2048));
    if (n_rows) {
        const DevColumns cols{track_id_dev, t_dev, {x_dev, y_dev, w_dev, h_dev, angle_dev}};
        hipLaunchKernelGGL(k_fmt_rows, dim3(grid), dim3(256), 0, st, rows_dev, n_rows, via_pandas, w + L.slots, (uint32_t *)(w + L.len),
                           cols, unserved_dev);
        ysmr::prim::inclusive_scan_u32(st, (const uint32_t *)(w + L.len), (uint32_t *)(w + L.incl), (size_t)n_rows,
                                       (uint32_t *)(w + L.temp));
    }
    hipLaunchKernelGGL(k_fmt_pack, dim3(grid), dim3(256), 0, st, w + L.slots, (const uint32_t *)(w + L.len), (const uint32_t *)(w + L.incl),
                       n_rows, with_header, csv_dev, csv_length_dev);
    YSMR_LAUNCH_CHECK();
    return YSMR_OK;
}

// the device formatter's arithmetic (fmt.h) on the HOST, a row at a time: what the CPU tests compare with ysmr_rows_format_csv
// (std::to_chars + the pandas restatement) on millions of values; *unserved counts the rows fmt.h leaves to the host path
int ysmr_rows_format_csv_devicelike(const ysmr_row *rows_host, long long n_rows, int with_header, int via_pandas, char *out,
                                    size_t out_capacity, size_t *out_length, double *columns5, long long *unserved)
{
    if (n_rows < 0 || (!rows_host && n_rows) || !out || !out_length || !unserved)
        return ysmr::fail(YSMR_ERR_ARG, "rows_host, out, out_length and unserved must be set");
    if (out_capacity < ysmr_rows_csv_bound(n_rows, with_header))
        return ysmr::fail(YSMR_ERR_CAPACITY, "csv buffer too small: %zu < %zu bytes", out_capacity, ysmr_rows_csv_bound(n_rows, with_header));
    char *p = out;
    if (with_header) { std::memcpy(p, CSV_HEADER, sizeof(CSV_HEADER) - 1); p += sizeof(CSV_HEADER) - 1; }
    long long bad = 0;
    for (long long i = 0; i < n_rows; ++i) {
        const ysmr_row &r = rows_host[i];
        double v[5] = {r.x, r.y, (double)r.w, (double)r.h, (double)r.angle};
        char *q = p;
        q = ysmr_fmt::put_u32(q, (uint32_t)r.track_id); *q++ = ',';
        q = ysmr_fmt::put_u32(q, (uint32_t)r.frame); *q++ = ',';
        bool ok = true;
        for (int k = 0; k < 5; ++k) { q = ysmr_fmt::put_value(q, &v[k], via_pandas != 0, ok); *q++ = k == 4 ? '\n' : ','; }
        if (ok) p = q; else ++bad;
        if (columns5)
            for (int k = 0; k < 5; ++k) columns5[(size_t)k * (size_t)n_rows + (size_t)i] = v[k];
    }
    *out_length = (size_t)(p - out);
    *unserved = bad;
    return YSMR_OK;
}

}  // extern "C"
