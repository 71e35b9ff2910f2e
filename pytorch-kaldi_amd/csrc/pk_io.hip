// pk_io.hip - host-side readers for the chunk loader: Kaldi binary matrix tables and the two
// whole-chunk transforms the reference applies after loading.  No device code in this file; it is part of
// libpk_amd.so so that the chunk loop needs no second library.
//
//   pk_ivec_*           data_io.py:790-838 (read_vec_int_ark / read_vec_int: alignments, pdf ids)
//   pk_ark_*            data_io.py:762-783 (read_key), :1062-1131 (read_mat_ark, read_mat, _read_mat_binary),
//                       :1150-1198 (_read_compressed_mat); 'CM2' / 'CM3' follow Kaldi's compressed-matrix.h, which the
//                       reference's reader refuses
//   pk_context_window   data_io.py:228-241 (np.roll based splicing of the concatenated chunk, edges trimmed)
//   pk_mean_var_norm    data_io.py:263 ((x - mean) / std per column, population std, double accumulation)
//
// Plain files only: the reference reads through Kaldi pipes ("ark:copy-feats scp:... ark:- |"), which stay outside.
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <new>
#include <vector>

#include "pk_common.h"

struct pk_ark {
    FILE* f;
    char kind;  // 'F' float32, 'D' float64, '1' CM, '2' CM2, '3' CM3, 'I' int vector: record announced, not yet read
    int64_t rows, cols;
    float cm_min, cm_range;
};

namespace {

bool read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

// bytes between the current position and the end of the file (-1: not a seekable file)
int64_t remaining_bytes(FILE* f) {
    struct stat st;
    const off_t pos = ftello(f);
    if (pos < 0 || fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode)) return -1;
    return (int64_t)st.st_size - (int64_t)pos;
}

// A damaged header must become an error code, not an allocation the size of its garbage dimensions: the payload a
// record announces has to fit into what is left of the file (and into 63 bits).
bool payload_fits(FILE* f, int64_t rows, int64_t cols, int64_t bytes_per_elem, int64_t extra) {
    if (rows < 0 || cols < 0) return false;
    if (cols != 0 && rows > (INT64_MAX / 16) / cols) return false;
    const int64_t need = rows * cols * bytes_per_elem + extra;
    const int64_t left = remaining_bytes(f);
    return left < 0 || need <= left;
}

// "<key> ": the reference's read_key (data_io.py:762-783) reads up to the blank and strips whitespace AROUND the key;
// white space inside a key is kept (Kaldi never writes any).  Returns the key length (0 at end of file), -1 when the
// key does not fit.
int read_key(FILE* f, char* key, int keycap) {
    int n = 0, c;
    while ((c = fgetc(f)) != EOF && c != ' ') {
        if (n == 0 && (c == '\n' || c == '\r' || c == '\t')) continue;  // leading white space (e.g. the newline of a text scp)
        if (key == nullptr || n + 1 >= keycap) return -1;
        key[n++] = (char)c;
    }
    while (n > 0 && (key[n - 1] == '\n' || key[n - 1] == '\r' || key[n - 1] == '\t')) --n;  // trailing white space
    if (key != nullptr && keycap > 0) key[n < keycap ? n : keycap - 1] = 0;
    return n;
}

}  // namespace

extern "C" pk_ark* pk_ark_open(const char* path, int64_t offset) {
    FILE* f = fopen(path, "rb");
    if (f == nullptr) {
        pk_set_error("pk_ark_open: cannot open %s: %s", path, strerror(errno));
        return nullptr;
    }
    if (offset > 0 && fseeko(f, (off_t)offset, SEEK_SET) != 0) {
        pk_set_error("pk_ark_open: cannot seek %s to %lld", path, (long long)offset);
        fclose(f);
        return nullptr;
    }
    pk_ark* a = new pk_ark();
    a->f = f;
    a->kind = 0;
    a->rows = a->cols = 0;
    a->cm_min = a->cm_range = 0.f;
    return a;
}

extern "C" void pk_ark_close(pk_ark* a) {
    if (a == nullptr) return;
    if (a->f) fclose(a->f);
    delete a;
}

// Reads "<key> " (unless key_expected == 0: a bare matrix, as an scp entry with an offset points at), the "\0B" marker
// and the matrix header.  Returns 1 with *rows / *cols set, 0 at a clean end of file, < 0 on a malformed table.
extern "C" int pk_ark_next(pk_ark* a, int key_expected, char* key, int keycap, int64_t* rows, int64_t* cols) {
    PK_REQUIRE(a != nullptr && a->f != nullptr, "pk_ark_next: closed table");
    PK_REQUIRE(a->kind == 0, "pk_ark_next: the previous matrix has not been read (pk_ark_read / pk_ark_skip)");
    if (key != nullptr && keycap > 0) key[0] = 0;
    if (key_expected) {
        const int n = read_key(a->f, key, keycap);
        PK_REQUIRE(n >= 0, "pk_ark_next: key longer than %d bytes", keycap);
        if (n == 0) return 0;  // end of file
    }
    char mark[2];
    PK_REQUIRE(read_exact(a->f, mark, 2), "pk_ark_next: truncated table");
    PK_REQUIRE(mark[0] == 0 && mark[1] == 'B', "pk_ark_next: not a binary Kaldi table (text tables are not supported)");
    char tag[3];
    PK_REQUIRE(read_exact(a->f, tag, 3), "pk_ark_next: truncated matrix header");
    if (tag[0] == 'C' && tag[1] == 'M') {
        a->kind = tag[2] == ' ' ? '1' : tag[2];
        if (a->kind != '1') {  // "CM2" / "CM3" are followed by the separating blank
            PK_REQUIRE((a->kind == '2' || a->kind == '3') && fgetc(a->f) == ' ', "pk_ark_next: unknown compressed header");
        }
        struct { float mn, range; int32_t rows, cols; } gh;
        PK_REQUIRE(read_exact(a->f, &gh, 16), "pk_ark_next: truncated compressed header");
        a->cm_min = gh.mn; a->cm_range = gh.range; a->rows = gh.rows; a->cols = gh.cols;
    } else {
        PK_REQUIRE((tag[0] == 'F' || tag[0] == 'D') && tag[1] == 'M' && tag[2] == ' ', "pk_ark_next: unknown matrix header '%c%c%c'",
                   tag[0], tag[1], tag[2]);
        a->kind = tag[0];
        unsigned char dims[10];
        PK_REQUIRE(read_exact(a->f, dims, 10) && dims[0] == 4 && dims[5] == 4, "pk_ark_next: bad dimension block");
        int32_t r, c;
        memcpy(&r, dims + 1, 4);
        memcpy(&c, dims + 6, 4);
        a->rows = r; a->cols = c;
    }
    PK_REQUIRE(a->rows >= 0 && a->cols >= 0, "pk_ark_next: negative dimensions");
    {
        const int64_t bpe = a->kind == 'F' ? 4 : a->kind == 'D' ? 8 : a->kind == '2' ? 2 : 1;
        const int64_t extra = a->kind == '1' ? a->cols * 8 : 0;
        if (!payload_fits(a->f, a->rows, a->cols, bpe, extra)) {
            const long long r_ = (long long)a->rows, c_ = (long long)a->cols;
            a->kind = 0;
            PK_REQUIRE(false, "pk_ark_next: a %lld x %lld matrix does not fit into the rest of the file (damaged table)", r_, c_);
        }
    }
    *rows = a->rows; *cols = a->cols;
    return 1;
}

// Integer-vector tables (alignments / pdf ids; data_io.py:790-838): "<key> \0B \4 <int32 n> (\4 <int32 v>) x n".
// pk_ivec_next returns 1 and *n, 0 at end of file, 2 on a malformed table; pk_ivec_read fills dst[n].
extern "C" int pk_ivec_next(pk_ark* a, char* key, int keycap, int64_t* n) {
    PK_REQUIRE(a != nullptr && a->f != nullptr, "pk_ivec_next: closed table");
    PK_REQUIRE(a->kind == 0, "pk_ivec_next: the previous record has not been read");
    const int len = read_key(a->f, key, keycap);
    PK_REQUIRE(len >= 0, "pk_ivec_next: key longer than %d bytes", keycap);
    if (len == 0) return 0;
    unsigned char hdr[7];
    PK_REQUIRE(read_exact(a->f, hdr, 7), "pk_ivec_next: truncated table");
    PK_REQUIRE(hdr[0] == 0 && hdr[1] == 'B' && hdr[2] == 4, "pk_ivec_next: not a binary int32 vector (text tables are not supported)");
    int32_t cnt;
    memcpy(&cnt, hdr + 3, 4);
    PK_REQUIRE(cnt >= 0, "pk_ivec_next: negative length");
    PK_REQUIRE(payload_fits(a->f, cnt, 1, 5, 0), "pk_ivec_next: a vector of %d elements does not fit into the rest of the file "
               "(damaged table)", (int)cnt);
    a->kind = 'I';
    a->rows = cnt;
    a->cols = 1;
    *n = cnt;
    return 1;
}

static int ivec_read_body(pk_ark* a, int32_t* dst);
extern "C" int pk_ivec_read(pk_ark* a, int32_t* dst) {
    // no C++ exception may cross the C ABI (ctypes would std::terminate the process)
    try {
        return ivec_read_body(a, dst);
    } catch (const std::bad_alloc&) {
        pk_set_error("pk_ivec_read: out of memory for the staging buffer");
        return 2;
    } catch (...) {
        pk_set_error("pk_ivec_read: unexpected exception");
        return 2;
    }
}
static int ivec_read_body(pk_ark* a, int32_t* dst) {
    PK_REQUIRE(a != nullptr && a->kind == 'I', "pk_ivec_read: no pending vector (call pk_ivec_next first)");
    const int64_t n = a->rows;
    a->kind = 0;
    std::vector<unsigned char> buf((size_t)n * 5);
    PK_REQUIRE(read_exact(a->f, buf.data(), (size_t)n * 5), "pk_ivec_read: truncated vector");
    for (int64_t i = 0; i < n; ++i) {
        PK_REQUIRE(buf[(size_t)i * 5] == 4, "pk_ivec_read: element %lld is not an int32", (long long)i);
        memcpy(dst + i, buf.data() + (size_t)i * 5 + 1, 4);
    }
    return 0;
}

// The matrix announced by pk_ark_next -> dst[rows*cols] (row-major float32).
static int ark_read_body(pk_ark* a, float* dst);
extern "C" int pk_ark_read(pk_ark* a, float* dst) {
    try {
        return ark_read_body(a, dst);
    } catch (const std::bad_alloc&) {
        pk_set_error("pk_ark_read: out of memory for the staging buffer");
        return 2;
    } catch (...) {
        pk_set_error("pk_ark_read: unexpected exception");
        return 2;
    }
}
static int ark_read_body(pk_ark* a, float* dst) {
    PK_REQUIRE(a != nullptr && a->kind != 0 && a->kind != 'I', "pk_ark_read: no pending matrix (call pk_ark_next first)");
    const int64_t R = a->rows, C = a->cols, n = R * C;
    const char kind = a->kind;
    a->kind = 0;
    if (kind == 'F') {
        PK_REQUIRE(read_exact(a->f, dst, (size_t)n * 4), "pk_ark_read: truncated float matrix");
    } else if (kind == 'D') {
        std::vector<double> buf((size_t)n);
        PK_REQUIRE(read_exact(a->f, buf.data(), (size_t)n * 8), "pk_ark_read: truncated double matrix");
        for (int64_t i = 0; i < n; ++i) dst[i] = (float)buf[(size_t)i];
    } else if (kind == '1') {
        // per-column percentile headers, then the bytes column-major (data_io.py:1168-1196)
        std::vector<uint16_t> hdr((size_t)C * 4);
        std::vector<uint8_t> bytes((size_t)n);
        PK_REQUIRE(read_exact(a->f, hdr.data(), (size_t)C * 8) && read_exact(a->f, bytes.data(), (size_t)n),
                   "pk_ark_read: truncated compressed matrix");
        for (int64_t c = 0; c < C; ++c) {
            float p[4];
            for (int q = 0; q < 4; ++q)
                p[q] = (float)((double)hdr[(size_t)c * 4 + q] * (double)a->cm_range * 1.52590218966964e-05 + (double)a->cm_min);
            const float s0 = (p[1] - p[0]) / 64.0f, s1 = (p[2] - p[1]) / 128.0f, s2 = (p[3] - p[2]) / 63.0f;
            const uint8_t* col = bytes.data() + (size_t)c * R;
            for (int64_t r = 0; r < R; ++r) {
                const int v = col[r];
                float x;
                if (v <= 64) x = p[0] + s0 * (float)v;
                else if (v <= 192) x = p[1] + s1 * (float)(v - 64);
                else x = p[2] + s2 * (float)(v - 192);
                dst[r * C + c] = x;
            }
        }
    } else if (kind == '2') {
        std::vector<uint16_t> buf((size_t)n);
        PK_REQUIRE(read_exact(a->f, buf.data(), (size_t)n * 2), "pk_ark_read: truncated CM2 matrix");
        const float inc = a->cm_range * (1.0f / 65535.0f);
        for (int64_t i = 0; i < n; ++i) dst[i] = a->cm_min + inc * (float)buf[(size_t)i];
    } else {
        std::vector<uint8_t> buf((size_t)n);
        PK_REQUIRE(read_exact(a->f, buf.data(), (size_t)n), "pk_ark_read: truncated CM3 matrix");
        const float inc = a->cm_range * (1.0f / 255.0f);
        for (int64_t i = 0; i < n; ++i) dst[i] = a->cm_min + inc * (float)buf[(size_t)i];
    }
    return 0;
}

extern "C" int pk_ark_skip(pk_ark* a) {
    PK_REQUIRE(a != nullptr && a->kind != 0, "pk_ark_skip: no pending matrix");
    const int64_t n = a->rows * a->cols;
    int64_t bytes = a->kind == 'F' ? n * 4 : a->kind == 'D' ? n * 8 : a->kind == '1' ? a->cols * 8 + n : a->kind == '2' ? n * 2 : n;
    a->kind = 0;
    PK_REQUIRE(fseeko(a->f, (off_t)bytes, SEEK_CUR) == 0, "pk_ark_skip: seek failed");
    return 0;
}

// out[(rows - left - right)][cols * (left + right + 1)]: block `lag + left` of output row i is input row
// i + left + lag, lag = -left .. right - exactly what the np.roll construction keeps after trimming (data_io.py:228-241).
extern "C" int pk_context_window(const float* x, int64_t rows, int64_t cols, int left, int right, float* out) {
    PK_REQUIRE(left >= 0 && right >= 0 && rows >= (int64_t)left + right, "pk_context_window: %lld rows cannot hold a -%d..+%d window",
               (long long)rows, left, right);
    const int64_t orows = rows - left - right, W = left + right + 1;
    for (int64_t i = 0; i < orows; ++i)
        for (int64_t b = 0; b < W; ++b) memcpy(out + (i * W + b) * cols, x + (i + b) * cols, (size_t)cols * sizeof(float));
    return 0;
}

// x <- (x - mean) / std per column (population std, as np.std), accumulated in double like the reference's float64 chunk
static int mean_var_norm_body(float* x, int64_t rows, int64_t cols);
extern "C" int pk_mean_var_norm(float* x, int64_t rows, int64_t cols) {
    try {
        return mean_var_norm_body(x, rows, cols);
    } catch (...) {
        pk_set_error("pk_mean_var_norm: out of memory");
        return 2;
    }
}
static int mean_var_norm_body(float* x, int64_t rows, int64_t cols) {
    PK_REQUIRE(rows > 0 && cols > 0, "pk_mean_var_norm: empty chunk");
    std::vector<double> mean((size_t)cols, 0.0), m2((size_t)cols, 0.0);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) mean[(size_t)c] += x[r * cols + c];
    for (int64_t c = 0; c < cols; ++c) mean[(size_t)c] /= (double)rows;
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) {
            const double d = x[r * cols + c] - mean[(size_t)c];
            m2[(size_t)c] += d * d;
        }
    for (int64_t c = 0; c < cols; ++c) m2[(size_t)c] = sqrt(m2[(size_t)c] / (double)rows);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < cols; ++c) x[r * cols + c] = (float)((x[r * cols + c] - mean[(size_t)c]) / m2[(size_t)c]);
    return 0;
}
