// Host-side post-decode re-alignment on the window form of the soft alignments: the Viterbi search of the OCR-D wrapper
// (wrapper/transcode.py:279-349, `_alignment2path`) -- same forward scores (float32, as the reference's numpy array), same
// back-tracking rule including numpy's wrap-around of index -1 at the borders and its argmax on NaN, same distance -- visiting
// only the cells inside the attention windows.  Pure host code (no device call): the Python restatement of the same loop
// (cor_asv_ann_amd/realign.py) took 0.4 ms per 60-character line, 16 ms of a 209 ms page of the OCR-D processor's call.
#include "engine.h"
#include <cmath>
#include <limits>

namespace {

struct Windows {
    int n, T, K; const int32_t* lo; const float* w;
    float value(int j, int i) const {                 // alignment[j][i]
        const int l = lo[j];
        if (l < 0) return std::numeric_limits<float>::quiet_NaN();
        const int k = i - l;
        return (k >= 0 && k < K && i < T) ? w[(size_t)j * K + k] : 0.0f;
    }
};

// numpy.argmax over v[0..n): the first NaN if there is one, otherwise the first maximum
int np_argmax(const float* v, int n, int stride) {
    int best = 0;
    for (int q = 0; q < n; ++q) {
        const float x = v[(size_t)q * stride];
        if (x != x) return q;
        if (x > v[(size_t)best * stride]) best = q;
    }
    return best;
}
// Python slice start for `a[start:]` on an axis of length n
int slice_start(int start, int n) { if (start < 0) { start += n; if (start < 0) start = 0; } return start > n ? n : start; }

}  // namespace

extern "C" int casv_realign_path(int32_t n_rows, int32_t T, int32_t K, const int32_t* lo, const float* w, int32_t i_max,
                                 int32_t j_max, float min_score, int32_t* path, double* dist_out) {
    if (!lo || !w || !path || !dist_out) return fail(CASV_ERR_ARG, "null argument");
    if (i_max < 1 || j_max < 1 || j_max > n_rows || K < 1 || T < 1) return fail(CASV_ERR_ARG, "bad shape i_max=%d j_max=%d rows=%d K=%d T=%d", i_max, j_max, n_rows, K, T);
    const Windows a{n_rows, T, K, lo, w};
    std::vector<float> fw((size_t)i_max * j_max, 0.0f);          // [i][j]
    auto F = [&](int i, int j) -> float& {                       // numpy indexing: -1 wraps to the last row / column
        if (i < 0) i += i_max;
        if (j < 0) j += j_max;
        return fw[(size_t)i * j_max + j];
    };
    auto visit = [&](int i, int j, float v) {
        const float im1 = i > 0 ? F(i - 1, j) : 0.0f, jm1 = j > 0 ? F(i, j - 1) : 0.0f, ijm1 = (i > 0 && j > 0) ? F(i - 1, j - 1) : 0.0f;
        // Python's max(im1, jm1, ijm1): the first argument unless a later one compares greater
        float m = im1;
        if (jm1 > m) m = jm1;
        if (ijm1 > m) m = ijm1;
        F(i, j) = v + m;
    };
    visit(0, 0, a.value(0, 0));                                  // the scan starts here whatever the score (transcode.py:296)
    for (int j = 0; j < j_max; ++j) {
        const int l = lo[j];
        if (l < 0) continue;                                     // an all-NaN row: no cell above min_score
        const int hi = std::min(std::min(l + K, T), (int)i_max);
        for (int i = l; i < hi; ++i) {
            const float v = w[(size_t)j * K + (i - l)];
            if (!(v > min_score) || (i == 0 && j == 0)) continue;
            visit(i, j, v);
        }
    }
    // backward pass (transcode.py:318-337)
    int i, j;
    if (i_max <= j_max) i = i_max - 1;
    else { const int s = slice_start(j_max - i_max - 2, i_max); i = j_max - 2 + np_argmax(&fw[(size_t)s * j_max + (j_max - 1)], i_max - s, j_max); }
    if (j_max <= i_max) j = j_max - 1;
    else { const int s = slice_start(i_max - j_max - 2, j_max); j = i_max - 2 + np_argmax(&fw[(size_t)(i_max - 1) * j_max + s], j_max - s, 1); }
    for (int q = 0; q <= i_max; ++q) path[q] = -1;
    path[i_max] = j_max;
    double dist = 0.0;
    while (i >= 0 && j >= 0) {
        if (i >= i_max || j >= j_max) return fail(CASV_ERR_STATE, "re-alignment walked out of the matrix");     // numpy would raise
        dist += 1.0 - (double)a.value(j, i);
        path[i] = j;
        const float up = F(i - 1, j), left = F(i, j - 1), diag = F(i - 1, j - 1);
        if (up > left) { if (up > diag) i -= 1; else { i -= 1; j -= 1; } }
        else if (left > diag) j -= 1;
        else { j -= 1; i -= 1; }
    }
    path[0] = 0;
    *dist_out = dist;
    return CASV_OK;
}
