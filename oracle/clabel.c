/* CPU oracle: skimage.measure.label on an INTEGER image (Cython `_ccomp.label_cython`).
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Contract (SK/measure/_label.py): two pixels belong to one component iff they are
 * neighbours (connectivity 1 = 4-connected, 2 = 8-connected) and carry the SAME non-zero value;
 * value 0 (background) is never labelled; components are numbered 1..K in raster order of their
 * first (lowest flat index) pixel.  R/masks.py:56 reaches this through clear_border
 * (SK/segmentation/_clear_border.py re-labels its input), R/masks.py:63 through label().
 *
 * One raster pass of unions with the already-visited neighbours (W, NW, N, NE), a union-find whose
 * root is always the smallest flat index of its set, then one pass that numbers roots in raster
 * order -- the cost scikit-image's own single-pass Cython labelling has, not one pass per value.
 */
#include <stdint.h>
#include <stdlib.h>

static inline int64_t find_root(int64_t* parent, int64_t i) {
    int64_t r = i;
    while (parent[r] != r) r = parent[r];
    while (parent[i] != r) { /* path compression */
        int64_t n = parent[i];
        parent[i] = r;
        i = n;
    }
    return r;
}

static inline void unite(int64_t* parent, int64_t a, int64_t b) {
    a = find_root(parent, a);
    b = find_root(parent, b);
    if (a < b) parent[b] = a;
    else if (b < a) parent[a] = b;
}

/* image: int64 (H,W) C-contiguous; out: int64 (H,W).  Returns the number of components, <0 on error. */
int64_t oracle_label_int64(const int64_t* image, int64_t* out, int64_t H, int64_t W, int connectivity) {
    int64_t n = H * W;
    int64_t* parent = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    if (!parent) return -1;
    for (int64_t y = 0; y < H; ++y) {
        for (int64_t x = 0; x < W; ++x) {
            int64_t i = y * W + x;
            int64_t v = image[i];
            parent[i] = i;
            if (v == 0) continue;
            if (x > 0 && image[i - 1] == v) unite(parent, i, i - 1);
            if (y > 0) {
                if (image[i - W] == v) unite(parent, i, i - W);
                if (connectivity >= 2) {
                    if (x > 0 && image[i - W - 1] == v) unite(parent, i, i - W - 1);
                    if (x + 1 < W && image[i - W + 1] == v) unite(parent, i, i - W + 1);
                }
            }
        }
    }
    int64_t count = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (image[i] == 0) { out[i] = 0; continue; }
        int64_t r = find_root(parent, i);
        if (r == i) out[i] = ++count;       /* the root is the set's first raster pixel */
        else out[i] = out[r];               /* r < i: already numbered */
    }
    free(parent);
    return count;
}
