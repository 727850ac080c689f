/* CPU oracle: skimage.segmentation.watershed (non-compact, no watershed line).
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Restates the sequential contract pinned in SURVEY.md Appendix A.1 (scikit-image
 * `_watershed_cy.pyx` + `heap_general.pxi`; python wrapper SK/segmentation/_watershed.py):
 *   - image -> float64, markers -> int32 * mask, mask -> int8, all zero-padded by one pixel;
 *   - neighbour visiting order on the padded raveled image (row stride Wp = W + 2):
 *       connectivity 1: -Wp, -1, +1, +Wp
 *       connectivity 2: -Wp, +1, -1, +Wp, -Wp-1, -Wp+1, +Wp-1, +Wp+1
 *   - heap element (value, age, index); smaller(a,b) := a.value < b.value ||
 *     (a.value == b.value && a.age < b.age);
 *   - the binary heap's push / pop mechanics are part of the contract because all markers
 *     enter with age 0 (equal keys are ordered by the heap's internal moves);
 *   - labels are assigned at push time.
 * Pinned against the real scikit-image 0.18.3 .so via tests/golden/watershed_*.npz.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double value;
    int64_t age;
    int64_t index;
} elem_t;

static inline int smaller(const elem_t* a, const elem_t* b) {
    if (a->value != b->value) return a->value < b->value;
    return a->age < b->age;
}

typedef struct {
    elem_t* data;
    int64_t items;
    int64_t space;
} heap_t;

static void heap_push(heap_t* h, const elem_t* e) {
    if (h->items == h->space) {
        h->space *= 2;
        h->data = (elem_t*)realloc(h->data, (size_t)h->space * sizeof(elem_t));
    }
    int64_t child = h->items;
    h->data[child] = *e;
    h->items++;
    while (child > 0) {
        int64_t parent = (child + 1) / 2 - 1;
        if (smaller(&h->data[child], &h->data[parent])) {
            elem_t t = h->data[parent];
            h->data[parent] = h->data[child];
            h->data[child] = t;
            child = parent;
        } else {
            break;
        }
    }
}

static void heap_pop(heap_t* h, elem_t* dst) {
    *dst = h->data[0];
    /* swap first and last, shrink */
    h->data[0] = h->data[h->items - 1];
    h->items--;
    int64_t parent = 0, child = 1;
    while (child < h->items) {
        if (child + 1 < h->items && smaller(&h->data[child + 1], &h->data[child])) child++;
        if (smaller(&h->data[child], &h->data[parent])) {
            elem_t t = h->data[parent];
            h->data[parent] = h->data[child];
            h->data[child] = t;
            parent = child;
            child = 2 * child + 1;
        } else {
            break;
        }
    }
}

/* image: H*W float64; markers: H*W int32; mask: H*W uint8 (may be NULL = all true);
 * out: H*W int32.  connectivity 1 or 2.  Returns 0, or -1 on allocation failure. */
int oracle_watershed(const double* image, const int32_t* markers, const uint8_t* mask, int32_t* out,
                     int64_t H, int64_t W, int connectivity) {
    const int64_t Hp = H + 2, Wp = W + 2, Np = Hp * Wp;
    double* pimg = (double*)calloc((size_t)Np, sizeof(double));
    int32_t* pout = (int32_t*)calloc((size_t)Np, sizeof(int32_t));
    int8_t* pmask = (int8_t*)calloc((size_t)Np, sizeof(int8_t));
    heap_t hp;
    hp.items = 0;
    hp.space = 1024;
    hp.data = (elem_t*)malloc((size_t)hp.space * sizeof(elem_t));
    if (!pimg || !pout || !pmask || !hp.data) return -1;

    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x) {
            int64_t s = y * W + x, d = (y + 1) * Wp + (x + 1);
            int8_t m = mask ? (mask[s] != 0) : 1;
            pimg[d] = image[s];
            pmask[d] = m;
            pout[d] = m ? markers[s] : 0; /* markers.astype(int32) * mask */
        }

    int64_t nb1[4] = {-Wp, -1, +1, +Wp};
    int64_t nb2[8] = {-Wp, +1, -1, +Wp, -Wp - 1, -Wp + 1, +Wp - 1, +Wp + 1};
    const int64_t* nb = connectivity == 1 ? nb1 : nb2;
    const int nnb = connectivity == 1 ? 4 : 8;

    elem_t e, n;
    for (int64_t i = 0; i < Np; ++i) {
        if (pout[i] != 0) { /* marker pixels in raster order of the padded image */
            e.value = pimg[i];
            e.age = 0;
            e.index = i;
            heap_push(&hp, &e);
        }
    }
    int64_t age = 0;
    while (hp.items > 0) {
        heap_pop(&hp, &e);
        for (int k = 0; k < nnb; ++k) {
            int64_t j = e.index + nb[k];
            if (!pmask[j]) continue;
            if (pout[j]) continue;
            age += 1;
            pout[j] = pout[e.index];
            n.value = pimg[j];
            n.age = age;
            n.index = j;
            heap_push(&hp, &n);
        }
    }
    for (int64_t y = 0; y < H; ++y)
        for (int64_t x = 0; x < W; ++x) out[y * W + x] = pout[(y + 1) * Wp + (x + 1)];
    free(pimg);
    free(pout);
    free(pmask);
    free(hp.data);
    return 0;
}
