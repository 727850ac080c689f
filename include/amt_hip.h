/* amt_hip.h -- C ABI of libamt_hip.so: the MI355X (gfx950) implementation of the
 * arcadia-microscopy-tools per-image preprocessing + segmentation + region-props hot path.
 *
 * Nothing native exists upstream: the reference (pure Python) reaches its arithmetic through
 * scikit-image / scipy.ndimage / numpy calls.  Each entry point below therefore cites the
 * REFERENCE CALL SITE it serves (R/ = src/arcadia_microscopy_tools/) and the scikit-image /
 * scipy function whose result it reproduces (SK/ = skimage 0.18.3 source, SP/ = scipy.ndimage).
 * INTEGRATION.md shows the ctypes stubs a maintainer of the reference would add.
 *
 * Conventions
 *   - every image argument is a DEVICE pointer to a batch of `nplanes` C-contiguous (H, W)
 *     planes of the stated element type; planes are independent (one per FOV-channel);
 *   - functions enqueue work on the context's HIP stream and return without synchronising
 *     unless stated; amt_sync() waits for the stream;
 *   - return value: 0 on success, a negative AMT_E* code otherwise; amt_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - a context is not thread-safe; use one context per host thread (they are cheap), which
 *     makes the library re-entrant for Pipeline(parallel=True) (R/pipeline.py:145-146);
 *   - STREAMS: every amt_* call runs on the stream of the context it is GIVEN -- never on the stream that
 *     produced its operands.  Memory written through context A may be read through context B only after B's
 *     stream has been ordered behind A's: amt_stream_wait(B, A) (everything A was given so far), or
 *     amt_event_record(A, ev) + amt_event_wait(B, ev) (exactly up to the record), or amt_sync(A).  Nothing in the
 *     library checks this (pointers carry no owner); a binding that keeps arrays with their context should refuse
 *     mixed-context calls, as the Python binding does for `out=` (tests/test_gpu_api.py::test_stream_rule_at_the_boundary).
 *     Host memory handed to amt_memcpy_h2d / amt_memcpy_d2h must stay valid until the stream has passed the copy.
 */
#ifndef AMT_HIP_H
#define AMT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMT_OK 0
#define AMT_EINVAL (-1)   /* bad argument */
#define AMT_EHIP (-2)     /* HIP runtime error */
#define AMT_ENOMEM (-3)   /* device allocation failed */
#define AMT_ENODEV (-4)   /* no usable gfx950 device */
#define AMT_ECAPACITY (-5)/* a caller-provided capacity was too small (see amt_last_error) */

/* element types */
#define AMT_U8 0
#define AMT_U16 1
#define AMT_I32 2
#define AMT_F64 3
#define AMT_I64 4
#define AMT_F32 5

/* scipy.ndimage boundary modes */
#define AMT_MODE_NEAREST 0
#define AMT_MODE_REFLECT 1  /* half-sample symmetric  d c b a | a b c d | d c b a */
#define AMT_MODE_MIRROR 2   /* whole-sample symmetric d c b | a b c d | c b a     */
#define AMT_MODE_CONSTANT 3
#define AMT_MODE_WRAP 4

typedef struct amt_ctx amt_ctx;

/* ---- context, memory, stream ------------------------------------------------------------- */
int amt_device_count(void);
int amt_ctx_create(int device, amt_ctx** out);
/* Share an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); 0 = null stream. */
int amt_ctx_create_on_stream(int device, void* hip_stream, amt_ctx** out);
int amt_ctx_destroy(amt_ctx* ctx);
/* Whether independent kernels inside one call (the watershed's flood classes) run on the context's auxiliary streams
 * (default 1; AMT_FORK=0 changes the default).  Hosts that drive several contexts concurrently switch it off. */
int amt_ctx_set_fork(amt_ctx* ctx, int enable);
/* The context's hipStream_t (e.g. to wrap it in torch.cuda.ExternalStream for an RCCL collective). */
int amt_ctx_stream(amt_ctx* ctx, void** hip_stream);
const char* amt_last_error(void);
const char* amt_version(void);
int amt_device_name(amt_ctx* ctx, char* buf, int buflen);
int amt_malloc(amt_ctx* ctx, size_t bytes, void** dptr);
int amt_free(amt_ctx* ctx, void* dptr);
int amt_memcpy_h2d(amt_ctx* ctx, void* dst, const void* src, size_t bytes); /* async on ctx stream */
int amt_memcpy_d2h(amt_ctx* ctx, void* dst, const void* src, size_t bytes); /* async; amt_sync before reading */
int amt_memcpy_d2d(amt_ctx* ctx, void* dst, const void* src, size_t bytes);
int amt_memset(amt_ctx* ctx, void* dst, int value, size_t bytes);
int amt_sync(amt_ctx* ctx);
/* Order `ctx`'s stream after everything enqueued so far on `other`'s stream (same device), without
 * blocking the host: joins batch parts processed on separate streams before a collective. */
int amt_stream_wait(amt_ctx* ctx, amt_ctx* other);
/* Events: amt_event_record marks a point on ctx's stream; amt_event_wait orders what is enqueued next on ANOTHER
 * context's stream after that point (not after everything the recording stream was given later).  Used by the
 * plate's staging ring: a block is rewritten only after the pack kernel that read it (plate.PlateTables). */
int amt_event_create(amt_ctx* ctx, void** event);
int amt_event_record(amt_ctx* ctx, void* event);
int amt_event_wait(amt_ctx* ctx, void* event);
int amt_event_sync(amt_ctx* ctx, void* event); /* host waits for the last record */
int amt_event_destroy(amt_ctx* ctx, void* event);
/* pinned host staging buffers for the FOV feeder */
int amt_host_alloc(size_t bytes, void** hptr);
int amt_host_free(void* hptr);
/* host memcpy with streaming stores, for filling a staging block the DMA engine reads next (thread-safe, no GPU call) */
int amt_host_copy(void* dst, const void* src, size_t bytes);
/* host helpers of the constructor checks and the label upload of SegmentationMask (R/masks.py:169-176: "non-negative",
   "contains no cells"): extrema of an integer array in one call -> out[2] = {min, max}; int64 -> int32 narrowing into a
   staging block with the source extrema from the same pass (minmax may be NULL) */
int amt_host_minmax_int(const void* src, int itemsize, int is_signed, size_t n, int64_t* out);
int amt_host_narrow_i64_i32(int32_t* dst, const int64_t* src, size_t n, int64_t* minmax);
/* HIP-event timing on the context's stream (bench.py roofline: kernel time measured live) */
int amt_timer_create(amt_ctx* ctx, void** timer);
int amt_timer_start(amt_ctx* ctx, void* timer);
int amt_timer_stop(amt_ctx* ctx, void* timer);
int amt_timer_elapsed_ms(amt_ctx* ctx, void* timer, float* ms); /* synchronises on the stop event */
int amt_timer_destroy(amt_ctx* ctx, void* timer);

/* ---- channel access: R/microscopy.py:241-282 (get_channel_intensities) --------------------- */
/* (C,Y,X) stacks are channel-major, so a channel is a pointer offset; ND2 frames are (Y,X,C)
 * interleaved: this de-interleaves nplanes frames into (C,Y,X). */
int amt_deinterleave_u16(amt_ctx* ctx, const uint16_t* yxc, uint16_t* cyx, int nplanes, int H, int W, int C);

/* ---- filters: R/operations.py:91 (difference_of_gaussians), SK/filters/_gaussian.py,
 *      SP/_filters.py:226-236,314-430 (gaussian_filter1d -> correlate1d) ------------------------ */
/* Separable symmetric correlation, axis 0 then axis 1, float64 accumulate in scipy's order
 * (centre tap, then pairs outermost->innermost); `weights` = HOST pointer to 2*radius+1 doubles as
 * scipy builds them (computed by the caller with numpy so that np.exp rounding is shared).
 * in_dtype AMT_U16 (scaled by `scale`, 1/65535 for img_as_float) or AMT_F64 (scale ignored if 1).
 * in_plane_stride = elements between consecutive INPUT planes (0 = H*W); C*H*W filters one channel of
 * every (C,Y,X) field of view of a batch in a single launch.  Output planes are always contiguous.
 * minmax_dev (nullable) receives [min, max] of every OUTPUT plane (2 doubles per plane), folded in while the
 * result is written -- hand it to amt_threshold_value to spare Otsu a full re-read of the image. */
int amt_gaussian(amt_ctx* ctx, const void* in, int in_dtype, double scale, double* out, int nplanes, int H, int W,
                 const double* weights, int radius, int mode, double cval, size_t in_plane_stride,
                 double* minmax_dev);
/* threshold_otsu(gaussian(in)) per plane WITHOUT materialising the float64 image -- the first three calls of the
 * nuclei recipes (ski.filters.gaussian -> ski.filters.threshold_otsu -> `>`; SURVEY.md A.7/A.8, reached through
 * R/pipeline.py:25-45).  uint16 input, fused radii (1..12), aligned rows: ask amt_gaussian_otsu_codes_supported first.
 * Outputs (all caller-owned): minmax_dev[2 n] = range of the smoothed plane, hist_dev[256 n] = np.histogram counts,
 * thr_dev[n] = the Otsu threshold (bit-identical to amt_gaussian + amt_threshold_value), codes[n H W] = uint16 code
 * per pixel with   gaussian(in) > thr  <=>  code > thr_code_dev[plane]   exactly, so amt_threshold_gt /
 * amt_threshold_open_close on (codes, AMT_U16, thr_code_dev) produce the mask of the separate operators. */
int amt_gaussian_otsu_codes_supported(int H, int W, int radius, int mode, size_t in_plane_stride);
int amt_gaussian_otsu_codes(amt_ctx* ctx, const uint16_t* in, double scale, int nplanes, int H, int W,
                            const double* weights, int radius, int mode, size_t in_plane_stride, double* minmax_dev,
                            uint32_t* hist_dev, double* thr_dev, double* thr_code_dev, uint16_t* codes);
/* out = G(w_lo) - G(w_hi) of the same converted input (SK/filters/_gaussian.py:284-290). */
/* One leading axis of an n-D Gaussian (skimage filters EVERY axis of an n-D image, leading axes first,
 * SP/_filters.py:423-427): the array is nplanes x L x inner, the 1-D filter runs along L; out is float64. */
int amt_convolve_axis0(amt_ctx* ctx, const void* in, int in_dtype, double scale, double* out, int nplanes, int L, int inner,
                       const double* weights, int radius, int mode, double cval);
int amt_dog(amt_ctx* ctx, const void* in, int in_dtype, double scale, double* out, int nplanes, int H, int W,
            const double* w_lo, int r_lo, const double* w_hi, int r_hi, int mode, double cval);

/* ---- elementwise (R/operations.py:50,97; SK/exposure/exposure.py:405-428) ------------------- */
/* out = max(in - level[plane], 0)  -- np.clip(dog - background_level, 0, None) */
int amt_sub_clip0_f64(amt_ctx* ctx, const double* in, const double* level_dev, double* out, int nplanes, size_t n);
/* rescale_intensity with tuple ranges: clip to [imin,imax]; (x-imin)/(imax-imin)*(omax-omin)+omin.
 * range_dev = nplanes x {imin, imax} doubles on the device. */
int amt_rescale(amt_ctx* ctx, const void* in, int in_dtype, const double* range_dev, double omin, double omax,
                double* out, int nplanes, size_t n);
int amt_convert_u16_f64(amt_ctx* ctx, const uint16_t* in, double scale, double* out, size_t n);
/* out = in + s (threshold_local subtracts its offset this way, SK/filters/thresholding.py:236) */
int amt_add_scalar_f64(amt_ctx* ctx, const double* in, double s, double* out, size_t n);

/* ---- statistics: np.percentile (R/operations.py:47,94), histogram (SK/exposure/exposure.py) -- */
/* One 65536-bin histogram per plane (uint32 counts), bin = pixel value. */
int amt_hist_u16(amt_ctx* ctx, const uint16_t* in, uint32_t* hist, int nplanes, size_t n);
/* One bin per integer value lo .. lo + nbins - 1 of an integer-valued float64 image (scikit-image's histogram of
 * integer images, SK/exposure/exposure.py:63-74, for ranges beyond uint16): hist = nplanes x nbins uint32; values outside
 * the range are not counted.  nbins <= 2^28. */
int amt_hist_range_f64(amt_ctx* ctx, const double* in, double lo, int64_t nbins, uint32_t* hist, int nplanes, size_t n);
/* minmax_dev[plane] = {min, max} */
int amt_minmax_f64(amt_ctx* ctx, const double* in, double* minmax_dev, int nplanes, size_t n);
/* np.histogram(x, bins=nbins, range=(min,max)) counts per plane; edges follow np.linspace. */
int amt_hist_f64(amt_ctx* ctx, const double* in, const double* minmax_dev, uint32_t* hist, int nbins, int nplanes,
                 size_t n);
/* np.percentile(x, q) (linear): q_host = nq percentiles in [0,100]; out_dev = nplanes x nq doubles.
 * u16: exact order statistics from the histogram; f64: exact radix select. */
int amt_percentile_u16(amt_ctx* ctx, const uint16_t* in, const double* q_host, int nq, double* out_dev, int nplanes,
                       size_t n);
int amt_percentile_f64(amt_ctx* ctx, const double* in, const double* q_host, int nq, double* out_dev, int nplanes,
                       size_t n);

/* out_dev[plane] = {sum(x <= t), count(x <= t), sum(x > t), count(x > t)} with t = thr_dev[plane]; a fixed
 * reduction tree (bit-reproducible).  Feeds threshold_mean / threshold_li on float images
 * (SK/filters/thresholding.py:830, :642-707). */
int amt_masked_sums_f64(amt_ctx* ctx, const double* in, const double* thr_dev, double* out_dev, int nplanes, size_t n);
/* dst[plane] (h x w) = src[plane][top:top+h, left:left+w]; crop_to_center (R/operations.py:100-132). */
/* dst[plane] ((H + 2 py) x (W + 2 px)) = np.pad(src[plane], ((py, py), (px, px)), mode='edge'): what scikit-image does to
 * the image before an opening / closing with an even-sized footprint (SK/morphology/grey.py:84-127). */
int amt_pad_edge(amt_ctx* ctx, const void* src, void* dst, int elem_size, int nplanes, int H, int W, int py, int px);
int amt_copy_rect(amt_ctx* ctx, const void* src, void* dst, int elem_size, int nplanes, int H, int W, int top,
                  int left, int h, int w);

/* ---- thresholds: R/operations.py:186-216 (apply_threshold), SK/filters/thresholding.py ------- */
#define AMT_THR_OTSU 0
#define AMT_THR_YEN 1
#define AMT_THR_ISODATA 2
#define AMT_THR_TRIANGLE 3
#define AMT_THR_MEAN 4
#define AMT_THR_MINIMUM 5
#define AMT_THR_LI 6
/* Global threshold value per plane (thr_dev[plane], float64 -- for integer images the bin centre).
 * status_dev[plane] != 0 flags "no threshold" (threshold_minimum: fewer/more than two maxima).
 * minmax_dev (nullable, float64 input only) = precomputed [min, max] per plane (amt_gaussian / amt_minmax_f64). */
int amt_threshold_value(amt_ctx* ctx, const void* in, int in_dtype, int method, int nbins, double* thr_dev,
                        int32_t* status_dev, int nplanes, size_t n, const double* minmax_dev);
/* out = in > thr[plane] (uint8 0/1) */
int amt_threshold_gt(amt_ctx* ctx, const void* in, int in_dtype, const double* thr_dev, uint8_t* out, int nplanes,
                     size_t n);
/* Niblack (method 0: m - k*s) / Sauvola (method 1: m*(1 + k*(s/r - 1))) threshold image from the mean m and
 * standard deviation s of the window_size^2 square around each pixel, mirror padding
 * (SK/filters/thresholding.py:910-964,1026-1027,1083-1087).  uint16 window sums are exact integers. */
int amt_window_threshold(amt_ctx* ctx, const void* in, int in_dtype, double* thr_image, int nplanes, int H, int W,
                         int window_size, int method, double k, double r);
/* The same with a window of window_y rows x window_x columns (scikit-image's per-axis `window_size` tuple). */
int amt_window_threshold_yx(amt_ctx* ctx, const void* in, int in_dtype, double* thr_image, int nplanes, int H, int W,
                            int window_y, int window_x, int method, double k, double r);
/* The same for ONE n-D image whose window spans every axis, as scikit-image's _mean_std does for stacks
 * (SK/filters/thresholding.py:910-964): nlead axes of lead_shape[] in front of (H, W), one odd window per axis. */
int amt_window_threshold_nd(amt_ctx* ctx, const void* in, int in_dtype, double* thr_image, int nlead,
                            const int* lead_shape, const int* lead_window, int H, int W, int window_y, int window_x,
                            int method, double k, double r);
/* out = in > thr_image (per-pixel thresholds: local / niblack / sauvola) */
int amt_threshold_gt_image(amt_ctx* ctx, const void* in, int in_dtype, const double* thr_image, uint8_t* out,
                           size_t n);

/* ---- morphology: SK/morphology/binary.py, grey.py (ImageOperation callables, north_star) ----- */
/* Footprint = HOST uint8 (fh, fw), odd sizes, anchor at the centre.
 * binary erosion: outside the image counts as `border_value` (skimage: 1); dilation: 0. */
int amt_binary_erode(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                     const uint8_t* footprint, int fh, int fw, int border_value);
int amt_binary_dilate(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                      const uint8_t* footprint, int fh, int fw, int border_value);
/* fused opening = dilate(erode(x)) and closing = erode(dilate(x)) with skimage's border rules */
int amt_binary_open(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                    const uint8_t* footprint, int fh, int fw);
int amt_binary_close(amt_ctx* ctx, const uint8_t* in, uint8_t* out, int nplanes, int H, int W,
                     const uint8_t* footprint, int fh, int fw);
/* out = binary_closing(binary_opening(in > thr[plane])) in one packed chain (the mask chain of BASELINE
 * configs[1]/[2]: R/operations.py:216 followed by SK/morphology/binary.py:82-147). */
int amt_threshold_open_close(amt_ctx* ctx, const void* in, int in_dtype, const double* thr_dev, uint8_t* out,
                             int nplanes, int H, int W, const uint8_t* footprint, int fh, int fw);
/* The same chain for a float64 image whose Otsu threshold came from amt_otsu_f64_bins: the comparison reads the byte
 * plane of bin indices (1 B/px) and the float64 value only inside the threshold's own bin -- identical masks. */
int amt_otsu_f64_bins(amt_ctx* ctx, const double* in, const double* minmax_dev, double* thr_dev, double* thr_code_dev,
                      uint8_t* bins, int nplanes, size_t n);
int amt_threshold_open_close_bins(amt_ctx* ctx, const double* in, const uint8_t* bins, const double* thr_dev,
                                  const double* thr_code_dev, uint8_t* out, int nplanes, int H, int W,
                                  const uint8_t* footprint, int fh, int fw);
/* grey erosion / dilation / median over a footprint (uint16 or float64 images), scipy boundary `mode`.
 * op: 0 = erosion (min), 1 = dilation (max, footprint already mirrored by the caller), 2 = median */
int amt_rank_filter(amt_ctx* ctx, const void* in, void* out, int dtype, int nplanes, int H, int W,
                    const uint8_t* footprint, int fh, int fw, int op, int mode, double cval);
/* out = minuend - rank_filter(in): the last step of ndi.white_tophat (R/ callers reach it through
 * skimage.morphology.white_tophat, SK/morphology/grey.py:425) without a separate subtraction pass; uint16 run
 * footprints subtract inside the filter kernel, everything else filters into `out` and subtracts in place. */
int amt_rank_filter_sub(amt_ctx* ctx, const void* in, const void* minuend, void* out, int dtype, int nplanes, int H,
                        int W, const uint8_t* footprint, int fh, int fw, int op, int mode, double cval);
/* out = a - b (same dtype; white_tophat = image - opening) */
int amt_subtract(amt_ctx* ctx, const void* a, const void* b, void* out, int dtype, size_t n);

/* ---- labelling: R/masks.py:56-65 (clear_border, label, relabel_sequential) ------------------- */
/* Connected components of equal-valued non-zero pixels (uint8 mask or int32 label image),
 * connectivity 1 (4-conn) or 2 (8-conn, skimage default in 2-D); labels 1..K per plane in raster
 * order of each component's first pixel; count_dev[plane] = K. */
int amt_label(amt_ctx* ctx, const void* in, int in_dtype, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
              int connectivity);
/* amt_label for a uint8 plane batch that is a TRUTH VALUE (foreground = byte != 0; what skimage.measure.label does with
 * a bool array, R/masks.py:63 on a mask): same labels as amt_label on the 0 / 1 image, without the launches that stand by
 * for "other byte values" (planes whose width is a multiple of 16; others take amt_label's path and REQUIRE 0 / 1 bytes). */
int amt_label_mask(amt_ctx* ctx, const uint8_t* mask, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
                   int connectivity);
/* Same result as amt_label for uint8 masks with at most `capacity` foreground pixels per plane (e.g. the EDT
 * peak markers): foreground is compacted in raster order and labelled on the compact list.  If a plane has
 * more foreground pixels than `capacity`, count_dev[plane] = -1 and that plane's labels are invalid. */
/* The same, for callers that label into the SAME plane batch again and again (a batch driver's marker planes): `out` must
 * be zero except at the pixels listed in keep_list[plane * capacity ...][0 .. keep_count[plane]) -- the state this
 * function leaves behind; start from a zeroed `out` and zeroed keep_count.  Only those pixels are cleared (no
 * full-plane memset), and the lists are replaced by this call's foreground pixels. */
int amt_label_sparse_reuse(amt_ctx* ctx, const uint8_t* in, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
                           int connectivity, int capacity, int32_t* keep_list, int32_t* keep_count);
int amt_label_sparse(amt_ctx* ctx, const uint8_t* in, int32_t* out, int32_t* count_dev, int nplanes, int H, int W,
                     int connectivity, int capacity);
/* skimage.segmentation.clear_border(labels) with buffer_size=0: zero every 8-connected component of
 * equal-valued pixels that touches the 1-px frame; other pixels keep their value. */
int amt_clear_border(amt_ctx* ctx, const int32_t* in, int32_t* out, int nplanes, int H, int W);
/* relabel_sequential: surviving labels -> 1..K in ascending order of the old label; count_dev = K.
 * max_label = upper bound of label values in `in` (any plane). */
int amt_relabel_sequential(amt_ctx* ctx, const int32_t* in, int32_t* out, int32_t* count_dev, int nplanes, size_t n,
                           int max_label);
/* clear_border followed by relabel_sequential (R/masks.py:56,65) in one pass, valid when every label of
 * `in` is a single connected component (outputs of amt_label and amt_watershed_*): a label is removed
 * iff one of its pixels lies on the 1-px frame; survivors -> 1..K in ascending old-label order.
 * nlabels_dev (nullable): the caller vouches that plane p holds exactly the labels 1..nlabels_dev[p] (true for a
 * watershed of a mask from markers numbered 1..K that lie inside the mask); the presence pass is then skipped. */
int amt_clear_border_relabel(amt_ctx* ctx, const int32_t* in, int32_t* out, int32_t* count_dev, int nplanes, int H,
                             int W, int max_label, const int32_t* nlabels_dev);
/* np.where(np.isin(labels, keep), labels, 0): keep_dev = nplanes x (max_label+1) uint8 flags */
int amt_keep_labels(amt_ctx* ctx, const int32_t* in, const uint8_t* keep_dev, int32_t* out, int nplanes, size_t n,
                    int max_label);
int amt_cast_i32_i64(amt_ctx* ctx, const int32_t* in, int64_t* out, size_t n);

/* ---- distance transform, markers, watershed (north_star) ------------ */
/* Exact squared Euclidean distance to the nearest zero pixel (int32), and its correctly rounded
 * float64 square root = scipy.ndimage.distance_transform_edt.  Either output may be NULL. */
int amt_edt(amt_ctx* ctx, const uint8_t* mask, int32_t* d2_out, double* edt_out, int nplanes, int H, int W);
/* peaks = (d2 == maximum_filter(d2, (2m+1)^2, constant 0)) & mask & (d2 > 0), border of width m cleared */
/* The same into a plane batch that is zero except at the pixels listed in prev_list / prev_count (the lists
 * amt_label_sparse_reuse keeps of the previous run's peaks): only those are cleared, no full-plane memset.
 * prev_status = that run's label counts (count_dev of amt_label_sparse_reuse): a plane whose count is -1 overflowed
 * its list and is cleared whole.  Start from zeroed planes, zeroed counts and zeroed status. */
int amt_peak_mask_reuse(amt_ctx* ctx, const int32_t* d2, const uint8_t* mask, uint8_t* peaks, int nplanes, int H, int W,
                        int min_distance, const int32_t* prev_list, const int32_t* prev_count, int capacity,
                        const int32_t* prev_status);
int amt_peak_mask(amt_ctx* ctx, const int32_t* d2, const uint8_t* mask, uint8_t* peaks, int nplanes, int H, int W,
                  int min_distance);
/* Priority flood restricted to `mask` (skimage.segmentation.watershed, no compactness, no watershed line;
 * SURVEY.md A.1).  Priority = (value, insertion age); labels are assigned at push time.
 * `mask` holds 0 / 1 bytes (what a bool array is on the device): scikit-image takes the mask as a truth value, and only
 * amt_watershed_edt_cleared_sparse on widths that are a multiple of 16 reads other non-zero bytes that way -- elsewhere
 * differing non-zero byte values would split a region.
 *   amt_watershed_edt : relief = -sqrt(d2) given as the exact integer d2 (bucket queue).  seeds_first = 1 is the
 *                       config-3 recipe (oracle/skops.py:seeded_flood_image): marker pixels are spread first, in
 *                       raster order -- i.e. the relief with every marker pixel lowered to a distinct lowest value;
 *   amt_watershed_f64 : general float64 relief (per-component binary heap).
 * Equal-valued age-0 MARKER pixels: scikit-image orders them by the moves of its single binary heap, which depends
 * on everything else in the image, so such planes cannot be flooded component by component.  tie_policy:
 *   AMT_WS_TIES_EXACT   planes in which two markers of one mask component (with more than one label) tie are
 *                       re-flooded by a sequential emulation of scikit-image's heap: bit-identical, one lane,
 *                       ~seconds per 2048 x 2048 plane.  Untied planes keep the parallel result (also bit-identical).
 *   AMT_WS_TIES_RASTER  ties are broken in raster order (fast; differs from scikit-image in tied planes).
 *   AMT_WS_TIES_REPORT  as RASTER, and ties_dev[plane] = 1 marks every plane whose result may differ.
 * ties_dev (nullable, nplanes ints) receives the per-plane tie flags under every policy.
 * connectivity 2 (8 neighbours, visited N, E, W, S, NW, NE, SW, SE as scikit-image 0.18.3 does) runs
 * the sequential emulation for every plane and requires AMT_WS_TIES_EXACT.  The two-argument-less entry points are
 * connectivity 1 with AMT_WS_TIES_EXACT. */
#define AMT_WS_TIES_EXACT 0
#define AMT_WS_TIES_RASTER 1
#define AMT_WS_TIES_REPORT 2
int amt_watershed_edt(amt_ctx* ctx, const int32_t* d2, const int32_t* markers, const uint8_t* mask, int32_t* out,
                      int nplanes, int H, int W, int seeds_first);
int amt_watershed_f64(amt_ctx* ctx, const double* relief, const int32_t* markers, const uint8_t* mask, int32_t* out,
                      int nplanes, int H, int W);
int amt_watershed_edt_ex(amt_ctx* ctx, const int32_t* d2, const int32_t* markers, const uint8_t* mask, int32_t* out,
                         int nplanes, int H, int W, int seeds_first, int connectivity, int tie_policy, int32_t* ties_dev);
int amt_watershed_f64_ex(amt_ctx* ctx, const double* relief, const int32_t* markers, const uint8_t* mask, int32_t* out,
                         int nplanes, int H, int W, int connectivity, int tie_policy, int32_t* ties_dev);
/* The config-3 tail in one call: amt_watershed_edt(seeds_first = 1) followed by amt_clear_border_relabel
 * (R/masks.py:56 + :65 on the watershed of a mask from markers numbered 1..nlabels_dev[plane]; labels that touch the
 * frame are dropped, the rest renumbered 1..count_dev[plane] in ascending order).  Same labels as the two calls, one
 * full-plane write less: the watershed image itself is never materialised -- ws_scratch (nplanes x H x W ints) only
 * receives the pixels of flooded components. */
int amt_watershed_edt_cleared(amt_ctx* ctx, const int32_t* d2, const int32_t* markers, const uint8_t* mask,
                              int32_t* ws_scratch, int32_t* labels_out, int32_t* count_dev, int nplanes, int H, int W,
                              int max_label, const int32_t* nlabels_dev);
/* The same for callers that hold the LIST of marker pixels (the keep_list / keep_count amt_label_sparse_reuse leaves
 * behind: marker_list[plane * list_capacity ...][0 .. marker_count[plane]) = flat indices of every non-zero pixel of the
 * marker plane): the per-component marker statistics come from the list and the dense statistics pass does not read the
 * marker plane (4 of its 12 bytes per pixel).  A plane whose list is incomplete (count above the capacity) gives
 * undefined labels -- amt_label_sparse_reuse reports that plane with count -1. */
int amt_watershed_edt_cleared_sparse(amt_ctx* ctx, const int32_t* d2, const int32_t* markers, const uint8_t* mask,
                                     int32_t* ws_scratch, int32_t* labels_out, int32_t* count_dev, int nplanes, int H,
                                     int W, int max_label, const int32_t* nlabels_dev, const int32_t* marker_list,
                                     const int32_t* marker_count, int list_capacity);

/* ---- region properties: R/masks.py:286-326 (regionprops_table) ------------------------------- */
/* Morphology columns per label 1..max_label (row = label-1), float64, column order: */
#define AMT_RP_AREA 0
#define AMT_RP_CENTROID_Y 1
#define AMT_RP_CENTROID_X 2
#define AMT_RP_BBOX_Y0 3
#define AMT_RP_BBOX_X0 4
#define AMT_RP_BBOX_Y1 5
#define AMT_RP_BBOX_X1 6
#define AMT_RP_PERIMETER 7
#define AMT_RP_AXIS_MAJOR 8
#define AMT_RP_AXIS_MINOR 9
#define AMT_RP_ECCENTRICITY 10
#define AMT_RP_ORIENTATION 11
#define AMT_RP_AREA_CONVEX 12
#define AMT_RP_SOLIDITY 13
#define AMT_RP_NCOLS 14
/* table_dev = nplanes x max_label x AMT_RP_NCOLS doubles.  Labels absent from a plane give area 0. */
int amt_regionprops(amt_ctx* ctx, const int32_t* labels, double* table_dev, int nplanes, int H, int W, int max_label);
/* Intensity columns per label and channel: {mean, max, min, std} (population std).
 * intensity = nplanes x C planes of uint16 (one (C,Y,X) FOV per label plane);
 * table_dev = nplanes x max_label x C x 4 doubles. */
int amt_regionprops_intensity_u16(amt_ctx* ctx, const int32_t* labels, const uint16_t* intensity, int C,
                                  double* table_dev, int nplanes, int H, int W, int max_label);
/* Both tables in one call (the bounding-box pass and the per-label scan are shared): what
 * SegmentationMask.cell_properties needs for a (C,Y,X) field of view (R/masks.py:286-326). */
int amt_regionprops_full_u16(amt_ctx* ctx, const int32_t* labels, const uint16_t* intensity, int C, double* table_dev,
                             double* itable_dev, int nplanes, int H, int W, int max_label);
/* The same {mean, max, min, std} for float64 intensity images (R/masks.py:178-190, :319-323 accept any 2-D ndarray):
 * two sweeps per label as np.mean / np.std make them; float64 sums, so agreement with numpy is ~1e-15 relative. */
int amt_regionprops_intensity_f64(amt_ctx* ctx, const int32_t* labels, const double* intensity, int C, double* table_dev,
                                  int nplanes, int H, int W, int max_label);
int amt_max_i32(amt_ctx* ctx, const int32_t* in, int32_t* max_dev, int nplanes, size_t n);

/* ---- per-plate feature rows (the reference's analogue is the list of cell_properties dicts a
 * user collects per image, R/masks.py:247-328, R/pipeline.py:145-149) ------------------------------------------
 * Compacts the dense tables of B fields of view (table_dev B x K x AMT_RP_NCOLS, itable_dev B x K x C x 4,
 * ncells_dev B int32) into one row per cell, fields of view in order, labels 1..ncells[b] in order:
 *   row = [fov index, label, the AMT_RP_NCOLS morphology columns, C x {mean, max, min, std}]  (16 + 4 C doubles)
 * The fov index of plane b is fov_index_dev[b], or fov_index0 + b when fov_index_dev is NULL.  rows_cap (in rows)
 * must be >= B * K.  *nrows_dev = total rows, or -1 when a field of view holds a count outside [0, K] (its table
 * overflowed).  This block is what a rank contributes to the plate's all-gather: counts first, then rows. */
int amt_pack_plate_rows(amt_ctx* ctx, const double* table_dev, const double* itable_dev, const int32_t* ncells_dev, int B,
                        int K, int C, const int32_t* fov_index_dev, int fov_index0, double* rows_dev, size_t rows_cap,
                        int64_t* nrows_dev);

/* ---- cell outlines: R/masks.py:82-115 (_extract_outlines_skimage), SK/measure/_find_contours.py ------
 * bbox_dev = nplanes x max_label x 4 ints {min row, min col, max row, max col} INCLUSIVE; labels absent from a
 * plane give max < min. */
int amt_label_bboxes(amt_ctx* ctx, const int32_t* labels, int32_t* bbox_dev, int nplanes, int H, int W, int max_label);
/* Marching squares at level 0.5 on the crop of every listed label of ONE label plane (H x W int32), longest
 * contour kept (first among equals), in scikit-image's vertex order.  Two calls with a host step in between
 * (the outline lengths size the output):
 *   boxes_dev   nlab x 5 ints {label, r_lo, c_lo, r_hi, c_hi}: the bounding box padded by one pixel and clamped
 *               to the image, half-open (R/masks.py:99-104)
 *   voff_dev    nlab + 1 offsets into visited_dev, one scratch byte per square ((h-1) x (w-1)) of each crop
 *   info_dev    nlab x 4 ints {points, closed, start square, start segment}; points = 0 for an empty outline
 *   poff_dev    nlab + 1 offsets (in points) into points_dev = (row, col) float64 pairs, image coordinates */
int amt_contours_find(amt_ctx* ctx, const int32_t* labels, int H, int W, int nlab, const int32_t* boxes_dev,
                      const int64_t* voff_dev, uint8_t* visited_dev, size_t visited_bytes, int32_t* info_dev);
int amt_contours_emit(amt_ctx* ctx, const int32_t* labels, int H, int W, int nlab, const int32_t* boxes_dev,
                      const int32_t* info_dev, const int64_t* poff_dev, double* points_dev);

/* Pixel-border outlines, the reference's default extractor (R/masks.py:68-79: cellpose.utils.outlines_list ->
 * cv2.findContours(masks == n, RETR_EXTERNAL, CHAIN_APPROX_NONE), contour with the most points, newest first among
 * equals).  Suzuki-Abe border following on ONE int32 label plane, one walk per outer border.  Parity unpinned: no
 * OpenCV / cellpose offline and no vector in the reference's tests (see oracle/contours.py).
 *   boxes_dev   nlab x 5 ints {label, r_lo, c_lo, r_hi, c_hi}: the TIGHT bounding box, half-open
 *   marks_dev   H x W scratch bytes (cleared by the call)
 *   info_dev    nlab x 3 ints {points, start row, start col} of the selected border
 *   poff_dev    nlab + 1 offsets (in points) into points_dev = (row, col) int32 pairs; a label whose slot is not
 *               exactly its point count is skipped (the host drops borders of fewer than five points) */
int amt_borders_find(amt_ctx* ctx, const int32_t* labels, int H, int W, int nlab, const int32_t* boxes_dev,
                     int8_t* marks_dev, int32_t* info_dev);
int amt_borders_emit(amt_ctx* ctx, const int32_t* labels, int H, int W, int nlab, const int32_t* boxes_dev,
                     const int32_t* info_dev, const int64_t* poff_dev, int32_t* points_dev);

/* ---- Cellpose post-processing: flows + cell probability -> labels (BASELINE configs[4]; the reference reaches it
 * through CellposeModel.eval, R/model.py:206-215, :270-290).  Restated from the published algorithm (cellpose
 * dynamics: follow the flow for niter Euler steps with bilinear sampling, histogram the end points, seeds = 5 x 5
 * maxima with more than 10 pixels, five rounds of 3 x 3 growth over bins with more than 2 pixels, labels by end point,
 * masks above max_size_fraction of the image or below min_size dropped, renumbered in raster order); the flow-error
 * filter and hole filling: amt_cellpose_masks_ex.  PARITY UNPINNED: cellpose is not available offline and the
 * reference holds no vector for it (oracle: oracle/cellpose_dynamics.py).
 *   dP = nplanes x 2 x H x W float32 (dY, dX), cellprob = nplanes x H x W float32, labels_out = nplanes x H x W int32,
 *   count_dev[plane] = number of masks, or -1 if the plane produced more than max_seeds seeds. */
int amt_cellpose_masks(amt_ctx* ctx, const float* dP, const float* cellprob, int32_t* labels_out, int32_t* count_dev,
                       int nplanes, int H, int W, float cellprob_threshold, int niter, int min_size,
                       float max_size_fraction, int max_seeds);
/* The same with the rest of cellpose's compute_masks, as CellposeModel.eval runs it with the parameters the reference
 * hands over (R/model.py:206-215: flow_threshold, default 0.4):
 *   flow_threshold > 0: remove_bad_flow_masks -- the flows are re-derived from the masks (float64 heat diffusion from
 *       each mask's centre, 2 * max(height + width + 2) steps) and a mask whose mean squared difference to dP / 5
 *       exceeds the threshold is dropped;
 *   then fill_holes_and_remove_small_masks: masks below min_size dropped, (fill_holes != 0) the others hole-filled
 *       inside their bounding box, renumbered 1..K in ascending order.
 * flow_threshold == 0 and fill_holes == 0 is amt_cellpose_masks.  PARITY UNPINNED like the above. */
int amt_cellpose_masks_ex(amt_ctx* ctx, const float* dP, const float* cellprob, int32_t* labels_out, int32_t* count_dev,
                          int nplanes, int H, int W, float cellprob_threshold, int niter, int min_size,
                          float max_size_fraction, int max_seeds, float flow_threshold, int fill_holes);
/* cellpose.utils.fill_holes_and_remove_small_masks on its own (the last step of amt_cellpose_masks_ex): labels_io =
 * nplanes x H x W int32 carrying labels 1..nlabels_dev[plane] (<= max_label; gaps allowed, larger values are treated as
 * background), in place: labels in ascending order, those with fewer than min_size pixels dropped, (fill_holes != 0)
 * the others hole-filled inside their bounding box -- written over whatever lies in the hole, in sequence, as the
 * package's loop does --, renumbered 1..K; count_dev[plane] = K. */
int amt_fill_holes_remove_small(amt_ctx* ctx, int32_t* labels_io, const int32_t* nlabels_dev, int32_t* count_dev,
                                int nplanes, int H, int W, int max_label, int min_size, int fill_holes);
/* metrics.flow_error of cellpose on its own: labels = nplanes x H x W int32 carrying 1..nlabels_dev[plane] (<= max_label),
 * dP as above; err_out = nplanes x max_label float64 (absent labels 0). */
int amt_cellpose_flow_error(amt_ctx* ctx, const int32_t* labels, const float* dP, const int32_t* nlabels_dev,
                            double* err_out, int nplanes, int H, int W, int max_label);

/* ---- channel overlay: R/blending.py:116-226 (create_overlay / overlay_channels) ------------------------
 * out_rgb = H x W x 3 float64 (interleaved).  background and every layer are H x W float64 planes on the device
 * (values outside [0, 1] are clipped, as the reference does).  layers_host = nlayers device pointers;
 * luts_host = nlayers x 256 x 4 float64 RGBA tables (matplotlib's LinearSegmentedColormap table of each layer);
 * opacity_host in [0, 1]; mode_host 0 = ALPHA ("over"), 1 = ADDITIVE.  At most 8 layers per call. */
int amt_overlay(amt_ctx* ctx, const double* background, const double* const* layers_host, int nlayers,
                const double* luts_host, const double* opacity_host, const int32_t* mode_host, double* out_rgb, int H,
                int W);

/* ---- glue of the config-5 flow network (BASELINE configs[4]; R/model.py:211: the network's convolutions run through
 * PyTorch-ROCm) ---- every convolution of Cellpose's residual U-Net is "batch norm -> [ReLU] -> conv" on a sum of up to
 * three terms; one pass prepares its input instead of one framework kernel per operation:
 *   s = (upsample ? x[n,h/2,w/2,c] : x[n,h,w,c]) + y[n,h,w,c] + pre_bias[c];  out = act((s + style[n,c]) * scale[c] + shift[c])
 * x, y (nullable), out, sum_out (nullable: receives s) are bf16 NHWC (channels-last) tensors, style (nullable) float32
 * [N][C], pre_bias (nullable: the biases of the convolutions that produced x and y, run without them), scale / shift
 * float32 [C] (the folded inference-time batch norm); C % 8 == 0. */
int amt_nn_affine_act_bf16(amt_ctx* ctx, const void* x, const void* y, const float* style, const float* pre_bias,
                           const float* scale, const float* shift, void* out, void* sum_out, int N, int H, int W, int C,
                           int relu, int upsample);

#ifdef __cplusplus
}
#endif
#endif /* AMT_HIP_H */
