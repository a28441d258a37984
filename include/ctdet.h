/* ctdet.h -- C ABI of libctdet.so: the MI355X (gfx950) detection hot path of
 * Ze-Yang/Context-Transformer (RFBNet-VGG forward, Context-Transformer attention,
 * prior-box decode / IoU matching, NMS), hand-written HIP behind plain pointers.
 *
 * Conventions
 *   - every entry point returns an int status (CT_OK == 0); ct_last_error_string()
 *     gives the text of the last failure on the calling thread;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); device
 *     entry points are asynchronous on it and never allocate: scratch memory comes
 *     from the caller through the matching ct_*_workspace_bytes() query;
 *   - all tensors are contiguous fp32 unless stated; "dev" = device pointer,
 *     "host" = host pointer; int = int32;
 *   - no torch / C++ types cross this boundary.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * reference repository).  The reference's only native symbol is `_nms`
 * (utils/nms/gpu_nms.hpp:1-2); everything else is stock PyTorch/ATen called from
 * Python, so for those the cited "interface" is the Python call site.
 */
#ifndef CTDET_H
#define CTDET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTDET_ABI_VERSION 1

enum {
    CT_OK = 0,
    CT_ERR_INVALID = 1,     /* bad argument / unsupported geometry            */
    CT_ERR_HIP = 2,         /* a HIP runtime call or kernel launch failed     */
    CT_ERR_WORKSPACE = 3,   /* caller-provided workspace too small            */
    CT_ERR_UNSUPPORTED = 4  /* valid request this build does not implement    */
};

typedef void* ct_stream_t;

int ct_abi_version(void);
const char* ct_last_error_string(void);
/* arch: e.g. "gfx950"; any out pointer may be NULL. */
int ct_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len);

/* Training scratch zeroing.  The accumulating training kernels (BatchNorm statistics / backward reductions, bias
 * gradients, split weight gradients, the Winograd weight-gradient workspace) zero their accumulation buffer with one
 * small memset per launch (~150 per training step).  ct_scratch_prezeroed(1) (per calling thread) makes the caller
 * responsible instead: ctdet/train_engine.py keeps those buffers in three arenas and zeroes each with ONE memset per
 * pass.  ct_scratch_prezeroed(0) restores the default. */
int ct_scratch_prezeroed(int on);

/* Batched weight packing for training steps (every optimizer step changes every weight, and each fused conv needs
 * its forward AND its data-gradient layout re-packed: ~130 small launches per step).  Between
 * ct_pack_record_begin() and ct_pack_record_end() the ct_conv_pack_weights* entry points called on THIS thread
 * record their arguments instead of launching; _end copies the table to caller memory (ct_pack_record_bytes());
 * ct_pack_run() then re-packs everything recorded in two launches.  The table holds the weight and output
 * pointers: record again when a parameter is re-allocated. */
int ct_pack_record_begin(void);
size_t ct_pack_record_bytes(void);
int ct_pack_record_end(void* table_dev, size_t table_bytes, int* num_direct, int* num_wino, ct_stream_t stream);
int ct_pack_run(const void* table_dev, int num_direct, int num_wino, ct_stream_t stream);

/* Measurement aid (bench.py `roofline.stages`): while enabled, the post-processing, score-fusion and attention
 * entry points bracket each of their kernel launches with HIP events on the launch stream.
 * ct_profile_enable(on) clears the records; ct_profile_collect waits for the recorded events and returns
 * (kernel name, milliseconds) per launch in launch order; *num_records = records available (may exceed max). */
typedef struct ct_profile_record {
    const char* name;   /* static string owned by the library */
    float ms;
} ct_profile_record;
int ct_profile_enable(int on);
int ct_profile_collect(ct_profile_record* out, int max_records, int* num_records);

/* ------------------------------------------------------------------ NMS ---- */

/* Drop-in for the reference's only native symbol:
 *   void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
 *             int boxes_dim, float nms_overlap_thresh, int device_id);
 *   (utils/nms/gpu_nms.hpp:1-2, utils/nms/nms_kernel.cu:91-144; caller utils/nms/gpu_nms.pyx:29)
 * Same contract: host buffers owned by the caller, boxes already sorted by descending
 * score, row = [x1,y1,x2,y2,score,...] (boxes_dim >= 4 floats per row), +1 pixel
 * convention, suppress IoU > thresh, keep_out = ascending indices into the sorted array.
 * Differences: returns a status instead of printing CUDA errors; does not change the
 * process-global current device (it is restored); the temporary device buffer is kept per thread between calls
 * (grown on demand) instead of being allocated and freed every time. */
int ct_nms_sorted_host(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
                       int boxes_dim, float nms_overlap_thresh, int device_id);
/* NMS mode flags (the `mode` / `ge` argument of the NMS entry points). */
#define CT_NMS_GT 0     /* suppress IoU >  thresh, +1 pixel convention (utils/nms/nms_kernel.cu:24-32,71) */
#define CT_NMS_GE 1     /* suppress IoU >= thresh                      (utils/nms/cpu_nms.pyx:65)          */
#define CT_NMS_PLAIN 2  /* no +1, union = (area_j - inter) + area_i    (utils/box_utils.py:238-302 `nms`)  */
/* Same with the rule selectable: mode = CT_NMS_GT | CT_NMS_GE, optionally | CT_NMS_PLAIN. */
int ct_nms_sorted_host_mode(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
                            int boxes_dim, float nms_overlap_thresh, int mode, int device_id);

/* Batched device NMS over independent segments (one per (image, class) of test.py:142-153).
 *   dets      dev [total,5]  rows [x1,y1,x2,y2,score], each segment sorted by descending score
 *   seg_off   dev [S+1]      segment s = rows seg_off[s] .. seg_off[s+1]-1
 *   keep      dev [total]    out: for segment s, keep[seg_off[s] + i], i < keep_count[s], are
 *                            the kept row positions RELATIVE to the segment, ascending
 *   keep_count dev [S]
 * max_seg_len bounds every segment length (sizes the launch). */
size_t ct_nms_batched_workspace_bytes(int total_boxes, int num_segments);
int ct_nms_batched_dev(const float* dets, const int* seg_off, int num_segments, int max_seg_len,
                       float thresh, int ge, int* keep, int* keep_count,
                       void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* CPU NMS of the reference's `--cpu` path: utils/nms/cpu_nms.pyx:17-68 (`cpu_nms`).
 * dets host [n,5] UNSORTED; keep_out host [n] receives original indices in descending
 * score order (ties: lower index first); suppress IoU >= thresh (ge=1) or > (ge=0). */
int ct_cpu_nms(const float* dets_host, int n, float thresh, int ge, int* keep_out, int* num_out);
/* utils/nms/cpu_nms.pyx:70-163 (`cpu_soft_nms`): mutates boxes [n,5] in place, *n_out = N'. */
int ct_cpu_soft_nms(float* boxes_host, int n, float sigma, float Nt, float threshold,
                    unsigned method, int* n_out);

/* ----------------------------------------------------- boxes / detection ---- */

/* utils/box_utils.py:184-202 `decode(loc, priors, variances)`, batched over images:
 * loc dev [B,P,4], priors dev [P,4] (cx,cy,w,h) -> boxes dev [B,P,4] (x1,y1,x2,y2),
 * each coordinate multiplied by scale4[c] if scale4 != NULL (dev [4] or per image [B,4],
 * `boxes *= scale` of test.py:136). */
int ct_decode(const float* loc, const float* priors, int batch, int num_priors,
              float var0, float var1, const float* scale4, int scale_per_image,
              float* boxes, ct_stream_t stream);
/* utils/box_utils.py:135-156 `encode(matched, priors, variances)`: [P,4],[P,4] -> [P,4]. */
int ct_encode(const float* matched, const float* priors, int num_priors, float var0, float var1,
              float* out, ct_stream_t stream);
/* layers/functions/detection.py:18-55 `Detect.forward`: decode + score fusion
 * scores[b,p,0] = obj[b,p,0]; scores[b,p,1+k] = obj[b,p,1]*conf[b,p,k].
 * If apply_softmax != 0, conf/obj are raw logits and the eval-time softmaxes of
 * models/RFB_Net_vgg.py:279-285 are fused in.  scale4 (dev [4] or [B,4], may be NULL) fuses
 * the `boxes *= scale` of test.py:136. */
int ct_detect_fused(const float* loc, const float* conf, const float* obj, const float* priors,
                    int batch, int num_priors, int num_fg, float var0, float var1,
                    int apply_softmax, const float* scale4, int scale_per_image,
                    float* boxes, float* scores, ct_stream_t stream);
/* torch.nn.functional.softmax(x, dim=-1) of models/RFB_Net_vgg.py:282-284: [rows, cols]. */
int ct_softmax_lastdim(const float* in, float* out, long rows, int cols, ct_stream_t stream);

/* utils/box_utils.py:50-68 `jaccard(box_a, box_b)`: a dev [A,4], b dev [Bn,4] point form
 * -> out dev [A,Bn]; no +1 convention. b_center_form != 0: b is (cx,cy,w,h) and
 * point_form (utils/box_utils.py:5-14) is applied on the fly. */
int ct_jaccard(const float* a, int na, const float* b, int nb, int b_center_form,
               float* out, ct_stream_t stream);

/* utils/box_utils.py:83-132 `match(...)` for a whole batch (the Python loop of
 * layers/modules/multibox_loss_combined.py:70-74):
 *   truths dev [sum G,6] rows [x1,y1,x2,y2,label,weight]; gt_off dev [B+1]
 *   priors dev [P,4] center form
 *   -> loc_t [B,P,4], conf_t [B,P,2] (label, weight), obj_t uint8 [B,P], overlap [B,P] or NULL
 * Force-match collisions keep the reference's "later GT wins" order (:122-123).
 * max_gt = the largest per-image box count in gt_off (any size); the workspace has max_gt slots per image, and an
 * image with MORE boxes than max_gt is matched against its first max_gt boxes only (never an out-of-bounds write). */
size_t ct_match_workspace_bytes(int batch, int num_priors, int max_gt);
int ct_match_batched(const float* truths, const int* gt_off, int batch, int max_gt,
                     const float* priors, int num_priors, float threshold, float var0, float var1,
                     float* loc_t, float* conf_t, uint8_t* obj_t, float* overlap,
                     void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* ---------------------------------------- batched test.py post-processing ---- */

/* test.py:136-161 for a batch: per (image, class>=1) select score > conf_thresh, order by
 * descending score (ties: lower prior index first), NMS, then the per-image top-k rule
 * (keep score >= k-th largest when more than max_per_image survive).
 *   boxes  dev [B,P,4] already scaled to pixels; scores dev [B,P,1+T]
 *   out_dets  dev [B, T, cap, 5]   kept rows [x1,y1,x2,y2,score] per (image,class), in
 *                                   descending score order, cap = out_cap rows reserved
 *   out_count dev [B, T]            rows valid in out_dets (after the top-k rule)
 *   out_index dev [B, T, cap]       prior index of each kept row (or NULL)
 * cap must be >= the largest number of boxes NMS keeps in any segment; on overflow the
 * call reports CT_ERR_WORKSPACE through *overflow (dev int, 0/1) for the host to check. */
size_t ct_postprocess_workspace_bytes(int batch, int num_priors, int num_fg);
int ct_postprocess_batched(const float* boxes, const float* scores, int batch, int num_priors,
                           int num_fg, float conf_thresh, float nms_thresh, int ge,
                           int max_per_image, int out_cap, float* out_dets, int* out_count,
                           int* out_index, int* overflow,
                           void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* ------------------------------------------------------------ convolution ---- */

/* One fused convolution of the RFBNet-VGG stack.  Replaces the ATen sequence
 * Conv2d [+ bias] [-> BatchNorm2d(eval)] [-> *res_scale + residual] [-> ReLU] [-> cat / permute]
 * of models/RFB_Net_vgg.py:7-22 (BasicConv), :53-64 / :100-112 (RFB tail), :219-227 (VGG),
 * :238-248 (heads: permute(0,2,3,1) + flatten + cat), as one implicit-GEMM kernel on the
 * fp32 MFMA path (v_mfma_f32_32x32x2_f32; exact fp32 products, fp32 accumulate).
 *
 * y[n,co,oh,ow] = act( (sum_k w[co,k] x[n,k@(oh,ow)]) * scale[co] + shift[co] ) with the
 * optional residual  y = act( (.)*res_scale + res[n,co,oh,ow] ).
 */
typedef struct ct_out_segment {
    float* ptr;          /* dev */
    int co_begin;        /* output channels [co_begin, co_end) go to this segment */
    int co_end;
    int pix_stride;      /* floats between consecutive pixels (= #channels of the segment) */
    long long img_stride;/* floats between consecutive images */
    long long base;      /* float offset of pixel 0 / channel co_begin inside an image */
} ct_out_segment;

typedef struct ct_conv_desc {
    /* input: NCHW buffer with in_ctot channels; the conv reads channels [in_coff, in_coff+cin) */
    const float* in;
    int batch, cin, h, w, in_ctot, in_coff;
    /* packed weights from ct_conv_pack_weights(): [k_pad][m_pad]; scale/shift: [m_pad] */
    const float* wpacked;
    const float* scale;
    const float* shift;
    int cout, m_pad, k_pad;
    int kh, kw, stride, pad_h, pad_w, dil;
    int oh, ow;
    /* output mode 0: NCHW buffer with out_ctot channels, written at [out_coff, out_coff+cout) */
    float* out;
    int out_ctot, out_coff;
    /* optional residual (NCHW, res_ctot channels, offset res_coff); NULL = none */
    const float* res;
    int res_ctot, res_coff;
    float res_scale;
    int relu;            /* 1: ReLU on every output channel */
    const float* lo;     /* optional dev [m_pad]: y = max(y, lo[co]) per channel (0 = ReLU, -inf = none);
                            overrides `relu`; lets convs with and without ReLU share one launch */
    /* output mode 1 (nseg > 0): channels-last scatter into up to 3 flattened head buffers */
    int nseg;
    ct_out_segment seg[3];
    /* tile configuration: 0 = heuristic; otherwise 1 + index into ct_conv_num_configs() */
    int config;
    /* 1: data gradient of the convolution (conv_transpose): `in` = dY [batch,cin=cout_fwd,h,w],
     * output = dX [batch,cout=cin_fwd,oh,ow] with (oh,ow) the forward INPUT size, weights packed by
     * ct_conv_pack_weights_dgrad; stride/pad/dil are the forward convolution's.  What autograd's
     * conv backward-data does for every Conv2d of models/RFB_Net_vgg.py in train.py:228. */
    int transposed;
    /* split-K for maps too small to fill the chip (ct_conv2d_fwd only): ksplit > 1 divides the reduction
     * (cin * kh * kw) over that many workgroups per output tile; each writes its partial sums to its own slab
     * of ksplit_ws ([ksplit][cout][batch*oh*ow] floats, no atomics) and a finishing kernel adds the slabs in
     * order and applies the epilogue, so results do not depend on scheduling.  ksplit_ws_floats = capacity of
     * the workspace (the factor is clamped to what fits).  0 / 1 or a null workspace = off; -1 = let the
     * library choose from the number of output tiles. */
    int ksplit;
    float* ksplit_ws;
    long long ksplit_ws_floats;
    /* Maxima of |activation| for the f16x2 operand form (csrc/ct_f16x2.h: binary16 pieces need a power-of-two scale, taken from
     * the tensor's maximum).  Both optional (NULL = off), both arrays of `batch` LINES of CT_ABSMAX_LINE_BYTES bytes of device
     * memory that the caller zeroes once per step: the bit pattern of max |x| of image n lives in the first word of line n.  PER
     * IMAGE, so that what a kernel computes for an image never depends on the other images of its batch.
     *   in_absmax   per image an upper bound of |x| over the input slice, left there by whoever produced the input (any kernel of
     *               this library run with out_absmax, or ct_absmax_f32); kernels that need the maxima and do not get them take
     *               them themselves in an extra pass (ct_conv2d_wino4s_fwd variant 3) or refuse (ct_conv2d_wino4f_pool_fwd_v
     *               variant 2, the "h2:" configurations of ct_conv2d_x3_fwd);
     *   out_absmax  the launch folds max |y| of everything it stores for image n into line n (atomic max): honoured by the shared
     *               epilogue of ct_conv2d_wino4s_fwd (every variant) and of the f16x2 variant of ct_conv2d_wino4f_pool_fwd_v, by
     *               ct_conv2d_x3_fwd and ct_conv2d_fwd (every configuration; split-K launches: in the finishing kernel); ignored
     *               by the other kernels (the fp32 / bf16x3 fused Winograd kernels, the bf16 path) and by data-gradient launches. */
    const unsigned* in_absmax;
    unsigned* out_absmax;
} ct_conv_desc;
#define CT_ABSMAX_LINE_BYTES 128
/* max |x| over the channel slice [per_image floats] of every image (images img_stride floats apart), folded into `lines`
 * (batch lines, atomic max; the caller zeroes them once per step). */
int ct_absmax_f32(const float* in, int batch, long long per_image, long long img_stride, unsigned* lines, ct_stream_t stream);

/* Rows of the packed weight matrix for a (cin, kh, kw) filter: k_pad. */
int ct_conv_kpad(int cin, int kh, int kw);
int ct_conv_mpad(int cout);
int ct_conv_num_configs(void);
/* Human-readable name of tile config i (0-based), e.g. "128x128".  The last one, "valu", is not an implicit-GEMM tile:
 * conv_valu3x3_f32, the 3x3 / 3-input-channel image layer (models/RFB_Net_vgg.py:219, conv1_1) on the vector ALU; it
 * is what config 0 picks for such a layer (cout % 8 == 0, NCHW output, no residual) and CT_ERR_UNSUPPORTED elsewhere. */
const char* ct_conv_config_name(int i);
/* Pack nparts (1..6) weight tensors w[i] dev [cout_i, cin, kh, kw] (concatenated along cout) into
 * wpacked dev [k_pad][m_pad] (k = ci*kh*kw + tap, zero padded). */
int ct_conv_pack_weights(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw,
                         float* wpacked, int m_pad, int k_pad, ct_stream_t stream);
/* Same for the data-gradient launch: wpacked [k_pad][m_pad] with k = co*kh*kw + tap over the
 * concatenated couts and m = ci;  k_pad = ct_conv_kpad(sum cout, kh, kw), m_pad = ct_conv_mpad(cin). */
int ct_conv_pack_weights_dgrad(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw,
                               float* wpacked, int m_pad, int k_pad, ct_stream_t stream);
/* Epilogue vectors. BatchNorm2d(eval) of models/RFB_Net_vgg.py:13: scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale.  With gamma == NULL: scale = 1, shift = bias (or 0 if bias NULL).
 * Written at [offset, offset+n) of the m_pad-long vectors. */
int ct_conv_fold_epilogue(const float* gamma, const float* beta, const float* mean, const float* var,
                          float eps, const float* bias, int n, int offset,
                          float* scale, float* shift, ct_stream_t stream);
int ct_conv2d_fwd(const ct_conv_desc* desc, ct_stream_t stream);

/* Winograd F(2x2,3x3) variant of ct_conv2d_fwd for 3x3 / stride 1 / dilation 1 / pad 1 convolutions with
 * cin % 8 == 0 and an NCHW output (the VGG trunk and the 3x3 BasicConv layers, models/RFB_Net_vgg.py:7-22,
 * 219-227): same descriptor, same fused epilogue, 2.25x fewer multiplications.  `desc->wpacked/k_pad/m_pad`
 * are ignored; the weights come pre-transformed (U = G g G^T) from ct_conv_pack_weights_wino. */
int ct_conv_wino_supported(const ct_conv_desc* desc);
size_t ct_conv_wino_packed_floats(int cin, int cout);
int ct_conv_pack_weights_wino(const float* const* w, const int* cout, int nparts, int cin, float* upacked,
                              ct_stream_t stream);
int ct_conv2d_wino_fwd(const ct_conv_desc* desc, const float* upacked, ct_stream_t stream);
/* Same with the following MaxPool2d(2, 2[, ceil_mode]) (models/RFB_Net_vgg.py:328-330) fused: the 2x2 Winograd
 * output tile is the pooling window, so the pooled activation [B, pool_ctot, pool_oh, pool_ow] (channels at
 * pool_coff) is written from registers; write_full = 0 skips the full-resolution output when nothing else reads it.
 * pool_oh/ow = floor or ceil of oh/2, ow/2 (ceil_mode windows are clipped to the map). */
int ct_conv2d_wino_pool_fwd(const ct_conv_desc* desc, const float* upacked, float* pool_out, int pool_ctot,
                            int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream);
/* Weights of the DATA-GRADIENT convolution of a 3x3 / stride 1 / pad 1 layer (input channels = sum cout,
 * output channels = cin, taps rotated by 180 degrees), for ct_conv2d_wino_fwd on dY:
 * upacked holds ct_conv_wino_packed_floats(sum cout, cin) floats. */
int ct_conv_pack_weights_wino_dgrad(const float* const* w, const int* cout, int nparts, int cin,
                                    float* upacked, ct_stream_t stream);

/* Winograd F(4x4,3x3): the large-tile variant of the five entry points above, same contracts (same layers:
 * models/RFB_Net_vgg.py:7-22,219-227; same descriptor, epilogue, pooling fusion -- a 4x4 tile holds four pooling
 * windows -- and head scatter), its own packed layout (ct_conv_wino4_packed_floats floats).  36 multiplications per
 * 16 outputs: 4x fewer than the direct convolution, 1.78x fewer than F(2x2,3x3); fp32 rounding error about 1e-5 of
 * the output range (interpolation points 0, +-1, +-2, inf) against 1e-6 for F(2x2,3x3).
 * desc->ksplit / ksplit_ws are honoured by ct_conv2d_wino4_fwd as a split over INPUT CHANNELS (ksplit -1: the library
 * decides, > 1: that many slices, else off): each slice stores its output-transformed partial sums in its slab
 * ws[slice][cout][batch*oh*ow] and a finishing kernel adds them in slice order -- for maps whose 32-tile x 64-cout
 * workgroup grid cannot fill the chip (small batches).  Not combined with the fused pooling.
 * ksplit -2 (with ksplit_ws >= 256*2*64*32*16 floats): stream-K -- the caller promises the launch runs alone on the
 * device; a persistent grid of 256 workgroups does the full rounds of work items whole and cuts the last, partial round
 * by input-channel chunks (partial sums in slabs, a fix-up kernel adds them in chunk order and applies the epilogue,
 * fused pooling included).  Deterministic, but the cut items are the last tiles of the batch: an image's rounding then
 * depends on its batch position, which is why ctdet/engine.py only uses it on request (CTDET_W4_STREAMK=1). */
int ct_conv_wino4_supported(const ct_conv_desc* desc);
size_t ct_conv_wino4_packed_floats(int cin, int cout);
int ct_conv_pack_weights_wino4(const float* const* w, const int* cout, int nparts, int cin, float* upacked,
                               ct_stream_t stream);
int ct_conv_pack_weights_wino4_dgrad(const float* const* w, const int* cout, int nparts, int cin,
                                     float* upacked, ct_stream_t stream);
int ct_conv2d_wino4_fwd(const ct_conv_desc* desc, const float* upacked, ct_stream_t stream);
int ct_conv2d_wino4_pool_fwd(const ct_conv_desc* desc, const float* upacked, float* pool_out, int pool_ctot,
                             int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream);

/* Winograd F(2x2,3x3) on the bf16 matrix pipe (csrc/ct_wino_x3.hip): the entry points of ct_conv2d_wino_fwd above with
 * the transform-domain products evaluated as six bf16 piece products ("bf16x3", see below) on v_mfma_f32_32x32x16_bf16
 * -- same layers (models/RFB_Net_vgg.py:7-22,219-227,238-248), descriptor, epilogue, pooling fusion and head scatter;
 * cin % 16 == 0.  The weights come pre-transformed AND pre-split from ct_conv_pack_weights_wino_x3
 * (ct_conv_wino_x3_packed_bytes bytes).  `variant` must be 1: eight-wave workgroups, the hi.hi products accumulate in their own
 * register block (the large sum sees cin / 16 roundings; error vs fp64 about a third of ct_conv2d_wino_fwd's, a tenth of
 * ct_conv2d_wino4_fwd's).  (Variant 2 of rounds 4-5 -- four-wave workgroups, one accumulator -- was removed in round 6: it
 * never won a layer inside the two-stream pipeline; CT_ERR_INVALID.) */
int ct_conv_wino_x3_supported(const ct_conv_desc* desc);
size_t ct_conv_wino_x3_packed_bytes(int cin, int cout);
int ct_conv_pack_weights_wino_x3(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                 ct_stream_t stream);
int ct_conv_pack_weights_wino_x3_dgrad(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                       ct_stream_t stream);
int ct_conv2d_wino_x3_fwd(const ct_conv_desc* desc, const void* upacked, int variant, ct_stream_t stream);
int ct_conv2d_wino_x3_pool_fwd(const ct_conv_desc* desc, const void* upacked, int variant, float* pool_out, int pool_ctot,
                               int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream);

/* Winograd F(4x4,3x3) in three kernels with the transform-domain GEMMs on the bf16 matrix pipe (csrc/ct_wino4s.hip): the
 * wide 3x3 layers of the VGG trunk and the multibox heads (models/RFB_Net_vgg.py:219-227 conv3_x .. conv5_x, :238-248)
 * -- same descriptor, transforms (interpolation points 0, +-3/4, +-3/2, inf), epilogue, pooling fusion and head scatter as
 * ct_conv2d_wino4_fwd, fp32 results equal to it up to summation order; cin % 16 == 0.  An input-transform kernel writes
 * V = B^T d B once per (tile, channel), split exactly into three bfloat16 pieces, as MFMA operand fragments; 36 GEMMs
 * M[xi] = U[xi] V[xi] (128 x 128 blocks, operands by LDS-DMA, six piece products per multiply-add, no vector-ALU work in
 * the loop); an output-transform kernel applies A^T M A and the epilogue.  V and M live in the caller's `workspace`
 * (ct_conv_wino4s_workspace_bytes(desc) bytes: 13.5 bytes per (output pixel, input channel) + 9 per (output pixel,
 * output channel); launches on different streams need different workspaces).  Weights: ct_conv_pack_weights_wino4s
 * (ct_conv_wino4s_packed_bytes bytes).  variant 1: bf16x3, the hi.hi products in their own accumulator (variant 2 of
 * rounds 4-5, one accumulator, was removed in round 6: CT_ERR_BAD_ARG).
 * variant 3 (round 6): the "f16x2" operand form -- every transform-domain value as TWO binary16 pieces (hi = rne16(x 2^e),
 * lo = rne16(x 2^e - hi)) and a multiply-add as the THREE piece products hi.hi, hi.lo, lo.hi on the f16 matrix pipe, the hi.hi
 * products in their own accumulator: the same 22-24 bits as three bfloat16 pieces / six products (same error against fp64), half
 * the matrix instructions, 9 instead of 13.5 bytes of V per (output pixel, input channel).  The power-of-two scales 2^e come from
 * the operands' own maxima (an absmax pass over the input slice in front of the input transform; max |g| of the layer at
 * packing time), so no binary16 piece overflows for any data; the output transform undoes them exactly.  Weights for variant 3:
 * ct_conv_pack_weights_wino4s_h2 (ct_conv_wino4s_h2_packed_bytes bytes; not recordable by ct_pack_record_begin). */
int ct_conv_wino4s_supported(const ct_conv_desc* desc);
size_t ct_conv_wino4s_packed_bytes(int cin, int cout);
size_t ct_conv_wino4s_h2_packed_bytes(int cin, int cout);
/* Batched f16x2 packing for callers that re-pack many layers per step (the training engine, once per optimizer step): fill one
 * item per layer on the host (ct_conv_wino_h2_pack_item_bytes() bytes each; tile 47 = the layout of ct_conv_pack_weights_wino4s_h2,
 * 48 = of ct_conv_pack_weights_wino4f_h2; dgrad != 0 = the data-gradient layouts), copy the items to the device as one array and
 * replay it with ct_conv_wino_h2_pack_run: three launches for the whole list (trailers, maxima, packing) instead of three per
 * layer.  Same bytes as the per-layer calls. */
size_t ct_conv_wino_h2_pack_item_bytes(void);
int ct_conv_wino_h2_pack_item(const float* const* w, const int* cout, int nparts, int cin, int dgrad, int tile, void* upacked,
                              void* item_out);
int ct_conv_wino_h2_pack_run(const void* items_dev, int n, ct_stream_t stream);
size_t ct_conv_wino4s_workspace_bytes(const ct_conv_desc* desc);
int ct_conv_pack_weights_wino4s(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                ct_stream_t stream);
int ct_conv_pack_weights_wino4s_dgrad(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                      ct_stream_t stream);
int ct_conv_pack_weights_wino4s_h2(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                   ct_stream_t stream);
int ct_conv_pack_weights_wino4s_h2_dgrad(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                         ct_stream_t stream);
int ct_conv2d_wino4s_fwd(const ct_conv_desc* desc, const void* upacked, void* workspace, size_t workspace_bytes,
                         int variant, ct_stream_t stream);
int ct_conv2d_wino4s_pool_fwd(const ct_conv_desc* desc, const void* upacked, void* workspace, size_t workspace_bytes,
                              int variant, float* pool_out, int pool_ctot, int pool_coff, int pool_oh, int pool_ow,
                              int write_full, ct_stream_t stream);

/* Winograd F(4x4,3x3) FUSED with the transform-domain products on the bf16 matrix pipe (csrc/ct_wino4f.hip): the narrow
 * 3x3 layers on the large maps (models/RFB_Net_vgg.py:323-336 conv1_2 .. conv3_1; conv3_2 / conv3_3 at 512 x 512), where the
 * three-kernel form's workspace traffic does not pay and the fp32-MFMA kernel is bound by the slowest matrix rate -- same
 * descriptor, transforms, epilogue, pooling fusion and head scatter as ct_conv2d_wino4_fwd, fp32 results equal to it up to
 * summation order; cin % 16 == 0.  One workgroup = 32 tiles x one block of 64 couts: patches arrive in LDS by DMA, the
 * transform runs once per (tile, channel, 64 couts), V stays in LDS as fp32 and is split into the three bfloat16 pieces by
 * the wave that multiplies it; no workspace.  Weights: ct_conv_pack_weights_wino4f (ct_conv_wino4f_packed_bytes bytes). */
int ct_conv_wino4f_supported(const ct_conv_desc* desc);
size_t ct_conv_wino4f_packed_bytes(int cin, int cout);
int ct_conv_pack_weights_wino4f(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                ct_stream_t stream);
int ct_conv_pack_weights_wino4f_dgrad(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                      ct_stream_t stream);
int ct_conv2d_wino4f_fwd(const ct_conv_desc* desc, const void* upacked, ct_stream_t stream);
int ct_conv2d_wino4f_pool_fwd(const ct_conv_desc* desc, const void* upacked, float* pool_out, int pool_ctot, int pool_coff,
                              int pool_oh, int pool_ow, int write_full, ct_stream_t stream);
/* The same kernel on the f16x2 operand form (variant 2; variant 1 = bf16x3 = ct_conv2d_wino4f_pool_fwd): two binary16 pieces per
 * transform-domain value, three piece products (csrc/ct_f16x2.h, see ct_conv2d_wino4s_fwd variant 3) -- 27 instead of 54 MFMAs and
 * 120 instead of 220 split instructions per wave and 16-channel chunk.  Needs desc->in_absmax (the fused kernel cannot take the
 * input's maximum itself) and weights from ct_conv_pack_weights_wino4f_h2 (ct_conv_wino4f_h2_packed_bytes bytes); pool_out == NULL
 * and write_full == 1 for a plain launch.  Honours desc->out_absmax. */
size_t ct_conv_wino4f_h2_packed_bytes(int cin, int cout);
int ct_conv_pack_weights_wino4f_h2(const float* const* w, const int* cout, int nparts, int cin, void* upacked, ct_stream_t stream);
int ct_conv_pack_weights_wino4f_h2_dgrad(const float* const* w, const int* cout, int nparts, int cin, void* upacked,
                                         ct_stream_t stream);
int ct_conv2d_wino4f_pool_fwd_v(const ct_conv_desc* desc, const void* upacked, int variant, float* pool_out, int pool_ctot,
                                int pool_coff, int pool_oh, int pool_ow, int write_full, ct_stream_t stream);

/* ---- "bf16x3": the fp32 convolution of ct_conv2d_fwd on the bf16 matrix pipe (csrc/ct_conv_x3.hip) ----
 * Same layers (models/RFB_Net_vgg.py:7-22 BasicConv, the plain Conv2d layers, the multibox heads :238-248), the same
 * descriptor (NCHW fp32 in / out, channel slices, residual, per-channel floor, head scatter, ksplit slabs) and the
 * same results to fp32 accuracy: every fp32 operand is split exactly into three bfloat16 pieces and the six leading
 * piece products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (six bf16 MFMAs = 0.375 of one fp32 MFMA;
 * per-layer error against fp64 no worse than ct_conv2d_fwd's, tests/test_gpu_x3.py).  Any filter size / stride /
 * dilation; cin must be a multiple of the config's k-step (CT_ERR_UNSUPPORTED otherwise).  desc->transposed = 1 (the
 * data gradient, ct_conv2d_fwd's contract: `in` = dY, weights from ct_conv_pack_weights_x3_dgrad, stride 1 or 2) is
 * what `losses.backward()` (train.py:228) needs for the same layers.  desc->wpacked / m_pad / k_pad / config are ignored: the split weights come from
 * ct_conv_pack_weights_x3 for the k-step length (16 or 32 channels) of the chosen tile config. */
int ct_conv_x3_num_configs(void);
const char* ct_conv_x3_config_name(int i);               /* e.g. "x3:128x128k16" */
int ct_conv_x3_config_bk(int i);                         /* channels per k-step of config i */
size_t ct_conv_x3_packed_bytes(int cin, int cout, int kh, int kw, int bk);
int ct_conv_pack_weights_x3(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw, int bk,
                            void* wx3, ct_stream_t stream);
/* Data-gradient layout: rows = cin (forward input channels), k-channels = the concatenated couts; holds
 * ct_conv_x3_packed_bytes(sum cout, cin, kh, kw, bk) bytes. */
int ct_conv_pack_weights_x3_dgrad(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw, int bk,
                                  void* wx3, ct_stream_t stream);
int ct_conv2d_x3_fwd(const ct_conv_desc* desc, const void* wx3, int config, ct_stream_t stream);
/* The f16x2 operand form of the same kernel (csrc/ct_f16x2.h; see ct_conv2d_wino4s_fwd variant 3): configurations named
 * "h2:<tile>" (ct_conv_x3_config_h2(i) == 1) take weights from ct_conv_pack_weights_x3h (ct_conv_x3h_packed_bytes bytes: two
 * binary16 pieces of w 2^eW and a trailer with eW) and need desc->in_absmax; forward only.  Every configuration honours
 * desc->out_absmax (split-K launches: in their finishing kernel). */
int ct_conv_x3_config_h2(int i);
size_t ct_conv_x3h_packed_bytes(int cin, int cout, int kh, int kw, int bk);
int ct_conv_pack_weights_x3h(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw, int bk, void* wx3,
                             ct_stream_t stream);
/* All weight splits of a training step (train.py:222-229 updates every weight every step) in ONE launch: the arguments
 * never change between steps, so build the list once -- ct_conv_x3_pack_item fills one item of
 * ct_conv_x3_pack_item_bytes() bytes in HOST memory from the arguments of ct_conv_pack_weights_x3 (dgrad = 0) or
 * ct_conv_pack_weights_x3_dgrad (dgrad = 1) without launching --, copy the items to the device, and replay them with
 * ct_conv_x3_pack_run(items_dev, n). */
size_t ct_conv_x3_pack_item_bytes(void);
int ct_conv_x3_pack_item(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw, int bk, void* wx3,
                         int dgrad, void* host_item);
int ct_conv_x3_pack_run(const void* items_dev, int n, ct_stream_t stream);

/* ---- bf16 channels-last convolutions (BASELINE.json configs[4]: "bf16 MFMA convs + fp32 NMS") ----
 * Same layers and epilogue as ct_conv2d_fwd (models/RFB_Net_vgg.py:7-22,219-248), other storage: activations
 * are [batch][h][w][channels] bfloat16 (round-to-nearest-even of the fp32 value), accumulation and epilogue
 * in fp32.  The descriptor is ct_conv_desc with `in`, `out`, `res` pointing at bf16 NHWC buffers (channel
 * slices: in_ctot / in_coff / cin multiples of 8), `wpacked` from ct_conv_pack_weights_bf16, `scale` / `shift`
 * / `lo` fp32, segments (nseg > 0) written in fp32 exactly as by ct_conv2d_fwd; ksplit / ksplit_ws as for
 * ct_conv2d_fwd (slabs [ksplit][batch*oh*ow][cout] fp32); m_pad / k_pad / config / transposed unused. */
int ct_conv_bf16_cin_pad(int cin);
int ct_conv_bf16_cout_pad(int cout);
/* bf16 elements of the packed weights of a (cin, cout, kh, kw) filter bank */
size_t ct_conv_bf16_packed_elems(int cin, int cout, int kh, int kw);
int ct_conv_pack_weights_bf16(const float* const* w, const int* cout, int nparts, int cin, int kh, int kw,
                              void* wpacked, ct_stream_t stream);
int ct_conv2d_bf16_fwd(const ct_conv_desc* d, ct_stream_t stream);
/* x dev [batch][channels][hw] fp32 -> y dev [batch][hw][c_pad] bf16 (channels >= `channels` zero) and back
 * (channel slice [coff, coff+channels) of a ctot-channel NHWC buffer -> dense NCHW fp32) */
int ct_nchw_f32_to_nhwc_bf16(const float* x, int batch, int channels, int hw, int c_pad, void* y, ct_stream_t stream);
int ct_nhwc_bf16_to_nchw_f32(const void* y, int batch, int channels, int hw, int ctot, int coff, float* x,
                             ct_stream_t stream);
/* nn.MaxPool2d on an NHWC bf16 map (models/RFB_Net_vgg.py:331-341); oh/ow decide floor or ceil mode */
int ct_maxpool2d_nhwc_bf16(const void* x, void* y, int batch, int channels, int h, int w, int oh, int ow, int k,
                           int stride, int pad, ct_stream_t stream);

/* ------------------------------------------------------------ training side ---- */
/* What `losses.backward()` (train.py:228) makes autograd/cuDNN do for the layers above.  The data
 * gradient of a convolution is ct_conv2d_fwd with desc.transposed = 1. */

/* Weight gradient of the convolution described by `d` (forward geometry; d->in = the forward input
 * X, d->wpacked/scale/shift/out unused):  dw[cout][cin][kh][kw] (dense fp32, overwritten) =
 * sum over batch and output pixels of dz[n][co][oh][ow] * X[n][ci][ih][iw];  dz is the channel slice
 * [dz_coff, dz_coff+cout) of an NCHW buffer with dz_ctot channels. */
int ct_conv2d_wgrad(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff, float* dw,
                    ct_stream_t stream);

/* The same weight gradient through Winograd F(3x3, 2x2) for 3x3 / stride 1 / dilation 1 / pad 1 layers
 * (2.25x fewer multiplications; same arguments and result layout as ct_conv2d_wgrad).  `workspace` holds
 * ct_conv_wgrad_wino_workspace_bytes(d) bytes (the 16 transform-domain partial sums [16][cout][cin]);
 * ct_conv_wgrad_wino_supported() tells whether the geometry qualifies (else CT_ERR_UNSUPPORTED). */
int ct_conv_wgrad_wino_supported(const ct_conv_desc* d);
size_t ct_conv_wgrad_wino_workspace_bytes(const ct_conv_desc* d);
int ct_conv2d_wgrad_wino(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff, float* dw,
                         void* workspace, ct_stream_t stream);
/* Winograd F(3x3, 4x4) weight gradient: the large-tile counterpart (4x4 dZ tiles, 6x6 input patches, 36 transform
 * points: 1.78x fewer multiplications than F(3x3, 2x2)), same contract; the workspace holds
 * ct_conv_wgrad_wino4_workspace_bytes(d) bytes ([36][cout][cin] partial sums). */
int ct_conv_wgrad_wino4_supported(const ct_conv_desc* d);
size_t ct_conv_wgrad_wino4_workspace_bytes(const ct_conv_desc* d);
int ct_conv2d_wgrad_wino4(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff, float* dw,
                          void* workspace, ct_stream_t stream);

/* The F(3x3, 4x4) weight gradient in the three-kernel form of ct_conv2d_wino4s_fwd (csrc/ct_wino4s.hip): dZ tiles and input
 * patches are transformed once (E = A e A^T, V = B^T d B), split exactly into three bfloat16 pieces and stored as MFMA operand
 * fragments with the TILE index as k; 36 GEMMs dU[xi][cout][cin] = sum_tiles E V^T on the bf16 matrix pipe (the forward form's
 * kernel, k split over several workgroups per block, one slab of dU per split, no atomics); a finishing kernel adds the slabs
 * in order and applies G^T . G.  Same arguments and result as ct_conv2d_wgrad_wino4 (train.py:228: autograd's conv
 * backward-weight for the wide 3x3 layers of models/RFB_Net_vgg.py:219-227,238-248); cin % 16 == 0; the workspace holds
 * ct_conv_wgrad_wino4s_workspace_bytes(d) bytes (E, V and the dU slabs). */
int ct_conv_wgrad_wino4s_supported(const ct_conv_desc* d);
size_t ct_conv_wgrad_wino4s_workspace_bytes(const ct_conv_desc* d);
int ct_conv2d_wgrad_wino4s(const ct_conv_desc* d, const float* dz, int dz_ctot, int dz_coff, float* dw,
                           void* workspace, size_t workspace_bytes, ct_stream_t stream);

/* nn.BatchNorm2d(eps 1e-5, momentum 0.01) in training mode (models/RFB_Net_vgg.py:13,19), split in
 * three launches around the conv output z (channel slice [z_coff, z_coff+channels) of an NCHW buffer
 * with z_ctot channels; dz uses the same slicing):
 *   stats    per-channel batch mean / biased variance (+ running-stat update with the unbiased
 *            variance when running_mean/var are given)
 *   apply    y = act( ((z-mean)/sqrt(var+eps)*gamma + beta) [* res_scale + res] ) into a channel slice
 *   backward dz, dgamma, dbeta (and the residual branch's gradient) from dy */
/* `scratch` (both BatchNorm entry points): optional device buffer of 2*channels doubles; with it the per-channel
 * reductions are split over several workgroups per channel (f64 atomics), without it one workgroup per channel. */
int ct_bn_train_stats(const float* z, int batch, int ctot, int coff, int channels, int hw,
                      float* mean, float* var, float momentum, float* running_mean, float* running_var,
                      void* scratch, ct_stream_t stream);
int ct_bn_train_apply(const float* z, const float* mean, const float* var, const float* gamma,
                      const float* beta, float eps, int relu, const float* lo, const float* res,
                      int res_ctot, int res_coff, float res_scale, float* y, int y_ctot, int y_coff,
                      int z_ctot, int z_coff, int batch, int channels, int hw, ct_stream_t stream);
int ct_bn_train_backward(const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot, int y_coff,
                         const float* z, const float* mean, const float* var, const float* gamma,
                         float eps, int relu, const float* lo, float res_scale,
                         float* dres, int dres_ctot, int dres_coff, int dres_accumulate,
                         float* dz, float* dgamma, float* dbeta, int z_ctot, int z_coff,
                         int batch, int channels, int hw, void* scratch, ct_stream_t stream);
/* The same backward for an nn.BatchNorm2d that sits in eval() mode inside a training network (frozen
 * statistics: ct_bn_train_apply was given running_mean / running_var): mean and variance are constants, so
 * dz = gamma / sqrt(var+eps) * dy_masked, dgamma = sum dy_masked * xhat, dbeta = sum dy_masked. */
int ct_bn_eval_backward(const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot, int y_coff,
                        const float* z, const float* running_mean, const float* running_var, const float* gamma,
                        float eps, int relu, const float* lo, float res_scale,
                        float* dres, int dres_ctot, int dres_coff, int dres_accumulate,
                        float* dz, float* dgamma, float* dbeta, int z_ctot, int z_coff,
                        int batch, int channels, int hw, void* scratch, ct_stream_t stream);
/* y = act(conv + bias): dz = dy * (y > 0 if relu) into a channel slice, dbias[c] = sum dz (may be NULL). */
int ct_bias_act_backward(const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot, int y_coff,
                         int relu, int batch, int channels, int hw, float* dz, int dz_ctot, int dz_coff,
                         float* dbias, ct_stream_t stream);
/* The same, and max |dz| per image folded into dz_absmax (lines as ct_conv_desc.out_absmax; NULL = off): dz is the INPUT of the
 * layer's data-gradient convolution, whose f16x2 form takes its exponent from these maxima (train.py:222-229 backward). */
int ct_bias_act_backward_amax(const float* dy, int dy_ctot, int dy_coff, const float* y, int y_ctot, int y_coff,
                              int relu, int batch, int channels, int hw, float* dz, int dz_ctot, int dz_coff,
                              float* dbias, unsigned* dz_absmax, ct_stream_t stream);
/* MaxPool2d(2, 2) backward and the bias + ReLU backward of the convolution under the pool in one pass (train.py:222-229 backward of
 * models/RFB_Net_vgg.py:323-336 conv + ReLU + 'M'): y = the convolution's (post-ReLU) output = the pool's input, dy = the gradient
 * of the POOLED map [batch][channels][oh*ow]; writes dz = (y > 0) * (first maximum of its window) * dy into a channel slice,
 * dbias[c] += sum dz (NULL = skip), max |dz| per image into dz_absmax (NULL = off).  For a convolution whose output only the pool reads. */
int ct_maxpool2x2_bias_relu_bwd(const float* y, int y_ctot, int y_coff, const float* dy, int batch, int channels, int h, int w,
                                int oh, int ow, float* dz, int dz_ctot, int dz_coff, float* dbias, unsigned* dz_absmax,
                                ct_stream_t stream);
/* max_pool2d backward: the gradient goes to the first maximum of each window (torch's rule). */
int ct_maxpool2d_bwd(const float* x, const float* dy, float* dx, long planes, int h, int w, int oh, int ow,
                     int k, int stride, int pad, int accumulate, ct_stream_t stream);
/* Gradient of the channels-last head scatter (models/RFB_Net_vgg.py:239-248 permute/view/cat):
 * dz[n][co][pix] gathered from the flattened loc/conf/obj gradient buffers described by segs. */
int ct_head_grad_gather(const ct_out_segment* segs, int nseg, int batch, int channels, int hw,
                        float* dz, ct_stream_t stream);

/* Test-time input transform, data/data_augment.py:224-266 (BaseTransform.__call__): bilinear
 * cv2.resize of uint8 HxWx3 images to size x size, float32, minus `means3`, HWC -> CHW.
 * src holds `batch` images back to back; image n starts at src + offsets[n] and is
 * hw[2n] x hw[2n+1] x 3 bytes (offsets, hw: device arrays).  out = float [batch][3][size][size].
 * Integer arithmetic follows OpenCV's 8-bit INTER_LINEAR path (11-bit coefficients). */
int ct_preproc_resize(const unsigned char* src, const long long* offsets, const int* hw, int batch,
                      int size, const float* means3, float* out, ct_stream_t stream);

/* Training-time augmentation, data/data_augment.py:164-221 (`preproc.__call__`): crop -> photometric distortion
 * (8-bit HSV space) -> expand on a mean-filled canvas -> mirror -> resize -> minus means -> CHW, one gather launch
 * per batch.  `plans` = device array of `batch` 96-byte records (ctdet/ops.py AugPlan: source offset / size, crop
 * rectangle, canvas size and placement, mirror flag, interpolation 0 linear / 1 nearest / 2 area, distortion flags
 * 1 brightness 2 contrast 4 hue 8 saturation with their parameters, canvas fill) holding the random DECISIONS the
 * host logic drew like the reference does.  Pixel arithmetic restates OpenCV's published 8-bit formulas (cv2 is
 * not in this image): tolerance-based parity. */
int ct_preproc_augment(const unsigned char* src, const void* plans, int batch, int size, const float* means3,
                       float* out, ct_stream_t stream);
/* mixup of two image batches, data/voc0712.py:262: out = img1 * lambd[n] + img2 * (1 - lambd[n]). */
int ct_mixup_blend(const float* img1, const float* img2, const float* lambd, int batch, long per_image,
                   float* out, ct_stream_t stream);

/* torch.nn.MaxPool2d of models/RFB_Net_vgg.py:328-330,338 (2x2 s2 [ceil], 3x3 s1 p1) on an NCHW
 * buffer: planes = batch*channels; windows are clipped to the input (ceil_mode semantics are
 * encoded in oh/ow by the caller). */
int ct_maxpool2d_fwd(const float* in, float* out, long planes, int h, int w, int oh, int ow,
                     int k, int stride, int pad, ct_stream_t stream);
/* Context pooling of models/RFB_Net_vgg.py:243-244 on the channels-last head output:
 * in  = conf logits of one source, per image [h*w, ch] at in + n*in_img_stride
 * out = max_pool2d(k, stride k, ceil_mode) -> [oh*ow, ch] at out + n*out_img_stride */
int ct_ctx_pool_fwd(const float* in, long long in_img_stride, float* out, long long out_img_stride,
                    int batch, int h, int w, int ch, int k, ct_stream_t stream);

/* ------------------------------------------------ Context-Transformer block ---- */

/* models/RFB_Net_vgg.py:253-271 fused (flash-style, W never materialised):
 *   theta = Lin_t(conf)+conf; phi = Lin_p(pool)+pool; g = Lin_g(pool)+pool
 *   delta = softmax(theta phi^T, dim=2) g * Wz;  nov = normalize(conf+delta) OBJ^T * scale
 *   setting 'incre' (fc_w != NULL): out = cat(Lin_fc(conf)+conf, nov)
 * conf dev [B,P,d], pool dev [B,M,d], out dev [B,P,(fc_w?d:0)+T]; d <= 64, T <= 32.
 * Linear weights are [d,d] row-major (out,in) as torch.nn.Linear stores them.
 * Both contractions carry their fp32 operands on the 16-bit matrix pipe: as three bfloat16 pieces / six products (default), or,
 * with CTDET_ATTN_H2=1 in the environment (read per call; round 6), as two binary16 pieces / three products (csrc/ct_f16x2.h: a
 * query row scaled by its own power of two, phi and g by one per image, the probabilities by 2^14) -- 1.2-1.34x faster, same error
 * against fp64, another rounding.  ct_ctx_attention_piece_products() returns the products per multiply-add in force (6 or 3);
 * ct_ctx_attention_fwd_train follows the same switch. */
typedef struct ct_ctx_params {
    const float *theta_w, *theta_b, *phi_w, *phi_b, *g_w, *g_b;
    const float *wz;          /* [d] */
    const float *obj_w;       /* [T,d] */
    const float *fc_w, *fc_b; /* optional ('incre'), [d,d],[d] */
    float scale;              /* the non-trainable `scale` parameter (5) */
    int d, t;
} ct_ctx_params;
size_t ct_ctx_attention_workspace_bytes(int batch, int num_priors, int num_ctx, int d);
int ct_ctx_attention_piece_products(void);
int ct_ctx_attention_fwd(const float* conf, const float* pool, int batch, int num_priors,
                         int num_ctx, const ct_ctx_params* prm, float* out,
                         void* workspace, size_t workspace_bytes, ct_stream_t stream);


/* ---- training of the Context-Transformer block ------------------------------------------------
 * The reference differentiates models/RFB_Net_vgg.py:253-271 with autograd (train.py:222-229).
 * Here: a forward that additionally saves, per query row, the aggregated context D = softmax(S) g
 * and the row log-sum-exp (`saved`, ct_ctx_attention_saved_bytes), and one backward entry point
 * that recomputes the affinity tiles on the MFMA path instead of storing the [P,M] matrix. */
size_t ct_ctx_attention_saved_bytes(int batch, int num_priors);
int ct_ctx_attention_fwd_train(const float* conf, const float* pool, int batch, int num_priors,
                               int num_ctx, const ct_ctx_params* prm, float* out, void* saved,
                               size_t saved_bytes, void* workspace, size_t workspace_bytes,
                               ct_stream_t stream);
/* Gradient buffers (device), same shapes as the parameters; overwritten by ct_ctx_attention_bwd. */
typedef struct ct_ctx_grads {
    float *theta_w, *theta_b, *phi_w, *phi_b, *g_w, *g_b;
    float *wz, *obj_w;
    float *fc_w, *fc_b;       /* required iff prm->fc_w */
} ct_ctx_grads;
size_t ct_ctx_attention_bwd_workspace_bytes(int batch, int num_priors, int num_ctx);
/* dout [B,P,(fc?d:0)+T] -> dconf [B,P,d] (direct + theta + fc_base paths; the pooled path is added
 * by ct_ctx_pool_bwd), dpool [B,M,d], parameter gradients in `grads`. */
int ct_ctx_attention_bwd(const float* conf, const float* pool, int batch, int num_priors, int num_ctx,
                         const ct_ctx_params* prm, const void* saved, const float* dout, float* dconf,
                         float* dpool, const ct_ctx_grads* grads, void* workspace,
                         size_t workspace_bytes, ct_stream_t stream);
/* Backward of ct_ctx_pool_fwd (max_pool2d, kernel = stride = k, ceil_mode, channels-last): adds the
 * window gradient dpool to din at the first maximum of each window (torch's tie rule). */
int ct_ctx_pool_bwd(const float* in, long long in_img_stride, const float* dpool,
                    long long dpool_img_stride, float* din, long long din_img_stride, int batch,
                    int h, int w, int ch, int k, ct_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CTDET_H */
