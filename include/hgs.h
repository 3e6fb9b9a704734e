/*
 * hgs.h -- C ABI of libhgs.so, the MI355X (gfx950) rasterizer + distCUDA2 library.
 *
 * This is the drop-in boundary for the reference's native layer.  Every entry point names the
 * reference interface it replaces (paths relative to /root/reference/submodules/):
 *
 *   hgs_forward_preprocess + hgs_forward_render
 *        <-> RasterizeGaussiansCUDA            diff-gaussian-rasterization/rasterize_points.cu:35-115
 *            CudaRasterizer::Rasterizer::forward  .../cuda_rasterizer/rasterizer_impl.cu:198-336
 *            (split at the one point where the reference needs `num_rendered` on the host, :280-285)
 *   hgs_backward
 *        <-> RasterizeGaussiansBackwardCUDA    diff-gaussian-rasterization/rasterize_points.cu:117-196
 *            CudaRasterizer::Rasterizer::backward .../cuda_rasterizer/rasterizer_impl.cu:340-434
 *   hgs_mark_visible
 *        <-> markVisible                       diff-gaussian-rasterization/rasterize_points.cu:198-217
 *   hgs_dist2
 *        <-> distCUDA2 / SimpleKNN::knn        simple-knn/spatial.cu:15-26, simple-knn/simple_knn.cu:186-222
 *   hgs_geom_bytes / hgs_image_bytes / hgs_binning_bytes
 *        <-> required<GeometryState|ImageState|BinningState>()  .../cuda_rasterizer/rasterizer_impl.h:67-71
 *            (the reference grows torch byte tensors through resize callbacks, rasterize_points.cu:27-33;
 *             here the caller asks for the size and allocates)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch), fp32/int32 contiguous, unless
 *     the name ends in _host;  "absent" optional inputs are NULL (the reference passes 0-element tensors);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work,
 *     except hgs_forward_preprocess which waits for the instance count when num_rendered_host != NULL;
 *   - return value 0 = ok, non-zero = error, message in hgs_last_error() (thread-local);
 *   - matrices are the reference's row-vector (transposed) 4x4s, read as m[0],m[4],m[8],m[12] per row
 *     (cuda_rasterizer/auxiliary.h:58-77);
 *   - the three opaque buffers play the roles of geomBuffer / imgBuffer / binningBuffer
 *     (rasterize_points.cu:72-78): written by forward, handed back unchanged to backward.
 *   - not re-entrant on the same buffers; distinct buffers + distinct streams may overlap.
 */
#ifndef HGS_H_INCLUDED
#define HGS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped with every incompatible change of a struct, a signature or a buffer layout (round 1: 1, round 2: 2, round 3: 3, then 4 with the one-launch parameters + preprocess entry points;
 * 5, round 5: hgs_backward_multi_params / hgs_hair_endpoint_gather, HgsPrologue.adam_prep and the in-lane Adam update, and the contract that HgsHeadParams.tile_used also limits
 * what hgs_loss_head_forward writes of d_extra_unit -- a caller of version 4 that read those planes everywhere must not);
 * the Python binding refuses a library whose version or struct sizes differ from its own */
#define HGS_ABI_VERSION 6
#define HGS_TILE 16 /* cuda_rasterizer/config.h:16-17 */

int hgs_abi_version(void);
const char* hgs_last_error(void);

/* ---- workspace sizes (bytes).  Layout is private; hgs_*_layout() exposes it for tests. ---- */
size_t hgs_geom_bytes(int P);
size_t hgs_image_bytes(int W, int H);
size_t hgs_binning_bytes(int R);
size_t hgs_backward_scratch_bytes(int P, int R);

/* Forward, part 1: per-Gaussian preprocess (cull, cov3D, EWA cov2D, conic, radius, tile rect, SH->RGB)
 * + per-tile instance counts + scans.  Writes radii[P].  If num_rendered_host != NULL the call blocks
 * until R = num_rendered is known and stores it there (reference: rasterizer_impl.cu:280-281);
 * with NULL nothing blocks and R stays on the device (pass a capacity to hgs_forward_render).
 * max_rendered (device, may be NULL): raised atomically to R -- a sticky maximum over all calls since the caller last
 * cleared it, which is what a captured HIP graph needs to validate its capacity after any number of replays without a
 * per-iteration device-to-host copy.  With num_rendered_host == NULL and max_rendered != NULL the scans are left to the
 * scatter kernel of hgs_forward_render (no one-workgroup scan launch in between): R and its maximum are then on the
 * device once THAT call has run on the stream; max_rendered has to stay valid until then.
 * prefiltered: bit 0 = the reference's flag (accepted, ignored); bit 1 (HGS_IMAGE_PREZEROED) = the caller has already
 * cleared the counters of image_buf on this stream (hgs_iteration_prologue over hgs_image_zero_range): the call's own
 * clearing launch is skipped; bit 2 (HGS_COUNT_ROW_RUNS) = count the tile rectangles of more than 16 tiles by their tile
 * ROWS (two marks per row, which the pass's scan turns into counts) instead of tile by tile: the same counts; pays
 * where Gaussians cover many tiles each (a Stage-I cloud at 1080p, a merged strand
 * model: callers set it from the instances per Gaussian they have seen -- diff_gaussian_rasterization/_C.py). */
#define HGS_IMAGE_PREZEROED 2
#define HGS_COUNT_ROW_RUNS 4
/* Value a max_rendered word takes when a workgroup of a pass gave up waiting for another one of the same launch (bounded
 * spins of the list-parallel sort / blend: never observed; would mean the dispatcher kept a predecessor from running).
 * The frame of that pass is invalid. */
#define HGS_WAIT_TIMED_OUT 0xFFFFFFFFu
int hgs_forward_preprocess(void* stream, int P, int D, int M, int W, int H,
                           const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, float scale_modifier,
                           const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* campos,
                           float tan_fovx, float tan_fovy, int prefiltered,
                           void* geom_buf, void* image_buf, int* radii, int* num_rendered_host,
                           unsigned int* max_rendered);

/* Forward, part 2: instance scatter (tile binning), per-tile depth sort, front-to-back blend.
 * `R_capacity` = number of instances binning_buf was sized for (== num_rendered in the blocking mode).
 * out_color is [3,H,W].  Fails (status in hgs_read_status) if the scene needs more than R_capacity. */
int hgs_forward_render(void* stream, int P, int W, int H, int R_capacity, const float* bg,
                       const float* colors_precomp, void* geom_buf, void* binning_buf, void* image_buf,
                       float* out_color);

/* Backward.  bg == NULL means a BLACK background (the same gradients as a zero vector; the background terms of the blend
 * backward are compiled out -- what a training step on a black background, train.py:94, should pass).
 * All nine gradient outputs are fully written by the call (zeros for culled Gaussians):
 * the caller does not need to zero-fill them (reference zero-allocates, rasterize_points.cu:151-159).
 * dL_dconic is [P,2,2] (element [1,0] is written as 0), dL_dmeans2D is [P,3] (z = 0).
 * `scratch` >= hgs_backward_scratch_bytes(P, R) bytes.  viewmatrix, projmatrix and campos are REQUIRED whatever the colour
 * source (the kernel reads them at entry; NULL is refused). */
int hgs_backward(void* stream, int P, int D, int M, int R, int W, int H, const float* bg,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                 const void* geom_buf, const void* binning_buf, const void* image_buf,
                 const float* dL_dpix, void* scratch,
                 float* dL_dmeans2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                 float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                 float* dL_drotations);

/* Single-pass mode: RGB + 4 extra unclamped per-Gaussian channels (the reference's mask value and
 * world-space direction) composited with the same weights in ONE traversal -- what the reference obtains from three
 * separate render() calls per iteration (train.py:146, loss/losses.py:247 and :312).  hgs_forward_preprocess is
 * shared; bg7 / out_color7 have 7 channels; dL_dpix_planes7 is a HOST array of 7 device pointers, one [H,W] gradient
 * plane per output channel (the planes need not be adjacent: the losses produce them separately); extra4 is [P,4] (16-byte aligned); workspace sizes from the
 * *_multi size functions.  dL_dmeans2D_rgb is the screen-space gradient of the RGB channels alone (the only one
 * the reference's densification statistics see); dL_dmeans3D uses the total. */
size_t hgs_binning_bytes_multi(int R);
size_t hgs_backward_scratch_bytes_multi(int P, int R);
int hgs_forward_render_multi(void* stream, int P, int W, int H, int R_capacity, const float* bg7,
                             const float* colors_precomp, const float* extra4, void* geom_buf, void* binning_buf,
                             void* image_buf, float* out_color7);
int hgs_backward_multi(void* stream, int P, int D, int M, int R, int W, int H, const float* bg7,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                       const void* geom_buf, const void* binning_buf, const void* image_buf,
                       const float* const* dL_dpix_planes7, void* scratch, float* dL_dextra4, float* dL_dmeans2D_rgb,
                       float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_dmeans3D,
                       float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations);

/* present[i] = (view-space z > 0.2)   (cuda_rasterizer/rasterizer_impl.cu:54-66) */
int hgs_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* present);

/* distCUDA2: out[i] = mean of the 3 smallest squared distances from point i to the other points. */
size_t hgs_dist2_scratch_bytes(int P);
int hgs_dist2(void* stream, int P, const float* points, float* out, void* scratch, size_t scratch_bytes);

/* ---- MI355X-native fusions of PyTorch-side work ON the training-step path.  The reference has
 * no native counterpart: it runs these as chains of small torch kernels.  Every pointer is a device pointer except
 * `window11_host` (the 11 normalised fp32 taps of the Gaussian window, host memory).
 *
 * hgs_ssim_l1_forward/backward <-> loss/losses.py:16-17 (l1_loss) + :43-84 (ssim: five grouped 11x11 conv2d) and
 *   their autograd.  forward writes 3*C*H*W floats of derivative maps + 2 floats per 32x32 block and channel
 *   (hgs_ssim_l1_num_blocks of them)
 *   (partial sums of the SSIM map and of |img1-img2|; the caller sums them and divides by C*H*W).
 *   backward: dL_dimg1 = g_ssim_mean/(CHW) * dSSIM/dimg1 + g_l1_mean/(CHW) * sign(img1-img2), g_* device scalars.
 * hgs_strand_geometry_forward/backward <-> scene/hair_gaussian_model.py:134-201 getters (get_xyz, get_scaling,
 *   get_rotation, get_orientation) + utils/transform.py:69-86, and their autograd (gather / Rodrigues /
 *   matrix_to_quaternion / scatter-add into shared endpoints).  endpoint_pairs is int64 [P,2] as torch stores it.
 *   backward: any of g_xyz/g_scale/g_quat/g_dir may be NULL (output unused); d_endpoints [E,3] is zeroed inside. ---- */
size_t hgs_ssim_l1_scratch_floats(int C, int H, int W);
int hgs_ssim_l1_num_blocks(int C, int H, int W);
int hgs_ssim_l1_forward(void* stream, int C, int H, int W, const float* window11_host, const float* img1,
                        const float* img2, float* dmaps, float* partials);
int hgs_ssim_l1_backward(void* stream, int C, int H, int W, const float* window11_host, const float* img1,
                         const float* img2, const float* dmaps, const float* g_ssim_mean, const float* g_l1_mean,
                         float* dL_dimg1);
/* hgs_orientation_loss_forward/backward <-> loss/losses.py:250-288 (everything after the orientation render):
 *   omap [3,H,W] rendered world-space directions; viewmatrix = world_view_transform [4,4] (device; rows 0-2, cols 0-1 used);
 *   mask uint8 [H,W] or NULL (then mask = any(omap != bg3_host)); partials: 2 floats per 256-pixel block
 *   (sum of diff*confidence over the mask, mask count); loss = sum0 / sum1.  backward: g_loss, mask_count are
 *   device scalars; d_omap [3,H,W] fully written. */
int hgs_orientation_loss_num_blocks(int H, int W);
int hgs_orientation_loss_forward(void* stream, int H, int W, const float* omap, const float* viewmatrix,
                                 const float* bg3_host, float min_val, const float* gt_theta, const float* confidence,
                                 const uint8_t* mask, float* partials);
int hgs_orientation_loss_backward(void* stream, int H, int W, const float* omap, const float* viewmatrix,
                                  const float* bg3_host, float min_val, const float* gt_theta, const float* confidence,
                                  const uint8_t* mask, const float* g_loss, const float* mask_count, float* d_omap);
/* hgs_adam_step <-> torch.optim.Adam(lr=0, eps=1e-15) as built at scene/gaussian_model.py:250 and
 *   scene/hair_gaussian_model.py:246: all parameter tensors (<= 8) updated by ONE launch.  The six arrays are HOST arrays
 *   of n_tensors DEVICE pointers; lr[k] and step[k] point to fp32 device scalars (step is incremented by the call, by
 *   the workgroup that takes the tensor's last ticket).  tickets: 8 uint32 words in device memory owned by the optimizer,
 *   zero before its first step (they return to zero at the end of every completed launch; an optimizer whose launch was
 *   aborted zeroes them again).  NULL: process-wide words -- all such calls must then be stream-ordered.
 * hgs_smoothness_forward/backward <-> loss/losses.py:175-221 angle_smoothness_loss: index_pairs is the int64 [N,2,2]
 *   table of consecutive strand segments (endpoint ids); partials: 2 floats per 256 pairs (sum of squared angles of the
 *   pairs bent more than the threshold, their count); loss = sum0 / max(sum1, 1).  backward zeroes d_endpoints [E,3]
 *   and scatters with fp32 atomics (order-dependent in the last bits). */
int hgs_adam_step(void* stream, int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const float* const* lr, float* const* step, const long long* numel,
                  float beta1, float beta2, float eps, unsigned int* tickets);
int hgs_smoothness_num_blocks(int N);
int hgs_smoothness_forward(void* stream, int N, const float* endpoints, const long long* index_pairs,
                           float cos_threshold, float eps, float* partials);
int hgs_smoothness_backward(void* stream, int N, int E, const float* endpoints, const long long* index_pairs,
                            float cos_threshold, float eps, const float* g_loss, const float* count, float* d_endpoints);
int hgs_strand_geometry_forward(void* stream, int P, const float* endpoints, const long long* endpoint_pairs,
                                const float* width, float dist_to_scale_factor, float* xyz, float* scale, float* quat,
                                float* dir);
int hgs_strand_geometry_backward(void* stream, int P, int E, const float* endpoints, const long long* endpoint_pairs,
                                 const float* width, float dist_to_scale_factor, const float* g_xyz, const float* g_scale,
                                 const float* g_quat, const float* g_dir, float* d_endpoints, float* d_width);

/* ---- the strand-Gaussian training iteration as a handful of launches -------------------------
 * The reference runs everything between the model parameters and the rasterizer, and between the rendered images and
 * the scalar loss, as ~60 small PyTorch kernels per iteration (train.py:135-168, loss/losses.py:224-355,
 * scene/hair_gaussian_model.py:134-201,1401-1408).  On MI355X those launches, not the math, bound the step; the
 * entry points below do the same arithmetic in one kernel per stage.
 *
 * HgsViewTargets: one training view's targets, resident in HBM for the whole run (288 GB holds every view of a capture
 *   session).  The loss-head kernels read it from DEVICE memory, so a captured HIP graph switches views by rewriting
 *   this one struct (hgs_select_view) instead of copying ~50 MB of images into a staging slot.  viewmatrix /
 *   projmatrix / campos are stored by value in the layout the reference's Camera holds them (world_view_transform,
 *   full_proj_transform: row-major, transposed; scene/cameras.py:59-62), so `&slot->viewmatrix[0]` is a valid
 *   `viewmatrix` argument of the rasterizer entry points. */
typedef struct HgsViewTargets {
  const float* image;           /* [3,H,W] ground-truth RGB (Camera.original_image) */
  const float* float_mask;      /* [H,W] mask as floats, BCE target (Camera.float_mask); NULL: no mask term */
  const float* orientation;     /* [H,W] ground-truth strand angle in [0,pi) (Camera.orientation_field) */
  const float* confidence;      /* [H,W] (Camera.orientation_confidence) */
  const unsigned char* mask;    /* [H,W] bool mask of the orientation term (Camera.mask); NULL: any(omap != bg) */
  float viewmatrix[16];
  float projmatrix[16];
  float campos[3];
  float mask_count;             /* number of set pixels of `mask` (0: unknown / no mask); lets the orientation term's
                                   gradient be formed in the forward pass */
} HgsViewTargets;
size_t hgs_view_targets_bytes(void);   /* sizeof(HgsViewTargets), for bindings that mirror the struct */
size_t hgs_head_params_bytes(void);    /* sizeof(HgsHeadParams) */
size_t hgs_strand_fusion_bytes(void);  /* sizeof(HgsStrandFusion) */
/* ---- Adam in the backward's own lanes (round 5; one rank).  The lane of the backward that has just finished the gradient of a
 * parameter element -- hgs_backward_multi_params' per-Gaussian lanes, hgs_hair_endpoint_gather's endpoint lanes -- applies that
 * element's Adam update at once (HgsAdamSlot: parameter, both moments, the step's two coefficients); an iteration then has no
 * optimizer launch.  The coefficients lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) and the advance of the step counters t are the
 * business of the iteration's PROLOGUE (HgsPrologue.adam_prep: one thread per tensor of a DEVICE-resident HgsAdamPrep), i.e. of
 * a launch in front of every kernel that reads them.  Same arithmetic as hgs_adam_step, bit for bit (csrc/hgs_adam.h): a run
 * may mix iterations of either form.  The gradients are written as without it. ---- */
#define HGS_ADAM_MAX_TENSORS 8
typedef struct HgsAdamSlot { float* p; float* m; float* v; const float* coef; } HgsAdamSlot;   /* p == NULL: not updated here */
typedef struct HgsAdamPrep {                     /* lives in DEVICE memory */
  int n; const float* lr[HGS_ADAM_MAX_TENSORS]; float* step[HGS_ADAM_MAX_TENSORS]; float beta1, beta2;
  float* coef;                                   /* [n][2] out: step_size, inv_sqrt_bc2 of tensor k (HgsAdamSlot.coef = coef + 2 k) */
} HgsAdamPrep;
/* slots by HgsParamBackward.kind -- HGS_PARAMS_HAIR: 0 width, 1 opacity, 2 mask, 3 SH DC ([P,1,3]);  HGS_PARAMS_CLOUD: 0 xyz,
 * 1 scaling, 2 rotation, 3 opacity, 4 mask, 5 SH DC;  hgs_hair_endpoint_gather: 0 endpoints */
typedef struct HgsAdamInline { HgsAdamSlot slot[6]; float beta1, beta2, eps; } HgsAdamInline;
size_t hgs_adam_prep_bytes(void);     /* sizeof(HgsAdamPrep) */
size_t hgs_adam_inline_bytes(void);   /* sizeof(HgsAdamInline) */

/* slot[0] = table[view]; if lr_dst != NULL also *lr_dst = lr (the position learning rate of this iteration, a by-value
 * kernel argument, so the host may run ahead of the device without racing on a staging buffer). */
int hgs_select_view(void* stream, const HgsViewTargets* table, int view, HgsViewTargets* slot, float lr, float* lr_dst);
/* Iteration prologue: hgs_select_view and, in the same launch, the clearing of zero_bytes bytes at zero_ptr -- meant for
 * the counters at the head of the image buffer of the coming hgs_forward_preprocess (range: hgs_image_zero_range; pass
 * HGS_IMAGE_PREZEROED to that call so that it does not clear them again).  One launch instead of three small ones in
 * front of every iteration. */
int hgs_iteration_prologue(void* stream, const HgsViewTargets* table, int view, HgsViewTargets* slot, float lr, float* lr_dst,
                           void* zero_ptr, size_t zero_bytes, const HgsAdamPrep* adam_prep /* NULL: none */);
int hgs_image_zero_range(int W, int H, size_t* offset, size_t* bytes);   /* of an image_buf of hgs_image_bytes(W, H) */
/* The same prologue as a rider of another launch: hgs_hair_params_forward / hgs_cloud_params_forward -- the first launch of
 * an iteration, which needs neither the view nor the counters -- run it in spare workgroups when HgsStrandFusion.prologue
 * is filled in (table != NULL): no launch of its own in front of the iteration (4 us of a 290 us iteration).  The graph
 * functions below find and re-point such a rider exactly like a stand-alone prologue launch. */
typedef struct HgsPrologue {
  const HgsViewTargets* table; int view; HgsViewTargets* slot; float lr; float* lr_dst;   /* as hgs_iteration_prologue */
  void* zero_ptr; size_t zero_bytes;
  const HgsAdamPrep* adam_prep;   /* NULL: none (see HgsAdamSlot) */
} HgsPrologue;
/* A captured HIP graph that holds exactly ONE hgs_iteration_prologue / hgs_select_view launch is re-pointed at another
 * view (and learning rate) without any launch between two replays: hgs_graph_find_prologue(hipGraph_t) returns that
 * kernel node once after the capture, hgs_graph_set_prologue(hipGraphExec_t, node, ...) rewrites its arguments (host
 * work only; takes effect at the next launch of the executable graph). */
int hgs_graph_find_prologue(void* graph, void** node_out);
/* several iterations captured in one graph: all its prologue nodes (any order) with the `lr` each was captured with, which
 * the caller uses as a tag to tell them apart */
int hgs_graph_find_prologues(void* graph, int max_nodes, void** nodes_out, float* lr_out, int* n_out);
/* (the node's HgsPrologue.adam_prep stays what it was captured with) */
int hgs_graph_set_prologue(void* graph_exec, void* node, const HgsViewTargets* table, int view, HgsViewTargets* slot, float lr,
                           float* lr_dst, void* zero_ptr, size_t zero_bytes);
/* Several views per optimizer step inside ONE captured graph (strong-scaling protocol, SURVEY.md 8e: a fixed global batch
 * of V views per step, rank r takes views r, r+N, ...).  The graph holds one hgs_select_view_queued launch per local view,
 * which reads the view index from DEVICE memory; before every replay the host writes the step's indices (and the
 * position learning rate) with ONE hgs_set_view_queue launch whose values are by-value kernel arguments (n <=
 * HGS_VIEW_QUEUE_MAX), so the host may run ahead of the device.
 *   hgs_set_view_queue:     queue[i] = views[i] (i < n), *lr_slot = lr
 *   hgs_select_view_queued: slot[0] = table[*view_index]; if lr_dst != NULL also *lr_dst = *lr_slot.  An index outside
 *                           [0, n_views) selects view 0 (a replay with an unwritten queue must not read wild memory). */
#define HGS_VIEW_QUEUE_MAX 16
int hgs_set_view_queue(void* stream, int* queue, int n, const int* views_host, float lr, float* lr_slot);
int hgs_select_view_queued(void* stream, const HgsViewTargets* table, int n_views, const int* view_index,
                           HgsViewTargets* slot, const float* lr_slot, float* lr_dst);

/* The tail of the loss head's reduction (hgs_loss_head_forward, below): the sums over the per-pixel kernel's partials
 *   and what depends on them --
 *   out[HGS_HEAD_MASK], out[HGS_HEAD_ORIENTATION], out[HGS_HEAD_ORI_COUNT] and the last three terms of out[HGS_HEAD_TOTAL]
 *   (every other entry of `out`, and total's first two terms, are written by the forward's own kernels).  The forward runs
 *   it as a one-workgroup launch of its own -- 5.7 us on the iteration's critical path -- unless HgsHeadParams.defer_tail is
 *   set: then `out` is complete only once a later launch has run the tail in a spare workgroup: hgs_hair_params_backward /
 *   hgs_cloud_params_backward do when HgsStrandFusion.head_tail is filled in (hgs_loss_head_tail), and
 *   hgs_loss_head_backward runs it first itself when its per-pixel pass needs out[HGS_HEAD_ORI_COUNT] (no
 *   HGS_HEAD_SKIP_PIXELS).  Same arithmetic either way: the values do not depend on who runs the tail. */
typedef struct HgsHeadTail {
  const float* pix_partials; int nb_pix;   /* [nb_pix][3]: orientation sum, orientation count, mask BCE sum */
  float* out;                              /* NULL: none */
  float inv_hw, l_mask, l_ori, l_smooth;
  int bce, ori, smooth;
} HgsHeadTail;

/* hgs_hair_params_forward/backward: hgs_strand_geometry_* plus the appearance activations of the same Gaussians
 *   (scene/gaussian_model.py:93-99 get_opacity / get_mask = sigmoid) and the 4 extra blended channels of the
 *   single-pass rasterizer, extra4 = [sigmoid(mask_raw), dir.xyz].  backward: g_extra4 [P,4] carries the gradient of
 *   the mask channel and of the direction (added to g_dir); `opacity`/`extra4` are the forward outputs;
 *   accumulate_endpoints != 0: d_endpoints is NOT zeroed first (it already holds the smoothness gradient). */
/* Optional work folded into the two launches (host struct, NULL = none): extra workgroups evaluate the smoothness term
 * over the same endpoints (forward: partial sums in the layout of hgs_smoothness_forward; backward: its gradient scattered
 * into d_endpoints, scaled by head_out[HGS_HEAD_G_SMOOTH] * *grad_out / max(head_out[HGS_HEAD_SMOOTH_COUNT], 1)), and the
 * backward also updates the densification statistics of hgs_densify_stats for its Gaussians.  Any group whose first
 * pointer is NULL is skipped. */
typedef struct HgsStrandFusion {
  const long long* smooth_pairs; int n_smooth; float cos_threshold, eps;
  float* smooth_partials;                          /* forward out */
  float* smooth_pair_grads;                        /* optional, [n_smooth][2][4]: forward out -- the pairs' unit gradients (g0, ok),
                                                      (g1, ok) -- and hgs_hair_endpoint_gather in: its endpoint lanes then read 16
                                                      bytes per pair role instead of evaluating the pair again (same bits) */
  const float* head_out; const float* grad_out;    /* backward in (device) */
  const int* radii; const float* dmean2D; int dmean2D_stride;   /* backward in: statistics inputs */
  float* max_radii2D; float* grad_accum; float* denom;          /* backward in/out */
  /* backward, gather mode: adjacency of the endpoints.  ep_segments [E][2]: 2 * segment + (0|1 = which end of it), -1 =
   * none; ep_pairs [E][4]: 4 * smoothness pair + role (0..3 = a0, a1, b0, b1), -1 = none.  When ep_segments is given,
   * d_endpoints is written with ONE plain store per endpoint (one extra workgroup range recomputes the adjacent segments'
   * and pairs' contributions): no float atomics, bitwise reproducible, and accumulate_endpoints is ignored. */
  const int* ep_segments; const int* ep_pairs; int n_endpoints;
  HgsHeadTail head_tail;   /* backward: out != NULL -> one spare workgroup of the launch runs the loss head's deferred tail */
  HgsPrologue prologue;    /* forward: table != NULL -> spare workgroups of the launch run the iteration prologue */
} HgsStrandFusion;
int hgs_hair_params_forward(void* stream, int P, const float* endpoints, const long long* endpoint_pairs,
                            const float* width, float dist_to_scale_factor, const float* opacity_raw,
                            const float* mask_raw, float* xyz, float* scale, float* quat, float* dir, float* opacity,
                            float* extra4, const HgsStrandFusion* fusion);
/* hgs_hair_forward_preprocess: hgs_hair_params_forward AND hgs_forward_preprocess of the same strand model as ONE launch
 *   (the iteration's first: every lane derives its segment's Gaussian -- written to xyz / scale / quat / opacity / extra4 for
 *   the backward and the render entry points, bit-identical to hgs_hair_params_forward's -- and preprocesses it from
 *   registers; SH colours only, scale_modifier 1).  Capacity mode only (max_rendered as in hgs_forward_preprocess; no
 *   blocking count), at most 8192 tiles.  `flags`: HGS_IMAGE_PREZEROED as in hgs_forward_preprocess.  The riders of
 *   `fusion` (smoothness partial sums, HgsPrologue) run BESIDE the preprocess workgroups, hence two rules:
 *     - viewmatrix / projmatrix / campos are read from fusion->prologue.table[view] when a prologue rides (the slot is being
 *       written by that very launch; the three pointers are used only without a prologue);
 *     - the prologue's zero range must start BEHIND the per-tile instance counters (hgs_image_layout: from
 *       HGS_IMG_TILE_CURSOR on): those must already be zero when the launch starts, which every pass of
 *       hgs_forward_render* in capacity mode leaves behind (its scan clears what it has read); after a pass in blocking mode,
 *       or on a buffer of unknown content, run hgs_iteration_prologue with the full range of hgs_image_zero_range first. */
#define HGS_FUSED_PREPROCESS_MAX_TILES 8192
int hgs_hair_forward_preprocess(void* stream, int P, int D, int M, int W, int H, const float* endpoints,
                                const long long* endpoint_pairs, const float* width, float dist_to_scale_factor,
                                const float* opacity_raw, const float* mask_raw, const float* shs, float* xyz, float* scale,
                                float* quat, float* opacity, float* extra4, const float* viewmatrix, const float* projmatrix,
                                const float* campos, float tan_fovx, float tan_fovy, int flags, void* geom_buf,
                                void* image_buf, int* radii, unsigned int* max_rendered, const HgsStrandFusion* fusion);
/* The same for the Stage-I cloud: hgs_cloud_params_forward AND hgs_forward_preprocess as one launch (means3D is the model's
 *   own parameter; scale / quat / opacity / extra4 are written as hgs_cloud_params_forward writes them, bit for bit); only the
 *   prologue group of `fusion` is used. */
int hgs_cloud_forward_preprocess(void* stream, int P, int D, int M, int W, int H, const float* means3D, const float* scaling_raw,
                                 const float* rotation_raw, const float* opacity_raw, const float* mask_raw, const float* shs,
                                 float* scale, float* quat, float* opacity, float* extra4, const float* viewmatrix,
                                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, int flags,
                                 void* geom_buf, void* image_buf, int* radii, unsigned int* max_rendered,
                                 const HgsStrandFusion* fusion);
int hgs_hair_params_backward(void* stream, int P, int E, const float* endpoints, const long long* endpoint_pairs,
                             const float* width, float dist_to_scale_factor, const float* opacity, const float* extra4,
                             const float* g_xyz, const float* g_scale, const float* g_quat, const float* g_dir,
                             const float* g_opacity, const float* g_extra4, int accumulate_endpoints,
                             float* d_endpoints, float* d_width, float* d_opacity_raw, float* d_mask_raw,
                             const HgsStrandFusion* fusion);

/* ---- hgs_backward_multi_params: the BACKWARD mirror of hgs_hair_forward_preprocess / hgs_cloud_forward_preprocess (round 5).
 *   hgs_backward_multi whose last per-Gaussian launch also applies the backward of the derivation parameters -> Gaussian, in the
 *   lane that has just finished the Gaussian's gradients (they never travel through memory):
 *     HGS_PARAMS_HAIR   what hgs_hair_params_backward's per-segment lanes compute -- d_width, d_opacity_raw, d_mask_raw, the
 *                       densification statistics -- and, in seg_contrib [P][2][4 floats], the gradient of the segment's first and
 *                       second endpoint (xyz, pad).  hgs_hair_endpoint_gather then sums, per endpoint, its (<= 2) segments'
 *                       contributions and (<= 4) smoothness-pair roles with one plain store (HgsStrandFusion.ep_segments /
 *                       ep_pairs are required: gather mode only) and runs the loss head's deferred tail if one is given.
 *     HGS_PARAMS_CLOUD  everything hgs_cloud_params_backward computes (d_means3D = the rasterizer's dL_dmeans3D); a deferred tail
 *                       (head_tail.out != NULL) rides in a spare workgroup of the launch.  No second launch.
 *   Results are those of hgs_backward_multi + hgs_hair_params_backward (gather mode) / hgs_cloud_params_backward bit for bit.
 *   SH colours, scale_modifier 1 (as the forward entry points).  dL_dsh [P,M,3] is written as hgs_backward_multi writes it.
 *   `extra4` / `opacity`-like inputs are the FORWARD's outputs of the same pass.  Optional (NULL = skipped): dL_dmeans2D_rgb
 *   [P,3], the statistics group (max_radii2D / grad_accum / denom, all three or none). ---- */
enum { HGS_PARAMS_HAIR = 1, HGS_PARAMS_CLOUD = 2 };
typedef struct HgsParamBackward {
  int kind;                                                                /* HGS_PARAMS_HAIR | HGS_PARAMS_CLOUD */
  const float* endpoints; const long long* endpoint_pairs; float dist_to_scale_factor;   /* hair: in */
  float* seg_contrib; float* d_width;                                                    /* hair: out */
  const float* rotation_raw;                                                             /* cloud: in [P,4], 16-byte aligned */
  float* d_means3D; float* d_scaling_raw; float* d_rotation_raw;                         /* cloud: out */
  const float* extra4;                                                     /* in: the forward's [P,4] (mask channel first) */
  float* d_opacity_raw; float* d_mask_raw;                                 /* out [P] */
  float* dL_dmeans2D_rgb;                                                  /* out [P,3], optional */
  float* max_radii2D; float* grad_accum; float* denom;                     /* in/out [P], optional (hgs_densify_stats) */
  HgsHeadTail head_tail;                                                   /* cloud: out != NULL -> a spare workgroup runs the tail */
  HgsAdamInline adam;                                                      /* slots with p != NULL: updated by the lane (see HgsAdamSlot) */
} HgsParamBackward;
size_t hgs_param_backward_bytes(void);   /* sizeof(HgsParamBackward) */
int hgs_backward_multi_params(void* stream, int P, int D, int M, int R, int W, int H, const float* bg7, const float* means3D,
                              const float* shs, const float* scales, const float* rotations, const float* viewmatrix,
                              const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                              const void* geom_buf, const void* binning_buf, const void* image_buf,
                              const float* const* dL_dpix_planes7, void* scratch, float* dL_dsh, const HgsParamBackward* params);
/* d_endpoints [E,3] fully written.  fusion: ep_segments (required), ep_pairs + the smoothness group (smooth_pairs, n_smooth,
 * cos_threshold, eps, head_out, grad_out) and head_tail as for hgs_hair_params_backward; its other groups are ignored. */
/* adam (may be NULL): slot 0 = the endpoints' Adam state; the lane then also applies endpoint i's update (HgsAdamSlot).  With a
 * smoothness term this requires fusion->smooth_pair_grads (no lane may read an endpoint while others update them: refused otherwise). */
int hgs_hair_endpoint_gather(void* stream, int E, const float* seg_contrib, const float* endpoints, float* d_endpoints,
                             const HgsStrandFusion* fusion, const HgsAdamInline* adam);

/* hgs_cloud_params_forward/backward: the Stage-I counterpart of hgs_hair_params_* -- the rasterizer-facing getters of the
 *   Gaussian cloud (scene/gaussian_model.py:118-157) and their autograd in one launch each:
 *   scale = exp(scaling_raw); quat = rotation_raw / max(|rotation_raw|, 1e-12) (F.normalize); opacity / mask = sigmoid;
 *   direction = column argmax(scale) of build_rotation(rotation_raw) (get_orientation: the longest axis in world space);
 *   extra4 = [mask, direction].  backward: g_scale / g_quat / g_opacity / g_extra4 are the rasterizer's gradients;
 *   `fusion` may carry the densification-statistics group of HgsStrandFusion (its smoothness group is ignored). */
int hgs_cloud_params_forward(void* stream, int P, const float* scaling_raw, const float* rotation_raw,
                             const float* opacity_raw, const float* mask_raw, float* scale, float* quat, float* opacity,
                             float* extra4, const HgsStrandFusion* fusion /* NULL, or its prologue group */);
int hgs_cloud_params_backward(void* stream, int P, const float* scaling_raw, const float* rotation_raw,
                              const float* opacity, const float* extra4, const float* g_scale, const float* g_quat,
                              const float* g_opacity, const float* g_extra4, float* d_scaling_raw, float* d_rotation_raw,
                              float* d_opacity_raw, float* d_mask_raw, const HgsStrandFusion* fusion);

/* hgs_loss_head_forward/backward <-> loss/losses.py:319-355 loss_function on the three rendered images:
 *   total = (1-l_dssim) L1 + l_dssim (1-SSIM) + l_mask BCEWithLogits(mask_img, float_mask) + l_orientation ORI
 *           + l_smooth SMOOTH(endpoints),  terms with weight 0 (or a NULL target) skipped.
 *   forward: 4 launches (SSIM/L1 map, BCE + orientation per pixel, smoothness per segment pair, one-block reduction);
 *   out[HGS_HEAD_*] device floats.  scratch: hgs_loss_head_scratch_floats() floats, kept for the backward.
 *   d_extra_unit ([4,H,W]: mask plane, then the 3 orientation planes; may be NULL): if given, the per-pixel pass also
 *   writes dL/d(mask_img) and dL/d(omap) FOR grad_out = 1 (requires targets->mask_count > 0 when the orientation term
 *   is on, since that gradient is normalised by the mask count); a caller whose upstream gradient is exactly 1 then
 *   passes HGS_HEAD_SKIP_PIXELS to the backward and uses those planes as they are.  With HgsHeadParams.tile_used given the
 *   planes are UNWRITTEN on the 16 x 16 tiles that are not read (ABI 5).
 *   smooth_partials_ext (may be NULL): smoothness partial sums already computed elsewhere (HgsStrandFusion): the head then
 *   launches no smoothness kernel of its own and reduces these.
 *   backward: d_image [3,H,W] fully written (but see HgsHeadParams.tile_used); d_mask_img [H,W], d_omap [3,H,W] fully written unless HGS_HEAD_SKIP_PIXELS;
 *   d_endpoints [E,3] zeroed, then (unless HGS_HEAD_SKIP_SMOOTH) the smoothness gradient scattered into it;
 *   grad_out = device scalar dL/dtotal. */
enum { HGS_HEAD_SKIP_PIXELS = 1, HGS_HEAD_SKIP_SMOOTH = 2 };
typedef struct HgsHeadParams {
  int H, W;
  float lambda_dssim, lambda_mask, lambda_orientation, lambda_smooth;
  float bg[3];                  /* background of the orientation render (loss/losses.py:249: black) */
  float min_val;                /* HairGaussianModel.min_val */
  float window[11];             /* SSIM window taps */
  int n_smooth;                 /* rows of the smoothness index table */
  float cos_threshold, eps;     /* loss/losses.py:175: threshold 30 deg, eps 1e-6 */
  int n_endpoints;
  int defer_tail;               /* != 0: the forward leaves the sums over the per-pixel partials (see HgsHeadTail) to the caller */
  /* Optional hint (NULL: none): tile_used[ty * tiles_x + tx] != 0 <=> the consumer of d_image reads the 16 x 16 tile
   * (tx, ty).  For the rasterizer backward that is the image buffer's per-tile contributor count (hgs_image_layout,
   * HGS_IMG_TILE_MAXC) of the forward pass that produced `image`: it touches dL/dpixel only where a pixel blended an entry.
   * With the hint, hgs_loss_head_backward leaves d_image UNWRITTEN on 32 x 32 blocks none of whose tiles is read (it
   * neither filters nor zero-fills them), and hgs_loss_head_forward leaves d_extra_unit UNWRITTEN on the tiles that are
   * not read. */
  const unsigned int* tile_used; int tiles_x, tiles_y;
} HgsHeadParams;
enum { HGS_HEAD_TOTAL = 0, HGS_HEAD_L1, HGS_HEAD_DSSIM, HGS_HEAD_MASK, HGS_HEAD_ORIENTATION, HGS_HEAD_SMOOTH,
       HGS_HEAD_ORI_COUNT, HGS_HEAD_SMOOTH_COUNT, HGS_HEAD_G_SSIM, HGS_HEAD_G_L1, HGS_HEAD_G_MASK, HGS_HEAD_G_ORI,
       HGS_HEAD_G_SMOOTH, HGS_HEAD_TOTAL_FWD, HGS_HEAD_NOUT = 16 };
/* (HGS_HEAD_TOTAL_FWD: total's first two terms -- what the tail, HgsHeadTail, starts from) */
size_t hgs_loss_head_scratch_floats(const HgsHeadParams* p);
/* fills `tail` for the forward that used (p, scratch, out) */
int hgs_loss_head_tail(const HgsHeadParams* p, const float* scratch, float* out, HgsHeadTail* tail);
int hgs_loss_head_forward(void* stream, const HgsHeadParams* p, const float* image, const float* mask_img,
                          const float* omap, const HgsViewTargets* targets, const float* endpoints,
                          const long long* smooth_pairs, float* scratch, float* out, float* d_extra_unit,
                          const float* smooth_partials_ext);
int hgs_loss_head_backward(void* stream, const HgsHeadParams* p, const float* image, const float* mask_img,
                           const float* omap, const HgsViewTargets* targets, const float* endpoints,
                           const long long* smooth_pairs, const float* scratch, const float* out,
                           const float* grad_out, int skip, float* d_image, float* d_mask_img,
                           float* d_omap, float* d_endpoints);

/* hgs_densify_stats <-> scene/hair_gaussian_model.py:1401-1408 / gaussian_model.py:675-682 add_densification_stats +
 *   train.py:170-171 max_radii2D update, for the Gaussians with radii > 0:
 *   max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += |dL_dmean2D.xy|; denom += 1.
 *   dL_dmean2D has `stride` floats per Gaussian (3 as the rasterizer returns it). */
int hgs_densify_stats(void* stream, int P, const int* radii, const float* dL_dmean2D, int stride, float* max_radii2D,
                      float* xyz_gradient_accum, float* denom);

/* ---- per-kernel device timing (bench.py roofline): when enabled every kernel launch of the library is bracketed
 * by hipEvents recorded on the launch stream.  hgs_prof_collect() synchronises those events, ADDS elapsed
 * milliseconds / launch counts per kernel id into the caller's arrays (length HGS_K_COUNT) and clears the log.
 * Enabling calibrates the fixed cost of an event pair with empty kernels once; every reading has it subtracted.
 * No reference counterpart (the reference only times whole iterations, train.py:81-82,133,156). ---- */
enum { HGS_K_PREPROCESS_FWD = 0, HGS_K_SCAN, HGS_K_SCATTER, HGS_K_SORT_TILES, HGS_K_BLEND_FWD, HGS_K_BLEND_BWD,
       HGS_K_PREPROCESS_BWD, HGS_K_KNN, HGS_K_SSIM_FWD, HGS_K_SSIM_BWD, HGS_K_STRAND_FWD, HGS_K_STRAND_BWD,
       HGS_K_ORI_FWD, HGS_K_ORI_BWD, HGS_K_ADAM, HGS_K_SMOOTH, HGS_K_HEAD, HGS_K_MISC, HGS_K_COUNT };
int hgs_prof_enable(int on);
double hgs_prof_bracket_overhead_ms(void);   /* calibrated cost of one event pair, subtracted from every reading */
int hgs_prof_collect(double* total_ms, long long* launches);
const char* hgs_prof_kernel_name(int kernel_id);

/* hgs_radius_pairs <-> the candidate search of strand merging, scipy cKDTree(pos).query_pairs(r) + the direction test at
 *   scene/hair_gaussian_model.py:1205-1290: all index pairs a < b of the N strand ends with
 *   |pos_a - pos_b| <= radius and -(dir_a . dir_b) >= min_cos (|.| instead if bidirectional).  Brute force, one
 *   256-point tile against all others through LDS: strand ends number in the thousands, where N^2/2 distance tests cost
 *   tens of microseconds and no tree has to be built or kept consistent with the moving endpoints.  sorted_by_x != 0: the
 *   caller passes the points in ascending order of x, and a block stops at the first tile that starts more than `radius`
 *   beyond its own last point (a model late in training has 10^5 strand ends within a few hundred radii of each other:
 *   52 ms -> 1 ms).
 *   Appends (a, b) to pairs[capacity][2] and the distance to dist[capacity] in unspecified order; *count (device, must be
 *   zero on entry) ends as the number of pairs FOUND, which may exceed capacity (the excess is dropped). */
int hgs_radius_pairs(void* stream, int N, const float* pos, const float* dir, float radius, float min_cos,
                     int bidirectional, int capacity, int* pairs, float* dist, int* count, int sorted_by_x);

/* hgs_knn3 <-> pytorch3d.ops.knn_points(p, p, K=3, return_sorted=True) as strand_joints_magnet_loss calls it on the strand
 *   ends (loss/losses.py:139-144; pytorch3d is a from-source dependency of the reference, README.md:25-30, absent from the tree):
 *   for every point the indices and squared distances of its 3 nearest points OF THE SAME SET, ascending, the point itself
 *   included (first, at distance 0); equal distances in ascending index order; fewer than 3 points: index -1, distance
 *   +inf.  Brute force through LDS tiles: the ends move every iteration and number in the thousands. */
int hgs_knn3(void* stream, int N, const float* points, int* idx /* [N,3] */, float* dist2 /* [N,3] */);
/* hgs_nearest_distance_f64 <-> scipy cKDTree(refs).query(points, k=1)[0] as compute_strands_info uses it on the strands' ends
 *   (scene/hair_gaussian_model.py:1466-1470): out[i] = min_m |points[i] - refs[m]| in float64 (points float32 [N,3], refs
 *   float64 [M,3], M >= 1). */
int hgs_nearest_distance_f64(void* stream, int N, int M, const float* points, const double* refs, double* out);
/* hgs_strand_walk_ends / hgs_strand_walk_fill <-> the walk over every open polyline of the segment table in compute_strands_info
 *   (scene/hair_gaussian_model.py:1410-1498; its per-strand Python loop :1432-1464).  pairs[n][2]: endpoint ids of the segments
 *   (every id of degree 1 or 2, ids < n_ep).
 *   _ends: builds the node table (nodes: n_ep x 16 bytes of scratch, 16-byte aligned; deg[n_ep]: degrees) and walks from every
 *   chain end: other[e] = the end its chain runs into (-1 for ids that are no chain end -- which makes `other` the reference's
 *   strand_endpoint_id_to_complementary), len[e] = the chain's segment count.  flags[0] != 0 afterwards: the table is not a set of
 *   chains (an id out of range or of degree > 2); the outputs are then undefined.
 *   _fill: for the S strands the caller has numbered (starts[s] = the end the strand is stored FROM, offsets[S+1] = prefix of their
 *   lengths, flip[s] != 0 = store it in the opposite direction), writes rows[total][2] = (current, next) endpoint id of every
 *   segment in strand order, seg_rows[total] = its row of `pairs`, id_to_strand[id] = s for every endpoint on a strand (the caller
 *   pre-fills -1).  Closed loops have no end and appear nowhere, like in the reference's walk. */
int hgs_strand_walk_ends(void* stream, int n, int n_ep, const long long* pairs, int* deg, void* nodes, int* other, int* len, int* flags);
int hgs_strand_walk_fill(void* stream, int S, const long long* starts, const long long* offsets, const unsigned char* flip,
                         const void* nodes, long long* rows, long long* seg_rows, int* id_to_strand);

/* Tile culling (default on).  The reference gives every Gaussian the tiles of its 3-sigma square (forward.cu:229-235,
 * auxiliary.h:46-56) although a pixel only blends it where opacity * exp(power) >= 1/255 (forward.cu:358): with culling on,
 * hgs_forward_preprocess keeps only the tiles that the bounding box of that ellipse reaches, so num_rendered, the tile
 * lists and n_contrib count fewer entries than the reference's -- the dropped ones are skipped by every pixel there too.
 * out_color, radii and all gradients are bit-identical either way.  Off reproduces the reference's lists exactly (the
 * parity tests of the binning stages use it).  Process-wide, takes effect at the next hgs_forward_preprocess; a forward
 * and its backward may run under different settings (the buffers carry the rectangles).  Returns the previous setting. */
int hgs_set_tile_cull(int on);

/* Development aid (tools/wg_trace.py): when a buffer of 8 uint64 per workgroup of the blend grid (2 T + 1024 is always
 * enough) is registered, blend_fwd / blend_bwd record per workgroup: start, phase marks 1..5, its work item, end (times
 * from s_memrealtime, 100 MHz); NULL (the default) switches it off. */
/* Tuning knob of the segment-parallel blend (csrc/hgs_blend.hip): tile lists longer than 1.5 segment lengths are walked
 * by one workgroup per segment (the forward walks a list of exactly two segments with one workgroup); the segment length of a pass with R instances is R / target_segments rounded up to a
 * multiple of 64 and clamped to [min_len, max_len] (multiples of 64, min_len >= 128; default 128, 1024, 2048).  Process-wide, takes
 * effect at the next forward pass (a captured graph keeps the policy it was captured with).  Results do not depend
 * on it beyond the association of the per-pixel transmittance product. */
int hgs_set_segment_policy(int min_len, int max_len, int target_segments);
/* The 7-channel backward (hgs_backward_multi / hgs_backward_multi_params) takes the per-Gaussian sums of the instance rows either
 * inside its per-Gaussian launch -- whose wavefronts wait for their longest Gaussian -- or with a launch of their own that is
 * balanced by ROWS (csrc/hgs_preprocess.hip row_reduce_kernel: runs of 512 rows per wavefront, whatever Gaussians they belong
 * to): what a pass with many instances per Gaussian wants (a Stage-I cloud at 1080p, the merged strand model: 4 - 12 x faster).
 * mode 1: always the launch; 0: never; -1 (default): where R >= 8 P.  R is the caller's argument -- in capacity mode a CAPACITY,
 * not a count --, so a caller that wants the choice to follow the model rather than its capacity sets 0 / 1 itself (the Python
 * layer does: from the instance counts of the warm-up passes of a capture, diff_gaussian_rasterization._C.set_row_reduce).
 * Process-wide, read by the next backward (a captured graph keeps what it was captured with).  The same terms either way,
 * associated differently (rounding); every form is a fixed sequence of additions for a given pass.  Returns the previous mode. */
int hgs_set_row_reduce(int mode);
/* How the blend kernels of the following forward passes (and their backwards) get the per-entry records: 0 = the sort kernel writes a
 * 48- / 64-byte record per instance which they stream (rounds 1-5), 1 = the sort kernel orders keys only and they build an entry's
 * record from its Gaussian's template through the sorted key (nothing is written for the entries behind a tile's last contributor:
 * two thirds of a dense Stage-I frame's), -1 (default) = 1 for passes with at least 128 entries per tile by capacity / exact count.
 * Images and gradients are the same bits either way.  Process-wide; returns the previous setting. */
int hgs_set_lazy_records(int mode);
int hgs_debug_set_wg_trace(void* device_buf_fwd, void* device_buf_bwd);

/* ---- introspection used by the parity tests (byte offsets of the sub-arrays of each buffer) ---- */
enum { HGS_GEOM_DEPTHS = 0, HGS_GEOM_CLAMPED, HGS_GEOM_MEANS2D, HGS_GEOM_COV3D, HGS_GEOM_CONIC_OPACITY,
       HGS_GEOM_RGB, HGS_GEOM_TILES_TOUCHED, HGS_GEOM_POINT_OFFSETS, HGS_GEOM_RECT, HGS_GEOM_BLOCK_SUMS,
       HGS_GEOM_NFIELDS };
enum { HGS_IMG_FINAL_T = 0, HGS_IMG_N_CONTRIB, HGS_IMG_RANGES, HGS_IMG_TILE_COUNT, HGS_IMG_TILE_CURSOR,
       HGS_IMG_TILE_MAXC, HGS_IMG_STATUS, HGS_IMG_TILE_ORDER, HGS_IMG_NFIELDS };
enum { HGS_BIN_KEYS = 0, HGS_BIN_POINT_LIST, HGS_BIN_PACKED, HGS_BIN_INV, HGS_BIN_KEYS_TMP, HGS_BIN_NFIELDS };
int hgs_geom_layout(int P, size_t* offsets /* [HGS_GEOM_NFIELDS] */);
int hgs_image_layout(int W, int H, size_t* offsets /* [HGS_IMG_NFIELDS] */);
int hgs_binning_layout(int R, size_t* offsets /* [HGS_BIN_NFIELDS] */);

/* status words written by the kernels into image_buf (HGS_IMG_STATUS): [0]=num_rendered, [1]=overflow flag (the pass
 * dropped instances: its image is incomplete and its backward returns exactly zero gradients), [8]=a cooperative wait
 * between workgroups timed out (never expected; results of that pass are invalid); the others are internal */
#define HGS_STATUS_WORDS 16
/* floats per packed instance record in HGS_BIN_PACKED, 3-channel mode: x,y, conic a,b,c, opacity, r,g,b, id, quadrant
 * mask, pad (the 7-channel mode uses 16: ..., 7 features, id, quadrant mask, pad) */
#define HGS_PACKED_FLOATS 12
/* floats per instance in the backward scratch: sums over the tile's pixels of u dx, u dy, u dx dx, u dx dy, u dy dy, u
 * (u = G dL/dalpha, d = mean - pixel; the moments from which dmean2D, dconic and dopacity follow per Gaussian), dcolor.rgb,
 * pad */
#define HGS_INST_GRAD_FLOATS 12

#ifdef __cplusplus
}
#endif
#endif /* HGS_H_INCLUDED */
