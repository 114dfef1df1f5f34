/* mireg.h -- C ABI of libmireg_hip.so (MI355X / gfx950 registration hot path).
 *
 * The reference (b1g-sw0rd/Self-supervised-Medical-Image-Registration-...) is pure Python and
 * defines no C ABI of its own; its device ops are torch calls and imports of un-vendored CUDA
 * extensions.  Each entry point below replaces the reference Python call site cited next to it
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer
 * adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocator in practice);
 *     kernels never allocate, free or retain pointers; workspaces are caller-provided;
 *   - plain C types only: pointers, int / long sizes and element strides, hipStream_t;
 *   - work is enqueued on `stream`, never synchronised; safe under hipGraph capture;
 *   - return 0 on success, MIREG_ERR_* (< 0) on bad arguments / launch failure; nothing throws;
 *   - "flow" tensors are addressed through element strides (sb batch, sc channel, sp pixel) so
 *     that NCHW (sc = h*w, sp = 1) and channel-interleaved NHWC (sc = 1, sp = 2) both stream.
 */
#ifndef MIREG_H_
#define MIREG_H_

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIREG_DTYPE_F32 0
#define MIREG_DTYPE_BF16 1

/* ---- library info ----------------------------------------------------------------------- */
int mireg_version(void);                 /* ABI version, bumped on any signature change */
const char* mireg_arch(void);            /* "gfx950" */

/* ---- K6: F.interpolate(mode='bilinear') -------------------------------------------------- */
/* FlowNetS/FlowNetS.py:82 (flow2 -> 256x256, align_corners=False, values NOT rescaled),
 * models.py:258 (moving -> (h,w), align_corners=True), loss.py:11,54 (fixed -> (h,w), False). */
int mireg_resize_bilinear_fwd(const float* in, float* out, int N, int C, int H, int W, int h, int w,
                              long isn, long isc, long isp, long osn, long osc, long osp,
                              int align_corners, hipStream_t stream);
int mireg_resize_bilinear_bwd(const float* gout, float* gin, int N, int C, int H, int W, int h, int w,
                              long isn, long isc, long isp, long osn, long osc, long osp,
                              int align_corners, float beta, hipStream_t stream);

/* Moment tables: per scale MIREG_SUM_SLOTS replicas of 8 doubles {Sx,Sy,Sxy,Sxx,Syy,Scharb,Ssmooth,-}; block b of
 * a reducing kernel adds into replica b % MIREG_SUM_SLOTS (spreads the float64 atomics over lines), the finalize
 * kernels sum the replicas.  The caller zeroes the table before the first accumulating launch of a step. */
#define MIREG_SUM_SLOTS 32
/* ---- K9 (+K11/K12 partials): opticalFlowReg.stn, models.py:256-268 ----------------------- */
/* frame is the moving image ALREADY resized to (h,w) (planar B,C,h,w).  When fixed != NULL the
 * kernel also accumulates the six loss moments {Sx,Sy,Sxy,Sxx,Syy,Scharb} of (warped, fixed)
 * into its replica of sums[MIREG_SUM_SLOTS][8] (entries 0..5) -- loss.py:9-14, 52-64 fused into the warp. */
int mireg_stn_warp_fwd(const float* flow, long fsb, long fsc, long fsp, const float* frame,
                       const float* fixed, float* warped, double* sums, int B, int C, int h, int w,
                       hipStream_t stream);
/* d warped / d flow (the moving image needs no gradient, train.py:44-50) */
int mireg_stn_warp_bwd(const float* flow, long fsb, long fsc, long fsp, const float* frame,
                       const float* gout, float* gflow, long gsb, long gsc, long gsp, float beta,
                       int B, int C, int h, int w, hipStream_t stream);

/* ---- K11/K12/K13 + OFEloss, loss.py:9-84 ------------------------------------------------- */
int mireg_loss_partials(const float* warped, const float* fixed, double* sums, long n, hipStream_t stream);
int mireg_loss_bwd(const float* warped, const float* fixed, const float* coef, float* gwarped, long n,
                   hipStream_t stream);
int mireg_smoothness_fwd(const float* flow, long fsb, long fsc, long fsp, double* sum, int B, int h, int w,
                         hipStream_t stream);
int mireg_smoothness_bwd(const float* flow, long fsb, long fsc, long fsp, const float* coef, float* gflow,
                         long gsb, long gsc, long gsp, float beta, int B, int h, int w, hipStream_t stream);
/* sums: [n][MIREG_SUM_SLOTS][8] doubles; npix[i] = B*h_i*w_i; out4 = (p,c,s,total)
 * as float64 (the reference returns float64 scalars, loss.py:71-73).  No host synchronisation:
 * the reference's two torch.equal() guards (loss.py:58-60) are evaluated on device. */
int mireg_ofe_finalize(const double* sums, const long* npix, int n, int B, double lamb_da, double gamma,
                       double zeta, double* out4, hipStream_t stream);
/* g4 = upstream d/d(p,c,s,total); coef: [n][8] floats {cp,k1,k2,mean_w,mean_f,cs,-,-} */
int mireg_ofe_bwd_coef(const double* sums, const long* npix, int n, int B, double lamb_da, double gamma,
                       double zeta, const double* g4, float* coef, hipStream_t stream);

/* ---- fused multi-scale tail of the training step (train.py:50-52 + loss.py:66-84 over all flow scales) ----
 * one launch per phase for ALL scales: job i = scale i with its own buffers; blk0 = first virtual block,
 * blocks_i = ceil(B*h*w / MIREG_TAIL_PIXELS_PER_BLOCK), total_blocks = sum.  x = the (B,2,H,W) fp32 batch [fixed, moving].
 *   mireg_tail_resize: moving_r = bilinear(moving, align_corners=True), fixed_r = bilinear(fixed, False)
 *   mireg_tail_fwd   : warped = stn(flow, moving_r); sums[slot][0..5] += moments(warped, fixed_r), [6] += smoothness
 *   mireg_tail_bwd   : gflow = d total / d flow from coef (mireg_ofe_bwd_coef row of the scale)
 * same per-element arithmetic as the per-scale entry points above. */
#define MIREG_TAIL_PIXELS_PER_BLOCK 1024
typedef struct mireg_tail_job {
  const float* flow; long fsb, fsc, fsp;
  float* moving_r; float* fixed_r; float* warped;
  float* gflow; long gsb, gsc, gsp;
  double* sums; const float* coef;
  int h, w, blk0, pad_;
} mireg_tail_job;
int mireg_tail_resize(const mireg_tail_job* jobs_dev, int njobs, int total_blocks, const float* x, int B, int H, int W,
                      hipStream_t stream);
int mireg_tail_fwd(const mireg_tail_job* jobs_dev, int njobs, int total_blocks, int B, hipStream_t stream);
int mireg_tail_bwd(const mireg_tail_job* jobs_dev, int njobs, int total_blocks, int B, hipStream_t stream);

/* ---- K19: F.grid_sample(vol, F.affine_grid(theta, vol.size())), models.py:187-188 (trilinear, zeros,
 * align_corners=False); planar (B,C,D,H,W) fp32 volumes, theta (B,3,4) ---------------------------------- */
int mireg_affine_sample3d(const float* vol, const float* theta, float* out, int B, int C, int D, int H, int W,
                          hipStream_t stream);
/* autograd of the above with respect to theta (the `para` output of affmodel, models.py:184-188):
 * gtheta[B][12] (+)= d/d theta of sum(gout * out).  workspace: B * MIREG_AFFINE3D_BWD_BLOCKS * 12 floats; two-stage,
 * fixed-order reduction (deterministic).  The moving volume is an input of the model, not a parameter: no d/d vol. */
#define MIREG_AFFINE3D_BWD_BLOCKS 512
int mireg_affine_sample3d_bwd(const float* vol, const float* theta, const float* gout, float* gtheta, float* workspace,
                              int accumulate, int B, int C, int D, int H, int W, hipStream_t stream);

/* ---- volume (3-D) counterparts of the registration tail: SURVEY section 8 row a14 / BASELINE config "3D FlowNetS on 128^3".
 * The reference has no dense 3-D flow; these follow its 2-D conventions axis by axis (flow channel 0/1/2 = x/y/z):
 *   resize  = F.interpolate(mode='trilinear', align_corners)      (models.py:258, loss.py:11, FlowNetS/FlowNetS.py:83)
 *   stn3d   = sample frame at (i + flow_i)(n_i - 1)/n_i per axis, zeros outside (models.py:256-268); d/d flow only
 *   smooth  = sum_c sum_axes charbonnier(f - f_shifted) / 3 / B   (loss.py:21-29 with three flow channels; the kernels
 *             add 2/3 of the raw sum into the smoothness slot so mireg_ofe_finalize / mireg_ofe_bwd_coef serve unchanged)
 * Volumes are planar fp32 (B,C,D,H,W); `in`/`gin` of the resize and every `flow` are addressed through element strides
 * (batch, channel, voxel), so channel-last predictor outputs are read in place; gflow is planar (B,3,d,h,w).
 * beta: 0 = overwrite, otherwise dst = dst*beta + value.  All backward kernels are gather-form (deterministic). */
int mireg_resize_trilinear_fwd(const float* in, long isn, long isc, long isp, float* out, int N, int C, int D, int H, int W,
                               int d, int h, int w, int align_corners, hipStream_t stream);
int mireg_resize_trilinear_bwd(const float* gout, float* gin, long isn, long isc, long isp, int N, int C, int D, int H, int W,
                               int d, int h, int w, int align_corners, float beta, hipStream_t stream);
/* the same adjoint as three 1-D passes (x, y, z) through a caller-owned workspace of N*C*d*h*W + N*C*d*H*W floats: reads the
 * fine gradient once, contiguously (sums in another order than the direct kernel: equal to fp32 rounding) */
int mireg_resize_trilinear_bwd_sep(const float* gout, float* gin, long isn, long isc, long isp, int N, int C, int D, int H, int W,
                                   int d, int h, int w, int align_corners, float beta, float* ws, long ws_elems, hipStream_t stream);
int mireg_stn3d_fwd(const float* flow, long fsb, long fsc, long fsp, const float* frame, float* warped, int B, int C, int d,
                    int h, int w, hipStream_t stream);
int mireg_stn3d_bwd(const float* flow, long fsb, long fsc, long fsp, const float* frame, const float* gout, float* gflow,
                    float beta, int B, int C, int d, int h, int w, hipStream_t stream);
int mireg_smoothness3d_fwd(const float* flow, long fsb, long fsc, long fsp, double* sum, int B, int d, int h, int w,
                           hipStream_t stream);
int mireg_smoothness3d_bwd(const float* flow, long fsb, long fsc, long fsp, const float* coef, float* gflow, float beta, int B,
                           int d, int h, int w, hipStream_t stream);
/* ---- the 3 -> 3 channel flow upsamplers of the volume predictor, ConvTranspose3d(3, 3, 4, 2, 1) (FlowNetS/FlowNetS.py:37-40 per axis; + autograd),
 * one thread per voxel on the fp32 weight w = [3 coarse][3 fine][4][4][4] (the ConvTranspose3d parameter as stored).  x_coarse / dx_coarse:
 * (B, Dc, Hc, Wc, ld_c) channel-last; y_fine / g_fine: (B, 2Dc, 2Hc, 2Wc, ld_f); three channels used.  bwd_weights writes nblocks =
 * mireg_tiny_deconv3d_blocks(...) partial slabs [nblocks][3][64 * Cpad] in the standard backward-weights layout (summed by mireg_wgrad_reduce). */
int mireg_tiny_deconv3d_blocks(int B, int Dc, int Hc, int Wc);
int mireg_tiny_deconv3d_fwd(const void* x_coarse, long ld_c, const float* w, void* y_fine, long ld_f, int B, int Dc, int Hc, int Wc,
                            int dtype, hipStream_t stream);
int mireg_tiny_deconv3d_bwd_data(const void* g_fine, long ld_f, const float* w, void* dx_coarse, long ld_c, int accumulate, int B, int Dc,
                                 int Hc, int Wc, int dtype, hipStream_t stream);
int mireg_tiny_deconv3d_bwd_weights(const void* g_fine, long ld_f, const void* x_coarse, long ld_c, float* slab, int nblocks, int Cpad, int B,
                                    int Dc, int Hc, int Wc, int dtype, hipStream_t stream);
/* x-axis im2col of a few-channel planar fp32 volume (the 7^3 / stride-2 / 2-channel input convolution of FlowNetS over
 * volumes, the 3-D counterpart of FlowNetS/FlowNetS.py:18): dst[b][z][y][xo][tx*C + c] = x[b][c][z][y][xo*stride + tx - pad]
 * (zeros outside and in the pad channels), dst channel-last with Cpad >= k*C channels (a multiple of 8), Wo = (W + 2*pad - k)/stride + 1.
 * The convolution then runs as a (k, k, 1) / stride (s, s, 1) Conv3d over k*C -> Cpad channels on mireg_conv_gemm. */
int mireg_stem3d_gather(const float* x, void* dst, int B, int C, int D, int H, int W, int k, int stride, int pad, int Cpad,
                        int dtype, hipStream_t stream);

/* ---- K14/K15: models.py:286 (rint + clip 0..3, on device), utils.py:72-91 (Dice) --------- */
int mireg_seg_round(const float* in, float* out, long n, hipStream_t stream);
/* counts: workspace B*9 floats; dice[b] = mean_l 2|A_l & B_l| / (|A_l| + |B_l|), l = 1..3 */
int mireg_dice(const float* y_true, const float* y_pred, float* counts, float* dice, int B, long n,
               hipStream_t stream);


/* ---- per-sample evaluation metrics of the inference loop (inference.py:66-75) ----
 * fixed / warped: B samples of n contiguous floats.  mireg_pair_metrics: out[b] = {MSE (utils.py:41-42), PSNR (utils.py:45-49:
 * 100 when mse < 1e-10), Pearson correlation (utils.py:58-59)} as float64; sums = workspace of B*8 doubles.
 * mireg_mutual_info: utils.py:52-55, sklearn.metrics.mutual_info_score (natural log) of the labels round(x * scale) clipped to
 * [0, bins); joint = B*bins*bins ints, marg = B*2*bins ints: workspaces that must be ZERO on entry and are zero again on
 * exit (only the touched counters are cleared, there is no dense memset or scan); out[b] float64.
 * The reference uses scale 1500 on [0,1] images: bins = 1501. */
int mireg_pair_metrics(const float* fixed, const float* warped, double* sums, double* out, int B, long n, hipStream_t stream);
int mireg_mutual_info(const float* fixed, const float* warped, int* joint, int* marg, double* out, int B, long n, int bins,
                      float scale, hipStream_t stream);
/* inference.py:70-71: skimage.metrics.structural_similarity(a, b, data_range) with its defaults -- win_size x win_size uniform
 * window (7), sample covariance, K1 = 0.01, K2 = 0.03, mean over the map cropped by (win_size-1)/2 per side; one float64 per
 * (H, W) sample.  skimage is absent from this image and unpinned in the reference: published definition, parity unpinned. */
int mireg_ssim(const float* a, const float* b, double* out, int B, int H, int W, int win_size, float data_range, hipStream_t stream);
/* utils.py:187-199 modified_hausdorff(A, B): max(mean_j min_i |A_i - B_j|, mean_i min_j |A_i - B_j|) for two point sets of
 * (row, col) float pairs (the contour points utils.py:154-170 extracts with skimage.measure.find_contours -- that extraction is
 * not part of this library); work = nA + nB floats; out = one float64. */
int mireg_modified_hausdorff(const float* A, int nA, const float* B, int nB, float* work, double* out, hipStream_t stream);
/* utils.py:155-170 extract_boundary_points on device for nmask binary masks seg == labels_dev[m] of (H, W) label maps seg_stride floats
 * apart: the marching-squares vertices of skimage.measure.find_contours(mask, 0.5) truncated to int = one (row, col) point per
 * 4-neighbour pixel pair with differing mask values, in raster order (deterministic).  counts[m] = number of points, points[m] =
 * counts[m] (row, col) float pairs at points + m * points_stride (points_stride >= 4 * H * W floats).  rowcnt / rowoff: nmask * H ints.
 * skimage repeats the first vertex of every closed contour; that duplicate is not reproduced (skimage absent: parity unpinned). */
int mireg_boundary_points(const float* seg, long seg_stride, int nmask, const float* labels_dev, int H, int W, int* rowcnt, int* rowoff,
                          int* counts, float* points, long points_stride, hipStream_t stream);
/* utils.py:201-211 dist_hausdorff without a host round trip: pair p = modified Hausdorff distance of point sets 2p and 2p+1 of the
 * buffers above (sizes read on device, at most cap points each); out[0] = mean over the pairs, out[1 + p] = the pairs' distances
 * (float64; nan when a contour is empty, as numpy's mean of an empty set).  work: npair * 2 * cap floats. */
int mireg_hausdorff_pairs(const float* points, long points_stride, const int* counts, int npair, int cap, float* work, double* out,
                          hipStream_t stream);

/* ---- K16-K18: the FlowNet2 stack's glue layers (flownet2/models.py:40-88,136-180) ----
 * Planar fp32 (B,C,H,W).  Resample2d and ChannelNorm are EXTERNAL custom layers of NVIDIA/flownet2-pytorch (sources absent
 * from the reference tree, unpinned): published definitions, parity unpinned.
 *   resample2d: out = bilinear(src, x + flow_x, y + flow_y) with the four tap indices clamped to the border; backward writes
 *               gflow (overwrite) and ADDS into gsrc with fp32 atomics (zero it first); either may be NULL.
 *   channelnorm: out[b,0] = sqrt(sum_c in[b,c]^2); backward gin = gout * in / (out + 1e-9).
 *   upsample_nearest: nn.Upsample(scale_factor=k, mode='nearest') over NC planes; backward != 0: in = gradient of the
 *               (H*k, W*k) tensor, out = gradient of the (H, W) tensor. */
int mireg_resample2d_fwd(const float* src, const float* flow, float* out, int B, int C, int H, int W, hipStream_t stream);
int mireg_resample2d_bwd(const float* src, const float* flow, const float* gout, float* gsrc, float* gflow, int B, int C, int H,
                         int W, hipStream_t stream);
int mireg_channelnorm_fwd(const float* in, float* out, int B, int C, long npix, hipStream_t stream);
int mireg_channelnorm_bwd(const float* in, const float* out, const float* gout, float* gin, int B, int C, long npix,
                          hipStream_t stream);
int mireg_upsample_nearest(const float* in, float* out, long NC, int H, int W, int k, int backward, hipStream_t stream);

/* ---- on-device elastic deformation of a batch (the Rand2DElasticd step of the reference's CPU data pipeline, dataset.py:78,
 * 150-152,205).  Planar fp32.  mireg_resize_bicubic_fwd = F.interpolate(mode='bicubic',
 * align_corners=True) over NC planes (control grid -> dense displacement, in pixels).  mireg_elastic_sample: out_img =
 * clamp(grid_sample(img, identity + disp, bicubic, zeros, align_corners=True), 0, 1), out_seg = the same grid with nearest
 * neighbour; the grid is linspace(-1,1)[i] + disp*2/size as the generator builds it; img or seg may be NULL. */
int mireg_resize_bicubic_fwd(const float* in, float* out, long NC, int H, int W, int h, int w, hipStream_t stream);
int mireg_elastic_sample(const float* img, const float* seg, const float* disp, float* out_img, float* out_seg, int B, int C, int Cs,
                         int H, int W, hipStream_t stream);
/* the RandAffined step (dataset.py:79,151): F.grid_sample(x, F.affine_grid(theta, x.size())) with torch's defaults
 * (align_corners=False, zeros): bilinear for img, nearest for seg; theta (B,2,3) row-major; img or seg may be NULL. */
int mireg_affine_sample2d(const float* img, const float* seg, const float* theta, float* out_img, float* out_seg, int B, int C,
                          int Cs, int H, int W, hipStream_t stream);

/* ---- the front of the data pipeline on device (dataset.py:52-57,73-77,83 and 141-153): Transposed /
 * SpatialCropd are views (base pointer + element strides per logical axis d, h, w), Resized = F.interpolate arithmetic (mode 0: linear,
 * align_corners=False, over every axis whose extent changes; mode 1: nearest), Rotate90d = numpy.rot90(k) on the (h, w) axes.  in:
 * N items of logical extent (D, H, W); out: N items of extent (d, h', w') with (h', w') = (h, w) for even k, (w, h) for odd k, written
 * through the element strides (osn, osd, osh, osw).  The 2-D slice pipeline is D = d = number of slices (no interpolation along d). */
int mireg_resample_volume(const float* in, long isn, long isd, long ish, long isw, int N, int D, int H, int W, float* out, long osn,
                          long osd, long osh, long osw, int d, int h, int w, int mode, int rot_k, hipStream_t stream);
/* ScaleIntensityd(minv, maxv) per item, in place: (x - min) / (max - min) * (maxv - minv) + minv over the item's n contiguous values
 * (a constant item becomes x * minv, as MONAI does).  workspace: items * MIREG_SCALE_INTENSITY_BLOCKS * 2 floats. */
#define MIREG_SCALE_INTENSITY_BLOCKS 64
int mireg_scale_intensity(float* x, int items, long n, float minv, float maxv, float* workspace, hipStream_t stream);

/* ---- K1-K4: implicit-GEMM convolution family on MFMA ------------------------------------- */
/* One descriptor drives three contractions (all NHWC, pixel stride `ld` in elements, so producers
 * write straight into channel slices of concat buffers -- replaces torch.cat, FlowNetS.py:64-79):
 *   mireg_conv_gemm  FWD   : Conv2d forward      (FlowNetS/util.py:17-46, PWCNet.py:24-31)
 *                    DGRAD : Conv2d backward-data == ConvTranspose2d forward
 *                            (FlowNetS/util.py:49-55, FlowNetS.py:39-42, PWCNet.py:33-34); the host
 *                            issues one launch per output-pixel parity class for stride 2
 *   mireg_conv_wgrad WGRAD : Conv2d backward-weights (autograd of the above)
 * GEMM view of mireg_conv_gemm: rows = logical grid (n_img, g_H, g_W); K = (ty, tx, c) with
 *   input pixel iy = gy*mul_y + off_y + ty*step_y (same in x), zero outside [0,x_H)x[0,x_W);
 *   cols = N output channels; weights packed [N][w_ld] with k = (ty*taps_x + tx)*x_C + c.
 * Output pixel of row (img, gy, gx) is (gy*y_mul_y + y_off_y, gx*y_mul_x + y_off_x) in a y_H x y_W
 * image; y (dtype `dtype`) and/or y32 (fp32) receive act(acc + bias) [+ previous y if accumulate].
 * split_k > 1: partial sums go to slab[split_k][M][N] (fp32) and a second pass finishes.
 * mireg_conv_wgrad reuses the struct: y/y_ld = dy rows over the same logical grid with N = Cout
 * channels, x = the forward input, w_bytes = readable bytes of dy; result slab[split_k][N][taps*x_C] (fp32).
 * x_C, x_ld, w_ld, y_ld (wgrad) must be multiples of 8 (bf16) / 4 (fp32); pad channels must hold zeros. */
typedef struct mireg_conv_cls {     /* one output-pixel (-voxel) parity class of a stride-2 DGRAD-form launch */
  int taps_y, taps_x, off_y, off_x, g_H, g_W, y_off_y, y_off_x;
  int taps_z, off_z, g_D, y_off_z;  /* depth axis of the class (Conv3d, up to 8 classes; ring kernel only); g_D == 0: the launch's own depth fields apply */
  const void* w; long w_ld, w_bytes;
} mireg_conv_cls;
typedef struct mireg_conv_desc {
  const void* x; long x_ld; int x_H, x_W, x_C;
  int taps_y, taps_x;
  int mul_y, mul_x, off_y, off_x, step_y, step_x;
  int g_H, g_W, n_img;
  const void* w; long w_ld; int N;
  void* y; long y_ld; int y_H, y_W, y_mul_y, y_mul_x, y_off_y, y_off_x;
  float* y32; long y32_ld;
  const float* bias; float slope; int accumulate; int dtype;
  int split_k; float* slab;
  long x_bytes, w_bytes;   /* readable bytes from x / w to the end of their allocations (buffer descriptors of the
                              LDS-DMA tile loads, each < 2 GiB) */
  int n_cls;               /* 0/1: the fields above describe the launch; 2..8: blockIdx.y picks cls[] (more than 4, or classes with a depth axis: ring kernel, algo 1) (taps, offsets,
                              sub-grid, output offset and weights per parity class), everything else is shared */
  mireg_conv_cls cls[8];
  long slab_cls_stride;    /* floats between the split-K slabs of consecutive classes */
  /* optional depth axis for Conv3d (reference models.py:39-43,160-165), NDHWC volumes; all zero for 2-D launches:
   * iz = gz*mul_z + off_z + tz*step_z in [0, x_D); rows run over (n_img, g_D, g_H, g_W); k = ((tz*taps_y+ty)*taps_x+tx)*x_C + c;
   * output voxel z = gz*y_mul_z + y_off_z in a y_D deep volume.  mireg_conv_gemm: single class (the host issues one
   * launch per output-voxel parity class for the backward-data form).  mireg_conv_wgrad: one launch per depth tap tz
   * (taps_z must be 0/1, off_z = the tap's source-slice offset), each writing its [Cout][taps_y*taps_x*x_C] column
   * block of a slab whose rows are slab_ld floats apart. */
  int x_D, taps_z, mul_z, off_z, step_z, g_D, y_D, y_mul_z, y_off_z;
  /* output-column tile width of mireg_conv_gemm: 0 = by N (128 / 64 / 32); 64 or 128 forces it (the host picks the
   * width whose tile count fills the 256 CUs most evenly) */
  int tile_n;
  /* mireg_conv_wgrad: LDS ring depth.  0 / 3 = three stages (48 KiB: three workgroups per CU, best stand-alone and when the
   * wgrad stream is the long pole); 4 = four stages (64 KiB: two per CU, leaves more of each CU to the concurrent main chain) */
  int stages;
  /* mireg_conv_wgrad: floats between consecutive Cout rows of the slab (0 = taps_y*taps_x*x_C); split-K slabs are
   * N*slab_ld apart */
  long slab_ld;
  /* mireg_conv_gemm / mireg_conv_wgrad kernel choice.  algo 0 = the halo-staged kernel (conv_halo.hip) whenever mireg_conv_halo_eligible says
   * so, else the ring kernel; 1 = ring kernel; 2 = halo kernel or MIREG_ERR_UNSUPPORTED; 3 = the 8-wave 256-row tiles (mireg_conv_gemm: conv_wide.hip,
   * tile_n 128 / 256 columns, 0 = by N; mireg_conv_wgrad: conv_wgrad_wide.hip) or MIREG_ERR_UNSUPPORTED.  tile_m = pixels per halo tile
   * (0 = by tile count, 128 or 256 forces it). */
  int algo, tile_m;
} mireg_conv_desc;
/* 1 when the halo-staged kernel applies to desc (unit-stride gather, grid 16/32/64 wide, >= 4 taps per class, N >= 64, no
 * split-K / y32 / depth); tiles_out[0] / [1] = pixel tiles per class at 128 / 256 pixels per tile (0 = not at that size). */
int mireg_conv_halo_eligible(const mireg_conv_desc* desc, long* tiles_out);
/* 1 when mireg_conv_wgrad takes the halo-staged backward-weights kernel (conv_wgrad_halo.hip) for desc: bf16, square kernel,
 * stride 1 or 2 with every tap-parity class 3x3 / 3x2 / 2x3 / 2x2 (3x3 s1, 5x5 s2, 4x4 s2), dy grid 16 / 32 / 64 wide. */
int mireg_conv_wgrad_halo_eligible(const mireg_conv_desc* desc);
/* 1 when mireg_conv_wgrad can run desc on the 256 x 256 8-wave tile of conv_wgrad_wide.hip (algo 3): bf16, 2-D, more than 128 output
 * channels. */
int mireg_conv_wgrad_wide_eligible(const mireg_conv_desc* desc);
/* 1 when algo 3 applies to desc (bf16, 2-D, x_C >= 64); tiles_out (may be NULL) = workgroups per class at the tile width
 * desc->tile_n selects. */
int mireg_conv_wide_eligible(const mireg_conv_desc* desc, long* tiles_out);
int mireg_conv_gemm(const mireg_conv_desc* desc, hipStream_t stream);
int mireg_conv_wgrad(const mireg_conv_desc* desc, hipStream_t stream);


/* ---- weight layout plumbing (torch layout <-> GEMM layouts), one launch for a table of jobs ---- */
/* pack (one job per layer; W = torch Conv2d layout [Co][Ci][kh][kw] fp32, every pad slot zeroed):
 *   FWD   pack  dst[co][(ky*kw+kx)*Cpad + ci]                                        (pitch ld)
 *   DGRAD packs cls[c].dst[ci][(ty*ntx+tx)*Cop + co], ky = ky0 + stride*ty, kx = kx0 + stride*tx,
 *         one per output-pixel parity class c (nclass = stride^2 <= 4)
 * ConvTranspose2d weights [Cin][Cout][kh][kw] are the Conv2d weights of the adjoint convolution, so
 * the same packs serve deconvolutions.
 * unpack: grad[co][ci][ky][kx] (+)= sum_{z<nsplit} slab[z][co][(ky*kw+kx)*Cpad + ci]; src = slab, dst = grad. */
typedef struct mireg_pack_class { void* dst; long ld; int ky0, kx0, nty, ntx; } mireg_pack_class;
typedef struct mireg_pack_job {
  const float* src; void* dst;
  int Co, Ci, kh, kw;
  int Cpad, Cop; long ld;
  int stride, nclass;
  mireg_pack_class cls[4];
  int nsplit, accumulate;
  int unit0;      /* first FWD-pack / unpack work unit of this job; units = Co * ceil(C/64), C = Cpad (pack) or Ci (unpack) */
  int dunit0;     /* first DGRAD-pack block: blocks = sum_c nty_c*ntx_c * ceil(Cop/64) * ceil(Ci/64) */
} mireg_pack_job;
int mireg_pack_weights(const mireg_pack_job* jobs_dev, int njobs, int total_units, int total_dgrad_units, int dtype,
                       hipStream_t stream);
int mireg_unpack_wgrad(const mireg_pack_job* jobs_dev, int njobs, int total_units, hipStream_t stream);

/* Conv3d backward-data packs (all parity classes of a layer from one pass over its torch-layout weight [Co][Ci][kd][kh][kw]):
 * class (cz,cy,cx), index (cz*sy + cy)*sx + cx, holds the taps t_a with (c_a + pad_a) % s_a == t_a % s_a as its tap
 * j_a = t_a / s_a:  dst[class][ci][(jz*nty + jy)*ntx + jx][Cop] = W[co][ci][tz][ty][tx], pad output channels zeroed.
 * units = Ci * ceil(Cop/64) per job.  ConvTranspose3d weights [Cin][Cout][k^3] are the Conv3d weights of the adjoint. */
typedef struct mireg_pack3d_job {
  const float* src; void* dst[8];
  int Co, Ci, Cop;
  int kd, kh, kw, sz, sy, sx, pz, py, px;   /* kernel extent, stride, padding per axis (z, y, x) */
  int unit0;
} mireg_pack3d_job;
int mireg_pack_dgrad3d(const mireg_pack3d_job* jobs_dev, int njobs, int total_units, int dtype, hipStream_t stream);
/* The same packs from the layers' FWD packs (mireg_pack_weights / mireg_adam_pack output, element type = dtype) instead of the fp32
 * weights: job.src = FWD pack [Co][taps*Cip], Cip = Ci rounded up to 8; units = taps * ceil(Cop/64) * ceil(Ci/64) per job. */
int mireg_pack_dgrad3d_fwd(const mireg_pack3d_job* jobs_dev, int njobs, int total_units, int dtype, hipStream_t stream);

/* ---- two-output-channel 3x3 / stride 1 / pad 1 convolutions (predict_flow heads: FlowNetS/util.py:33-34,
 * flownet2/networks/submodules.py:32-33, PWC/models/PWCNet.py:31-32) on the vector ALUs.  w = the FWD pack
 * [2][9*Cpad] of mireg_pack_weights; x / dx = wide NHWC tensors (Cpad channels walked, 16-byte aligned rows);
 * y / dy = two-channel NHWC tensors.  wgrad writes `ntiles` partial slabs [ntiles][2][9*Cpad] (fp32), summed by
 * mireg_wgrad_reduce / mireg_unpack_wgrad; ntiles must be what mireg_thin_conv_wgrad_tiles returns. */
int mireg_thin_conv_fwd(const void* x, long ld_x, const void* w, long ld_w, const float* bias, void* y, long ld_y,
                        float* y32, long ld_y32, int B, int H, int W, int Cpad, int dtype, hipStream_t stream);
int mireg_thin_conv_dgrad(const void* dy, long ld_dy, const void* w, long ld_w, void* dx, long ld_dx, int accumulate,
                          int B, int H, int W, int Cpad, int dtype, hipStream_t stream);
int mireg_thin_conv_wgrad_tiles(int B, int H, int W, int Cpad, int dtype, int* pixels_per_lane);
int mireg_thin_conv_wgrad(const void* x, long ld_x, const void* dy, long ld_dy, float* slab, int ntiles, int B, int H,
                          int W, int Cpad, int dtype, hipStream_t stream);
/* GEMM formulation of the same heads at the fine levels (the host runs it for >= 16k pixels): the FWD pack [2][9*Cpad] is
 * [18][Cpad] row-major, so z[p][co*9+tap] = x[p] . w[co][tap] is a 1x1 mireg_conv_gemm to 18 columns that reads x once;
 * mireg_thin_shift_sum: y[p][co] = bias[co] + sum_tap z[p + tap - (1,1)][co*9+tap]  (z fp32 [B*H*W][ld_z >= 18]);
 * mireg_thin_gather18:  dz[q][co*9+tap] = dy[q - tap + (1,1)][co] (zero outside), the dy operand of a 1x1 mireg_conv_wgrad
 * whose slab [z][18][Cpad] is the heads' standard slab [z][2][9*Cpad]. */
int mireg_thin_shift_sum(const float* z, long ld_z, const float* bias, void* y, long ld_y, float* y32, long ld_y32, int B, int H,
                         int W, int dtype, hipStream_t stream);
int mireg_thin_gather18(const void* dy, long ld_dy, void* dz, long ld_dz, int B, int H, int W, int dtype, hipStream_t stream);

/* ---- the 2 -> 2 channel flow / feature upsamplers ConvTranspose2d(2, 2, 4, 2, 1) (FlowNetS/FlowNetS.py:37-40,
 * flownet2/networks/FlowNetC.py:53-56, PWC/models/PWCNet.py upfeat / deconv): pixel-parallel kernels on the fp32 master
 * weight w = the module's weight [2][2][4][4] (no packs).  x_coarse / dx_coarse: NHWC (B, Hc, Wc, ld_c), y_fine / g_fine:
 * (B, 2Hc, 2Wc, ld_f), two channels each.  bwd_weights writes mireg_tiny_deconv_blocks(B,Hc,Wc) partial slabs in the standard
 * layout [blk][2][16*Cpad] (pad slots untouched: hand in zeroed memory), summed by mireg_wgrad_reduce / mireg_unpack_wgrad. */
int mireg_tiny_deconv_blocks(int B, int Hc, int Wc);
int mireg_tiny_deconv_fwd(const void* x_coarse, long ld_c, const float* w, const float* bias, void* y_fine, long ld_f, float* y32,
                          long ld_y32, int B, int Hc, int Wc, int dtype, hipStream_t stream);   /* y_fine and / or an fp32 copy y32 */
/* dx_coarse = conv_s2(g_fine, w) [+ add_nchw, a planar fp32 (B, 2, Hc, Wc) term such as the loss gradient of that flow; may be
 * NULL] [+ dx_coarse when accumulate] */
int mireg_tiny_deconv_bwd_data(const void* g_fine, long ld_f, const float* w, void* dx_coarse, long ld_c, int accumulate,
                               const float* add_nchw, int B, int Hc, int Wc, int dtype, hipStream_t stream);
int mireg_tiny_deconv_bwd_weights(const void* g_fine, long ld_f, const void* x_coarse, long ld_c, float* slab, int nblocks, int Cpad,
                                  int B, int Hc, int Wc, int dtype, hipStream_t stream);

/* ---- 7x7 / stride 2 / pad 3 input convolutions with 1 or 2 input channels and 64 outputs, bf16 (FlowNetS conv1
 * `FlowNetS/FlowNetS.py:18`, FlowNetC's siamese conv1 `flownet2/networks/FlowNetC.py:20`): K re-ordered to (ky, kx, ci)
 * so the input patch is staged once per 8x16 output tile.  x = NHWC bf16 input (channels 0..Ci-1 real), w = the
 * standard FWD pack [64][49*Cpad], y / dy = NHWC bf16 [..][64]; wgrad writes mireg_stem_conv_blocks(B,H,W) partial
 * slabs [blk][64][ld_w] in the standard layout (pad slots are never written: hand in zeroed memory). */
int mireg_stem_conv_blocks(int B, int H, int W);
int mireg_stem_conv_fwd(const void* x, long ld_x, const void* w, long ld_w, int Cpad, const float* bias, float slope,
                        void* y, long ld_y, int B, int H, int W, int Ci, int Co, hipStream_t stream);
int mireg_stem_conv_wgrad(const void* x, long ld_x, const void* dy, long ld_dy, float* slab, long ld_w, int Cpad,
                          int nblocks, int B, int H, int W, int Ci, int Co, hipStream_t stream);

/* ---- packed-domain optimizer for the convolution weights (train.py:55-57,129 zero_grad/backward/step) ----
 * The torch-layout gradient is never materialised on the training path:
 *   mireg_wgrad_reduce : g[co][k] = sum_{z<nsplit} slab[z][co][k]       (fixed order; jobs with nsplit == 0 own no units)
 *   mireg_adam_pack    : torch.optim.Adam step on p/m/v (torch layout [Co][Ci][taps], fp32) with the gradient read
 *                        from the packed g[co][tap*Cpad+ci], then F[co][tap*Cpad+ci] = (dtype)p (pad slots zero).
 * g of every layer lives in one flat buffer, so under data parallelism it is what RCCL all-reduces. */
typedef struct mireg_wopt_job {
  const float* slab; long slab_stride; int nsplit;
  float* g;
  float* p; float* m; float* v;
  void* F;
  int Co, Ci, taps, Cpad; long ld;
  int runit0;     /* first reduce block: blocks = ceil(Co*ld / 256), 0 blocks when nsplit == 0 */
  int unit0;      /* first optimizer block: blocks = Co * ceil(Cpad/64) */
} mireg_wopt_job;
int mireg_wgrad_reduce(const mireg_wopt_job* jobs_dev, int njobs, int total_runits, hipStream_t stream);
/* max_taps = largest kd*kh*kw in the table (<= 125: sizes the kernel's LDS tile); tick != 0 increments *step_dev first (as mireg_adam_step does) */
int mireg_adam_pack(const mireg_wopt_job* jobs_dev, int njobs, int total_units, int max_taps, int* step_dev, int tick,
                    float lr, float beta1, float beta2, float eps, float grad_scale, int dtype, hipStream_t stream);

/* ---- K1 tail: BatchNorm2d (batch statistics in train mode) + LeakyReLU, FlowNetS/util.py:17-30 ---- */
/* y = raw convolution output [M][ld_y]; out = lrelu(bn(y)); ss = [scale | shift | mean | rstd] (4*C floats,
 * kept for the backward pass); partial = workspace of MIREG_BN_MAX_BLOCKS * 2 * C floats (two-stage,
 * deterministic reduction); running stats are updated in place when training (momentum, unbiased var). */
#define MIREG_BN_MAX_BLOCKS 512
int mireg_bn_forward(const void* y, long ld_y, void* out, long ld_o, long M, int C, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                     int training, float slope, float* partial, float* ss, int dtype, hipStream_t stream);
/* da = gradient wrt the activated output; dy = gradient wrt the raw convolution output; red = 2*C floats */
int mireg_bn_backward(const void* y, long ld_y, const void* da, long ld_da, void* dy, long ld_dy, const float* ss,
                      float* partial, float* red, float* dgamma, float* dbeta, int acc_param_grads, long M,
                      int C, float slope, int dtype, hipStream_t stream);
int mireg_lrelu_bwd(void* g, long ld_g, const void* a, long ld_a, long M, int C, float slope, int dtype,
                    hipStream_t stream);

/* ---- dtype / layout staging ---------------------------------------------------------------- */
int mireg_cast_from_f32(void* dst, long ld_d, const float* src, long ld_s, long M, int C, float alpha, float beta,
                        int dtype, hipStream_t stream);
int mireg_cast_to_f32(float* dst, long ld_d, const void* src, long ld_s, long M, int C, float alpha, float beta,
                      int dtype, hipStream_t stream);
/* x[:, c0:c0+nc] of an NCHW fp32 batch (train.py:44-46 hands NCHW) -> NHWC rows with pixel stride ld */
int mireg_nchw_to_nhwc(const float* src, void* dst, int B, int Ctot, int c0, int nc, long HW, long ld, int dtype,
                       hipStream_t stream);
/* clear a table of 16-byte aligned device buffers in one launch (the gradient accumulators PWCNet.py's autograd would
 * allocate fresh per backward); unit0 = first 16 KiB block of the job, units = ceil(bytes / 16384) */
typedef struct mireg_zero_job { void* p; long bytes; int unit0; int pad_; } mireg_zero_job;
int mireg_zero_many(const mireg_zero_job* jobs_dev, int njobs, int total_units, hipStream_t stream);

/* bias gradients: out[c] (+)= sum_m g[m][c], fixed summation order on the vector path (C % 8 == 0 for bf16 / % 4 fp32).
 * workspace: MIREG_COLSUM_MAX_SEGMENTS * C floats, needed once M > 8192 (row segments + finalize); may be NULL. */
#define MIREG_COLSUM_MAX_SEGMENTS 64
int mireg_colsum(const void* g, long ld, long M, int C, float* out, int accumulate, float* workspace, int dtype,
                 hipStream_t stream);

/* ---- K20: Adam exactly as train.py:129 builds it (betas .9/.999, eps = lrMin = 1e-4) ---------- */
typedef struct mireg_adam_job { float* p; const float* g; float* m; float* v; long n; } mireg_adam_job;
/* tick != 0: *step_dev is incremented on device first (graph-replayable; exactly one launch per optimizer step
 * ticks, the others of the same step pass 0); grad_scale multiplies every gradient (1/world_size after a sum
 * all-reduce). */
int mireg_adam_step(const mireg_adam_job* jobs_dev, int njobs, int* step_dev, int tick, float lr, float beta1,
                    float beta2, float eps, float grad_scale, hipStream_t stream);


/* ---- K7/K8: cost volume == external correlation_package.Correlation(pad=md, k=1, md, s1=1, s2) ------- */
/* call sites: flownet2/networks/FlowNetC.py:31,88 (md 20, s2 2); PWC/models/PWCNet.py:69,200-259 (md 4, s2 1);
 * same values as FlowNetS/util.py:58-72.  NHWC views; out channel (dy+R)*D+(dx+R), R = md/s2, D = 2R+1;
 * out = lrelu((1/c_norm) * <f1[y,x,:], f2[y+s2*dy, x+s2*dx, :]>, slope), zeros outside f2.  C = channels
 * walked (padded to 8 bf16 / 4 fp32 with zeros), c_norm = true channel count of the 1/C factor.
 * The third-party op has no source in the reference tree: parity unpinned, published definition. */
int mireg_correlation_fwd(const void* f1, long ld1, const void* f2, long ld2, void* out, long ldo, int B, int H,
                          int W, int C, int c_norm, int max_displacement, int stride2, float slope, int dtype,
                          hipStream_t stream);
/* backward of the cost volume: g = d loss / d out with the LeakyReLU backward already applied (NHWC, D*D channels);
 * df1 / df2 may be NULL; accumulate flags add into existing gradients (f1 usually has other consumers). */
int mireg_correlation_bwd(const void* g, long ldg, const void* f1, long ld1, const void* f2, long ld2, void* df1,
                          long ldd1, void* df2, long ldd2, int B, int H, int W, int C, int c_norm,
                          int max_displacement, int stride2, int accumulate1, int accumulate2, int dtype,
                          hipStream_t stream);
/* ---- K10: PWCDCNet.warp, PWC/models/PWCNet.py:143-179 (flow fp32 [pix][ldf], pre-scaled by flow_scale) ---- */
int mireg_pwc_warp_fwd(const void* x, long ldx, const float* flow, long ldf, float flow_scale, void* out, long ldo,
                       int B, int H, int W, int C, int dtype, hipStream_t stream);
/* concat staging for producers that cannot write in place (PWCNet.py:217: cat(corr, c1, up_flow, up_feat)) */
int mireg_copy_channels(const void* src, long ld_s, void* dst, long ld_d, long M, int C, int accumulate, int dtype,
                        hipStream_t stream);
/* backward of the PWC warp: dx32 (fp32 NHWC scratch, zeroed by the caller) and dflow (fp32 [pix][lddf], zeroed by the
 * caller) receive atomic adds; the validity mask carries no gradient (it is overwritten in place in the reference). */
int mireg_pwc_warp_bwd(const void* x, long ldx, const float* flow, long ldf, float flow_scale, const void* g, long ldg,
                       float* dx32, long lddx, float* dflow, long lddf, int B, int H, int W, int C, int dtype,
                       hipStream_t stream);
/* The same gradients without fp32 scatter atomics (bit-reproducible): taps are bucketed by source pixel (integer atomics, scan,
 * fill) and every source sums its bucket in ascending (destination, tap) order.  dx32 is OVERWRITTEN (all B*H*W*C values), dflow is
 * added onto (hand in zeros).  Workspaces (caller-owned, P = B*H*W): ws_cnt P ints, ZERO on entry and zero again on exit; ws_off
 * P + 1 + ceil(P/2048) ints; ws_entries 4*P 8-byte records.  C <= 512. */
int mireg_pwc_warp_bwd_det(const void* x, long ldx, const float* flow, long ldf, float flow_scale, const void* g, long ldg,
                           float* dx32, long lddx, float* dflow, long lddf, int* ws_cnt, int* ws_off, void* ws_entries,
                           int B, int H, int W, int C, int dtype, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MIREG_H_ */
