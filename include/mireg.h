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

/* ---- K9 (+K11/K12 partials): opticalFlowReg.stn, models.py:256-268 ----------------------- */
/* frame is the moving image ALREADY resized to (h,w) (planar B,C,h,w).  When fixed != NULL the
 * kernel also accumulates the six loss moments {Sx,Sy,Sxy,Sxx,Syy,Scharb} of (warped, fixed)
 * into sums[0..5] (doubles, caller zeroes them) -- loss.py:9-14, 52-64 fused into the warp. */
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
/* sums: [n][8] doubles {Sx,Sy,Sxy,Sxx,Syy,Scharb,Ssmooth,-}; npix[i] = B*h_i*w_i; out4 = (p,c,s,total)
 * as float64 (the reference returns float64 scalars, loss.py:71-73).  No host synchronisation:
 * the reference's two torch.equal() guards (loss.py:58-60) are evaluated on device. */
int mireg_ofe_finalize(const double* sums, const long* npix, int n, int B, double lamb_da, double gamma,
                       double zeta, double* out4, hipStream_t stream);
/* g4 = upstream d/d(p,c,s,total); coef: [n][8] floats {cp,k1,k2,mean_w,mean_f,cs,-,-} */
int mireg_ofe_bwd_coef(const double* sums, const long* npix, int n, int B, double lamb_da, double gamma,
                       double zeta, const double* g4, float* coef, hipStream_t stream);

/* ---- K14/K15: models.py:286 (rint + clip 0..3, on device), utils.py:72-91 (Dice) --------- */
int mireg_seg_round(const float* in, float* out, long n, hipStream_t stream);
/* counts: workspace B*9 floats; dice[b] = mean_l 2|A_l & B_l| / (|A_l| + |B_l|), l = 1..3 */
int mireg_dice(const float* y_true, const float* y_pred, float* counts, float* dice, int B, long n,
               hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MIREG_H_ */
