// metrics.hip -- per-sample evaluation metrics of the inference loop on device (reference inference.py:66-75 calls them once
// per sample and pulls every result to the host):
//   MSE  utils.py:41-42    mean((warped - fixed)^2)
//   PSNR utils.py:45-49    100 if mse < 1e-10 else 10*log10(1/mse)
//   CORR utils.py:58-59    Pearson correlation of the flattened images (torchmetrics.pearson_corrcoef)
//   MI   utils.py:52-55    sklearn.metrics.mutual_info_score of round(x*1500) labels (natural log)
//   SSIM inference.py:70-71 skimage.metrics.structural_similarity(a, b, data_range=1.0), defaults (7x7 uniform window)
// One launch covers the whole batch for the moment-based three; MI counts into a dense joint table per sample (bins^2 int32),
// sums log(N n_ij / (a_i b_j)) over the PIXELS (not the cells) and zeroes the touched counters again: three launches per batch.
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

namespace {

constexpr int kThreads = 256;

// sums[b][8] = {S a, S b, S ab, S aa, S bb, S (a-b)^2, -, -}; a = fixed, b = warped
__global__ void __launch_bounds__(kThreads)
pair_moments_kernel(const float* __restrict__ fixed, const float* __restrict__ warped, double* __restrict__ sums, long n) {
  __shared__ float red[6 * (kThreads / 64)];
  const int b = blockIdx.y;
  const float* f = fixed + (long)b * n;
  const float* w = warped + (long)b * n;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = f[i], y = w[i], d = y - x;
    acc[0] += x; acc[1] += y; acc[2] += x * y; acc[3] += x * x; acc[4] += y * y; acc[5] += d * d;
  }
  block_sum<6>(acc, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) atomicAdd(&sums[b * 8 + k], (double)acc[k]);
  }
}

// out[b][3] = {mse, psnr, pearson}
__global__ void pair_metrics_finalize_kernel(const double* __restrict__ sums, double* __restrict__ out, int B, long n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* s = sums + b * 8;
  const double N = (double)n;
  const double mse = s[5] / N;
  const double cov = s[2] - s[0] * s[1] / N, va = s[3] - s[0] * s[0] / N, vb = s[4] - s[1] * s[1] / N;
  out[b * 3 + 0] = mse;
  out[b * 3 + 1] = mse < 1.0e-10 ? 100.0 : 10.0 * log10(1.0 / mse);
  out[b * 3 + 2] = cov / sqrt(va * vb);                  // 0/0 -> NaN for a constant image, as the reference's dependency gives
}

__device__ __forceinline__ int label_of(float v, float scale, int bins) {
  const int q = (int)rintf(v * scale);                   // torch.round = round-half-even
  return min(max(q, 0), bins - 1);
}

// counter += 1 per lane, with the lanes of a wave that hit the same counter combined first: smooth images put thousands of
// pixels (the background above all) into one cell, and one-by-one atomics on a single address serialise in L2.  Two leader
// rounds catch the dominant keys; whatever is left goes one atomic per lane.
__device__ __forceinline__ void count_key(int* __restrict__ table, int key, bool active) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(active);
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (!todo) break;
    const int leader = __ffsll((long long)todo) - 1;
    const int lk = __shfl(key, leader, 64);
    const unsigned long long same = __ballot(active && key == lk) & todo;
    if (lane == leader) atomicAdd(&table[lk], (int)__popcll(same));
    todo &= ~same;
  }
  if ((todo >> lane) & 1ull) atomicAdd(&table[key], 1);
}

// pass 1: joint[b][a*bins + c] and the two marginals of every sample (blockIdx.y = sample)
__global__ void __launch_bounds__(kThreads)
mi_count_kernel(const float* __restrict__ fixed, const float* __restrict__ warped, int* __restrict__ joint, int* __restrict__ marg,
                long n, int bins, float scale) {
  const int s = blockIdx.y;
  const float* f = fixed + (long)s * n;
  const float* w = warped + (long)s * n;
  int* jt = joint + (long)s * bins * bins;
  int* mg = marg + (long)s * 2 * bins;
  const long span = (long)gridDim.x * blockDim.x, trips = (n + span - 1) / span;      // uniform trip count: ballots need every lane
  for (long t = 0; t < trips; ++t) {
    const long i = t * span + (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    const int a = ok ? label_of(f[i], scale, bins) : 0, c = ok ? label_of(w[i], scale, bins) : 0;
    count_key(jt, a * bins + c, ok);
    count_key(mg, a, ok);
    count_key(mg, bins + c, ok);
  }
}

// pass 2: MI = sum_ij (n_ij/N) log(N n_ij / (a_i b_j)) = (1/N) sum over PIXELS of log(N n_ij / (a_i b_j)) at the pixel's cell
// (sklearn.metrics.mutual_info_score, natural log) -- no scan of the dense table
__global__ void __launch_bounds__(kThreads)
mi_sum_kernel(const float* __restrict__ fixed, const float* __restrict__ warped, const int* __restrict__ joint,
              const int* __restrict__ marg, double* __restrict__ out, long n, int bins, float scale) {
  __shared__ double red[kThreads / 64];
  const int s = blockIdx.y;
  const float* f = fixed + (long)s * n;
  const float* w = warped + (long)s * n;
  const int* jt = joint + (long)s * bins * bins;
  const int* mg = marg + (long)s * 2 * bins;
  const double logN = log((double)n);
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int a = label_of(f[i], scale, bins), c = label_of(w[i], scale, bins);
    acc += log((double)jt[(long)a * bins + c]) + logN - log((double)mg[a]) - log((double)mg[bins + c]);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kThreads / 64; ++k) t += red[k];
    atomicAdd(out + s, t / (double)n);
  }
}

// pass 3: put the touched counters back to zero (the tables stay clean between calls; no dense memset)
__global__ void __launch_bounds__(kThreads)
mi_clear_kernel(const float* __restrict__ fixed, const float* __restrict__ warped, int* __restrict__ joint, int* __restrict__ marg,
                long n, int bins, float scale) {
  const int s = blockIdx.y;
  const float* f = fixed + (long)s * n;
  const float* w = warped + (long)s * n;
  int* jt = joint + (long)s * bins * bins;
  int* mg = marg + (long)s * 2 * bins;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int a = label_of(f[i], scale, bins), c = label_of(w[i], scale, bins);
    jt[(long)a * bins + c] = 0;
    mg[a] = 0;
    mg[bins + c] = 0;
  }
}


// SSIM as skimage.metrics.structural_similarity(a, b, data_range=R) computes it with its defaults (inference.py:70-71): 7x7
// uniform window, sample covariance (x 49/48), K1 = 0.01, K2 = 0.03, mean of the SSIM map cropped by 3 pixels per side
// (only windows that lie fully inside the image count, so the filter's border mode never matters).  One thread per
// interior pixel; window sums in double, rounded to float like the float32 filter output, map arithmetic in float.
__global__ void __launch_bounds__(kThreads)
ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ out, int H, int W, int win, float c1,
            float c2) {
  __shared__ double red[kThreads / 64];
  const int s = blockIdx.y, pad = (win - 1) / 2;
  const float* x = a + (long)s * H * W;
  const float* y = b + (long)s * H * W;
  const int ih = H - 2 * pad, iw = W - 2 * pad;
  const long cnt = (long)ih * iw;
  const double np_ = (double)win * win;
  const float cov_norm = (float)(np_ / (np_ - 1.0));
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (long)gridDim.x * blockDim.x) {
    const int py = (int)(i / iw), px = (int)(i - (long)py * iw);      // window = rows py..py+win-1, cols px..px+win-1
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int dy = 0; dy < win; ++dy) {
      const float* rx = x + (long)(py + dy) * W + px;
      const float* ry = y + (long)(py + dy) * W + px;
      for (int dx = 0; dx < win; ++dx) {
        const double u = rx[dx], v = ry[dx];
        sx += u; sy += v; sxx += u * u; syy += v * v; sxy += u * v;
      }
    }
    const float ux = (float)(sx / np_), uy = (float)(sy / np_), uxx = (float)(sxx / np_), uyy = (float)(syy / np_), uxy = (float)(sxy / np_);
    const float vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
    const float A1 = 2.f * ux * uy + c1, A2 = 2.f * vxy + c2, B1 = ux * ux + uy * uy + c1, B2 = vx + vy + c2;
    acc += (double)((A1 * A2) / (B1 * B2));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kThreads / 64; ++k) t += red[k];
    atomicAdd(out + s, t / (double)cnt);
  }
}


// modified Hausdorff distance of two point sets (utils.py:187-199, Dubuisson & Jain): D = cdist(A, B);
// max(mean_j min_i D_ij, mean_i min_j D_ij).  Thread per point: nearest neighbour in the other set (sets are contour
// points, a few thousand each), then a one-block mean / max.  Points are (row, col) pairs, any float values.
__global__ void __launch_bounds__(kThreads)
mhd_nearest_kernel(const float* __restrict__ A, int nA, const float* __restrict__ Bp, int nB, float* __restrict__ work) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nA + nB) return;
  const bool fromA = i < nA;
  const float* me = fromA ? A + 2 * i : Bp + 2 * (i - nA);
  const float* other = fromA ? Bp : A;
  const int n = fromA ? nB : nA;
  const float y = me[0], x = me[1];
  float best = 3.4e38f;
  for (int j = 0; j < n; ++j) {
    const float dy = other[2 * j] - y, dx = other[2 * j + 1] - x;
    best = fminf(best, dy * dy + dx * dx);
  }
  work[i] = sqrtf(best);
}

__global__ void __launch_bounds__(kThreads)
mhd_finalize_kernel(const float* __restrict__ work, int nA, int nB, double* __restrict__ out) {
  __shared__ double red[2][kThreads / 64];
  double sa = 0.0, sb = 0.0;
  for (int i = threadIdx.x; i < nA; i += blockDim.x) sa += (double)work[i];
  for (int i = threadIdx.x; i < nB; i += blockDim.x) sb += (double)work[nA + i];
  sa = wave_sum(sa); sb = wave_sum(sb);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sa; red[1][threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int k = 0; k < kThreads / 64; ++k) { ta += red[0][k]; tb += red[1][k]; }
    const double rhd = ta / (double)nA, fhd = tb / (double)nB;      // rows of D belong to A, columns to B
    out[0] = fhd > rhd ? fhd : rhd;
  }
}


// ---- contour points of a label mask (utils.py:155-170: skimage.measure.find_contours(mask, 0.5), vstack, astype(int)).  For a binary
// mask the marching-squares vertices are the midpoints of the 4-neighbour pixel pairs whose mask values differ, (r, c + 0.5) and
// (r + 0.5, c); truncated to int they are (r, c).  One point per crossing, in raster order (row, then horizontal before vertical
// crossing per pixel): deterministic compaction = per-row counts, a single-block scan, per-row fill.  skimage additionally repeats the
// first vertex of every closed contour; that duplicate is not reproduced (skimage is absent from this image: parity unpinned).
__device__ __forceinline__ bool is_lab(const float* seg, int W, int r, int c, float lab) { return seg[(long)r * W + c] == lab; }

__global__ void __launch_bounds__(kThreads)
boundary_count_kernel(const float* __restrict__ seg, long sstride, int nmask, const float* __restrict__ labels, int H, int W, int* __restrict__ rowcnt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nmask * H) return;
  const int m = t / H, r = t - m * H;
  const float* sg = seg + (long)m * sstride;
  const float lab = labels[m];
  int n = 0;
  for (int c = 0; c < W; ++c) {
    const bool a = is_lab(sg, W, r, c, lab);
    if (c + 1 < W && a != is_lab(sg, W, r, c + 1, lab)) ++n;
    if (r + 1 < H && a != is_lab(sg, W, r + 1, c, lab)) ++n;
  }
  rowcnt[t] = n;
}
__global__ void __launch_bounds__(kThreads)
boundary_scan_kernel(const int* __restrict__ rowcnt, int H, int* __restrict__ rowoff, int* __restrict__ total) {   // one block per mask
  __shared__ int part[kThreads];
  const int m = blockIdx.x;
  const int per = (H + kThreads - 1) / kThreads, r0 = threadIdx.x * per, r1 = min(H, r0 + per);
  int s = 0;
  for (int r = r0; r < r1; ++r) s += rowcnt[m * H + r];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int acc = 0; for (int k = 0; k < kThreads; ++k) { const int v = part[k]; part[k] = acc; acc += v; } total[m] = acc; }
  __syncthreads();
  int acc = part[threadIdx.x];
  for (int r = r0; r < r1; ++r) { rowoff[m * H + r] = acc; acc += rowcnt[m * H + r]; }
}
__global__ void __launch_bounds__(kThreads)
boundary_fill_kernel(const float* __restrict__ seg, long sstride, int nmask, const float* __restrict__ labels, int H, int W,
                     const int* __restrict__ rowoff, float* __restrict__ pts, long pstride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nmask * H) return;
  const int m = t / H, r = t - m * H;
  const float* sg = seg + (long)m * sstride;
  const float lab = labels[m];
  float* o = pts + (long)m * pstride + 2L * rowoff[t];
  for (int c = 0; c < W; ++c) {
    const bool a = is_lab(sg, W, r, c, lab);
    if (c + 1 < W && a != is_lab(sg, W, r, c + 1, lab)) { o[0] = (float)r; o[1] = (float)c; o += 2; }
    if (r + 1 < H && a != is_lab(sg, W, r + 1, c, lab)) { o[0] = (float)r; o[1] = (float)c; o += 2; }
  }
}
// modified Hausdorff distance of point sets whose sizes live on the device (no host round trip between extraction and distance)
__global__ void __launch_bounds__(kThreads)
mhd_nearest_dev_kernel(const float* __restrict__ A, const float* __restrict__ Bp, long pstride, const int* __restrict__ cnt, int npair, int cap,
                       float* __restrict__ work) {                   // grid (blocks, pairs): pair p uses masks 2p (A) and 2p+1 (B)
  const int p = blockIdx.y;
  const int nA = min(cnt[2 * p], cap), nB = min(cnt[2 * p + 1], cap);
  const float* a = A + (long)(2 * p) * pstride;
  const float* b = Bp + (long)(2 * p + 1) * pstride;
  float* wk = work + (long)p * 2 * cap;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nA + nB; i += gridDim.x * blockDim.x) {
    const bool fromA = i < nA;
    const float* me = fromA ? a + 2 * i : b + 2 * (i - nA);
    const float* other = fromA ? b : a;
    const int n = fromA ? nB : nA;
    const float y = me[0], x = me[1];
    float best = 3.4e38f;
    for (int j = 0; j < n; ++j) { const float dy = other[2 * j] - y, dx = other[2 * j + 1] - x; best = fminf(best, dy * dy + dx * dx); }
    wk[fromA ? i : cap + (i - nA)] = sqrtf(best);
  }
}
__global__ void __launch_bounds__(kThreads)
mhd_finalize_dev_kernel(const float* __restrict__ work, const int* __restrict__ cnt, int npair, int cap, double* __restrict__ out) {
  __shared__ double red[2][kThreads / 64];
  __shared__ double acc_mean;
  if (threadIdx.x == 0) acc_mean = 0.0;
  __syncthreads();
  for (int p = 0; p < npair; ++p) {
    const int nA = min(cnt[2 * p], cap), nB = min(cnt[2 * p + 1], cap);
    const float* wk = work + (long)p * 2 * cap;
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < nA; i += blockDim.x) sa += (double)wk[i];
    for (int i = threadIdx.x; i < nB; i += blockDim.x) sb += (double)wk[cap + i];
    sa = wave_sum(sa); sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sa; red[1][threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double ta = 0.0, tb = 0.0;
      for (int k = 0; k < kThreads / 64; ++k) { ta += red[0][k]; tb += red[1][k]; }
      // an empty contour on either side: numpy's mean of an empty min() is nan in the reference; reported as nan here too
      const double rhd = nA > 0 && nB > 0 ? ta / (double)nA : __longlong_as_double(0x7ff8000000000000LL);
      const double fhd = nA > 0 && nB > 0 ? tb / (double)nB : __longlong_as_double(0x7ff8000000000000LL);
      out[1 + p] = fhd > rhd ? fhd : rhd;
      acc_mean += out[1 + p];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = acc_mean / (double)npair;
}

}  // namespace

extern "C" {

int mireg_pair_metrics(const float* fixed, const float* warped, double* sums, double* out, int B, long n, hipStream_t stream) {
  MIREG_CHECK_ARG(fixed && warped && sums && out && B > 0 && n > 0);
  if (hipMemsetAsync(sums, 0, (size_t)B * 8 * sizeof(double), stream) != hipSuccess) return MIREG_ERR_LAUNCH;
  long g = (n + kThreads * 4 - 1) / (kThreads * 4);
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(pair_moments_kernel, dim3((unsigned)g, B), dim3(kThreads), 0, stream, fixed, warped, sums, n);
  hipLaunchKernelGGL(pair_metrics_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, sums, out, B, n);
  MIREG_LAUNCH_RET();
}

int mireg_mutual_info(const float* fixed, const float* warped, int* joint, int* marg, double* out, int B, long n, int bins,
                      float scale, hipStream_t stream) {
  MIREG_CHECK_ARG(fixed && warped && joint && marg && out && B > 0 && n > 0 && bins > 1 && bins <= 8192);
  if (hipMemsetAsync(out, 0, (size_t)B * sizeof(double), stream) != hipSuccess) return MIREG_ERR_LAUNCH;
  long g = (n + kThreads - 1) / kThreads;
  if (g > 256) g = 256;
  const dim3 grid((unsigned)g, B);
  hipLaunchKernelGGL(mi_count_kernel, grid, dim3(kThreads), 0, stream, fixed, warped, joint, marg, n, bins, scale);
  hipLaunchKernelGGL(mi_sum_kernel, grid, dim3(kThreads), 0, stream, fixed, warped, joint, marg, out, n, bins, scale);
  hipLaunchKernelGGL(mi_clear_kernel, grid, dim3(kThreads), 0, stream, fixed, warped, joint, marg, n, bins, scale);
  MIREG_LAUNCH_RET();
}


int mireg_ssim(const float* a, const float* b, double* out, int B, int H, int W, int win_size, float data_range, hipStream_t stream) {
  MIREG_CHECK_ARG(a && b && out && B > 0 && win_size >= 3 && (win_size & 1) && H >= win_size && W >= win_size && data_range > 0.f);
  if (hipMemsetAsync(out, 0, (size_t)B * sizeof(double), stream) != hipSuccess) return MIREG_ERR_LAUNCH;
  const long cnt = (long)(H - win_size + 1) * (W - win_size + 1);
  long g = (cnt + kThreads - 1) / kThreads;
  if (g > 512) g = 512;
  const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
  hipLaunchKernelGGL(ssim_kernel, dim3((unsigned)g, B), dim3(kThreads), 0, stream, a, b, out, H, W, win_size, c1, c2);
  MIREG_LAUNCH_RET();
}


int mireg_modified_hausdorff(const float* A, int nA, const float* B, int nB, float* work, double* out, hipStream_t stream) {
  MIREG_CHECK_ARG(A && B && work && out && nA > 0 && nB > 0);
  hipLaunchKernelGGL(mhd_nearest_kernel, dim3((nA + nB + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, A, nA, B, nB, work);
  hipLaunchKernelGGL(mhd_finalize_kernel, dim3(1), dim3(kThreads), 0, stream, work, nA, nB, out);
  MIREG_LAUNCH_RET();
}


int mireg_boundary_points(const float* seg, long seg_stride, int nmask, const float* labels_dev, int H, int W, int* rowcnt, int* rowoff,
                          int* counts, float* points, long points_stride, hipStream_t stream) {
  MIREG_CHECK_ARG(seg && labels_dev && rowcnt && rowoff && counts && points && nmask > 0 && H > 0 && W > 0 && points_stride >= 4L * H * W);
  const int g = (nmask * H + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(boundary_count_kernel, dim3(g), dim3(kThreads), 0, stream, seg, seg_stride, nmask, labels_dev, H, W, rowcnt);
  hipLaunchKernelGGL(boundary_scan_kernel, dim3(nmask), dim3(kThreads), 0, stream, rowcnt, H, rowoff, counts);
  hipLaunchKernelGGL(boundary_fill_kernel, dim3(g), dim3(kThreads), 0, stream, seg, seg_stride, nmask, labels_dev, H, W, rowoff, points, points_stride);
  MIREG_LAUNCH_RET();
}

int mireg_hausdorff_pairs(const float* points, long points_stride, const int* counts, int npair, int cap, float* work, double* out,
                          hipStream_t stream) {
  MIREG_CHECK_ARG(points && counts && work && out && npair > 0 && cap > 0 && points_stride >= 2L * cap);
  int g = (2 * cap + kThreads - 1) / kThreads;
  g = g > 64 ? 64 : g;
  hipLaunchKernelGGL(mhd_nearest_dev_kernel, dim3(g, npair), dim3(kThreads), 0, stream, points, points, points_stride, counts, npair, cap, work);
  hipLaunchKernelGGL(mhd_finalize_dev_kernel, dim3(1), dim3(kThreads), 0, stream, work, counts, npair, cap, out);
  MIREG_LAUNCH_RET();
}

}  // extern "C"
