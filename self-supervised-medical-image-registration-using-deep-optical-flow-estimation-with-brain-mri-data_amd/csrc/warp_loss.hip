// warp_loss.hip -- HBM-bound tail of the registration hot path for gfx950:
//   K6  bilinear resize (F.interpolate)                 FlowNetS/FlowNetS.py:82, models.py:258, loss.py:11,54
//   K9  stn flow-warp (grid_sample, zeros, align=True)  models.py:256-268
//   K11 Charbonnier photometric partial sums            loss.py:9-14,33-35
//   K12 global-NCC partial sums (5 moments)             loss.py:52-64
//   K13 smoothness stencil                              loss.py:23-30
//   a12 OFEloss finalisation on device (float64)        loss.py:66-84
//   K14 seg round/clip, K15 Dice counters               models.py:286, utils.py:72-91
// All kernels are streaming kernels: one pass over the pixels, coalesced
// 16-byte accesses where the layout allows, per-block reduction (wave
// shuffles -> LDS) and ONE double atomic per moment per block.
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

#pragma clang fp contract(off)   // keep the reference's mul-then-add order in coordinate maths

namespace {

// flat index -> (sample, pixel, row, column).  The launches here cover a few million pixels: 32-bit divisions (a dozen instructions) instead
// of two 64-bit ones (a hundred each) whenever the index fits, which it always does below 2^32 pixels per call.
__device__ __forceinline__ void split_pixel(long i, long npix, int w, int& b, long& pix, int& y, int& x) {
  if (((unsigned long)i >> 32) == 0 && ((unsigned long)npix >> 32) == 0) {
    const unsigned iu = (unsigned)i, nu = (unsigned)npix;
    const unsigned bu = iu / nu, pu = iu - bu * nu, yu = pu / (unsigned)w;
    b = (int)bu; pix = (long)pu; y = (int)yu; x = (int)(pu - yu * (unsigned)w);
  } else {
    b = (int)(i / npix);
    pix = i - (long)b * npix;
    y = (int)(pix / w);
    x = (int)(pix - (long)y * w);
  }
}

constexpr int kThreads = 256;
constexpr int kSlots = MIREG_SUM_SLOTS;   // moment tables are replicated: block b adds into slot b % kSlots (no hot line)

// ---------------------------------------------------------------------------------------------
// bilinear source index, identical op order to ATen's area_pixel_compute_source_index
__device__ __forceinline__ void src_coord(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
  float s = align ? (float)dst * scale : fmaxf(((float)dst + 0.5f) * scale - 0.5f, 0.f);
  i0 = min((int)s, in_size - 1);
  i1 = min(i0 + 1, in_size - 1);
  l1 = s - (float)i0;
}

__global__ void __launch_bounds__(kThreads)
resize_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int H, int W, int h, int w,
                  long isn, long isc, long isp, long osn, long osc, long osp, float sy, float sx, int align) {
  const long total = (long)N * C * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r0, r1, r2;
    const long q0 = fast_divmod(i, w, r0), q1 = fast_divmod(q0, h, r1);
    const int x = (int)r0, y = (int)r1, n = (int)fast_divmod(q1, C, r2), c = (int)r2;
    int y0, y1, x0, x1;
    float ly, lx;
    src_coord(y, sy, align, H, y0, y1, ly);
    src_coord(x, sx, align, W, x0, x1, lx);
    const float* p = in + n * isn + c * isc;
    const float v00 = p[((long)y0 * W + x0) * isp], v01 = p[((long)y0 * W + x1) * isp];
    const float v10 = p[((long)y1 * W + x0) * isp], v11 = p[((long)y1 * W + x1) * isp];
    const float top = v00 * (1.f - lx) + v01 * lx;
    const float bot = v10 * (1.f - lx) + v11 * lx;
    out[n * osn + c * osc + ((long)y * w + x) * osp] = top * (1.f - ly) + bot * ly;
  }
}

// Backward of the resize in gather form (deterministic, no atomics): each INPUT pixel scans the
// output indices that can reference it.  beta: 0 = overwrite gin, 1 = accumulate into gin.
__device__ __forceinline__ void out_range(int i, float scale, int align, int out_size, int& lo, int& hi) {
  // outputs whose floor(src) is i-1 or i (plus one of slack on both sides)
  float inv = scale > 0.f ? 1.f / scale : 0.f;
  float a = align ? ((float)i - 1.f) * inv : (((float)i - 1.f) + 0.5f) * inv - 0.5f;
  float b = align ? ((float)i + 1.f) * inv : (((float)i + 1.f) + 0.5f) * inv - 0.5f;
  lo = max((int)floorf(a) - 1, 0);
  hi = min((int)ceilf(b) + 1, out_size - 1);
  if (scale <= 0.f) { lo = 0; hi = out_size - 1; }
}

__global__ void __launch_bounds__(kThreads)
resize_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, int N, int C, int H, int W, int h, int w,
                  long isn, long isc, long isp, long osn, long osc, long osp, float sy, float sx, int align,
                  float beta) {
  const long total = (long)N * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r0, r1, r2;
    const long q0 = fast_divmod(i, W, r0), q1 = fast_divmod(q0, H, r1);
    const int X = (int)r0, Y = (int)r1, n = (int)fast_divmod(q1, C, r2), c = (int)r2;
    int ylo, yhi, xlo, xhi;
    out_range(Y, sy, align, h, ylo, yhi);
    out_range(X, sx, align, w, xlo, xhi);
    const float* g = gout + n * osn + c * osc;
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
      int y0, y1; float ly;
      src_coord(y, sy, align, H, y0, y1, ly);
      const float wy = (y0 == Y ? 1.f - ly : 0.f) + (y1 == Y ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int x = xlo; x <= xhi; ++x) {
        int x0, x1; float lx;
        src_coord(x, sx, align, W, x0, x1, lx);
        const float wx = (x0 == X ? 1.f - lx : 0.f) + (x1 == X ? lx : 0.f);
        if (wx != 0.f) acc += wy * wx * g[((long)y * w + x) * osp];
      }
    }
    float* dst = gin + n * isn + c * isc + ((long)Y * W + X) * isp;
    *dst = beta != 0.f ? *dst * beta + acc : acc;
  }
}

// ---------------------------------------------------------------------------------------------
// stn sampling coordinate, same fp32 op order as the reference + ATen unnormalize (align_corners=True)
__device__ __forceinline__ float stn_coord(float pix, float disp, float two_over, float sizem1) {
  const float g = (disp + pix) * two_over - 1.f;
  return ((g + 1.f) / 2.f) * sizem1;
}

struct Taps {
  int x0, y0;
  float wx1, wy1;
};

__device__ __forceinline__ float tap(const float* __restrict__ img, int x, int y, int w, int h) {
  return (x >= 0 && x < w && y >= 0 && y < h) ? img[(long)y * w + x] : 0.f;
}

// One thread = VEC consecutive pixels of one image row.  frame / fixed / warped are planar
// (B,C,h,w); flow is addressed through (sb, sc, sp) so NCHW (sp=1) and the conv engine's
// interleaved NHWC (sc=1, sp=2) both stream coalesced.
template <int VEC, bool SUMS>
__global__ void __launch_bounds__(kThreads)
stn_warp_fwd_kernel(const float* __restrict__ flow, long fsb, long fsc, long fsp,
                    const float* __restrict__ frame, const float* __restrict__ fixed, float* __restrict__ warped,
                    double* __restrict__ sums, int B, int C, int h, int w) {
  __shared__ float red[6 * (kThreads / 64)];
  const long npix = (long)h * w;
  const long groups = (long)B * npix / VEC;
  const float two_w = (float)(2.0 / (double)w), two_h = (float)(2.0 / (double)h);
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long gidx = (long)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += (long)gridDim.x * blockDim.x) {
    const long p0 = gidx * VEC;
    int b, y, x;
    long pix;
    split_pixel(p0, npix, w, b, pix, y, x);
    float u[VEC], v[VEC];
    const float* fu = flow + b * fsb + pix * fsp;
    if (VEC == 4 && fsp == 1) {
      const float4 a = *reinterpret_cast<const float4*>(fu);
      const float4 c = *reinterpret_cast<const float4*>(fu + fsc);
      u[0] = a.x; u[1 % VEC] = a.y; u[2 % VEC] = a.z; u[3 % VEC] = a.w;
      v[0] = c.x; v[1 % VEC] = c.y; v[2 % VEC] = c.z; v[3 % VEC] = c.w;
    } else if (VEC == 4 && fsp == 2 && fsc == 1) {
      const float4 a = *reinterpret_cast<const float4*>(fu);
      const float4 c = *reinterpret_cast<const float4*>(fu + 4);
      u[0] = a.x; v[0] = a.y; u[1 % VEC] = a.z; v[1 % VEC] = a.w;
      u[2 % VEC] = c.x; v[2 % VEC] = c.y; u[3 % VEC] = c.z; v[3 % VEC] = c.w;
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { u[j] = fu[j * fsp]; v[j] = fu[j * fsp + fsc]; }
    }
    Taps t[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float px = stn_coord((float)(x + j), u[j], two_w, (float)(w - 1));
      const float py = stn_coord((float)y, v[j], two_h, (float)(h - 1));
      const float fx = floorf(px), fy = floorf(py);
      t[j].x0 = (int)fx; t[j].y0 = (int)fy; t[j].wx1 = px - fx; t[j].wy1 = py - fy;
    }
    for (int c = 0; c < C; ++c) {
      const float* img = frame + ((long)b * C + c) * npix;
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float nw = tap(img, t[j].x0, t[j].y0, w, h), ne = tap(img, t[j].x0 + 1, t[j].y0, w, h);
        const float sw = tap(img, t[j].x0, t[j].y0 + 1, w, h), se = tap(img, t[j].x0 + 1, t[j].y0 + 1, w, h);
        const float wx0 = 1.f - t[j].wx1, wy0 = 1.f - t[j].wy1;
        o[j] = nw * (wx0 * wy0) + ne * (t[j].wx1 * wy0) + sw * (wx0 * t[j].wy1) + se * (t[j].wx1 * t[j].wy1);
      }
      float* dst = warped + ((long)b * C + c) * npix + pix;
      if (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
      else dst[0] = o[0];
      if (SUMS && c == 0) {
        float f[VEC];
        const float* fp = fixed + (long)b * npix + pix;
        if (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(fp); f[0] = q.x; f[1 % VEC] = q.y; f[2 % VEC] = q.z; f[3 % VEC] = q.w; }
        else f[0] = fp[0];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          acc[0] += o[j]; acc[1] += f[j]; acc[2] += o[j] * f[j]; acc[3] += o[j] * o[j]; acc[4] += f[j] * f[j];
          acc[5] += charb(f[j] - o[j]);
        }
      }
    }
  }
  if (SUMS) {
    block_sum<6>(acc, red);
    if (threadIdx.x == 0) {
      double* dst = sums + (blockIdx.x % kSlots) * 8;
#pragma unroll
      for (int i = 0; i < 6; ++i) atomicAdd(&dst[i], (double)acc[i]);
    }
  }
}

// d(warped)/d(flow): gflow[b,0] = sum_c gout * dO/dpx * (w-1)/w ; gflow[b,1] likewise in y.
// beta = 0 overwrites gflow, beta = 1 accumulates (flows also feed the smoothness term / upsamplers).
__global__ void __launch_bounds__(kThreads)
stn_warp_bwd_kernel(const float* __restrict__ flow, long fsb, long fsc, long fsp,
                    const float* __restrict__ frame, const float* __restrict__ gout,
                    float* __restrict__ gflow, long gsb, long gsc, long gsp, float beta, int B, int C, int h, int w) {
  const long npix = (long)h * w, total = (long)B * npix;
  const float two_w = (float)(2.0 / (double)w), two_h = (float)(2.0 / (double)h);
  const float kx = (float)(w - 1) / (float)w * 1.f, ky = (float)(h - 1) / (float)h * 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b, y, x;
    long pix;
    split_pixel(i, npix, w, b, pix, y, x);
    const float* fu = flow + b * fsb + pix * fsp;
    const float px = stn_coord((float)x, fu[0], two_w, (float)(w - 1));
    const float py = stn_coord((float)y, fu[fsc], two_h, (float)(h - 1));
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = px - fx, wy1 = py - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float gx = 0.f, gy = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* img = frame + ((long)b * C + c) * npix;
      const float go = gout[((long)b * C + c) * npix + pix];
      const float nw = tap(img, x0, y0, w, h), ne = tap(img, x0 + 1, y0, w, h);
      const float sw = tap(img, x0, y0 + 1, w, h), se = tap(img, x0 + 1, y0 + 1, w, h);
      gx += go * ((ne - nw) * wy0 + (se - sw) * wy1);
      gy += go * ((sw - nw) * wx0 + (se - ne) * wx1);
    }
    float* g = gflow + b * gsb + pix * gsp;
    gx *= kx; gy *= ky;
    if (beta != 0.f) { g[0] = g[0] * beta + gx; g[gsc] = g[gsc] * beta + gy; }
    else { g[0] = gx; g[gsc] = gy; }
  }
}

// ---------------------------------------------------------------------------------------------
// Stand-alone moment pass (used when warp and loss are not fused): x = warped, y = fixed (same size)
__global__ void __launch_bounds__(kThreads)
loss_partials_kernel(const float* __restrict__ warped, const float* __restrict__ fixed, double* __restrict__ sums, long n) {
  __shared__ float red[6 * (kThreads / 64)];
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(warped)[i];
    const float4 f = reinterpret_cast<const float4*>(fixed)[i];
    const float xs[4] = {a.x, a.y, a.z, a.w}, ys[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0] += xs[j]; acc[1] += ys[j]; acc[2] += xs[j] * ys[j]; acc[3] += xs[j] * xs[j]; acc[4] += ys[j] * ys[j];
      acc[5] += charb(ys[j] - xs[j]);
    }
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = warped[i], y = fixed[i];
    acc[0] += x; acc[1] += y; acc[2] += x * y; acc[3] += x * x; acc[4] += y * y; acc[5] += charb(y - x);
  }
  block_sum<6>(acc, red);
  if (threadIdx.x == 0) {
    double* dst = sums + (blockIdx.x % kSlots) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&dst[i], (double)acc[i]);
  }
}

// gwarped = -cp*charb'(y-x) - k1*(y-my) + k2*(x-mx); coef = {cp,k1,k2,mx,my,cs,..} on device
__global__ void __launch_bounds__(kThreads)
loss_bwd_kernel(const float* __restrict__ warped, const float* __restrict__ fixed, const float* __restrict__ coef,
                float* __restrict__ gwarped, long n) {
  const float cp = coef[0], k1 = coef[1], k2 = coef[2], mx = coef[3], my = coef[4];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = warped[i], y = fixed[i];
    gwarped[i] = -cp * charb_grad(y - x) - k1 * (y - my) + k2 * (x - mx);
  }
}

// smoothness: sum over (b,c,y,x) of charb(f - f_down) + charb(f - f_right), neighbours outside = 0
__global__ void __launch_bounds__(kThreads)
smooth_fwd_kernel(const float* __restrict__ flow, long fsb, long fsc, long fsp, double* __restrict__ sum, int B, int h, int w) {
  __shared__ float red[kThreads / 64];
  const long npix = (long)h * w, total = (long)B * npix;
  float acc[1] = {0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b, y, x;
    long pix;
    split_pixel(i, npix, w, b, pix, y, x);
    const float* f = flow + b * fsb + pix * fsp;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float v = f[c * fsc];
      const float dn = (y + 1 < h) ? f[c * fsc + (long)w * fsp] : 0.f;
      const float rt = (x + 1 < w) ? f[c * fsc + fsp] : 0.f;
      acc[0] += charb(v - dn) + charb(v - rt);
    }
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) atomicAdd(sum + (blockIdx.x % kSlots) * 8, (double)acc[0]);
}

__global__ void __launch_bounds__(kThreads)
smooth_bwd_kernel(const float* __restrict__ flow, long fsb, long fsc, long fsp, const float* __restrict__ coef,
                  float* __restrict__ gflow, long gsb, long gsc, long gsp, float beta, int B, int h, int w) {
  const float cs = coef[5];
  const long npix = (long)h * w, total = (long)B * npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b, y, x;
    long pix;
    split_pixel(i, npix, w, b, pix, y, x);
    const float* f = flow + b * fsb + pix * fsp;
    float* g = gflow + b * gsb + pix * gsp;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float v = f[c * fsc];
      const float dn = (y + 1 < h) ? f[c * fsc + (long)w * fsp] : 0.f;
      const float rt = (x + 1 < w) ? f[c * fsc + fsp] : 0.f;
      float acc = charb_grad(v - dn) + charb_grad(v - rt);
      if (y > 0) acc -= charb_grad(f[c * fsc - (long)w * fsp] - v);
      if (x > 0) acc -= charb_grad(f[c * fsc - fsp] - v);
      acc *= cs;
      g[c * gsc] = beta != 0.f ? g[c * gsc] * beta + acc : acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// OFEloss finalisation, float64 like the reference (its weights are a float64 tensor).
// sums: [n][8] = {Sx, Sy, Sxy, Sxx, Syy, Scharb, Ssmooth, -};  npix[i] = B*h_i*w_i
__device__ __forceinline__ void gather_slots(const double* __restrict__ sums, int scale, double (&q)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = 0.0;
  for (int sl = 0; sl < kSlots; ++sl)
#pragma unroll
    for (int i = 0; i < 7; ++i) q[i] += sums[((long)scale * kSlots + sl) * 8 + i];
}

__device__ __forceinline__ void ncc_moments(const double* s, double N, double& sxx, double& syy, double& sxy,
                                            double& mx, double& my, bool& degenerate) {
  mx = s[0] / N; my = s[1] / N;
  sxy = s[2] - s[0] * s[1] / N;
  sxx = s[3] - s[0] * s[0] / N;
  syy = s[4] - s[1] * s[1] / N;
  // reference guard: centred tensor identically zero (loss.py:58-60) -> corr := 1
  degenerate = !(sxx > 1e-12 * fmax(s[3], 1e-300)) || !(syy > 1e-12 * fmax(s[4], 1e-300));
}

__global__ void ofe_finalize_kernel(const double* __restrict__ sums, const long* __restrict__ npix, int n, int B,
                                    double lamb, double gamma, double zeta, double* __restrict__ out4) {
  // slot gather in parallel (lane = scale x moment, same slot order as gather_slots), closed form on lane 0
  __shared__ double qs[16][8];
  for (int t = threadIdx.x; t < n * 8; t += blockDim.x) {
    const int i = t >> 3, k = t & 7;
    double v = 0.0;
    if (k < 7)
      for (int sl = 0; sl < kSlots; ++sl) v += sums[((long)i * kSlots + sl) * 8 + k];
    qs[i][k] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double p = 0, c = 0, s = 0;
  for (int i = 0; i < n; ++i) {
    const double* q = qs[i];
    const double wgt = 0.05 * (double)(i + 1);
    double sxx, syy, sxy, mx, my; bool deg;
    ncc_moments(q, (double)npix[i], sxx, syy, sxy, mx, my, deg);
    const double corr = deg ? 1.0 : (1.0 / B) * sxy / (sqrt(sxx) * sqrt(syy));
    p += wgt * (q[5] / B);
    c += wgt * (1.0 - corr);
    s += wgt * (q[6] / 2.0 / B);
  }
  p = 1.0 / n * gamma * p; c = 1.0 / n * zeta * c; s = 1.0 / n * lamb * s;
  out4[0] = p; out4[1] = c; out4[2] = s; out4[3] = p + s + c;
}

// upstream grads g4 = d/d(p, c, s, total) -> per-scale coefficients coef[n][8] (fp32)
__global__ void ofe_bwd_coef_kernel(const double* __restrict__ sums, const long* __restrict__ npix, int n, int B,
                                    double lamb, double gamma, double zeta, const double* __restrict__ g4,
                                    float* __restrict__ coef) {
  const int i = threadIdx.x;
  if (blockIdx.x != 0 || i >= n) return;
  double q[8];
  gather_slots(sums, i, q);
  const double wgt = 0.05 * (double)(i + 1);
  const double gp = g4[0] + g4[3], gc = g4[1] + g4[3], gs = g4[2] + g4[3];
  double sxx, syy, sxy, mx, my; bool deg;
  ncc_moments(q, (double)npix[i], sxx, syy, sxy, mx, my, deg);
  float* o = coef + 8 * i;
  o[0] = (float)(gp * gamma / n * wgt / B);
  const double wc = gc * zeta / n * wgt / B;
  o[1] = deg ? 0.f : (float)(wc / sqrt(sxx * syy));
  o[2] = deg ? 0.f : (float)(wc * sxy / (sxx * sqrt(sxx * syy)));
  o[3] = (float)mx; o[4] = (float)my;
  o[5] = (float)(gs * lamb / n * wgt / (2.0 * B));
  o[6] = 0.f; o[7] = 0.f;
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
seg_round_kernel(const float* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = fminf(fmaxf(rintf(in[i]), 0.f), 3.f);   // rintf = round-half-even = numpy rint
}

// per-sample Dice over labels 1..3: counts[b][9] = {|A_l|, |B_l|, |A_l & B_l|}_l ; one block row per sample
__global__ void __launch_bounds__(kThreads)
dice_counts_kernel(const float* __restrict__ y_true, const float* __restrict__ y_pred, float* __restrict__ counts, long n) {
  __shared__ float red[9 * (kThreads / 64)];
  const int b = blockIdx.y;
  const float* t = y_true + (long)b * n;
  const float* p = y_pred + (long)b * n;
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float a = t[i], q = p[i];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const float ia = a == (float)(l + 1) ? 1.f : 0.f, iq = q == (float)(l + 1) ? 1.f : 0.f;
      acc[3 * l] += ia; acc[3 * l + 1] += iq; acc[3 * l + 2] += ia * iq;
    }
  }
  block_sum<9>(acc, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) atomicAdd(&counts[b * 9 + i], acc[i]);
  }
}

__global__ void dice_finalize_kernel(const float* __restrict__ counts, float* __restrict__ dice, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float m = 0.f;
  for (int l = 0; l < 3; ++l) {
    const float* c = counts + b * 9 + 3 * l;
    m += (2.f * c[2]) / (c[0] + c[1]);    // 0/0 -> NaN exactly like the reference
  }
  dice[b] = m / 3.f;
}

// F.grid_sample(vol, F.affine_grid(theta, vol.size())) with both torch defaults (align_corners=False, trilinear,
// zeros) for planar volumes (B,C,D,H,W): reference models.py:187-188 (affmodel).  theta: (B,3,4) row-major.
__global__ void __launch_bounds__(kThreads)
affine_sample3d_kernel(const float* __restrict__ vol, const float* __restrict__ theta, float* __restrict__ out,
                       int B, int C, int D, int H, int W) {
  const long nvox = (long)D * H * W, total = (long)B * nvox;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long v, r0, r1;
    const int b = (int)fast_divmod(i, nvox, v);
    const long q0 = fast_divmod(v, W, r0);
    const int x = (int)r0, z = (int)fast_divmod(q0, H, r1), y = (int)r1;
    const float* t = theta + b * 12;
    // affine_grid base coordinates (align_corners=False): (2i + 1)/n - 1
    const float bx = (2.f * x + 1.f) / (float)W - 1.f, by = (2.f * y + 1.f) / (float)H - 1.f, bz = (2.f * z + 1.f) / (float)D - 1.f;
    const float gx = t[0] * bx + t[1] * by + t[2] * bz + t[3];
    const float gy = t[4] * bx + t[5] * by + t[6] * bz + t[7];
    const float gz = t[8] * bx + t[9] * by + t[10] * bz + t[11];
    const float px = ((gx + 1.f) * (float)W - 1.f) / 2.f, py = ((gy + 1.f) * (float)H - 1.f) / 2.f, pz = ((gz + 1.f) * (float)D - 1.f) / 2.f;
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx1 = px - fx, wy1 = py - fy, wz1 = pz - fz;
    for (int c = 0; c < C; ++c) {
      const float* src = vol + ((long)b * C + c) * nvox;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int xi = x0 + (k & 1), yi = y0 + ((k >> 1) & 1), zi = z0 + (k >> 2);
        if (xi < 0 || xi >= W || yi < 0 || yi >= H || zi < 0 || zi >= D) continue;
        const float wgt = ((k & 1) ? wx1 : 1.f - wx1) * (((k >> 1) & 1) ? wy1 : 1.f - wy1) * ((k >> 2) ? wz1 : 1.f - wz1);
        acc += src[((long)zi * H + yi) * W + xi] * wgt;
      }
      out[((long)b * C + c) * nvox + v] = acc;
    }
  }
}

// d(loss)/d(theta) of the sampler above (autograd of models.py:187-188 with respect to `para`): per voxel the trilinear
// weights' derivatives give d out / d (px, py, pz); px = ((gx + 1) W - 1)/2 and gx = theta[0, :] . (bx, by, bz, 1), so
// gtheta[b][r][:] = sum_voxels (d/d p_r) * (extent_r / 2) * (bx, by, bz, 1).  Two stages, fixed order (deterministic):
// block (blk, b) writes partial[b][blk][12]; the finalize adds them in double.
__global__ void __launch_bounds__(kThreads)
affine_sample3d_bwd_kernel(const float* __restrict__ vol, const float* __restrict__ theta, const float* __restrict__ gout,
                           float* __restrict__ partial, int C, int D, int H, int W) {
  __shared__ float red[12 * (kThreads / 64)];
  const int b = blockIdx.y;
  const long nvox = (long)D * H * W;
  const float* t = theta + b * 12;
  const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5], t6 = t[6], t7 = t[7], t8 = t[8], t9 = t[9],
              t10 = t[10], t11 = t[11];
  float acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (long)gridDim.x * blockDim.x) {
    const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long)W * H));
    const float bx = (2.f * x + 1.f) / (float)W - 1.f, by = (2.f * y + 1.f) / (float)H - 1.f, bz = (2.f * z + 1.f) / (float)D - 1.f;
    const float gx = t0 * bx + t1 * by + t2 * bz + t3;
    const float gy = t4 * bx + t5 * by + t6 * bz + t7;
    const float gz = t8 * bx + t9 * by + t10 * bz + t11;
    const float px = ((gx + 1.f) * (float)W - 1.f) / 2.f, py = ((gy + 1.f) * (float)H - 1.f) / 2.f, pz = ((gz + 1.f) * (float)D - 1.f) / 2.f;
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx1 = px - fx, wy1 = py - fy, wz1 = pz - fz;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* src = vol + ((long)b * C + c) * nvox;
      const float g = gout[((long)b * C + c) * nvox + v];
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int xi = x0 + (k & 1), yi = y0 + ((k >> 1) & 1), zi = z0 + (k >> 2);
        if (xi < 0 || xi >= W || yi < 0 || yi >= H || zi < 0 || zi >= D) continue;
        const float s = src[((long)zi * H + yi) * W + xi];
        const float wx = (k & 1) ? wx1 : 1.f - wx1, wy = ((k >> 1) & 1) ? wy1 : 1.f - wy1, wz = (k >> 2) ? wz1 : 1.f - wz1;
        ax += ((k & 1) ? s : -s) * wy * wz;
        ay += (((k >> 1) & 1) ? s : -s) * wx * wz;
        az += ((k >> 2) ? s : -s) * wx * wy;
      }
      dx += g * ax; dy += g * ay; dz += g * az;
    }
    dx *= 0.5f * (float)W; dy *= 0.5f * (float)H; dz *= 0.5f * (float)D;
    acc[0] += dx * bx; acc[1] += dx * by; acc[2] += dx * bz; acc[3] += dx;
    acc[4] += dy * bx; acc[5] += dy * by; acc[6] += dy * bz; acc[7] += dy;
    acc[8] += dz * bx; acc[9] += dz * by; acc[10] += dz * bz; acc[11] += dz;
  }
  block_sum<12>(acc, red);
  if (threadIdx.x == 0) {
    float* dst = partial + ((long)b * gridDim.x + blockIdx.x) * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) dst[i] = acc[i];
  }
}

__global__ void affine_sample3d_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ gtheta, int nblk,
                                                    int accumulate) {
  const int b = blockIdx.x, j = threadIdx.x;
  if (j >= 12) return;
  double s = 0.0;
  for (int k = 0; k < nblk; ++k) s += (double)partial[((long)b * nblk + k) * 12 + j];
  gtheta[b * 12 + j] = (float)s + (accumulate ? gtheta[b * 12 + j] : 0.f);
}

// =============================================================================================
// Multi-scale fused tail of the training step (reference train.py:50-52 + loss.py:66-84 over the six
// flow scales): ONE launch per phase instead of one per scale and op.  Job i = scale i; virtual block
// ranges [blk0, blk0 + ceil(B*h*w / 256)) are laid out back to back.  Per-element arithmetic is the same
// as the per-scale kernels above (same op order), only the partition of the moment sums differs.
//   resize : moving_r = bilinear(moving, align_corners=True) [models.py:258], fixed_r = bilinear(fixed, False) [loss.py:11,54]
//   fwd    : warped = stn(flow, moving_r); moments of (warped, fixed_r); smoothness sum of flow
//   bwd    : gflow = d(total)/d(flow) = warp-backward(d loss/d warped) + smoothness gradient
// =============================================================================================
__device__ __forceinline__ int tail_find(const mireg_tail_job* __restrict__ jobs, int n, int blk) {
  int lo = 0;
  for (int i = 1; i < n; ++i) if (jobs[i].blk0 <= blk) lo = i;
  return lo;
}

__global__ void __launch_bounds__(kThreads)
tail_resize_kernel(const mireg_tail_job* __restrict__ jobs, int n, const float* __restrict__ x, int B, int H, int W) {
  const mireg_tail_job j = jobs[tail_find(jobs, n, blockIdx.x)];
  const long npix = (long)j.h * j.w;
  for (int chunk = 0; chunk < MIREG_TAIL_PIXELS_PER_BLOCK / kThreads; ++chunk) {
  const long i = ((long)(blockIdx.x - j.blk0) * (MIREG_TAIL_PIXELS_PER_BLOCK / kThreads) + chunk) * kThreads + threadIdx.x;
  if (i >= (long)B * npix) return;
  int b, y, xx;
  long pix;
  split_pixel(i, npix, j.w, b, pix, y, xx);
  const float* fixed = x + (long)b * 2 * H * W;
  const float* moving = fixed + (long)H * W;
#pragma unroll
  for (int align = 0; align < 2; ++align) {
    const float sy = align ? (j.h > 1 ? (float)(H - 1) / (float)(j.h - 1) : 0.f) : (float)H / (float)j.h;
    const float sx = align ? (j.w > 1 ? (float)(W - 1) / (float)(j.w - 1) : 0.f) : (float)W / (float)j.w;
    int y0, y1, x0, x1;
    float ly, lx;
    src_coord(y, sy, align, H, y0, y1, ly);
    src_coord(xx, sx, align, W, x0, x1, lx);
    const float* p = align ? moving : fixed;
    const float v00 = p[(long)y0 * W + x0], v01 = p[(long)y0 * W + x1], v10 = p[(long)y1 * W + x0], v11 = p[(long)y1 * W + x1];
    const float top = v00 * (1.f - lx) + v01 * lx, bot = v10 * (1.f - lx) + v11 * lx;
    (align ? j.moving_r : j.fixed_r)[i] = top * (1.f - ly) + bot * ly;
  }
  }
}

__global__ void __launch_bounds__(kThreads)
tail_fwd_kernel(const mireg_tail_job* __restrict__ jobs, int n, int B) {
  __shared__ float red[7 * (kThreads / 64)];
  const mireg_tail_job j = jobs[tail_find(jobs, n, blockIdx.x)];
  const int h = j.h, w = j.w;
  const long npix = (long)h * w;
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int chunk = 0; chunk < MIREG_TAIL_PIXELS_PER_BLOCK / kThreads; ++chunk) {     // fewer blocks = fewer f64 atomics per moment
    const long i = ((long)(blockIdx.x - j.blk0) * (MIREG_TAIL_PIXELS_PER_BLOCK / kThreads) + chunk) * kThreads + threadIdx.x;
    if (i >= (long)B * npix) continue;
    int b, y, x;
    long pix;
    split_pixel(i, npix, w, b, pix, y, x);
    const float* f = j.flow + b * j.fsb + pix * j.fsp;
    const float u = f[0], v = f[j.fsc];
    const float px = stn_coord((float)x, u, (float)(2.0 / (double)w), (float)(w - 1));
    const float py = stn_coord((float)y, v, (float)(2.0 / (double)h), (float)(h - 1));
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = px - fx, wy1 = py - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float* img = j.moving_r + (long)b * npix;
    const float nw = tap(img, x0, y0, w, h), ne = tap(img, x0 + 1, y0, w, h);
    const float sw = tap(img, x0, y0 + 1, w, h), se = tap(img, x0 + 1, y0 + 1, w, h);
    const float o = nw * (wx0 * wy0) + ne * (wx1 * wy0) + sw * (wx0 * wy1) + se * (wx1 * wy1);
    j.warped[i] = o;
    const float fv = j.fixed_r[i];
    acc[0] += o; acc[1] += fv; acc[2] += o * fv; acc[3] += o * o; acc[4] += fv * fv; acc[5] += charb(fv - o);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float val = f[c * j.fsc];
      const float dn = (y + 1 < h) ? f[c * j.fsc + (long)w * j.fsp] : 0.f;
      const float rt = (x + 1 < w) ? f[c * j.fsc + j.fsp] : 0.f;
      acc[6] += charb(val - dn) + charb(val - rt);
    }
  }
  block_sum<7>(acc, red);
  if (threadIdx.x == 0) {
    double* dst = j.sums + (blockIdx.x % kSlots) * 8;
#pragma unroll
    for (int k = 0; k < 7; ++k) atomicAdd(&dst[k], (double)acc[k]);
  }
}

__global__ void __launch_bounds__(kThreads)
tail_bwd_kernel(const mireg_tail_job* __restrict__ jobs, int n, int B) {
  const mireg_tail_job j = jobs[tail_find(jobs, n, blockIdx.x)];
  const int h = j.h, w = j.w;
  const long npix = (long)h * w;
  const float cp = j.coef[0], k1 = j.coef[1], k2 = j.coef[2], mx = j.coef[3], my = j.coef[4], cs = j.coef[5];
  for (int chunk = 0; chunk < MIREG_TAIL_PIXELS_PER_BLOCK / kThreads; ++chunk) {
  const long i = ((long)(blockIdx.x - j.blk0) * (MIREG_TAIL_PIXELS_PER_BLOCK / kThreads) + chunk) * kThreads + threadIdx.x;
  if (i >= (long)B * npix) return;
  int b, y, x;
  long pix;
  split_pixel(i, npix, w, b, pix, y, x);
  const float* f = j.flow + b * j.fsb + pix * j.fsp;
  // d loss / d warped (loss.py:38-50 photometric + NCC through the scale's moments)
  const float xv = j.warped[i], yv = j.fixed_r[i];
  const float go = -cp * charb_grad(yv - xv) - k1 * (yv - my) + k2 * (xv - mx);
  // through the sampler
  const float px = stn_coord((float)x, f[0], (float)(2.0 / (double)w), (float)(w - 1));
  const float py = stn_coord((float)y, f[j.fsc], (float)(2.0 / (double)h), (float)(h - 1));
  const float fx = floorf(px), fy = floorf(py);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = px - fx, wy1 = py - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const float* img = j.moving_r + (long)b * npix;
  const float nw = tap(img, x0, y0, w, h), ne = tap(img, x0 + 1, y0, w, h);
  const float sw = tap(img, x0, y0 + 1, w, h), se = tap(img, x0 + 1, y0 + 1, w, h);
  float g2[2];
  g2[0] = 0.f + go * ((ne - nw) * wy0 + (se - sw) * wy1);
  g2[1] = 0.f + go * ((sw - nw) * wx0 + (se - ne) * wx1);
  g2[0] *= (float)(w - 1) / (float)w * 1.f;
  g2[1] *= (float)(h - 1) / (float)h * 1.f;
  float* g = j.gflow + b * j.gsb + pix * j.gsp;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float val = f[c * j.fsc];
    const float dn = (y + 1 < h) ? f[c * j.fsc + (long)w * j.fsp] : 0.f;
    const float rt = (x + 1 < w) ? f[c * j.fsc + j.fsp] : 0.f;
    float acc = charb_grad(val - dn) + charb_grad(val - rt);
    if (y > 0) acc -= charb_grad(f[c * j.fsc - (long)w * j.fsp] - val);
    if (x > 0) acc -= charb_grad(f[c * j.fsc - j.fsp] - val);
    acc *= cs;
    g[c * j.gsc] = g2[c] * 1.f + acc;
  }
  }
}

inline int grid_for(long work, int cap = 2048) {
  long g = (work + kThreads - 1) / kThreads;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int mireg_resize_bilinear_fwd(const float* in, float* out, int N, int C, int H, int W, int h, int w,
                              long isn, long isc, long isp, long osn, long osc, long osp, int align_corners,
                              hipStream_t stream) {
  MIREG_CHECK_ARG(in && out && N > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  const float sy = align_corners ? (h > 1 ? (float)(H - 1) / (float)(h - 1) : 0.f) : (float)H / (float)h;
  const float sx = align_corners ? (w > 1 ? (float)(W - 1) / (float)(w - 1) : 0.f) : (float)W / (float)w;
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(grid_for((long)N * C * h * w)), dim3(kThreads), 0, stream, in, out, N, C,
                     H, W, h, w, isn, isc, isp, osn, osc, osp, sy, sx, align_corners);
  MIREG_LAUNCH_RET();
}

int mireg_resize_bilinear_bwd(const float* gout, float* gin, int N, int C, int H, int W, int h, int w,
                              long isn, long isc, long isp, long osn, long osc, long osp, int align_corners,
                              float beta, hipStream_t stream) {
  MIREG_CHECK_ARG(gout && gin && N > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  const float sy = align_corners ? (h > 1 ? (float)(H - 1) / (float)(h - 1) : 0.f) : (float)H / (float)h;
  const float sx = align_corners ? (w > 1 ? (float)(W - 1) / (float)(w - 1) : 0.f) : (float)W / (float)w;
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(grid_for((long)N * C * H * W)), dim3(kThreads), 0, stream, gout, gin, N,
                     C, H, W, h, w, isn, isc, isp, osn, osc, osp, sy, sx, align_corners, beta);
  MIREG_LAUNCH_RET();
}

int mireg_stn_warp_fwd(const float* flow, long fsb, long fsc, long fsp, const float* frame, const float* fixed,
                       float* warped, double* sums, int B, int C, int h, int w, hipStream_t stream) {
  MIREG_CHECK_ARG(flow && frame && warped && B > 0 && C > 0 && h > 0 && w > 0);
  MIREG_CHECK_ARG((fixed == nullptr) == (sums == nullptr));
  const bool vec = (w % 4 == 0) && ((fsp == 1 && fsc % 4 == 0 && fsb % 4 == 0) || (fsp == 2 && fsc == 1 && fsb % 4 == 0)) &&
                   ((uintptr_t)flow % 16 == 0) && ((uintptr_t)warped % 16 == 0) && (!fixed || (uintptr_t)fixed % 16 == 0);
  const long npix = (long)B * h * w;
  if (vec) {
    const int g = grid_for(npix / 4);
    if (sums) hipLaunchKernelGGL((stn_warp_fwd_kernel<4, true>), dim3(g), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, frame, fixed, warped, sums, B, C, h, w);
    else hipLaunchKernelGGL((stn_warp_fwd_kernel<4, false>), dim3(g), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, frame, fixed, warped, sums, B, C, h, w);
  } else {
    const int g = grid_for(npix);
    if (sums) hipLaunchKernelGGL((stn_warp_fwd_kernel<1, true>), dim3(g), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, frame, fixed, warped, sums, B, C, h, w);
    else hipLaunchKernelGGL((stn_warp_fwd_kernel<1, false>), dim3(g), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, frame, fixed, warped, sums, B, C, h, w);
  }
  MIREG_LAUNCH_RET();
}

int mireg_stn_warp_bwd(const float* flow, long fsb, long fsc, long fsp, const float* frame, const float* gout,
                       float* gflow, long gsb, long gsc, long gsp, float beta, int B, int C, int h, int w,
                       hipStream_t stream) {
  MIREG_CHECK_ARG(flow && frame && gout && gflow && B > 0 && C > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(stn_warp_bwd_kernel, dim3(grid_for((long)B * h * w)), dim3(kThreads), 0, stream, flow, fsb, fsc,
                     fsp, frame, gout, gflow, gsb, gsc, gsp, beta, B, C, h, w);
  MIREG_LAUNCH_RET();
}

int mireg_loss_partials(const float* warped, const float* fixed, double* sums, long n, hipStream_t stream) {
  MIREG_CHECK_ARG(warped && fixed && sums && n > 0);
  MIREG_CHECK_ARG((uintptr_t)warped % 16 == 0 && (uintptr_t)fixed % 16 == 0);
  hipLaunchKernelGGL(loss_partials_kernel, dim3(grid_for(n / 4 + 1)), dim3(kThreads), 0, stream, warped, fixed, sums, n);
  MIREG_LAUNCH_RET();
}

int mireg_loss_bwd(const float* warped, const float* fixed, const float* coef, float* gwarped, long n,
                   hipStream_t stream) {
  MIREG_CHECK_ARG(warped && fixed && coef && gwarped && n > 0);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, warped, fixed, coef, gwarped, n);
  MIREG_LAUNCH_RET();
}

int mireg_smoothness_fwd(const float* flow, long fsb, long fsc, long fsp, double* sum, int B, int h, int w,
                         hipStream_t stream) {
  MIREG_CHECK_ARG(flow && sum && B > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(smooth_fwd_kernel, dim3(grid_for((long)B * h * w)), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, sum, B, h, w);
  MIREG_LAUNCH_RET();
}

int mireg_smoothness_bwd(const float* flow, long fsb, long fsc, long fsp, const float* coef, float* gflow, long gsb,
                         long gsc, long gsp, float beta, int B, int h, int w, hipStream_t stream) {
  MIREG_CHECK_ARG(flow && coef && gflow && B > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(smooth_bwd_kernel, dim3(grid_for((long)B * h * w)), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, coef, gflow, gsb, gsc, gsp, beta, B, h, w);
  MIREG_LAUNCH_RET();
}

int mireg_ofe_finalize(const double* sums, const long* npix, int n, int B, double lamb_da, double gamma, double zeta,
                       double* out4, hipStream_t stream) {
  MIREG_CHECK_ARG(sums && npix && out4 && n > 0 && n <= 16 && B > 0);
  hipLaunchKernelGGL(ofe_finalize_kernel, dim3(1), dim3(128), 0, stream, sums, npix, n, B, lamb_da, gamma, zeta, out4);
  MIREG_LAUNCH_RET();
}

int mireg_ofe_bwd_coef(const double* sums, const long* npix, int n, int B, double lamb_da, double gamma, double zeta,
                       const double* g4, float* coef, hipStream_t stream) {
  MIREG_CHECK_ARG(sums && npix && g4 && coef && n > 0 && n <= 16 && B > 0);
  hipLaunchKernelGGL(ofe_bwd_coef_kernel, dim3(1), dim3(64), 0, stream, sums, npix, n, B, lamb_da, gamma, zeta, g4, coef);
  MIREG_LAUNCH_RET();
}

int mireg_affine_sample3d(const float* vol, const float* theta, float* out, int B, int C, int D, int H, int W,
                          hipStream_t stream) {
  MIREG_CHECK_ARG(vol && theta && out && B > 0 && C > 0 && D > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(affine_sample3d_kernel, dim3(grid_for((long)B * D * H * W)), dim3(kThreads), 0, stream, vol, theta, out, B, C, D, H, W);
  MIREG_LAUNCH_RET();
}

int mireg_affine_sample3d_bwd(const float* vol, const float* theta, const float* gout, float* gtheta, float* workspace,
                              int accumulate, int B, int C, int D, int H, int W, hipStream_t stream) {
  if (!vol || !theta || !gout || !gtheta || !workspace || B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return MIREG_ERR_ARG;
  const long nvox = (long)D * H * W;
  int nblk = (int)((nvox + kThreads - 1) / kThreads);
  if (nblk > MIREG_AFFINE3D_BWD_BLOCKS) nblk = MIREG_AFFINE3D_BWD_BLOCKS;
  hipLaunchKernelGGL(affine_sample3d_bwd_kernel, dim3(nblk, B), dim3(kThreads), 0, stream, vol, theta, gout, workspace, C, D, H, W);
  hipLaunchKernelGGL(affine_sample3d_bwd_finalize_kernel, dim3(B), dim3(64), 0, stream, workspace, gtheta, nblk, accumulate);
  return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
}

int mireg_tail_resize(const mireg_tail_job* jobs_dev, int njobs, int total_blocks, const float* x, int B, int H, int W,
                      hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 16 && total_blocks > 0 && x && B > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(tail_resize_kernel, dim3(total_blocks), dim3(kThreads), 0, stream, jobs_dev, njobs, x, B, H, W);
  MIREG_LAUNCH_RET();
}

int mireg_tail_fwd(const mireg_tail_job* jobs_dev, int njobs, int total_blocks, int B, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 16 && total_blocks > 0 && B > 0);
  hipLaunchKernelGGL(tail_fwd_kernel, dim3(total_blocks), dim3(kThreads), 0, stream, jobs_dev, njobs, B);
  MIREG_LAUNCH_RET();
}

int mireg_tail_bwd(const mireg_tail_job* jobs_dev, int njobs, int total_blocks, int B, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 16 && total_blocks > 0 && B > 0);
  hipLaunchKernelGGL(tail_bwd_kernel, dim3(total_blocks), dim3(kThreads), 0, stream, jobs_dev, njobs, B);
  MIREG_LAUNCH_RET();
}

int mireg_seg_round(const float* in, float* out, long n, hipStream_t stream) {
  MIREG_CHECK_ARG(in && out && n > 0);
  hipLaunchKernelGGL(seg_round_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, in, out, n);
  MIREG_LAUNCH_RET();
}

int mireg_dice(const float* y_true, const float* y_pred, float* counts, float* dice, int B, long n, hipStream_t stream) {
  MIREG_CHECK_ARG(y_true && y_pred && counts && dice && B > 0 && n > 0);
  if (hipMemsetAsync(counts, 0, sizeof(float) * 9 * B, stream) != hipSuccess) return MIREG_ERR_LAUNCH;
  hipLaunchKernelGGL(dice_counts_kernel, dim3(grid_for(n, 64), B), dim3(kThreads), 0, stream, y_true, y_pred, counts, n);
  hipLaunchKernelGGL(dice_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, counts, dice, B);
  MIREG_LAUNCH_RET();
}

}  // extern "C"
