// flownet2_ops.hip -- the two custom layers of the FlowNet2 stack besides Correlation, and its nearest upsampling
// (kernels K16-K18; call sites reference flownet2/models.py:40-88,136-180).  The CUDA sources of
// NVIDIA/flownet2-pytorch's resample2d_package / channelnorm_package are not part of the reference tree and no version is
// pinned (README.md:18-19): these follow the published definitions of those layers -- parity unpinned, see DESIGN.md section 2.
//   Resample2d  out[b,c,y,x] = bilinear(src[b,c], x + flow_x, y + flow_y), the four tap INDICES clamped to the border
//               (not grid_sample's zero padding); gradients to src (scatter, fp32 atomics) and to flow (gather)
//   ChannelNorm out[b,0,y,x] = sqrt(sum_c in[b,c,y,x]^2); backward g * in / (out + 1e-9)
//   Upsample    nn.Upsample(scale_factor=k, mode='nearest') (the bilinear variant is mireg_resize_bilinear_*)
// Planar fp32 tensors (B,C,H,W), as at the reference's call sites.
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

namespace {

constexpr int kThreads = 256;

inline int grid_for(long work, int cap = 4096) {
  long g = (work + kThreads - 1) / kThreads;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

struct Tap {
  int xL, xR, yT, yB;
  float a, b;                                            // fractional parts along x / y
};

__device__ __forceinline__ Tap tap_for(int x, int y, float dx, float dy, int H, int W) {
  const float xf = (float)x + dx, yf = (float)y + dy;
  const float fx = floorf(xf), fy = floorf(yf);
  Tap t;
  t.a = xf - fx; t.b = yf - fy;
  t.xL = max(min((int)fx, W - 1), 0); t.xR = max(min((int)fx + 1, W - 1), 0);
  t.yT = max(min((int)fy, H - 1), 0); t.yB = max(min((int)fy + 1, H - 1), 0);
  return t;
}

__global__ void __launch_bounds__(kThreads)
resample2d_fwd_kernel(const float* __restrict__ src, const float* __restrict__ flow, float* __restrict__ out, int B, int C, int H, int W) {
  const long npix = (long)H * W, total = (long)B * npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npix);
    const long p = i - (long)b * npix;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    const Tap t = tap_for(x, y, flow[((long)b * 2) * npix + p], flow[((long)b * 2 + 1) * npix + p], H, W);
    for (int c = 0; c < C; ++c) {
      const float* s = src + ((long)b * C + c) * npix;
      out[((long)b * C + c) * npix + p] = (1.f - t.a) * (1.f - t.b) * s[(long)t.yT * W + t.xL] + t.a * (1.f - t.b) * s[(long)t.yT * W + t.xR] +
                                          (1.f - t.a) * t.b * s[(long)t.yB * W + t.xL] + t.a * t.b * s[(long)t.yB * W + t.xR];
    }
  }
}

// gflow (gather, one thread per pixel) and gsrc (scatter with fp32 atomics; gsrc must be zeroed by the caller or hold the
// value to accumulate into).  Either output may be null.
__global__ void __launch_bounds__(kThreads)
resample2d_bwd_kernel(const float* __restrict__ src, const float* __restrict__ flow, const float* __restrict__ gout,
                      float* __restrict__ gsrc, float* __restrict__ gflow, int B, int C, int H, int W) {
  const long npix = (long)H * W, total = (long)B * npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npix);
    const long p = i - (long)b * npix;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    const Tap t = tap_for(x, y, flow[((long)b * 2) * npix + p], flow[((long)b * 2 + 1) * npix + p], H, W);
    float gx = 0.f, gy = 0.f;
    for (int c = 0; c < C; ++c) {
      const long base = ((long)b * C + c) * npix;
      const float g = gout[base + p];
      const float* s = src + base;
      const float tl = s[(long)t.yT * W + t.xL], tr = s[(long)t.yT * W + t.xR], bl = s[(long)t.yB * W + t.xL], br = s[(long)t.yB * W + t.xR];
      gx += g * ((1.f - t.b) * (tr - tl) + t.b * (br - bl));
      gy += g * ((1.f - t.a) * (bl - tl) + t.a * (br - tr));
      if (gsrc) {
        float* d = gsrc + base;
        atomicAdd(&d[(long)t.yT * W + t.xL], g * (1.f - t.a) * (1.f - t.b));
        atomicAdd(&d[(long)t.yT * W + t.xR], g * t.a * (1.f - t.b));
        atomicAdd(&d[(long)t.yB * W + t.xL], g * (1.f - t.a) * t.b);
        atomicAdd(&d[(long)t.yB * W + t.xR], g * t.a * t.b);
      }
    }
    if (gflow) { gflow[((long)b * 2) * npix + p] = gx; gflow[((long)b * 2 + 1) * npix + p] = gy; }
  }
}

__global__ void __launch_bounds__(kThreads)
channelnorm_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, long npix) {
  const long total = (long)B * npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npix);
    const long p = i - (long)b * npix;
    float s = 0.f;
    for (int c = 0; c < C; ++c) { const float v = in[((long)b * C + c) * npix + p]; s += v * v; }
    out[i] = sqrtf(s);
  }
}

__global__ void __launch_bounds__(kThreads)
channelnorm_bwd_kernel(const float* __restrict__ in, const float* __restrict__ out, const float* __restrict__ gout,
                       float* __restrict__ gin, int B, int C, long npix) {
  const long total = (long)B * C * npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % npix;
    const int b = (int)(i / (npix * C));
    const long o = (long)b * npix + p;
    gin[i] = gout[o] * in[i] / (out[o] + 1e-9f);
  }
}

// nearest: src index = floor(dst / k) (ATen nearest with an integer scale factor)
__global__ void __launch_bounds__(kThreads)
upsample_nearest_kernel(const float* __restrict__ in, float* __restrict__ out, long NC, int H, int W, int k, int bwd) {
  const int Ho = H * k, Wo = W * k;
  if (!bwd) {
    const long total = NC * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
      const long n = i / ((long)Wo * Ho);
      out[i] = in[(n * H + y / k) * W + x / k];
    }
  } else {                                               // in = gradient of the big tensor, out = gradient of the small one
    const long total = NC * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int x = (int)(i % W), y = (int)((i / W) % H);
      const long n = i / ((long)W * H);
      float s = 0.f;
      for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) s += in[(n * Ho + y * k + dy) * Wo + x * k + dx];
      out[i] = s;
    }
  }
}

}  // namespace

extern "C" {

int mireg_resample2d_fwd(const float* src, const float* flow, float* out, int B, int C, int H, int W, hipStream_t stream) {
  MIREG_CHECK_ARG(src && flow && out && B > 0 && C > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(resample2d_fwd_kernel, dim3(grid_for((long)B * H * W)), dim3(kThreads), 0, stream, src, flow, out, B, C, H, W);
  MIREG_LAUNCH_RET();
}

int mireg_resample2d_bwd(const float* src, const float* flow, const float* gout, float* gsrc, float* gflow, int B, int C, int H,
                         int W, hipStream_t stream) {
  MIREG_CHECK_ARG(src && flow && gout && (gsrc || gflow) && B > 0 && C > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(resample2d_bwd_kernel, dim3(grid_for((long)B * H * W)), dim3(kThreads), 0, stream, src, flow, gout, gsrc, gflow,
                     B, C, H, W);
  MIREG_LAUNCH_RET();
}

int mireg_channelnorm_fwd(const float* in, float* out, int B, int C, long npix, hipStream_t stream) {
  MIREG_CHECK_ARG(in && out && B > 0 && C > 0 && npix > 0);
  hipLaunchKernelGGL(channelnorm_fwd_kernel, dim3(grid_for((long)B * npix)), dim3(kThreads), 0, stream, in, out, B, C, npix);
  MIREG_LAUNCH_RET();
}

int mireg_channelnorm_bwd(const float* in, const float* out, const float* gout, float* gin, int B, int C, long npix,
                          hipStream_t stream) {
  MIREG_CHECK_ARG(in && out && gout && gin && B > 0 && C > 0 && npix > 0);
  hipLaunchKernelGGL(channelnorm_bwd_kernel, dim3(grid_for((long)B * C * npix)), dim3(kThreads), 0, stream, in, out, gout, gin, B, C, npix);
  MIREG_LAUNCH_RET();
}

int mireg_upsample_nearest(const float* in, float* out, long NC, int H, int W, int k, int backward, hipStream_t stream) {
  MIREG_CHECK_ARG(in && out && NC > 0 && H > 0 && W > 0 && k > 0);
  const long work = backward ? NC * H * W : NC * H * W * k * k;
  hipLaunchKernelGGL(upsample_nearest_kernel, dim3(grid_for(work)), dim3(kThreads), 0, stream, in, out, NC, H, W, k, backward);
  MIREG_LAUNCH_RET();
}

}  // extern "C"
