// augment.hip -- on-device elastic deformation of a batch: the step the reference's MONAI pipeline
// runs on the CPU per batch (Rand2DElasticd, dataset.py:78,150-152,205) and that bench / tests use to make moving images:
// a coarse control grid of displacements is upsampled bicubically to a dense field, images are resampled through it
// bicubically (zero padding, clamped to [0,1]) and label maps with nearest neighbour.  Arithmetic = ATen's
// upsample_bicubic2d(align_corners=True) and grid_sampler_2d(bicubic | nearest, zeros, align_corners=True), cubic
// convolution with A = -0.75, so the CPU generator (mireg/synth.py, torch ops) is the oracle.
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

__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  c[0] = cc2(t + 1.f, A); c[1] = cc1(t, A);
  const float u = 1.f - t;
  c[2] = cc1(u, A); c[3] = cc2(u + 1.f, A);
}

// F.interpolate(in, (h, w), mode='bicubic', align_corners=True): tap indices clamped to the border
__global__ void __launch_bounds__(kThreads)
bicubic_upsample_kernel(const float* __restrict__ in, float* __restrict__ out, long NC, int H, int W, int h, int w, float sy, float sx) {
  const long total = NC * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const long n = i / ((long)w * h);
    const float rx = sx * (float)x, ry = sy * (float)y;
    const float fx = floorf(rx), fy = floorf(ry);
    const int ix = (int)fx, iy = (int)fy;
    float cx[4], cy[4];
    cubic_coeffs(rx - fx, cx);
    cubic_coeffs(ry - fy, cy);
    const float* p = in + n * H * W;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = min(max(iy - 1 + j, 0), H - 1);
      float row = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) row += p[(long)yy * W + min(max(ix - 1 + k, 0), W - 1)] * cx[k];
      acc += row * cy[j];
    }
    out[i] = acc;
  }
}

// moving[b,c] = clamp(grid_sample(img[b,c], identity + disp, bicubic, zeros, align_corners=True), 0, 1);
// seg_out[b,c] = grid_sample(seg[b,c], same grid, nearest, zeros).  disp (B,2,H,W) in pixels of an H x W image; the grid is
// built like the generator does: g = lin(-1,1)[i] + disp * 2 / size, then ATen's unnormalisation ((g + 1) / 2) * (size - 1).
__global__ void __launch_bounds__(kThreads)
elastic_sample_kernel(const float* __restrict__ img, const float* __restrict__ seg, const float* __restrict__ disp,
                      float* __restrict__ out_img, float* __restrict__ out_seg, int B, int C, int Cs, int H, int W) {
  const long npix = (long)H * W, total = (long)B * npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npix);
    const long p = i - (long)b * npix;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    // torch.linspace(-1, 1, n)[i]: symmetric evaluation around the midpoint, step = 2/(n-1)
    const float stx = 2.f / (float)(W - 1), sty = 2.f / (float)(H - 1);
    const float lx = x < W / 2 ? -1.f + stx * (float)x : 1.f - stx * (float)(W - 1 - x);
    const float ly = y < H / 2 ? -1.f + sty * (float)y : 1.f - sty * (float)(H - 1 - y);
    const float gx = lx + disp[((long)b * 2) * npix + p] * 2.f / (float)W;
    const float gy = ly + disp[((long)b * 2 + 1) * npix + p] * 2.f / (float)H;
    const float fxr = ((gx + 1.f) / 2.f) * (float)(W - 1), fyr = ((gy + 1.f) / 2.f) * (float)(H - 1);
    if (img) {
      const float fx = floorf(fxr), fy = floorf(fyr);
      const int ix = (int)fx, iy = (int)fy;
      float cx[4], cy[4];
      cubic_coeffs(fxr - fx, cx);
      cubic_coeffs(fyr - fy, cy);
      for (int c = 0; c < C; ++c) {
        const float* s = img + ((long)b * C + c) * npix;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int yy = iy - 1 + j;
          float row = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int xx = ix - 1 + k;
            const float v = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? s[(long)yy * W + xx] : 0.f;
            row += v * cx[k];
          }
          acc += row * cy[j];
        }
        out_img[((long)b * C + c) * npix + p] = fminf(fmaxf(acc, 0.f), 1.f);
      }
    }
    if (seg) {
      const int xn = (int)nearbyintf(fxr), yn = (int)nearbyintf(fyr);          // round-half-even like ATen
      const bool ok = xn >= 0 && xn < W && yn >= 0 && yn < H;
      for (int c = 0; c < Cs; ++c)
        out_seg[((long)b * Cs + c) * npix + p] = ok ? seg[((long)b * Cs + c) * npix + (long)yn * W + xn] : 0.f;
    }
  }
}


// affine resampling of a batch (the RandAffined step, dataset.py:79,151): F.grid_sample(x, F.affine_grid(theta, x.size())) with
// torch's defaults (align_corners=False): bilinear + zeros for images, nearest + zeros for label maps.  theta: (B,2,3).
__global__ void __launch_bounds__(kThreads)
affine_sample2d_kernel(const float* __restrict__ img, const float* __restrict__ seg, const float* __restrict__ theta,
                       float* __restrict__ out_img, float* __restrict__ out_seg, int B, int C, int Cs, int H, int W) {
  const long npix = (long)H * W, total = (long)B * npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / npix);
    const long p = i - (long)b * npix;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    const float* t = theta + b * 6;
    const float bx = (2.f * x + 1.f) / (float)W - 1.f, by = (2.f * y + 1.f) / (float)H - 1.f;
    const float gx = t[0] * bx + t[1] * by + t[2], gy = t[3] * bx + t[4] * by + t[5];
    const float px = ((gx + 1.f) * (float)W - 1.f) / 2.f, py = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    if (img) {
      const float fx = floorf(px), fy = floorf(py);
      const int x0 = (int)fx, y0 = (int)fy;
      const float wx1 = px - fx, wy1 = py - fy;
      for (int c = 0; c < C; ++c) {
        const float* s = img + ((long)b * C + c) * npix;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int xi = x0 + (k & 1), yi = y0 + (k >> 1);
          if (xi < 0 || xi >= W || yi < 0 || yi >= H) continue;
          acc += s[(long)yi * W + xi] * ((k & 1) ? wx1 : 1.f - wx1) * ((k >> 1) ? wy1 : 1.f - wy1);
        }
        out_img[((long)b * C + c) * npix + p] = acc;
      }
    }
    if (seg) {
      const int xn = (int)nearbyintf(px), yn = (int)nearbyintf(py);
      const bool ok = xn >= 0 && xn < W && yn >= 0 && yn < H;
      for (int c = 0; c < Cs; ++c)
        out_seg[((long)b * Cs + c) * npix + p] = ok ? seg[((long)b * Cs + c) * npix + (long)yn * W + xn] : 0.f;
    }
  }
}


// ---- the front of the reference's data pipeline on device (dataset.py:52-57,73-77,141-148: Transposed -> SpatialCropd -> per-slice
// Resized(bilinear | nearest) -> Rotate90d; volume_ds: Resized(trilinear) -> Rotate90d(k=2)) as ONE gather kernel: the source is
// addressed through element strides per logical axis (so Transposed and the crop are views: a base pointer and strides), the
// resize is torch's F.interpolate arithmetic (linear: align_corners=False, src = (dst + 0.5) * in/out - 0.5 clamped at 0;
// nearest: src = floor(dst * in/out)), the rotation is numpy's rot90 on the last two logical axes.
__device__ __forceinline__ void lin_tap(int dst, int in, int out, int& i0, int& i1, float& w1) {
  if (in == out) { i0 = i1 = dst; w1 = 0.f; return; }
  float sc = (float)in / (float)out;
  float s = ((float)dst + 0.5f) * sc - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 > in - 1 ? in - 1 : i0;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = s - (float)i0;
}
__device__ __forceinline__ int near_tap(int dst, int in, int out) {
  if (in == out) return dst;
  const int i = (int)floorf((float)dst * ((float)in / (float)out));
  return i > in - 1 ? in - 1 : i;
}

__global__ void __launch_bounds__(kThreads)
resample_volume_kernel(const float* __restrict__ in, long isn, long isd, long ish, long isw, int N, int D, int H, int W,
                       float* __restrict__ out, long osn, long osd, long osh, long osw, int d, int h, int w, int mode, int rot_k) {
  // output extents after the rotation: (d, oh, ow) with (oh, ow) = (h, w) for even k, (w, h) for odd k
  const int oh = (rot_k & 1) ? w : h, ow = (rot_k & 1) ? h : w;
  const long total = (long)N * d * oh * ow;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int j = (int)(t % ow), i = (int)((t / ow) % oh), z = (int)((t / ((long)ow * oh)) % d);
    const long n = t / ((long)ow * oh * d);
    // numpy.rot90(m, k) on (h, w): k=1: out[i][j] = m[j][w-1-i]; k=2: m[h-1-i][w-1-j]; k=3: m[h-1-j][i]
    int r, c;
    switch (rot_k & 3) {
      case 0: r = i; c = j; break;
      case 1: r = j; c = w - 1 - i; break;
      case 2: r = h - 1 - i; c = w - 1 - j; break;
      default: r = h - 1 - j; c = i; break;
    }
    const float* src = in + n * isn;
    float v;
    if (mode == 1) {
      v = src[(long)near_tap(z, D, d) * isd + (long)near_tap(r, H, h) * ish + (long)near_tap(c, W, w) * isw];
    } else {
      int z0, z1, y0, y1, x0, x1;
      float wz, wy, wx;
      lin_tap(z, D, d, z0, z1, wz); lin_tap(r, H, h, y0, y1, wy); lin_tap(c, W, w, x0, x1, wx);
      auto at = [&](int zz, int yy, int xx) { return src[(long)zz * isd + (long)yy * ish + (long)xx * isw]; };
      // torch's order: lerp along x, then y, then z
      const float a00 = at(z0, y0, x0) * (1.f - wx) + at(z0, y0, x1) * wx, a01 = at(z0, y1, x0) * (1.f - wx) + at(z0, y1, x1) * wx;
      const float a0 = a00 * (1.f - wy) + a01 * wy;
      if (z1 == z0) v = a0;
      else {
        const float a10 = at(z1, y0, x0) * (1.f - wx) + at(z1, y0, x1) * wx, a11 = at(z1, y1, x0) * (1.f - wx) + at(z1, y1, x1) * wx;
        v = a0 * (1.f - wz) + (a10 * (1.f - wy) + a11 * wy) * wz;
      }
    }
    out[n * osn + (long)z * osd + (long)i * osh + (long)j * osw] = v;
  }
}

// ScaleIntensityd(minv, maxv) per item (dataset.py:83,153,209): (x - min) / (max - min) * (maxv - minv) + minv over the item's n values;
// a constant item becomes x * minv, as MONAI does.  Two launches: per-item partial extrema (fixed-order tree), then the map.
__global__ void __launch_bounds__(kThreads)
minmax_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {   // grid (blocks, items); part[item][block][2]
  __shared__ float smin[kThreads], smax[kThreads];
  const float* p = x + (long)blockIdx.y * n;
  float lo = 3.4e38f, hi = -3.4e38f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float v = p[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
  __syncthreads();
  for (int s2 = kThreads / 2; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) { smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s2]); smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s2]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { float* o = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2; o[0] = smin[0]; o[1] = smax[0]; }
}
__global__ void __launch_bounds__(kThreads)
scale_intensity_kernel(float* __restrict__ x, long n, const float* __restrict__ part, int nblk, float minv, float maxv) {
  const float* pp = part + (long)blockIdx.y * nblk * 2;
  float lo = 3.4e38f, hi = -3.4e38f;
  for (int b = 0; b < nblk; ++b) { lo = fminf(lo, pp[2 * b]); hi = fmaxf(hi, pp[2 * b + 1]); }
  float* p = x + (long)blockIdx.y * n;
  const bool flat = hi == lo;
  const float k = flat ? 0.f : (maxv - minv) / (hi - lo);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] = flat ? p[i] * minv : (p[i] - lo) * k + minv;
}

}  // namespace

extern "C" {

int mireg_resize_bicubic_fwd(const float* in, float* out, long NC, int H, int W, int h, int w, hipStream_t stream) {
  MIREG_CHECK_ARG(in && out && NC > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  const float sy = h > 1 ? (float)(H - 1) / (float)(h - 1) : 0.f, sx = w > 1 ? (float)(W - 1) / (float)(w - 1) : 0.f;
  hipLaunchKernelGGL(bicubic_upsample_kernel, dim3(grid_for(NC * h * w)), dim3(kThreads), 0, stream, in, out, NC, H, W, h, w, sy, sx);
  MIREG_LAUNCH_RET();
}

int mireg_elastic_sample(const float* img, const float* seg, const float* disp, float* out_img, float* out_seg, int B, int C, int Cs,
                         int H, int W, hipStream_t stream) {
  MIREG_CHECK_ARG(disp && (img || seg) && B > 0 && H > 1 && W > 1);
  MIREG_CHECK_ARG((!img || (out_img && C > 0)) && (!seg || (out_seg && Cs > 0)));
  hipLaunchKernelGGL(elastic_sample_kernel, dim3(grid_for((long)B * H * W)), dim3(kThreads), 0, stream, img, seg, disp, out_img, out_seg,
                     B, C, Cs, H, W);
  MIREG_LAUNCH_RET();
}


int mireg_affine_sample2d(const float* img, const float* seg, const float* theta, float* out_img, float* out_seg, int B, int C,
                          int Cs, int H, int W, hipStream_t stream) {
  MIREG_CHECK_ARG(theta && (img || seg) && B > 0 && H > 0 && W > 0);
  MIREG_CHECK_ARG((!img || (out_img && C > 0)) && (!seg || (out_seg && Cs > 0)));
  hipLaunchKernelGGL(affine_sample2d_kernel, dim3(grid_for((long)B * H * W)), dim3(kThreads), 0, stream, img, seg, theta, out_img,
                     out_seg, B, C, Cs, H, W);
  MIREG_LAUNCH_RET();
}


int mireg_resample_volume(const float* in, long isn, long isd, long ish, long isw, int N, int D, int H, int W, float* out, long osn,
                          long osd, long osh, long osw, int d, int h, int w, int mode, int rot_k, hipStream_t stream) {
  MIREG_CHECK_ARG(in && out && N > 0 && D > 0 && H > 0 && W > 0 && d > 0 && h > 0 && w > 0 && (mode == 0 || mode == 1) && rot_k >= 0 && rot_k <= 3);
  hipLaunchKernelGGL(resample_volume_kernel, dim3(grid_for((long)N * d * h * w)), dim3(kThreads), 0, stream, in, isn, isd, ish, isw, N, D, H, W,
                     out, osn, osd, osh, osw, d, h, w, mode, rot_k);
  MIREG_LAUNCH_RET();
}

int mireg_scale_intensity(float* x, int items, long n, float minv, float maxv, float* workspace, hipStream_t stream) {
  MIREG_CHECK_ARG(x && workspace && items > 0 && n > 0);
  int nblk = (int)((n + kThreads * 8 - 1) / (kThreads * 8));
  nblk = nblk < 1 ? 1 : (nblk > MIREG_SCALE_INTENSITY_BLOCKS ? MIREG_SCALE_INTENSITY_BLOCKS : nblk);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nblk, items), dim3(kThreads), 0, stream, x, n, workspace);
  hipLaunchKernelGGL(scale_intensity_kernel, dim3(nblk, items), dim3(kThreads), 0, stream, x, n, workspace, nblk, minv, maxv);
  MIREG_LAUNCH_RET();
}

}  // extern "C"
