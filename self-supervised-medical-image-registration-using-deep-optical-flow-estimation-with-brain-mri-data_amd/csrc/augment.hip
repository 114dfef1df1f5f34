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

}  // extern "C"
