// volume_ops.hip -- the 3-D (volume) counterparts of the per-scale registration tail, SURVEY section 8 row a14 / BASELINE config
// "3D FlowNetS on 128^3": the reference has no dense 3-D flow, so these follow its 2-D conventions axis by axis
//   resize  : F.interpolate(mode='trilinear', align_corners=...)           (models.py:258 / loss.py:11 / FlowNetS.py:83 per axis)
//   stn3d   : sample at (i + flow_i) (n_i - 1) / n_i per axis, zero padding  (models.py:256-268)
//   smooth  : sum_c sum_axes charb(f - f_shifted) / C                        (loss.py:21-29 with C = 3 flow channels)
// Planar fp32 volumes (B,C,D,H,W); flows are addressed through (sb, sc, sp) strides so the predictor's channel-last
// outputs are read in place.  All backward kernels are gather-form (one thread per destination element, no atomics).
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

namespace {

constexpr int kThreads = 256;
constexpr int kSlots = MIREG_SUM_SLOTS;

inline int grid_for(long work, int cap = 4096) {
  long g = (work + kThreads - 1) / kThreads;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// Grid-stride loops over (plane, z, y, x) without per-element 64-bit divisions (each costs tens of instructions; five of them per
// voxel made these kernels ALU-bound at 128^3): the flat index is split once and then advanced by the loop stride in mixed radix.
struct Vox { int x, y, z, p; };
inline Vox vox_step(long stride, int w, int h, int d) {               // host: the stride's digits
  Vox s;
  s.x = (int)(stride % w); stride /= w;
  s.y = (int)(stride % h); stride /= h;
  s.z = (int)(stride % d);
  s.p = (int)(stride / d);
  return s;
}
__device__ __forceinline__ Vox vox_split(long i, int w, int h, int d) {
  Vox v;
  v.x = (int)(i % w); i /= w;
  v.y = (int)(i % h); i /= h;
  v.z = (int)(i % d);
  v.p = (int)(i / d);
  return v;
}
__device__ __forceinline__ void vox_advance(Vox& v, const Vox& s, int w, int h, int d) {
  v.x += s.x; int c = v.x >= w; v.x -= c ? w : 0;
  v.y += s.y + c; c = v.y >= h; v.y -= c ? h : 0;
  v.z += s.z + c; c = v.z >= d; v.z -= c ? d : 0;
  v.p += s.p + c;
}

// same op order as ATen's area_pixel_compute_source_index (linear modes)
__device__ __forceinline__ void src_coord(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
  float s = align ? (float)dst * scale : fmaxf(((float)dst + 0.5f) * scale - 0.5f, 0.f);
  i0 = min((int)s, in_size - 1);
  i1 = min(i0 + 1, in_size - 1);
  l1 = s - (float)i0;
}

__device__ __forceinline__ void out_range(int i, float scale, int align, int out_size, int& lo, int& hi) {
  const float inv = scale > 0.f ? 1.f / scale : 0.f;
  const float a = align ? ((float)i - 1.f) * inv : (((float)i - 1.f) + 0.5f) * inv - 0.5f;
  const float b = align ? ((float)i + 1.f) * inv : (((float)i + 1.f) + 0.5f) * inv - 0.5f;
  lo = max((int)floorf(a) - 1, 0);
  hi = min((int)ceilf(b) + 1, out_size - 1);
  if (scale <= 0.f) { lo = 0; hi = out_size - 1; }
}

inline float host_scale(int in, int out, int align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}

// ---- trilinear resize -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
resize3d_fwd_kernel(const float* __restrict__ in, long isn, long isc, long isp, float* __restrict__ out, int N, int C,
                    int D, int H, int W, int d, int h, int w, float sz, float sy, float sx, int align, Vox step) {
  const long ovox = (long)d * h * w, total = (long)N * C * ovox;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox q = vox_split(i, w, h, d);
  for (; i < total; i += (long)gridDim.x * blockDim.x, vox_advance(q, step, w, h, d)) {
    const int x = q.x, y = q.y, z = q.z;
    const int n = q.p / C, c = q.p - n * C;
    int z0, z1, y0, y1, x0, x1;
    float lz, ly, lx;
    src_coord(z, sz, align, D, z0, z1, lz);
    src_coord(y, sy, align, H, y0, y1, ly);
    src_coord(x, sx, align, W, x0, x1, lx);
    const float* p = in + n * isn + c * isc;
    auto at = [&](int zz, int yy, int xx) { return p[(((long)zz * H + yy) * W + xx) * isp]; };
    const float a00 = at(z0, y0, x0) * (1.f - lx) + at(z0, y0, x1) * lx, a01 = at(z0, y1, x0) * (1.f - lx) + at(z0, y1, x1) * lx;
    const float a10 = at(z1, y0, x0) * (1.f - lx) + at(z1, y0, x1) * lx, a11 = at(z1, y1, x0) * (1.f - lx) + at(z1, y1, x1) * lx;
    const float b0 = a00 * (1.f - ly) + a01 * ly, b1 = a10 * (1.f - ly) + a11 * ly;
    out[i] = b0 * (1.f - lz) + b1 * lz;
  }
}

// gin[(Z,Y,X)] = sum over the output voxels whose taps include it; gin addressed through (isn, isc, isp)
__global__ void __launch_bounds__(kThreads)
resize3d_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, long isn, long isc, long isp, int N, int C,
                    int D, int H, int W, int d, int h, int w, float sz, float sy, float sx, int align, float beta) {
  const long ivox = (long)D * H * W, ovox = (long)d * h * w, total = (long)N * C * ivox;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W), Y = (int)((i / W) % H), Z = (int)((i / ((long)W * H)) % D);
    const int c = (int)((i / ivox) % C), n = (int)(i / (ivox * C));
    int zlo, zhi, ylo, yhi, xlo, xhi;
    out_range(Z, sz, align, d, zlo, zhi);
    out_range(Y, sy, align, h, ylo, yhi);
    out_range(X, sx, align, w, xlo, xhi);
    const float* g = gout + ((long)n * C + c) * ovox;
    float acc = 0.f;
    for (int z = zlo; z <= zhi; ++z) {
      int z0, z1; float lz;
      src_coord(z, sz, align, D, z0, z1, lz);
      const float wz = (z0 == Z ? 1.f - lz : 0.f) + (z1 == Z ? lz : 0.f);
      if (wz == 0.f) continue;
      for (int y = ylo; y <= yhi; ++y) {
        int y0, y1; float ly;
        src_coord(y, sy, align, H, y0, y1, ly);
        const float wy = (y0 == Y ? 1.f - ly : 0.f) + (y1 == Y ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int x = xlo; x <= xhi; ++x) {
          int x0, x1; float lx;
          src_coord(x, sx, align, W, x0, x1, lx);
          const float wx = (x0 == X ? 1.f - lx : 0.f) + (x1 == X ? lx : 0.f);
          if (wx != 0.f) acc += wz * wy * wx * g[((long)z * h + y) * w + x];
        }
      }
    }
    float* dst = gin + n * isn + c * isc + (((long)Z * H + Y) * W + X) * isp;
    *dst = beta != 0.f ? *dst * beta + acc : acc;
  }
}

// Separable form of the same backward: trilinear interpolation is three 1-D linear interpolations, so its adjoint is three 1-D
// gathers (x, then y, then z).  The direct kernel above visits up to 8 x 8 x 8 output voxels per source voxel with a 16-byte lane
// stride (x4 upsampling of the 3-channel flow: 0.6-1.0 ms for 8 x 3 x 128^3 gradients); the passes below read the big tensor once,
// contiguously, and shrink it 4x per pass.  src / dst are [outer][L][inner] planes; the last pass writes through the strided
// (batch, channel, voxel) addressing of gin and applies beta.
__global__ void __launch_bounds__(kThreads)
resize1d_bwd_kernel(const float* __restrict__ src, float* __restrict__ dst, long outer, int Lout, int Lin, long inner, float scale,
                    int align, int strided, long isn, long isc, long isp, int C, float beta, Vox step) {
  const long total = outer * Lin * inner;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox qv = vox_split(i, (int)inner, Lin, 0x7fffffff);                  // x = inner index, y = P, z = outer index (no carry out)
  for (; i < total; i += (long)gridDim.x * blockDim.x, vox_advance(qv, step, (int)inner, Lin, 0x7fffffff)) {
    const long in_ = qv.x, o = qv.z;
    const int P = qv.y;
    int lo, hi;
    out_range(P, scale, align, Lout, lo, hi);
    const float* s = src + (o * Lout) * inner + in_;
    float acc = 0.f;
    for (int q = lo; q <= hi; ++q) {
      int q0, q1; float l;
      src_coord(q, scale, align, Lin, q0, q1, l);
      const float wq = (q0 == P ? 1.f - l : 0.f) + (q1 == P ? l : 0.f);
      if (wq != 0.f) acc += wq * s[(long)q * inner];
    }
    if (!strided) dst[i] = acc;
    else {
      const int n = (int)o / C, c = (int)o - n * C;
      float* d = dst + n * isn + c * isc + ((long)P * inner + in_) * isp;
      *d = beta != 0.f ? *d * beta + acc : acc;
    }
  }
}

// ---- dense 3-D warp ---------------------------------------------------------------------------------------------------
// sampling coordinate per axis, same fp32 op order as the 2-D path: g = (i + f) * (2/n) - 1; p = ((g + 1)/2) * (n - 1)
__device__ __forceinline__ float stn_coord(float pix, float disp, float two_over, float sizem1) {
  const float g = (disp + pix) * two_over - 1.f;
  return ((g + 1.f) / 2.f) * sizem1;
}

struct Taps3 {
  int x0, y0, z0;
  float wx1, wy1, wz1;
};

__device__ __forceinline__ Taps3 taps_for(const float* __restrict__ f, long fsc, int x, int y, int z, int d, int h, int w) {
  const float two_w = (float)(2.0 / (double)w), two_h = (float)(2.0 / (double)h), two_d = (float)(2.0 / (double)d);
  const float px = stn_coord((float)x, f[0], two_w, (float)(w - 1));
  const float py = stn_coord((float)y, f[fsc], two_h, (float)(h - 1));
  const float pz = stn_coord((float)z, f[2 * fsc], two_d, (float)(d - 1));
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  Taps3 t;
  t.x0 = (int)fx; t.y0 = (int)fy; t.z0 = (int)fz;
  t.wx1 = px - fx; t.wy1 = py - fy; t.wz1 = pz - fz;
  return t;
}

__device__ __forceinline__ float tap3(const float* __restrict__ img, int x, int y, int z, int w, int h, int d) {
  return (x >= 0 && x < w && y >= 0 && y < h && z >= 0 && z < d) ? img[((long)z * h + y) * w + x] : 0.f;
}

__global__ void __launch_bounds__(kThreads)
stn3d_fwd_kernel(const float* __restrict__ flow, long fsb, long fsc, long fsp, const float* __restrict__ frame,
                 float* __restrict__ warped, int B, int C, int d, int h, int w, Vox step) {
  const long nvox = (long)d * h * w, total = (long)B * nvox;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox q = vox_split(i, w, h, d);
  for (; i < total; i += (long)gridDim.x * blockDim.x, vox_advance(q, step, w, h, d)) {
    const int b = q.p, x = q.x, y = q.y, z = q.z;
    const long v = ((long)z * h + y) * w + x;
    const Taps3 t = taps_for(flow + b * fsb + v * fsp, fsc, x, y, z, d, h, w);
    for (int c = 0; c < C; ++c) {
      const float* img = frame + ((long)b * C + c) * nvox;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float wgt = ((k & 1) ? t.wx1 : 1.f - t.wx1) * (((k >> 1) & 1) ? t.wy1 : 1.f - t.wy1) * ((k >> 2) ? t.wz1 : 1.f - t.wz1);
        acc += tap3(img, t.x0 + (k & 1), t.y0 + ((k >> 1) & 1), t.z0 + (k >> 2), w, h, d) * wgt;
      }
      warped[((long)b * C + c) * nvox + v] = acc;
    }
  }
}

// gflow planar (B,3,d,h,w): channel a = d/d flow_a = sum_c gout_c * d sample / d p_a * (n_a - 1)/n_a
__global__ void __launch_bounds__(kThreads)
stn3d_bwd_kernel(const float* __restrict__ flow, long fsb, long fsc, long fsp, const float* __restrict__ frame,
                 const float* __restrict__ gout, float* __restrict__ gflow, float beta, int B, int C, int d, int h, int w, Vox step) {
  const long nvox = (long)d * h * w, total = (long)B * nvox;
  const float kx = (float)(w - 1) / (float)w, ky = (float)(h - 1) / (float)h, kz = (float)(d - 1) / (float)d;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox q = vox_split(i, w, h, d);
  for (; i < total; i += (long)gridDim.x * blockDim.x, vox_advance(q, step, w, h, d)) {
    const int b = q.p, x = q.x, y = q.y, z = q.z;
    const long v = ((long)z * h + y) * w + x;
    const Taps3 t = taps_for(flow + b * fsb + v * fsp, fsc, x, y, z, d, h, w);
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* img = frame + ((long)b * C + c) * nvox;
      const float go = gout[((long)b * C + c) * nvox + v];
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float s = tap3(img, t.x0 + (k & 1), t.y0 + ((k >> 1) & 1), t.z0 + (k >> 2), w, h, d);
        const float wx = (k & 1) ? t.wx1 : 1.f - t.wx1, wy = ((k >> 1) & 1) ? t.wy1 : 1.f - t.wy1, wz = (k >> 2) ? t.wz1 : 1.f - t.wz1;
        ax += ((k & 1) ? s : -s) * wy * wz;
        ay += (((k >> 1) & 1) ? s : -s) * wx * wz;
        az += ((k >> 2) ? s : -s) * wx * wy;
      }
      gx += go * ax; gy += go * ay; gz += go * az;
    }
    float* g = gflow + (long)b * 3 * nvox + v;
    gx *= kx; gy *= ky; gz *= kz;
    if (beta != 0.f) { g[0] = g[0] * beta + gx; g[nvox] = g[nvox] * beta + gy; g[2 * nvox] = g[2 * nvox] * beta + gz; }
    else { g[0] = gx; g[nvox] = gy; g[2 * nvox] = gz; }
  }
}

// ---- smoothness over three axes, three flow channels ------------------------------------------------------------------
// the moment table's smoothness slot is finalised as sum / 2 (two flow channels in the reference); with three channels
// the mean over channels is sum / 3, so the kernels carry the factor 2/3
constexpr float kChan = 2.f / 3.f;

__global__ void __launch_bounds__(kThreads)
smooth3d_fwd_kernel(const float* __restrict__ flow, long fsb, long fsc, long fsp, double* __restrict__ sum, int B, int d, int h, int w,
                    Vox step) {
  __shared__ float red[kThreads / 64];
  const long nvox = (long)d * h * w, total = (long)B * nvox;
  float acc[1] = {0.f};
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox q = vox_split(i, w, h, d);
  for (; i < total; i += (long)gridDim.x * blockDim.x, vox_advance(q, step, w, h, d)) {
    const int b = q.p, x = q.x, y = q.y, z = q.z;
    const long v = ((long)z * h + y) * w + x;
    const float* f = flow + b * fsb + v * fsp;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float val = f[c * fsc];
      const float nz = (z + 1 < d) ? f[c * fsc + (long)h * w * fsp] : 0.f;
      const float ny = (y + 1 < h) ? f[c * fsc + (long)w * fsp] : 0.f;
      const float nx = (x + 1 < w) ? f[c * fsc + fsp] : 0.f;
      acc[0] += charb(val - nz) + charb(val - ny) + charb(val - nx);
    }
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) atomicAdd(sum + (blockIdx.x % kSlots) * 8, (double)(acc[0] * kChan));
}

__global__ void __launch_bounds__(kThreads)
smooth3d_bwd_kernel(const float* __restrict__ flow, long fsb, long fsc, long fsp, const float* __restrict__ coef,
                    float* __restrict__ gflow, float beta, int B, int d, int h, int w, Vox step) {
  const float cs = coef[5] * kChan;
  const long nvox = (long)d * h * w, total = (long)B * nvox;
  const long sz = (long)h * w * fsp, sy = (long)w * fsp;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox q = vox_split(i, w, h, d);
  for (; i < total; i += (long)gridDim.x * blockDim.x, vox_advance(q, step, w, h, d)) {
    const int b = q.p, x = q.x, y = q.y, z = q.z;
    const long v = ((long)z * h + y) * w + x;
    const float* f = flow + b * fsb + v * fsp;
    float* g = gflow + (long)b * 3 * nvox + v;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* fc = f + c * fsc;
      const float val = fc[0];
      float acc = charb_grad(val - ((z + 1 < d) ? fc[sz] : 0.f)) + charb_grad(val - ((y + 1 < h) ? fc[sy] : 0.f)) +
                  charb_grad(val - ((x + 1 < w) ? fc[fsp] : 0.f));
      if (z > 0) acc -= charb_grad(fc[-sz] - val);
      if (y > 0) acc -= charb_grad(fc[-sy] - val);
      if (x > 0) acc -= charb_grad(fc[-fsp] - val);
      acc *= cs;
      g[c * nvox] = beta != 0.f ? g[c * nvox] * beta + acc : acc;
    }
  }
}


// ---- x-axis im2col of a few-channel volume (the stem of FlowNetS over volumes) ---------------------------------------------
// dst[b][z][y][xo][tx*C + c] = x[b][c][z][y][xo*stride + tx - pad] (zero outside, pad channels zero): a k^3 / C-channel
// convolution becomes a (k, k, 1) convolution over k*C -> Cpad channels, so the GEMM's 8-channel K granules carry
// k*C/Cpad real data instead of C/8 (2 of 8 for the two-channel input: 4x less gathered bytes and MFMA work).
template <typename T>
__global__ void __launch_bounds__(kThreads)
stem3d_gather_kernel(const float* __restrict__ x, T* __restrict__ dst, int B, int C, int D, int H, int W, int Wo, int k, int stride,
                     int pad, int Cpad, Vox step) {
  // one thread per (voxel, 8-channel granule): consecutive threads write consecutive 16/32-byte granules
  const int gpv = Cpad / 8;
  const long total = (long)B * D * H * Wo * gpv;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox q = vox_split(i, Wo * gpv, H, D);                                // x digit = (xo, granule)
  for (; i < total; i += (long)gridDim.x * blockDim.x, vox_advance(q, step, Wo * gpv, H, D)) {
    const int xo = q.x / gpv, g = q.x - xo * gpv;
    const int y = q.y, z = q.z, b = q.p;
    const float* row = x + (((long)b * C * D + z) * H + y) * W;          // channel c adds c * D*H*W
    alignas(16) T v[8];
    int tx = (g * 8) / C, c = g * 8 - tx * C;                          // (tap, channel) of the granule's first element, then counted up
    const long cstride = (long)D * H * W;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = 0.f;
      const int xi = xo * stride + tx - pad;
      if (tx < k && xi >= 0 && xi < W) f = row[c * cstride + xi];
      v[e] = (T)f;
      if (++c == C) { c = 0; ++tx; }
    }
    if constexpr (sizeof(T) == 2) *reinterpret_cast<uint4*>(dst + i * 8) = *reinterpret_cast<const uint4*>(v);   // one 16-byte store
    else {
      T* d = dst + i * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = v[e];
    }
  }
}

// ---- the 3 -> 3 channel flow upsamplers of FlowNetS over volumes, ConvTranspose3d(3, 3, 4, 2, 1) (FlowNetS/FlowNetS.py:37-40 per axis) -----------
// A few MFLOP each; through the 128-wide GEMM tiles they were 8 parity-class launches forward, 1 backward-data and 4 backward-weights launches per
// level (56 launches, ~0.9 ms per step).  Here: one thread per voxel on the fp32 master weight Wc[co][ci][tz][ty][tx] (the Conv3d weight of the
// adjoint, stride-2 convolution fine(ci) -> coarse(co); 576 floats), as the 2-D tiny_* kernels of thin_conv.hip do for 2 -> 2 channels.
template <typename T>
__global__ void __launch_bounds__(kThreads)
tiny_deconv3d_fwd_kernel(const T* __restrict__ xc, long ld_c, const float* __restrict__ w, T* __restrict__ yf, long ld_f, int B, int Dc, int Hc,
                         int Wc, Vox step) {
  const int Df = 2 * Dc, Hf = 2 * Hc, Wf = 2 * Wc;
  const long n = (long)B * Df * Hf * Wf;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox q = vox_split(i, Wf, Hf, Df);
  for (; i < n; i += (long)gridDim.x * blockDim.x, vox_advance(q, step, Wf, Hf, Df)) {
    float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int tz = 0; tz < 2; ++tz) {
      const int kz = ((q.z + 1) & 1) + 2 * tz, oz = (q.z + 1 - kz) >> 1;           // fine slice z receives taps kz = (z + 1) mod 2, + 2
      if ((unsigned)oz >= (unsigned)Dc) continue;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        const int ky = ((q.y + 1) & 1) + 2 * ty, oy = (q.y + 1 - ky) >> 1;
        if ((unsigned)oy >= (unsigned)Hc) continue;
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
          const int kx = ((q.x + 1) & 1) + 2 * tx, ox = (q.x + 1 - kx) >> 1;
          if ((unsigned)ox >= (unsigned)Wc) continue;
          const T* src = xc + ((((long)q.p * Dc + oz) * Hc + oy) * Wc + ox) * ld_c;
          const int t = (kz * 4 + ky) * 4 + kx;
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            const float c = (float)src[co];
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) a[ci] += c * w[(co * 3 + ci) * 64 + t];
          }
        }
      }
    }
    T* d = yf + i * ld_f;
    d[0] = (T)a[0]; d[1] = (T)a[1]; d[2] = (T)a[2];
  }
}

// backward-data of the upsampler = the stride-2 convolution itself: coarse[o][co] (+)= sum_{ci, k} fine[2 o + k - 1][ci] * Wc[co][ci][k]
template <typename T>
__global__ void __launch_bounds__(kThreads)
tiny_conv3d_fwd_kernel(const T* __restrict__ xf, long ld_f, const float* __restrict__ w, T* __restrict__ yc, long ld_c, int accumulate, int B,
                       int Dc, int Hc, int Wc, Vox step) {
  const int Df = 2 * Dc, Hf = 2 * Hc, Wf = 2 * Wc;
  const long n = (long)B * Dc * Hc * Wc;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Vox q = vox_split(i, Wc, Hc, Dc);
  for (; i < n; i += (long)gridDim.x * blockDim.x, vox_advance(q, step, Wc, Hc, Dc)) {
    float a[3] = {0.f, 0.f, 0.f};
    for (int kz = 0; kz < 4; ++kz) {
      const int iz = 2 * q.z + kz - 1;
      if ((unsigned)iz >= (unsigned)Df) continue;
      for (int ky = 0; ky < 4; ++ky) {
        const int iy = 2 * q.y + ky - 1;
        if ((unsigned)iy >= (unsigned)Hf) continue;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const int ix = 2 * q.x + kx - 1;
          if ((unsigned)ix >= (unsigned)Wf) continue;
          const T* src = xf + ((((long)q.p * Df + iz) * Hf + iy) * Wf + ix) * ld_f;
          const int t = (kz * 4 + ky) * 4 + kx;
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) {
            const float f = (float)src[ci];
#pragma unroll
            for (int co = 0; co < 3; ++co) a[co] += f * w[(co * 3 + ci) * 64 + t];
          }
        }
      }
    }
    T* d = yc + i * ld_c;
#pragma unroll
    for (int co = 0; co < 3; ++co) d[co] = (T)(accumulate ? (float)d[co] + a[co] : a[co]);
  }
}

// backward-weights: block = a contiguous chunk of coarse voxels, thread = (tap, fine channel ci), three accumulators (co); partial slab
// [block][co][tap * Cpad + ci] in the layers' standard slab layout (summed by mireg_wgrad_reduce in fixed order: run-to-run identical)
constexpr int kTinyW3Threads = 192;
template <typename T>
__global__ void __launch_bounds__(kTinyW3Threads)
tiny_wgrad3d_kernel(const T* __restrict__ gf, long ld_f, const T* __restrict__ xc, long ld_c, float* __restrict__ slab, int Cpad, int B,
                    int Dc, int Hc, int Wc, int per_block) {
  const int Df = 2 * Dc, Hf = 2 * Hc, Wf = 2 * Wc;
  const long n = (long)B * Dc * Hc * Wc;
  const int t = threadIdx.x / 3, ci = threadIdx.x - 3 * t;
  const int kz = t >> 4, ky = (t >> 2) & 3, kx = t & 3;
  float a[3] = {0.f, 0.f, 0.f};
  const long v0 = (long)blockIdx.x * per_block, v1 = v0 + per_block < n ? v0 + per_block : n;
  Vox q = vox_split(v0, Wc, Hc, Dc);
  const Vox one = {1, 0, 0, 0};
  for (long v = v0; v < v1; ++v, vox_advance(q, one, Wc, Hc, Dc)) {
    const int iz = 2 * q.z + kz - 1, iy = 2 * q.y + ky - 1, ix = 2 * q.x + kx - 1;
    if ((unsigned)iz >= (unsigned)Df || (unsigned)iy >= (unsigned)Hf || (unsigned)ix >= (unsigned)Wf) continue;
    const float f = (float)gf[((((long)q.p * Df + iz) * Hf + iy) * Wf + ix) * ld_f + ci];
    const T* c = xc + v * ld_c;
    a[0] += f * (float)c[0]; a[1] += f * (float)c[1]; a[2] += f * (float)c[2];
  }
  float* d = slab + (long)blockIdx.x * 3 * 64 * Cpad + (long)t * Cpad + ci;
#pragma unroll
  for (int co = 0; co < 3; ++co) d[(long)co * 64 * Cpad] = a[co];
}

}  // namespace

extern "C" {

int mireg_resize_trilinear_fwd(const float* in, long isn, long isc, long isp, float* out, int N, int C, int D, int H, int W,
                               int d, int h, int w, int align_corners, hipStream_t stream) {
  MIREG_CHECK_ARG(in && out && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && d > 0 && h > 0 && w > 0);
  const int g = grid_for((long)N * C * d * h * w);
  hipLaunchKernelGGL(resize3d_fwd_kernel, dim3(g), dim3(kThreads), 0, stream, in, isn, isc, isp, out,
                     N, C, D, H, W, d, h, w, host_scale(D, d, align_corners), host_scale(H, h, align_corners),
                     host_scale(W, w, align_corners), align_corners, vox_step((long)g * kThreads, w, h, d));
  MIREG_LAUNCH_RET();
}

int mireg_resize_trilinear_bwd(const float* gout, float* gin, long isn, long isc, long isp, int N, int C, int D, int H, int W,
                               int d, int h, int w, int align_corners, float beta, hipStream_t stream) {
  MIREG_CHECK_ARG(gout && gin && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && d > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(resize3d_bwd_kernel, dim3(grid_for((long)N * C * D * H * W)), dim3(kThreads), 0, stream, gout, gin, isn, isc,
                     isp, N, C, D, H, W, d, h, w, host_scale(D, d, align_corners), host_scale(H, h, align_corners),
                     host_scale(W, w, align_corners), align_corners, beta);
  MIREG_LAUNCH_RET();
}

int mireg_resize_trilinear_bwd_sep(const float* gout, float* gin, long isn, long isc, long isp, int N, int C, int D, int H, int W,
                                   int d, int h, int w, int align_corners, float beta, float* ws, long ws_elems, hipStream_t stream) {
  MIREG_CHECK_ARG(gout && gin && ws && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && d > 0 && h > 0 && w > 0);
  const long nc = (long)N * C, e1 = nc * d * h * W, e2 = nc * d * H * W;
  MIREG_CHECK_ARG(ws_elems >= e1 + e2);
  float* t1 = ws;                                                    // [N*C*d*h][W]
  float* t2 = ws + e1;                                               // [N*C*d][H][W]
  MIREG_CHECK_ARG((long)H * W < 0x7fffffffL && nc * d * h < 0x7fffffffL);
  const int g1 = grid_for(e1), g2 = grid_for(e2), g3 = grid_for(nc * D * H * W);
  hipLaunchKernelGGL(resize1d_bwd_kernel, dim3(g1), dim3(kThreads), 0, stream, gout, t1, nc * d * h, w, W, 1L,
                     host_scale(W, w, align_corners), align_corners, 0, 0L, 0L, 0L, C, 0.f, vox_step((long)g1 * kThreads, 1, W, 0x7fffffff));
  hipLaunchKernelGGL(resize1d_bwd_kernel, dim3(g2), dim3(kThreads), 0, stream, (const float*)t1, t2, nc * d, h, H, (long)W,
                     host_scale(H, h, align_corners), align_corners, 0, 0L, 0L, 0L, C, 0.f, vox_step((long)g2 * kThreads, W, H, 0x7fffffff));
  hipLaunchKernelGGL(resize1d_bwd_kernel, dim3(g3), dim3(kThreads), 0, stream, (const float*)t2, gin, nc, d, D,
                     (long)H * W, host_scale(D, d, align_corners), align_corners, 1, isn, isc, isp, C, beta,
                     vox_step((long)g3 * kThreads, H * W, D, 0x7fffffff));
  MIREG_LAUNCH_RET();
}

int mireg_stn3d_fwd(const float* flow, long fsb, long fsc, long fsp, const float* frame, float* warped, int B, int C, int d,
                    int h, int w, hipStream_t stream) {
  MIREG_CHECK_ARG(flow && frame && warped && B > 0 && C > 0 && d > 0 && h > 0 && w > 0);
  const int g = grid_for((long)B * d * h * w);
  hipLaunchKernelGGL(stn3d_fwd_kernel, dim3(g), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, frame,
                     warped, B, C, d, h, w, vox_step((long)g * kThreads, w, h, d));
  MIREG_LAUNCH_RET();
}

int mireg_stn3d_bwd(const float* flow, long fsb, long fsc, long fsp, const float* frame, const float* gout, float* gflow,
                    float beta, int B, int C, int d, int h, int w, hipStream_t stream) {
  MIREG_CHECK_ARG(flow && frame && gout && gflow && B > 0 && C > 0 && d > 0 && h > 0 && w > 0);
  const int g = grid_for((long)B * d * h * w);
  hipLaunchKernelGGL(stn3d_bwd_kernel, dim3(g), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, frame,
                     gout, gflow, beta, B, C, d, h, w, vox_step((long)g * kThreads, w, h, d));
  MIREG_LAUNCH_RET();
}

int mireg_smoothness3d_fwd(const float* flow, long fsb, long fsc, long fsp, double* sum, int B, int d, int h, int w,
                           hipStream_t stream) {
  MIREG_CHECK_ARG(flow && sum && B > 0 && d > 0 && h > 0 && w > 0);
  const int g = grid_for((long)B * d * h * w, 1024);
  hipLaunchKernelGGL(smooth3d_fwd_kernel, dim3(g), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp,
                     sum, B, d, h, w, vox_step((long)g * kThreads, w, h, d));
  MIREG_LAUNCH_RET();
}

int mireg_smoothness3d_bwd(const float* flow, long fsb, long fsc, long fsp, const float* coef, float* gflow, float beta, int B,
                           int d, int h, int w, hipStream_t stream) {
  MIREG_CHECK_ARG(flow && coef && gflow && B > 0 && d > 0 && h > 0 && w > 0);
  const int g = grid_for((long)B * d * h * w);
  hipLaunchKernelGGL(smooth3d_bwd_kernel, dim3(g), dim3(kThreads), 0, stream, flow, fsb, fsc, fsp, coef,
                     gflow, beta, B, d, h, w, vox_step((long)g * kThreads, w, h, d));
  MIREG_LAUNCH_RET();
}


int mireg_stem3d_gather(const float* x, void* dst, int B, int C, int D, int H, int W, int k, int stride, int pad, int Cpad,
                        int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(x && dst && B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0 && Cpad >= k * C && Cpad % 8 == 0);
  MIREG_CHECK_ARG(dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
  const int Wo = (W + 2 * pad - k) / stride + 1;
  MIREG_CHECK_ARG(Wo > 0);
  const long total = (long)B * D * H * Wo * (Cpad / 8);
  const int g = grid_for(total, 16384);
  const Vox step = vox_step((long)g * kThreads, Wo * (Cpad / 8), H, D);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((stem3d_gather_kernel<__bf16>), dim3(g), dim3(kThreads), 0, stream, x,
                       reinterpret_cast<__bf16*>(dst), B, C, D, H, W, Wo, k, stride, pad, Cpad, step);
  else
    hipLaunchKernelGGL((stem3d_gather_kernel<float>), dim3(g), dim3(kThreads), 0, stream, x,
                       reinterpret_cast<float*>(dst), B, C, D, H, W, Wo, k, stride, pad, Cpad, step);
  MIREG_LAUNCH_RET();
}

int mireg_tiny_deconv3d_blocks(int B, int Dc, int Hc, int Wc) {
  const long n = (long)B * Dc * Hc * Wc;
  const long g = (n + 63) / 64;
  return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}

int mireg_tiny_deconv3d_fwd(const void* x_coarse, long ld_c, const float* w, void* y_fine, long ld_f, int B, int Dc, int Hc, int Wc,
                            int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(x_coarse && w && y_fine && ld_c >= 3 && ld_f >= 3 && B > 0 && Dc > 0 && Hc > 0 && Wc > 0);
  MIREG_CHECK_ARG(dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
  const int g = grid_for((long)B * Dc * Hc * Wc * 8);
  const Vox step = vox_step((long)g * kThreads, 2 * Wc, 2 * Hc, 2 * Dc);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((tiny_deconv3d_fwd_kernel<__bf16>), dim3(g), dim3(kThreads), 0, stream, (const __bf16*)x_coarse, ld_c, w, (__bf16*)y_fine, ld_f, B, Dc, Hc, Wc, step);
  else
    hipLaunchKernelGGL((tiny_deconv3d_fwd_kernel<float>), dim3(g), dim3(kThreads), 0, stream, (const float*)x_coarse, ld_c, w, (float*)y_fine, ld_f, B, Dc, Hc, Wc, step);
  MIREG_LAUNCH_RET();
}

int mireg_tiny_deconv3d_bwd_data(const void* g_fine, long ld_f, const float* w, void* dx_coarse, long ld_c, int accumulate, int B, int Dc,
                                 int Hc, int Wc, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(g_fine && w && dx_coarse && ld_c >= 3 && ld_f >= 3 && B > 0 && Dc > 0 && Hc > 0 && Wc > 0);
  MIREG_CHECK_ARG(dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
  const int g = grid_for((long)B * Dc * Hc * Wc);
  const Vox step = vox_step((long)g * kThreads, Wc, Hc, Dc);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((tiny_conv3d_fwd_kernel<__bf16>), dim3(g), dim3(kThreads), 0, stream, (const __bf16*)g_fine, ld_f, w, (__bf16*)dx_coarse, ld_c, accumulate, B, Dc, Hc, Wc, step);
  else
    hipLaunchKernelGGL((tiny_conv3d_fwd_kernel<float>), dim3(g), dim3(kThreads), 0, stream, (const float*)g_fine, ld_f, w, (float*)dx_coarse, ld_c, accumulate, B, Dc, Hc, Wc, step);
  MIREG_LAUNCH_RET();
}

int mireg_tiny_deconv3d_bwd_weights(const void* g_fine, long ld_f, const void* x_coarse, long ld_c, float* slab, int nblocks, int Cpad, int B,
                                    int Dc, int Hc, int Wc, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(g_fine && x_coarse && slab && ld_c >= 3 && ld_f >= 3 && B > 0 && Dc > 0 && Hc > 0 && Wc > 0 && Cpad >= 3);
  MIREG_CHECK_ARG(nblocks == mireg_tiny_deconv3d_blocks(B, Dc, Hc, Wc) && (dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32));
  const long n = (long)B * Dc * Hc * Wc;
  const int per = (int)((n + nblocks - 1) / nblocks);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((tiny_wgrad3d_kernel<__bf16>), dim3(nblocks), dim3(kTinyW3Threads), 0, stream, (const __bf16*)g_fine, ld_f, (const __bf16*)x_coarse, ld_c, slab, Cpad, B, Dc, Hc, Wc, per);
  else
    hipLaunchKernelGGL((tiny_wgrad3d_kernel<float>), dim3(nblocks), dim3(kTinyW3Threads), 0, stream, (const float*)g_fine, ld_f, (const float*)x_coarse, ld_c, slab, Cpad, B, Dc, Hc, Wc, per);
  MIREG_LAUNCH_RET();
}


}  // extern "C"
