// correlation.hip -- cost volume (K7/K8) and PWC feature warp (K10) for gfx950, NHWC views.
//
// Correlation(pad=md, kernel=1, max_displacement=md, stride1=1, stride2=s2) as the reference uses it at
// flownet2/networks/FlowNetC.py:31,88 (md=20, s2=2 -> 441 ch) and PWC/models/PWCNet.py:69,200-259 (md=4, s2=1 -> 81 ch):
//   out[b,y,x,(dy+R)*D+(dx+R)] = act( (1/C) * sum_c f1[b,y,x,c] * f2[b,y+s2*dy,x+s2*dx,c] ),  zero outside f2
// The third-party CUDA kernel (one 32-thread block per output pixel, serial over displacements) is latency bound
// and re-reads f2 D^2 times.  Here every (b, y, dy) is a 32 x 32 x C product  f1row . f2row^T  on the matrix
// cores (one wave each, operands straight from L1/L2 as 16-byte fragments): the band |x'-x| <= md with matching
// stride2 parity is then picked out of the 32x32 tile through LDS and written as coalesced channel runs.  That
// turns 441 (81) scalar dot products per pixel into 21 (9) MFMA tiles and leaves the kernel bound by the output
// write (HBM), which a VALU + shuffle formulation cannot reach at fp32 rate.
#include "mireg_common.h"
#include "../../include/mireg.h"

#include <cstdlib>

using namespace mireg;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

#if defined(__HIP_DEVICE_COMPILE__)
#define GPTR(T, p) (reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p)))
#else
#define GPTR(T, p) (reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p)))
#endif

__device__ __forceinline__ float ldf(const float* p) { return *GPTR(const float, p); }
__device__ __forceinline__ float ldf(const __bf16* p) { return (float)*GPTR(const __bf16, p); }
__device__ __forceinline__ void stf(float* p, float v) { *GPTR(float, p) = v; }
__device__ __forceinline__ void stf(__bf16* p, float v) { *GPTR(__bf16, p) = (__bf16)v; }
// one 16-byte channel granule -> floats
__device__ __forceinline__ void ld_granule(const float* p, float (&v)[4]) {
  const float4 q = *GPTR(const float4, p);
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void ld_granule(const __bf16* p, float (&v)[8]) {
  const uint4 q = *GPTR(const uint4, p);
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

// one block = one (b, y, 32-pixel x tile): its 4 waves split the D displacement rows, every wave turns a
// (dy, x' tile) into one 32x32xC MFMA product and drops the band into the block's LDS output tile [32 px][D*D];
// the finished tile leaves as whole channel rows (D*D contiguous elements per pixel), i.e. full cache lines.
template <typename T>
__global__ void __launch_bounds__(256)
correlation_fwd_kernel(const T* __restrict__ f1, long ld1, const T* __restrict__ f2, long ld2, T* __restrict__ out, long ldo,
                       int B, int H, int W, int C, int c_norm, int R, int s2, float slope) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int D = 2 * R + 1, DD = D * D;
  float (*tile)[33] = reinterpret_cast<float (*)[33]>(dyn + wid * 32 * 33 * 4);     // per-wave 32x32 staging
  constexpr int VEC = 16 / (int)sizeof(T);
  const int DDp = (DD + VEC - 1) / VEC * VEC;                                       // row stride: 16-byte aligned rows
  T* otile = reinterpret_cast<T*>(dyn + 4 * 32 * 33 * 4);                            // [32][DDp]
  const int xt = (W + 31) / 32;
  const float inv_c = 1.f / (float)c_norm;
  const int u = blockIdx.x;
  const int x0 = (u % xt) * 32;
  const int y = (u / xt) % H;
  const int b = u / (xt * H);
  const int xa = x0 + r;                                     // this lane's f1 pixel (A row)
  const T* a_row = f1 + (((long)b * H + y) * W + min(xa, W - 1)) * ld1;
  const bool a_ok = xa < W;
  for (int e = threadIdx.x; e < 32 * DDp / VEC; e += 256)                 // out-of-image displacements stay zero
    reinterpret_cast<uint4*>(otile)[e] = make_uint4(0, 0, 0, 0);
  uint4 a_frag[16];                                          // f1 fragments of channels 0..255 (bf16 path)
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kk = 16 * i + 8 * h;
      a_frag[i] = make_uint4(0, 0, 0, 0);
      if (16 * i < C) {                                        // block-uniform: PWC's 32..196-channel levels use 2..13 of the 16
        const bool ok = a_ok && kk < C;
        const uint4 v = *GPTR(const uint4, a_row + (ok ? kk : 0));
        if (ok) a_frag[i] = v;
      }
    }
  }
  __syncthreads();
  for (int dyi = wid; dyi < D; dyi += 4) {
    const int yy = y + (dyi - R) * s2;
    if (yy < 0 || yy >= H) continue;
    const int t_lo = max(0, (x0 - R * s2) >> 5), t_hi = min(xt - 1, (x0 + 31 + R * s2) >> 5);
    for (int t = t_lo; t <= t_hi; ++t) {
      const int xb = t * 32 + r;                             // this lane's f2 pixel (B column)
      const T* b_row = f2 + (((long)b * H + yy) * W + min(xb, W - 1)) * ld2;
      const bool b_ok = xb < W;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      if constexpr (sizeof(T) == 2) {
        // 256 channels at a time: all 16 f2 fragments of the row are requested before the first MFMA (one round of
        // memory latency per displacement row instead of one per K-step); the f1 fragments (same for every dy and
        // f2 tile) live in registers across the whole block
        for (int kb = 0; kb < C; kb += 256) {
          uint4 bv[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int kk = kb + 16 * i + 8 * h;
            bv[i] = make_uint4(0, 0, 0, 0);
            if (kb + 16 * i < C) {                                 // block-uniform
              const bool ok = b_ok && kk < C;
              const uint4 v = *GPTR(const uint4, b_row + (ok ? kk : 0));
              if (ok) bv[i] = v;
            }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (kb + 16 * i < C) {                             // block-uniform
              uint4 av;
              if (kb == 0) av = a_frag[i];
              else {
                const int kk = kb + 16 * i + 8 * h;
                const bool ok = a_ok && kk < C;
                const uint4 v = *GPTR(const uint4, a_row + (ok ? kk : 0));
                av = ok ? v : make_uint4(0, 0, 0, 0);
              }
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv[i]), acc, 0, 0, 0);
            }
          }
        }
      } else {
#pragma unroll 2
        for (int k = 0; k < C; k += 8) {                     // lane (r,h) feeds K slots k+4h..k+4h+3, one per MFMA
          const int kk = k + 4 * h;
          float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
          if (a_ok && kk < C) av = *GPTR(const float4, a_row + kk);
          if (b_ok && kk < C) bv = *GPTR(const float4, b_row + kk);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
      }
      // acc[e] = <f1[x0 + row], f2[32 t + col]>, col = lane&31, row = (e&3) + 8*(e>>2) + 4*h
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[(e & 3) + 8 * (e >> 2) + 4 * h][r] = acc[e];
      __builtin_amdgcn_wave_barrier();
      for (int px = h; px < 32; px += 2) {                   // lanes 0..31 of each half walk dx for one pixel
        if (r < D) {
          const int xx = x0 + px + (r - R) * s2;             // f2 pixel of displacement dxi = r
          if (xx >= t * 32 && xx < t * 32 + 32 && xx < W) {
            float v = tile[px][xx - t * 32] * inv_c;
            v = v > 0.f ? v : v * slope;
            otile[px * DDp + dyi * D + r] = (T)v;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  T* o_base = out + (((long)b * H + y) * W + x0) * ldo;
  // one pixel row per wave pass: 16-byte stores for the full granules (whole cache lines leave per instruction), scalar tail
  const bool vec = (reinterpret_cast<uintptr_t>(out) % 16) == 0 && (ldo % VEC) == 0;
  const int full = vec ? DD / VEC : 0;
  for (int px = wid; px < 32; px += 4) {
    if (x0 + px >= W) break;
    T* dst = o_base + (long)px * ldo;
    const T* src = otile + px * DDp;
    for (int j = lane; j < full; j += 64) *GPTR(uint4, dst + j * VEC) = *reinterpret_cast<const uint4*>(src + j * VEC);
    for (int ch = full * VEC + lane; ch < DD; ch += 64) *GPTR(T, dst + ch) = src[ch];
  }
}

// ---- vector-ALU forward for the few-channel, few-displacement cost volumes (PWC: 32..196 channels, 81 displacements) --------
// With C = 32 a 32x32xC MFMA tile is two instructions of useful work per staged row while the band extraction, the per-tile
// loads and the block bookkeeping stay the same: the MFMA kernel above takes 217 us for PWC's level 2 (57 MB in and out).  Here a
// block owns one output row and 64 pixels: the f1 row segment sits in LDS, the f2 row of each displacement row is staged next to it
// (zero outside the image), and every (pixel, dx) pair is one dot product over the channel granules (v_dot2_f32_bf16 / fma) read
// as 16-byte LDS vectors; results collect in an LDS output tile that leaves as whole rows.  LDS rows are padded by one granule.
__device__ __forceinline__ float dot_granule(const uint4 a, const uint4 b, float acc, __bf16) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a.x), __builtin_bit_cast(bf2_t, b.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a.y), __builtin_bit_cast(bf2_t, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a.z), __builtin_bit_cast(bf2_t, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a.w), __builtin_bit_cast(bf2_t, b.w), acc, false);
  return acc;
}
__device__ __forceinline__ float dot_granule(const uint4 a, const uint4 b, float acc, float) {
  acc = fmaf(__uint_as_float(a.x), __uint_as_float(b.x), acc);
  acc = fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
  acc = fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
  acc = fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
  return acc;
}

constexpr int kCorrXT = 64;                                          // output pixels per block of the vector-ALU kernel

template <typename T>
__global__ void __launch_bounds__(256)
correlation_fwd_valu_kernel(const T* __restrict__ f1, long ld1, const T* __restrict__ f2, long ld2, T* __restrict__ out, long ldo,
                            int B, int H, int W, int C, int c_norm, int R, int s2, float slope) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  constexpr int VEC = 16 / (int)sizeof(T), XT = kCorrXT;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int D = 2 * R + 1, DD = D * D, DDp = (DD + VEC - 1) / VEC * VEC;
  const int G = C / VEC, GS = G + 1, XW = XT + 2 * R * s2;
  uint4* at = reinterpret_cast<uint4*>(dyn);                        // [XT][GS]  f1 row segment
  uint4* bt = at + XT * GS;                                          // [XW][GS]  f2 row of the current displacement row
  T* otile = reinterpret_cast<T*>(bt + XW * GS);                    // [XT][DDp]
  const int xt = (W + XT - 1) / XT;
  const int u = blockIdx.x;
  const int x0 = (u % xt) * XT, y = (u / xt) % H, b = u / (xt * H);
  const float inv_c = 1.f / (float)c_norm;
  const T* row1 = f1 + (((long)b * H + y) * W) * ld1;
  for (int e = threadIdx.x; e < XT * G; e += 256) {
    const int px = e / G, gq = e - px * G, x = x0 + px;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (x < W) v = *GPTR(const uint4, row1 + (long)x * ld1 + gq * VEC);
    at[px * GS + gq] = v;
  }
  for (int e = threadIdx.x; e < XT * DDp / VEC; e += 256) reinterpret_cast<uint4*>(otile)[e] = make_uint4(0, 0, 0, 0);
  for (int dyi = 0; dyi < D; ++dyi) {
    const int yy = y + (dyi - R) * s2;
    if (yy < 0 || yy >= H) continue;                                 // block-uniform: that displacement row stays zero
    __syncthreads();                                                 // the previous row's dot products are done with bt
    const T* row2 = f2 + (((long)b * H + yy) * W) * ld2;
    for (int e = threadIdx.x; e < XW * G; e += 256) {
      const int kk = e / G, gq = e - kk * G, x = x0 - R * s2 + kk;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (x >= 0 && x < W) v = *GPTR(const uint4, row2 + (long)x * ld2 + gq * VEC);
      bt[kk * GS + gq] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < XT * D; i += 256) {
      const int px = i / D, dxi = i - px * D;                       // f2 pixel x0 + px + (dxi - R) s2 = staged column px + dxi s2
      const uint4* pa = at + px * GS;
      const uint4* pb = bt + (px + dxi * s2) * GS;
      float acc = 0.f;
      for (int gq = 0; gq < G; ++gq) acc = dot_granule(pa[gq], pb[gq], acc, T());
      float v = acc * inv_c;
      v = v > 0.f ? v : v * slope;
      otile[px * DDp + dyi * D + dxi] = (T)v;
    }
  }
  __syncthreads();
  T* o_base = out + (((long)b * H + y) * W + x0) * ldo;
  const bool vec = (reinterpret_cast<uintptr_t>(out) % 16) == 0 && (ldo % VEC) == 0;
  const int full = vec ? DD / VEC : 0;
  for (int px = wid; px < XT; px += 4) {
    if (x0 + px >= W) break;
    T* dst = o_base + (long)px * ldo;
    const T* src = otile + px * DDp;
    for (int j = lane; j < full; j += 64) *GPTR(uint4, dst + j * VEC) = *reinterpret_cast<const uint4*>(src + j * VEC);
    for (int ch = full * VEC + lane; ch < DD; ch += 64) *GPTR(T, dst + ch) = src[ch];
  }
}

// Cost-volume backward.  With G = d loss / d corr (LeakyReLU backward already applied by the caller):
//   WHICH 0: dF1[b,y,x,c]   = (1/C) sum_{dy,dx} G[b,y,x,(dy,dx)]               * F2[b, y+s2*dy, x+s2*dx, c]
//   WHICH 1: dF2[b,y',x',c] = (1/C) sum_{dy,dx} G[b,y'-s2*dy,x'-s2*dx,(dy,dx)] * F1[b, y'-s2*dy, x'-s2*dx, c]
// Per (b, row, dy) this is  Gband[32 x 32] . Frow[32 x C]  with Gband the banded matrix of displacement gradients,
// run on the exact-fp32 MFMA (32x32x2, operands converted on load) so that bf16 storage needs no LDS transpose:
// lane (r,h) feeds A[m=r][k=h] (a 2-byte/4-byte gather from G) and B[k=h][n=r] (32 consecutive channels).
template <typename T, int WHICH>
__global__ void __launch_bounds__(256)
correlation_bwd_kernel(const T* __restrict__ g, long ldg, const T* __restrict__ fo, long ldo_, T* __restrict__ dout, long ldd,
                       int B, int H, int W, int C, int c_norm, int R, int s2, int accumulate) {
  constexpr int NT = 8;                                        // up to 256 channels per pass
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int D = 2 * R + 1;
  const int xt = (W + 31) / 32;
  const long units = (long)B * H * xt;
  const float inv_c = 1.f / (float)c_norm;
  for (long u = (long)blockIdx.x * 4 + wid; u < units; u += (long)gridDim.x * 4) {
    const int x0 = (int)(u % xt) * 32;
    const int y = (int)((u / xt) % H);
    const int b = (int)(u / ((long)xt * H));
    const int xm = x0 + r;                                     // the pixel of accumulator row r
    for (int c0 = 0; c0 < C; c0 += 32 * NT) {
      const int nt = min(NT, (C - c0 + 31) / 32);
      f32x16 acc[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      for (int dyi = 0; dyi < D; ++dyi) {
        const int yy = WHICH == 0 ? y + (dyi - R) * s2 : y - (dyi - R) * s2;     // row of the other feature map
        if (yy < 0 || yy >= H) continue;
        const int gy = WHICH == 0 ? y : yy;                    // row of G
        const int k_lo = max(0, x0 - R * s2), k_hi = min(W - 1, x0 + 31 + R * s2);
        for (int k0 = k_lo & ~1; k0 <= k_hi; k0 += 2) {
          const int k = k0 + h;                                // this lane's K index = pixel of the other map
          float a = 0.f;
          if (xm < W && k >= 0 && k < W) {
            const int diff = WHICH == 0 ? k - xm : xm - k;     // s2 * dx
            const int gx = WHICH == 0 ? xm : k;                // pixel of G
            if (diff % s2 == 0) {
              const int dxi = diff / s2 + R;
              if (dxi >= 0 && dxi < D) a = ldf(g + (((long)b * H + gy) * W + gx) * ldg + dyi * D + dxi);
            }
          }
          const T* frow = fo + (((long)b * H + yy) * W + min(max(k, 0), W - 1)) * ldo_ + c0;
          const bool kok = k >= 0 && k < W;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (j < nt) {
              const int c = j * 32 + r;
              const float bv = (kok && c0 + c < C) ? ldf(frow + c) : 0.f;
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j >= nt) continue;
        const int c = c0 + j * 32 + r;
        if (c >= C) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int xo = x0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (xo >= W) continue;
          T* d = dout + (((long)b * H + y) * W + xo) * ldd + c;
          float v = acc[j][e] * inv_c;
          if (accumulate) v += ldf(d);
          stf(d, v);
        }
      }
    }
  }
}

// bf16 cost-volume backward on the bf16 matrix cores.  Block = one (b, output row y, 32-pixel x tile); per valid
// displacement row dyi the 4 waves cooperatively stage in LDS
//   Bt[kk][c]  = the other feature map's row yy, pixels klo..klo+KW-1 (channel-contiguous, as in HBM), and
//   At[m][kk]  = the banded displacement-gradient matrix (G gathered once per block, shared by all channel tiles);
// A fragments are plain 16-byte LDS reads, B fragments come through the gfx950 transposing read ds_read_b64_tr_b16
// (K = pixels is the strided dimension in memory), and every wave owns the 32-channel tiles j = wid, wid+4, ...
// so one A fragment feeds all of a wave's MFMAs of a K-step.  Same sums as correlation_bwd_kernel, fp32 accumulate.
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int WHICH, int NTW, int NB>
__global__ void __launch_bounds__(256)
correlation_bwd_mfma_kernel(const __bf16* __restrict__ g, long ldg, const __bf16* __restrict__ fo, long ldo_, __bf16* __restrict__ dout,
                            long ldd, int B, int H, int W, int C, int c_norm, int R, int s2, int accumulate, int KW, int LDB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  __bf16* Bt = reinterpret_cast<__bf16*>(dyn);                    // [KW][LDB]
  __bf16* At = Bt + (long)KW * LDB;                               // [32][KW + 8]
  const int LDA = KW + 8;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int D = 2 * R + 1;
  const int xt = (W + 31) / 32;
  const int u = blockIdx.x;
  const int x0 = (u % xt) * 32, y = (u / xt) % H, b = u / (xt * H);
  const int klo = max(0, x0 - R * s2) & ~15;                      // first staged pixel of the other map (16-aligned)
  const int nsteps = KW / 16;
  const int ntiles = (C + 31) / 32;
  const float inv_c = 1.f / (float)c_norm;
  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  // transposing-read geometry (see conv_wgrad_kernel): 16-lane group gq covers columns 16*(gq&1).., rows 8*(gq>>1)+q(+4)
  const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int rowoff = 8 * (gq >> 1) + q, coloff = 16 * (gq & 1) + 4 * pp;
  const int cgran = LDB / 8;                                       // 16-byte granules per staged row (incl. zero pad columns)

  // The band's nonzero pattern (which (m, kk) pair up) does not depend on dyi: zero the tile once, rewrite only the band.
  for (int e = threadIdx.x; e < 32 * LDA; e += 256) At[e] = (__bf16)0.f;
  constexpr int NA = 4;                                            // register staging: 256 * NB granules >= KW * cgran, 256 * NA >= 32 * D
  // two register sets: the loads of row dyi + 2 are issued while row dyi is multiplied, so a row's global-load latency is hidden
  // behind two rows of staging + MFMA instead of one (the loop was one memory round trip per displacement row)
  // (only where a row is few granules per thread: with 14 slots a second set would not fit the register file)
  constexpr bool TWO = NB <= 5;
  uint4 breg0[NB], breg1[TWO ? NB : 1];
  __bf16 areg0[NA], areg1[TWO ? NA : 1];
  // Per-slot addressing is fixed for the whole block: source offset (0 = a harmless in-range dummy for masked slots), LDS destination
  // (-1 = none) and the zero mask are computed once, so fetch() is nothing but unconditional loads (straight-line code: the compiler
  // turned per-slot conditions into a branch and a full s_waitcnt per slot) and commit() a counted wait plus LDS stores.
  int bsrc[NB], bdst[NB], asrc[NA], adst[NA];
  unsigned bzero = 0u, azero = 0u;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int kk = e / cgran, cg = e - kk * cgran, k = klo + kk;
    const bool in = e < KW * cgran, ok = in && k < W && cg * 8 < C;
    bsrc[i] = ok ? k * (int)ldo_ + cg * 8 : 0;
    bdst[i] = in ? kk * LDB + cg * 8 : -1;
    if (!ok) bzero |= 1u << i;
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int m = e / D, dxi = e - m * D, xm = x0 + m;
    const int k = WHICH == 0 ? xm + (dxi - R) * s2 : xm - (dxi - R) * s2;       // the other map's pixel this entry pairs with
    const bool ok = e < 32 * D && xm < W && k >= 0 && k < W;
    asrc[i] = ok ? (WHICH == 0 ? xm : k) * (int)ldg + dxi : 0;
    adst[i] = (e < 32 * D && k >= klo && k < klo + KW) ? m * LDA + (k - klo) : -1;
    if (!ok) azero |= 1u << i;
  }
  auto fetch = [&](int dyi, auto& breg, auto& areg) {              // global loads only, consumed one or two iterations later
    const int yy = WHICH == 0 ? y + (dyi - R) * s2 : y - (dyi - R) * s2;
    const __bf16* frow = fo + (((long)b * H + yy) * W) * ldo_;
#pragma unroll
    for (int i = 0; i < NB; ++i) breg[i] = *GPTR(const uint4, frow + bsrc[i]);
    const __bf16* grow = g + (((long)b * H + (WHICH == 0 ? y : yy)) * W) * ldg + dyi * D;
#pragma unroll
    for (int i = 0; i < NA; ++i) areg[i] = *GPTR(const __bf16, grow + asrc[i]);
  };
  auto commit = [&](const auto& breg, const auto& areg) {          // registers -> LDS tiles
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (bdst[i] >= 0) *reinterpret_cast<uint4*>(Bt + bdst[i]) = ((bzero >> i) & 1u) ? make_uint4(0u, 0u, 0u, 0u) : breg[i];
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (adst[i] >= 0) At[adst[i]] = ((azero >> i) & 1u) ? (__bf16)0.f : areg[i];
  };
  // valid displacement rows form one interval [d_lo, d_hi]
  int d_lo = 0, d_hi = D - 1;
  while (d_lo < D) { const int yy = WHICH == 0 ? y + (d_lo - R) * s2 : y - (d_lo - R) * s2; if (yy >= 0 && yy < H) break; ++d_lo; }
  while (d_hi >= 0) { const int yy = WHICH == 0 ? y + (d_hi - R) * s2 : y - (d_hi - R) * s2; if (yy >= 0 && yy < H) break; --d_hi; }
  auto multiply = [&]() {
    for (int ks = 0; ks < nsteps; ++ks) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(At + r * LDA + ks * 16 + 8 * h);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int tile = wid + 4 * j;
        if (tile < ntiles) {                                                      // wave-uniform
          const __bf16* base = Bt + (long)(ks * 16 + rowoff) * LDB + tile * 32 + coloff;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 4 * LDB));
          const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, v), acc[j], 0, 0, 0);
        }
      }
    }
  };
  if (d_lo <= d_hi) fetch(d_lo, breg0, areg0);
  if constexpr (TWO) {
    if (d_lo + 1 <= d_hi) fetch(d_lo + 1, breg1, areg1);
    for (int dyi = d_lo; dyi <= d_hi; dyi += 2) {
      __syncthreads();                                             // previous row's fragments are consumed (and the zero fill landed)
      commit(breg0, areg0);
      __syncthreads();
      if (dyi + 2 <= d_hi) fetch(dyi + 2, breg0, areg0);
      multiply();
      if (dyi + 1 > d_hi) break;
      __syncthreads();
      commit(breg1, areg1);
      __syncthreads();
      if (dyi + 3 <= d_hi) fetch(dyi + 3, breg1, areg1);
      multiply();
    }
  } else {
    for (int dyi = d_lo; dyi <= d_hi; ++dyi) {
      __syncthreads();
      commit(breg0, areg0);
      __syncthreads();
      if (dyi < d_hi) fetch(dyi + 1, breg0, areg0);                // next row's loads fly under this row's MFMAs
      multiply();
    }
  }
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int c = (wid + 4 * j) * 32 + r;
    if (wid + 4 * j >= ntiles || c >= C) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int xo = x0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (xo >= W) continue;
      __bf16* d = dout + (((long)b * H + y) * W + xo) * ldd + c;
      float v = acc[j][e] * inv_c;
      if (accumulate) v += ldf(d);
      stf(d, v);
    }
  }
}

// ---- vector-ALU backward for PWC's cost volumes (81 displacements, up to 128 channels) ------------------------------------------
// One thread per (output pixel, 16-byte channel granule), XT = floor(256 / granules) pixels per block; per displacement row the other
// map's row segment (and, for dF2, that row's gradient entries) is staged in LDS, and every (dy, dx) adds  g * granule  to the
// thread's fp32 accumulators.  Same sums as the kernels above (fp32 accumulate, fixed order dy-major), one 16-byte store per thread.
template <typename T, int WHICH>
__global__ void __launch_bounds__(256)
correlation_bwd_valu_kernel(const T* __restrict__ g, long ldg, const T* __restrict__ fo, long ldo_, T* __restrict__ dout, long ldd,
                            int B, int H, int W, int C, int c_norm, int R, int s2, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  constexpr int VEC = 16 / (int)sizeof(T);
  const int D = 2 * R + 1, DD = D * D;
  const int G = C / VEC, XT = 256 / G, XW = XT + 2 * R * s2, GS = G + 1;
  uint4* bt = reinterpret_cast<uint4*>(dyn);                        // [XW][GS]  the other map's row segment
  T* gt = reinterpret_cast<T*>(bt + XW * GS);                       // WHICH 0: [XT][DD] own gradient rows; WHICH 1: [XW][D] entries of row dyi
  const int xt = (W + XT - 1) / XT;
  const int u = blockIdx.x;
  const int x0 = (u % xt) * XT, y = (u / xt) % H, b = u / (xt * H);
  const int px = min(threadIdx.x / G, XT - 1), gq = threadIdx.x - (threadIdx.x / G) * G;
  const bool active = threadIdx.x < XT * G;                         // 256 is not a multiple of every granule count
  const float inv_c = 1.f / (float)c_norm;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (WHICH == 0) {
    const T* grow = g + (((long)b * H + y) * W) * ldg;
    for (int e = threadIdx.x; e < XT * DD; e += 256) {
      const int p = e / DD, ch = e - p * DD;
      gt[e] = x0 + p < W ? *GPTR(const T, grow + (long)(x0 + p) * ldg + ch) : (T)0.f;
    }
  }
  for (int dyi = 0; dyi < D; ++dyi) {
    const int yy = WHICH == 0 ? y + (dyi - R) * s2 : y - (dyi - R) * s2;
    if (yy < 0 || yy >= H) continue;                                 // block-uniform
    __syncthreads();
    const T* frow = fo + (((long)b * H + yy) * W) * ldo_;
    for (int e = threadIdx.x; e < XW * G; e += 256) {
      const int kk = e / G, q = e - kk * G, x = x0 - R * s2 + kk;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (x >= 0 && x < W) v = *GPTR(const uint4, frow + (long)x * ldo_ + q * VEC);
      bt[kk * GS + q] = v;
    }
    if (WHICH == 1) {
      const T* grow = g + (((long)b * H + yy) * W) * ldg + dyi * D;
      for (int e = threadIdx.x; e < XW * D; e += 256) {
        const int kk = e / D, dxi = e - kk * D, x = x0 - R * s2 + kk;
        gt[e] = (x >= 0 && x < W) ? *GPTR(const T, grow + (long)x * ldg + dxi) : (T)0.f;
      }
    }
    __syncthreads();
    for (int dxi = 0; dxi < D; ++dxi) {
      // dF1: pairs with f2 pixel x + (dxi - R) s2 = staged column px + dxi s2;  dF2: source pixel x - (dxi - R) s2 = column px + (2R - dxi) s2
      const int col = WHICH == 0 ? px + dxi * s2 : px + (2 * R - dxi) * s2;
      const float gv = (float)(WHICH == 0 ? gt[px * DD + dyi * D + dxi] : gt[col * D + dxi]);
      const uint4 v = bt[col * GS + gq];
      if constexpr (sizeof(T) == 2) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] = fmaf(gv, __uint_as_float(w[i] << 16), acc[2 * i]);
          acc[2 * i + 1] = fmaf(gv, __uint_as_float(w[i] & 0xffff0000u), acc[2 * i + 1]);
        }
      } else {
        acc[0] = fmaf(gv, __uint_as_float(v.x), acc[0]); acc[1] = fmaf(gv, __uint_as_float(v.y), acc[1]);
        acc[2] = fmaf(gv, __uint_as_float(v.z), acc[2]); acc[3] = fmaf(gv, __uint_as_float(v.w), acc[3]);
      }
    }
  }
  if (!active || x0 + px >= W) return;
  T* d = dout + (((long)b * H + y) * W + x0 + px) * ldd + gq * VEC;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    float v = acc[e] * inv_c;
    if (accumulate) v += ldf(d + e);
    stf(d + e, v);
  }
}

// PWCDCNet.warp (PWC/models/PWCNet.py:143-179): grid normalised with (W-1) but sampled with align_corners=False,
// so the tap coordinate is ((2(x+u)/(W-1) - 1 + 1) * W - 1) / 2; output * (bilinear(ones) >= 0.9999).
template <typename T>
__global__ void __launch_bounds__(256)
pwc_warp_fwd_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ flow, long ldf_, float flow_scale,
                    T* __restrict__ out, long ldo, int B, int H, int W, int C) {
  constexpr int V = 16 / (int)sizeof(T);
  const int cpr = C / V;
  const long total = (long)B * H * W * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
    const long pix = i / cpr;
    const int c0 = (int)(i - pix * cpr) * V;
    const int xq = (int)(pix % W), yq = (int)((pix / W) % H);
    const long img = pix / ((long)W * H);
    const float u = flow[pix * ldf_] * flow_scale, v = flow[pix * ldf_ + 1] * flow_scale;
    const float gx = 2.0f * ((float)xq + u) / (float)max(W - 1, 1) - 1.0f;
    const float gy = 2.0f * ((float)yq + v) / (float)max(H - 1, 1) - 1.0f;
    const float px = ((gx + 1.f) * (float)W - 1.f) / 2.f, py = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = px - fx, wy1 = py - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float acc[V], msk = 0.f;
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
      const float wgt = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
      if (xi < 0 || xi >= W || yi < 0 || yi >= H) continue;
      msk += wgt;
      const T* src = x + ((img * H + yi) * W + xi) * ldx + c0;
#pragma unroll
      for (int q = 0; q < V; ++q) acc[q] += ldf(src + q) * wgt;
    }
    const float m = msk < 0.9999f ? 0.f : 1.f;
    T* dst = out + pix * ldo + c0;
#pragma unroll
    for (int q = 0; q < V; ++q) stf(dst + q, acc[q] * m);
  }
}

// PWC warp backward (mask is a constant of the graph: PWCNet.py:174-177 overwrites it in place):
//   dx[tap_t][c] += m * w_t * g[c]            (fp32 atomics into a zeroed NHWC scratch, cast/added by the caller)
//   dflow       += scale * W/(W-1) * m * sum_c g[c] * sum_t dw_t/dpx * x[tap_t][c]     (likewise for y)
template <typename T>
__global__ void __launch_bounds__(256)
pwc_warp_bwd_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ flow, long ldf_, float flow_scale,
                    const T* __restrict__ g, long ldg, float* __restrict__ dx32, long lddx, float* __restrict__ dflow, long lddf,
                    int B, int H, int W, int C) {
  constexpr int V = 16 / (int)sizeof(T);
  const int cpr = C / V;
  const long total = (long)B * H * W * cpr;
  for (long i0 = (long)blockIdx.x * blockDim.x; i0 < total; i0 += (long)gridDim.x * blockDim.x) {   // wave-uniform trip count
#pragma clang fp contract(off)
    const bool active = i0 + threadIdx.x < total;
    const long i = active ? i0 + threadIdx.x : total - 1;
    const long pix = i / cpr;
    const int c0 = (int)(i - pix * cpr) * V;
    const int xq = (int)(pix % W), yq = (int)((pix / W) % H);
    const long img = pix / ((long)W * H);
    const float u = flow[pix * ldf_] * flow_scale, v = flow[pix * ldf_ + 1] * flow_scale;
    const float gx = 2.0f * ((float)xq + u) / (float)max(W - 1, 1) - 1.0f;
    const float gy = 2.0f * ((float)yq + v) / (float)max(H - 1, 1) - 1.0f;
    const float px = ((gx + 1.f) * (float)W - 1.f) / 2.f, py = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = px - fx, wy1 = py - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float msk = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
      if (xi >= 0 && xi < W && yi >= 0 && yi < H) msk += ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
    }
    const bool live = active && msk >= 0.9999f;          // masked pixel: no gradient at all
    float gv[V];
    ld_granule(g + pix * ldg + c0, gv);
    float dpx = 0.f, dpy = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int xi = x0 + (t & 1), yi = y0 + (t >> 1);
      if (!live || xi < 0 || xi >= W || yi < 0 || yi >= H) continue;
      const float wx = (t & 1) ? wx1 : wx0, wy = (t >> 1) ? wy1 : wy0;
      const float dwx = (t & 1) ? 1.f : -1.f, dwy = (t >> 1) ? 1.f : -1.f;
      const long tp = ((img * H + yi) * W + xi);
      float xv[V];
      ld_granule(x + tp * ldx + c0, xv);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        atomicAdd(dx32 + tp * lddx + c0 + q, gv[q] * wx * wy);
        dpx += gv[q] * xv[q] * dwx * wy;
        dpy += gv[q] * xv[q] * wx * dwy;
      }
    }
    // the channel granules of one pixel sit in consecutive lanes: segmented shuffle reduction, then ONE atomic pair per
    // pixel and wave instead of one per granule (they all hit the same two addresses)
    const int lane = threadIdx.x & 63;
    long key = active ? pix : -1 - (long)threadIdx.x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float ox = __shfl_down(dpx, off, 64), oy = __shfl_down(dpy, off, 64);
      const long ok_ = __shfl_down(key, off, 64);
      if (lane + off < 64 && ok_ == key) { dpx += ox; dpy += oy; }
    }
    const long prev = __shfl_up(key, 1, 64);
    if (live && (lane == 0 || prev != key)) {
      atomicAdd(dflow + pix * lddf, dpx * flow_scale * (float)W / (float)max(W - 1, 1));
      atomicAdd(dflow + pix * lddf + 1, dpy * flow_scale * (float)H / (float)max(H - 1, 1));
    }
  }
}

// ---- deterministic PWC warp backward (reference PWC/models/PWCNet.py:143-179 through autograd) ------------------------------
// d x is a scatter: destination pixel d adds w_t * g[d] to its four bilinear taps.  Instead of fp32 atomics (run-to-run
// different sums) the taps are bucketed by SOURCE pixel with integer atomics only -- count, exclusive scan, fill -- and a
// gather kernel sums every source's bucket in ascending (destination, tap) order: bit-reproducible, and the fp32 adds leave
// the memory side.  Buckets longer than kDetSort entries (flows collapsing hundreds of pixels onto one) are summed in fill
// order instead of paying an n^2 selection: reproducible whenever no bucket is that long.
constexpr int kDetSort = 48;

struct WarpTaps { int x0, y0; float wx0, wx1, wy0, wy1; bool live; };
__device__ __forceinline__ WarpTaps warp_taps(const float* __restrict__ flow, long ldf_, float flow_scale, long pix, int xq, int yq,
                                              int H, int W) {
#pragma clang fp contract(off)
  WarpTaps t;
  const float u = flow[pix * ldf_] * flow_scale, v = flow[pix * ldf_ + 1] * flow_scale;
  const float gx = 2.0f * ((float)xq + u) / (float)max(W - 1, 1) - 1.0f;
  const float gy = 2.0f * ((float)yq + v) / (float)max(H - 1, 1) - 1.0f;
  const float px = ((gx + 1.f) * (float)W - 1.f) / 2.f, py = ((gy + 1.f) * (float)H - 1.f) / 2.f;
  const float fx = floorf(px), fy = floorf(py);
  t.x0 = (int)fx; t.y0 = (int)fy;
  t.wx1 = px - fx; t.wy1 = py - fy; t.wx0 = 1.f - t.wx1; t.wy0 = 1.f - t.wy1;
  float msk = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int xi = t.x0 + (k & 1), yi = t.y0 + (k >> 1);
    if (xi >= 0 && xi < W && yi >= 0 && yi < H) msk += ((k & 1) ? t.wx1 : t.wx0) * ((k >> 1) ? t.wy1 : t.wy0);
  }
  t.live = msk >= 0.9999f;
  return t;
}

// pass 1 (FILL = false): cnt[src] += 1 per live tap;  pass 3 (FILL = true): entries[off[src] + --cnt[src]] = (dst*4 + tap, weight)
// -- the second pass counts cnt back down to zero, so the counters are clean for the next call
template <bool FILL>
__global__ void __launch_bounds__(256)
warp_bucket_kernel(const float* __restrict__ flow, long ldf_, float flow_scale, int* __restrict__ cnt, const int* __restrict__ off,
                   int2* __restrict__ entries, int B, int H, int W) {
  const long P = (long)B * H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < P; pix += (long)gridDim.x * blockDim.x) {
    const int xq = (int)(pix % W), yq = (int)((pix / W) % H);
    const long img = pix / ((long)W * H);
    const WarpTaps t = warp_taps(flow, ldf_, flow_scale, pix, xq, yq, H, W);
    if (!t.live) continue;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xi = t.x0 + (k & 1), yi = t.y0 + (k >> 1);
      if (xi < 0 || xi >= W || yi < 0 || yi >= H) continue;
      const long tp = (img * H + yi) * W + xi;
      if constexpr (!FILL) atomicAdd(cnt + tp, 1);
      else {
        const int slot = off[tp] + atomicSub(cnt + tp, 1) - 1;
        const float wgt = ((k & 1) ? t.wx1 : t.wx0) * ((k >> 1) ? t.wy1 : t.wy0);
        entries[slot] = make_int2((int)(pix * 4 + k), __float_as_int(wgt));
      }
    }
  }
}

// exclusive scan of cnt[0..P) into off[0..P] in three small launches: block sums, scan of the block sums, local scans
constexpr int kScanBlock = 2048;                                    // elements per block (256 threads x 8)
__global__ void __launch_bounds__(256) scan_sums_kernel(const int* __restrict__ cnt, int* __restrict__ bsum, long P) {
  __shared__ int red[256];
  const long base = (long)blockIdx.x * kScanBlock;
  int s = 0;
  for (int i = threadIdx.x; i < kScanBlock; i += 256) s += base + i < P ? cnt[base + i] : 0;
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(256) scan_top_kernel(int* __restrict__ bsum, int nb, int* __restrict__ total) {    // in place, exclusive; nb <= 256 * 64
  __shared__ int part[256];
  const int per = (nb + 255) / 256, b0 = threadIdx.x * per;
  int s = 0;
  for (int i = 0; i < per; ++i) s += b0 + i < nb ? bsum[b0 + i] : 0;
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; } *total = run; }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int i = 0; i < per && b0 + i < nb; ++i) { const int v = bsum[b0 + i]; bsum[b0 + i] = run; run += v; }
}
__global__ void __launch_bounds__(256) scan_local_kernel(const int* __restrict__ cnt, const int* __restrict__ bsum, int* __restrict__ off, long P) {
  __shared__ int part[256];
  const long base = (long)blockIdx.x * kScanBlock + threadIdx.x * 8;
  int v[8], s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = base + i < P ? cnt[base + i] : 0; s += v[i]; }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int run = bsum[blockIdx.x]; for (int i = 0; i < 256; ++i) { const int t = part[i]; part[i] = run; run += t; } }
  __syncthreads();
  int run = part[threadIdx.x];
#pragma unroll
  for (int i = 0; i < 8; ++i) { if (base + i < P) off[base + i] = run; run += v[i]; }
}

// dx32[src][c] = sum over the bucket of w * g[dst][c], ascending key order; one thread per (source pixel, channel granule)
template <typename T>
__global__ void __launch_bounds__(256)
warp_gather_kernel(const T* __restrict__ g, long ldg, const int* __restrict__ off, const int2* __restrict__ entries,
                   float* __restrict__ dx32, long lddx, long P, int C) {
  constexpr int V = 16 / (int)sizeof(T);
  const int cpr = C / V;
  const long total = P * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
    const long src = i / cpr;
    const int c0 = (int)(i - src * cpr) * V;
    const int b0 = off[src], n = off[src + 1] - b0;
    float acc[V];
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
    int last = -1;
    for (int it = 0; it < n; ++it) {
      int2 e;
      if (n <= kDetSort) {                                           // next entry in ascending key order
        int best = 0x7fffffff, bw = 0;
        for (int j = 0; j < n; ++j) { const int2 c = entries[b0 + j]; if (c.x > last && c.x < best) { best = c.x; bw = c.y; } }
        e = make_int2(best, bw);
        last = best;
      } else e = entries[b0 + it];
      const float w = __int_as_float(e.y);
      float gv[V];
      ld_granule(g + (long)(e.x >> 2) * ldg + c0, gv);
#pragma unroll
      for (int q = 0; q < V; ++q) acc[q] += gv[q] * w;
    }
    float* d = dx32 + src * lddx + c0;
#pragma unroll
    for (int q = 0; q < V; ++q) d[q] = acc[q];
  }
}

// d flow (per destination pixel, no scatter): the channel granules of a pixel sit in consecutive lanes -> segmented shuffle sum
template <typename T>
__global__ void __launch_bounds__(256)
warp_dflow_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ flow, long ldf_, float flow_scale,
                  const T* __restrict__ g, long ldg, float* __restrict__ dflow, long lddf, int B, int H, int W, int C) {
  constexpr int V = 16 / (int)sizeof(T);
  const int cpr = C / V;
  const long total = (long)B * H * W * cpr;
  for (long i0 = (long)blockIdx.x * blockDim.x; i0 < total; i0 += (long)gridDim.x * blockDim.x) {   // wave-uniform trip count
#pragma clang fp contract(off)
    const bool active = i0 + threadIdx.x < total;
    const long i = active ? i0 + threadIdx.x : total - 1;
    const long pix = i / cpr;
    const int c0 = (int)(i - pix * cpr) * V;
    const int xq = (int)(pix % W), yq = (int)((pix / W) % H);
    const long img = pix / ((long)W * H);
    const WarpTaps t = warp_taps(flow, ldf_, flow_scale, pix, xq, yq, H, W);
    const bool live = active && t.live;
    float gv[V];
    ld_granule(g + pix * ldg + c0, gv);
    float dpx = 0.f, dpy = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xi = t.x0 + (k & 1), yi = t.y0 + (k >> 1);
      if (!live || xi < 0 || xi >= W || yi < 0 || yi >= H) continue;
      const float wx = (k & 1) ? t.wx1 : t.wx0, wy = (k >> 1) ? t.wy1 : t.wy0;
      const float dwx = (k & 1) ? 1.f : -1.f, dwy = (k >> 1) ? 1.f : -1.f;
      float xv[V];
      ld_granule(x + ((img * H + yi) * W + xi) * ldx + c0, xv);
#pragma unroll
      for (int q = 0; q < V; ++q) { dpx += gv[q] * xv[q] * dwx * wy; dpy += gv[q] * xv[q] * wx * dwy; }
    }
    const int lane = threadIdx.x & 63;
    long key = active ? pix : -1 - (long)threadIdx.x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float ox = __shfl_down(dpx, o, 64), oy = __shfl_down(dpy, o, 64);
      const long ok_ = __shfl_down(key, o, 64);
      if (lane + o < 64 && ok_ == key) { dpx += ox; dpy += oy; }
    }
    const long prev = __shfl_up(key, 1, 64);
    // a pixel's granules span at most two waves (C <= 512): two adds onto the caller's zeros commute exactly
    if (live && (lane == 0 || prev != key)) {
      atomicAdd(dflow + pix * lddf, dpx * flow_scale * (float)W / (float)max(W - 1, 1));
      atomicAdd(dflow + pix * lddf + 1, dpy * flow_scale * (float)H / (float)max(H - 1, 1));
    }
  }
}

// dst[m][0..C) (+)= src[m][0..C)  (concat staging where a producer cannot write in place; any channel offset)
template <typename T>
__global__ void __launch_bounds__(256)
copy_channels_kernel(const T* __restrict__ src, long lds_, T* __restrict__ dst, long ldd, long M, int C, int accumulate) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C; const int c = (int)(i - m * C);
    const float v = ldf(src + m * lds_ + c) + (accumulate ? ldf(dst + m * ldd + c) : 0.f);
    stf(dst + m * ldd + c, v);
  }
}

}  // namespace

extern "C" {

int mireg_pwc_warp_bwd(const void* x, long ldx, const float* flow, long ldf_, float flow_scale, const void* g, long ldg,
                       float* dx32, long lddx, float* dflow, long lddf, int B, int H, int W, int C, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(x && flow && g && dx32 && dflow && B > 0 && H > 0 && W > 0 && C > 0 && ldf_ >= 2 && lddf >= 2);
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  MIREG_CHECK_ARG(C % V == 0 && ldx % V == 0 && ldg % V == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)g % 16 == 0);
  const long total = (long)B * H * W * (C / V);
  long gr = (total + 255) / 256;
  if (gr > 4096) gr = 4096;
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((pwc_warp_bwd_kernel<__bf16>), dim3((unsigned)gr), dim3(256), 0, stream, (const __bf16*)x, ldx, flow, ldf_, flow_scale, (const __bf16*)g, ldg, dx32, lddx, dflow, lddf, B, H, W, C);
  else if (dtype == MIREG_DTYPE_F32)
    hipLaunchKernelGGL((pwc_warp_bwd_kernel<float>), dim3((unsigned)gr), dim3(256), 0, stream, (const float*)x, ldx, flow, ldf_, flow_scale, (const float*)g, ldg, dx32, lddx, dflow, lddf, B, H, W, C);
  else return MIREG_ERR_ARG;
  MIREG_LAUNCH_RET();
}

int mireg_copy_channels(const void* src, long ld_s, void* dst, long ld_d, long M, int C, int accumulate, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(src && dst && M > 0 && C > 0);
  long g = (M * C + 255) / 256;
  if (g > 4096) g = 4096;
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((copy_channels_kernel<__bf16>), dim3((unsigned)g), dim3(256), 0, stream, (const __bf16*)src, ld_s, (__bf16*)dst, ld_d, M, C, accumulate);
  else hipLaunchKernelGGL((copy_channels_kernel<float>), dim3((unsigned)g), dim3(256), 0, stream, (const float*)src, ld_s, (float*)dst, ld_d, M, C, accumulate);
  MIREG_LAUNCH_RET();
}


int mireg_correlation_fwd(const void* f1, long ld1, const void* f2, long ld2, void* out, long ldo, int B, int H, int W,
                          int C, int c_norm, int max_displacement, int stride2, float slope, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(f1 && f2 && out && B > 0 && H > 0 && W > 0 && C > 0 && c_norm > 0 && stride2 > 0 && max_displacement >= 0);
  MIREG_CHECK_ARG(max_displacement % stride2 == 0 && max_displacement / stride2 <= 15);
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  MIREG_CHECK_ARG(C % V == 0 && ld1 % V == 0 && ld2 % V == 0 && (uintptr_t)f1 % 16 == 0 && (uintptr_t)f2 % 16 == 0);
  const long units = (long)B * H * ((W + 31) / 32);
  MIREG_CHECK_ARG(units < (1L << 30));
  const long g = units;
  const int R = max_displacement / stride2;
  const int DD = (2 * R + 1) * (2 * R + 1);
  const int DDp = (DD + V - 1) / V * V;                             // output tile rows padded to 16 bytes
  // few channels and few displacements (PWC's levels): dot products on the vector ALUs; the MFMA formulation wins from
  // 64 channels up (scratch/mb_corr_levels.py, profiles/README.md); MIREG_CORR_VALU=0/1 forces one for A/B runs
  static const int force = getenv("MIREG_CORR_VALU") ? atoi(getenv("MIREG_CORR_VALU")) : -1;
  const size_t esz = dtype == MIREG_DTYPE_BF16 ? 2 : 4;
  const size_t lds_valu = ((size_t)(kCorrXT + kCorrXT + 2 * R * stride2) * (C / V + 1)) * 16 + (size_t)kCorrXT * DDp * esz;
  const bool valu = force >= 0 ? force == 1 : (C <= 32 && DD <= 81);
  if (valu && lds_valu <= 160 * 1024 - 1024 && (dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32)) {
    const long gv = (long)B * H * ((W + kCorrXT - 1) / kCorrXT);
    if (dtype == MIREG_DTYPE_BF16) {
      if (lds_valu > 48 * 1024) (void)hipFuncSetAttribute((const void*)correlation_fwd_valu_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_valu);
      hipLaunchKernelGGL((correlation_fwd_valu_kernel<__bf16>), dim3((unsigned)gv), dim3(256), lds_valu, stream, (const __bf16*)f1, ld1, (const __bf16*)f2, ld2, (__bf16*)out, ldo, B, H, W, C, c_norm, R, stride2, slope);
    } else {
      if (lds_valu > 48 * 1024) (void)hipFuncSetAttribute((const void*)correlation_fwd_valu_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_valu);
      hipLaunchKernelGGL((correlation_fwd_valu_kernel<float>), dim3((unsigned)gv), dim3(256), lds_valu, stream, (const float*)f1, ld1, (const float*)f2, ld2, (float*)out, ldo, B, H, W, C, c_norm, R, stride2, slope);
    }
    MIREG_LAUNCH_RET();
  }
  const size_t lds = 4 * 32 * 33 * 4 + (size_t)32 * DDp * (dtype == MIREG_DTYPE_BF16 ? 2 : 4);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((correlation_fwd_kernel<__bf16>), dim3((unsigned)g), dim3(256), lds, stream, (const __bf16*)f1, ld1, (const __bf16*)f2, ld2, (__bf16*)out, ldo, B, H, W, C, c_norm, R, stride2, slope);
  else if (dtype == MIREG_DTYPE_F32)
    hipLaunchKernelGGL((correlation_fwd_kernel<float>), dim3((unsigned)g), dim3(256), lds, stream, (const float*)f1, ld1, (const float*)f2, ld2, (float*)out, ldo, B, H, W, C, c_norm, R, stride2, slope);
  else return MIREG_ERR_ARG;
  MIREG_LAUNCH_RET();
}

int mireg_correlation_bwd(const void* g, long ldg, const void* f1, long ld1, const void* f2, long ld2, void* df1, long ldd1,
                          void* df2, long ldd2, int B, int H, int W, int C, int c_norm, int max_displacement, int stride2,
                          int accumulate1, int accumulate2, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(g && f1 && f2 && (df1 || df2) && B > 0 && H > 0 && W > 0 && C > 0 && c_norm > 0 && stride2 > 0);
  MIREG_CHECK_ARG(max_displacement % stride2 == 0 && max_displacement / stride2 <= 15);
  const long units = (long)B * H * ((W + 31) / 32);
  long gr = (units + 3) / 4;
  if (gr > 4096) gr = 4096;
  const int R = max_displacement / stride2;
  // few channels (PWC level 2): one thread per (pixel, channel granule) on the vector ALUs (see correlation_fwd_valu_kernel)
  {
    static const int force = getenv("MIREG_CORR_VALU") ? atoi(getenv("MIREG_CORR_VALU")) : -1;
    const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4, D = 2 * R + 1;
    const int G = C % V == 0 ? C / V : 0;
    const bool shape_ok = G > 0 && G <= 32 && ld1 % V == 0 && ld2 % V == 0 && (uintptr_t)f1 % 16 == 0 && (uintptr_t)f2 % 16 == 0 &&
                          (!df1 || (ldd1 % V == 0 && (uintptr_t)df1 % 16 == 0)) && (!df2 || (ldd2 % V == 0 && (uintptr_t)df2 % 16 == 0)) &&
                          (dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
    const bool valu = shape_ok && (force >= 0 ? force == 1 : (C <= 128 && D * D <= 81));    // measured: wins up to 128 channels, 256 loses 2x
    if (valu) {
      const int XT = 256 / G, XW = XT + 2 * R * stride2;
      const size_t esz = dtype == MIREG_DTYPE_BF16 ? 2 : 4;
      const size_t lds0 = (size_t)XW * (G + 1) * 16 + (size_t)XT * D * D * esz, lds1 = (size_t)XW * (G + 1) * 16 + (size_t)XW * D * esz;
      const long gv = (long)B * H * ((W + XT - 1) / XT);
      if (lds0 <= 150 * 1024 && lds1 <= 150 * 1024 && gv < (1L << 30)) {
#define MIREG_CORR_BWD_VALU(T, WHICH, fo_, ldo__, dd, lddd, acc_, lds_) { \
          if (lds_ > 48 * 1024) (void)hipFuncSetAttribute((const void*)correlation_bwd_valu_kernel<T, WHICH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
          hipLaunchKernelGGL((correlation_bwd_valu_kernel<T, WHICH>), dim3((unsigned)gv), dim3(256), lds_, stream, (const T*)g, ldg, (const T*)fo_, ldo__, \
                             (T*)dd, lddd, B, H, W, C, c_norm, R, stride2, acc_); }
        if (dtype == MIREG_DTYPE_BF16) {
          if (df1) MIREG_CORR_BWD_VALU(__bf16, 0, f2, ld2, df1, ldd1, accumulate1, lds0)
          if (df2) MIREG_CORR_BWD_VALU(__bf16, 1, f1, ld1, df2, ldd2, accumulate2, lds1)
        } else {
          if (df1) MIREG_CORR_BWD_VALU(float, 0, f2, ld2, df1, ldd1, accumulate1, lds0)
          if (df2) MIREG_CORR_BWD_VALU(float, 1, f1, ld1, df2, ldd2, accumulate2, lds1)
        }
#undef MIREG_CORR_BWD_VALU
        MIREG_LAUNCH_RET();
      }
    }
  }
  // bf16 with 16-byte aligned channel rows: bf16 matrix cores + LDS-staged operands
  if (dtype == MIREG_DTYPE_BF16 && C <= 512 && C % 8 == 0 && ld1 % 8 == 0 && ld2 % 8 == 0 && (uintptr_t)f1 % 16 == 0 && (uintptr_t)f2 % 16 == 0 &&
      (uintptr_t)g % 2 == 0 && units < (1L << 30)) {
    const int span = 32 + 2 * R * stride2 + 15;                                  // pixels a tile can pair with, from a 16-aligned start
    // staged pixels start at a 16-aligned klo >= 0 and pixels >= W are zero: a narrow image needs no more than its own width
    const int KW = ((span + 15) / 16 * 16) < ((W + 15) / 16 * 16) ? ((span + 15) / 16 * 16) : ((W + 15) / 16 * 16);
    const int LDB = (C + 31) / 32 * 32 + 8;
    const size_t lds = ((size_t)KW * LDB + 32 * (KW + 8)) * 2;
    const int ntw = ((C + 31) / 32 + 3) / 4;
    if (lds <= 150 * 1024 && (long)KW * (LDB / 8) <= 256 * 14 && 32 * (2 * R + 1) <= 256 * 4 && (long)W * ld1 < (1L << 30) && (long)W * ld2 < (1L << 30) &&
        (long)W * ldg < (1L << 30)) {
      const long slots = ((long)KW * (LDB / 8) + 255) / 256;                        // register slots per staged row (granules per thread)
#define MIREG_CORR_BWD_L(WHICH, NTW, NB, gp, fo_, ldo__, dd, lddd, acc_) { \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)correlation_bwd_mfma_kernel<WHICH, NTW, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((correlation_bwd_mfma_kernel<WHICH, NTW, NB>), dim3((unsigned)units), dim3(256), lds, stream, (const __bf16*)gp, ldg, \
                           (const __bf16*)fo_, ldo__, (__bf16*)dd, lddd, B, H, W, C, c_norm, R, stride2, acc_, KW, LDB); }
#define MIREG_CORR_BWD(WHICH, NTW, gp, fo_, ldo__, dd, lddd, acc_) { \
        if (slots <= 2) MIREG_CORR_BWD_L(WHICH, NTW, 2, gp, fo_, ldo__, dd, lddd, acc_) \
        else if (slots <= 5) MIREG_CORR_BWD_L(WHICH, NTW, 5, gp, fo_, ldo__, dd, lddd, acc_) \
        else MIREG_CORR_BWD_L(WHICH, NTW, 14, gp, fo_, ldo__, dd, lddd, acc_) }
      if (df1) { if (ntw <= 1) MIREG_CORR_BWD(0, 1, g, f2, ld2, df1, ldd1, accumulate1) else if (ntw == 2) MIREG_CORR_BWD(0, 2, g, f2, ld2, df1, ldd1, accumulate1) else MIREG_CORR_BWD(0, 4, g, f2, ld2, df1, ldd1, accumulate1) }
      if (df2) { if (ntw <= 1) MIREG_CORR_BWD(1, 1, g, f1, ld1, df2, ldd2, accumulate2) else if (ntw == 2) MIREG_CORR_BWD(1, 2, g, f1, ld1, df2, ldd2, accumulate2) else MIREG_CORR_BWD(1, 4, g, f1, ld1, df2, ldd2, accumulate2) }
#undef MIREG_CORR_BWD
#undef MIREG_CORR_BWD_L
      MIREG_LAUNCH_RET();
    }
  }
  if (dtype == MIREG_DTYPE_BF16) {
    if (df1) hipLaunchKernelGGL((correlation_bwd_kernel<__bf16, 0>), dim3((unsigned)gr), dim3(256), 0, stream, (const __bf16*)g, ldg, (const __bf16*)f2, ld2, (__bf16*)df1, ldd1, B, H, W, C, c_norm, R, stride2, accumulate1);
    if (df2) hipLaunchKernelGGL((correlation_bwd_kernel<__bf16, 1>), dim3((unsigned)gr), dim3(256), 0, stream, (const __bf16*)g, ldg, (const __bf16*)f1, ld1, (__bf16*)df2, ldd2, B, H, W, C, c_norm, R, stride2, accumulate2);
  } else if (dtype == MIREG_DTYPE_F32) {
    if (df1) hipLaunchKernelGGL((correlation_bwd_kernel<float, 0>), dim3((unsigned)gr), dim3(256), 0, stream, (const float*)g, ldg, (const float*)f2, ld2, (float*)df1, ldd1, B, H, W, C, c_norm, R, stride2, accumulate1);
    if (df2) hipLaunchKernelGGL((correlation_bwd_kernel<float, 1>), dim3((unsigned)gr), dim3(256), 0, stream, (const float*)g, ldg, (const float*)f1, ld1, (float*)df2, ldd2, B, H, W, C, c_norm, R, stride2, accumulate2);
  } else return MIREG_ERR_ARG;
  MIREG_LAUNCH_RET();
}

int mireg_pwc_warp_bwd_det(const void* x, long ldx, const float* flow, long ldf_, float flow_scale, const void* g, long ldg,
                           float* dx32, long lddx, float* dflow, long lddf, int* ws_cnt, int* ws_off, void* ws_entries,
                           int B, int H, int W, int C, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(x && flow && g && dx32 && dflow && ws_cnt && ws_off && ws_entries && B > 0 && H > 0 && W > 0 && C > 0 && C <= 512 &&
                  ldf_ >= 2 && lddf >= 2);
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  MIREG_CHECK_ARG(C % V == 0 && ldx % V == 0 && ldg % V == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)g % 16 == 0);
  MIREG_CHECK_ARG(dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
  const long P = (long)B * H * W;
  MIREG_CHECK_ARG(P * 4 < (1L << 31));
  const int nb = (int)((P + kScanBlock - 1) / kScanBlock);
  MIREG_CHECK_ARG(nb <= 256 * 64);
  int* bsum = ws_off + P + 1;                                        // ws_off holds P + 1 + ceil(P / 2048) ints
  const unsigned gp = (unsigned)((P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048);
  hipLaunchKernelGGL((warp_bucket_kernel<false>), dim3(gp), dim3(256), 0, stream, flow, ldf_, flow_scale, ws_cnt, (const int*)nullptr, (int2*)nullptr, B, H, W);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(256), 0, stream, (const int*)ws_cnt, bsum, P);
  hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(256), 0, stream, bsum, nb, ws_off + P);        // off[P] = number of entries
  hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(256), 0, stream, (const int*)ws_cnt, (const int*)bsum, ws_off, P);
  hipLaunchKernelGGL((warp_bucket_kernel<true>), dim3(gp), dim3(256), 0, stream, flow, ldf_, flow_scale, ws_cnt, (const int*)ws_off, (int2*)ws_entries, B, H, W);
  const long total = P * (C / V);
  const unsigned gg = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == MIREG_DTYPE_BF16) {
    hipLaunchKernelGGL((warp_gather_kernel<__bf16>), dim3(gg), dim3(256), 0, stream, (const __bf16*)g, ldg, (const int*)ws_off, (const int2*)ws_entries, dx32, lddx, P, C);
    hipLaunchKernelGGL((warp_dflow_kernel<__bf16>), dim3(gg < 4096 ? gg : 4096), dim3(256), 0, stream, (const __bf16*)x, ldx, flow, ldf_, flow_scale, (const __bf16*)g, ldg, dflow, lddf, B, H, W, C);
  } else {
    hipLaunchKernelGGL((warp_gather_kernel<float>), dim3(gg), dim3(256), 0, stream, (const float*)g, ldg, (const int*)ws_off, (const int2*)ws_entries, dx32, lddx, P, C);
    hipLaunchKernelGGL((warp_dflow_kernel<float>), dim3(gg < 4096 ? gg : 4096), dim3(256), 0, stream, (const float*)x, ldx, flow, ldf_, flow_scale, (const float*)g, ldg, dflow, lddf, B, H, W, C);
  }
  MIREG_LAUNCH_RET();
}

int mireg_pwc_warp_fwd(const void* x, long ldx, const float* flow, long ldf_, float flow_scale, void* out, long ldo, int B,
                       int H, int W, int C, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(x && flow && out && B > 0 && H > 0 && W > 0 && C > 0 && ldf_ >= 2);
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  MIREG_CHECK_ARG(C % V == 0);
  const long total = (long)B * H * W * (C / V);
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((pwc_warp_fwd_kernel<__bf16>), dim3((unsigned)g), dim3(256), 0, stream, (const __bf16*)x, ldx, flow, ldf_, flow_scale, (__bf16*)out, ldo, B, H, W, C);
  else if (dtype == MIREG_DTYPE_F32)
    hipLaunchKernelGGL((pwc_warp_fwd_kernel<float>), dim3((unsigned)g), dim3(256), 0, stream, (const float*)x, ldx, flow, ldf_, flow_scale, (float*)out, ldo, B, H, W, C);
  else return MIREG_ERR_ARG;
  MIREG_LAUNCH_RET();
}

}  // extern "C"
