// stem_conv.hip -- the 7x7 / stride-2 / pad-3 input convolutions with ONE or TWO input channels (bf16):
// FlowNetS conv1 (2 -> 64, FlowNetS/FlowNetS.py:18) and the siamese FlowNetC conv1 (1 -> 64,
// flownet2/networks/FlowNetC.py:20).  In the generic implicit GEMM the 1-2 real channels sit in an 8-channel
// granule, so 3/4 (7/8) of the gathered bytes and of the MFMA work are zeros (55-80 TFLOP/s).  Here the K axis is
// re-ordered to (ky, kx, ci) with kx padded 7 -> 8, which makes the 8 K-values of an MFMA lane CONTIGUOUS pixels of
// one image row: the block stages the input patch of its 8x16 output tile in LDS once and forms A fragments with
// four 4-byte LDS reads -- K = 112 (64) instead of 392.
//   forward : y[pix][co] = act(b[co] + sum_k P[pix][k] W[co][k])                 (+ fused bias / LeakyReLU)
//   wgrad   : slab[blk][co][(ky*7+kx)*Cpad + ci] = sum_{pix in the block's tiles} dy[pix][co] P[pix][k]
// Both consume / produce the STANDARD packs (F[co][(ky*7+kx)*Cpad + ci], slabs in the same layout), so packing,
// the split-K reduce and the optimizer are untouched.
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace {

#if defined(__HIP_DEVICE_COMPILE__)
#define GPTR(T, p) (reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p)))
#else
#define GPTR(T, p) (reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p)))
#endif

constexpr int TH = 8, TW = 16;                 // output tile (128 pixels = MFMA rows)
constexpr int PH = 2 * TH + 6, PW = 2 * TW + 6;   // input patch incl. the kx = 7 / ky = 7 pad taps: 22 x 38
constexpr int CO = 64;

template <int CI> struct Stem;
template <> struct Stem<2> { static constexpr int K = 112, KSTEPS = 7; };   // k = ky*16 + kx*2 + ci
template <> struct Stem<1> { static constexpr int K = 64, KSTEPS = 4; };    // k = ky*8 + kx (ky, kx padded to 8)

struct StemArgs {
  const __bf16* x; long ld_x;          // NHWC input, channels 0..CI-1 real
  const __bf16* w; long ld_w; int Cpad;   // standard FWD pack [CO][49*Cpad]
  const float* bias; float slope;
  __bf16* y; long ld_y;                // fwd output / wgrad: incoming dy, NHWC [.., CO]
  float* slab; long slab_stride;       // wgrad partial slabs [blocks][CO][ld_w]
  int B, H, W, Ho, Wo;
};

// input patch of tile t -> registers (global loads only, branch-free) -> LDS [PH][PW][CI]; zeros outside the image.
// Split in two so the next tile's loads fly under the current tile's MFMAs.
constexpr int NPATCH = (PH * PW + 255) / 256;                     // 4 pixels per thread

struct TileId { int b, oy0, ox0; };
__device__ __forceinline__ TileId tile_of(long t, int tiles_x, int tiles_y) {
  return {(int)(t / ((long)tiles_x * tiles_y)), (int)((t / tiles_x) % tiles_y) * TH, (int)(t % tiles_x) * TW};
}

template <int CI>
__device__ __forceinline__ void fetch_patch(const StemArgs& a, const TileId id, uint32_t (&reg)[NPATCH]) {
#pragma unroll
  for (int i = 0; i < NPATCH; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int py = e / PW, px = e - py * PW;
    const int iy = 2 * id.oy0 - 3 + py, ix = 2 * id.ox0 - 3 + px;
    const bool ok = e < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const __bf16* src = a.x + (((long)id.b * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * a.ld_x;
    const uint32_t v = CI == 2 ? *GPTR(const uint32_t, src) : (uint32_t)*GPTR(const unsigned short, src);
    reg[i] = ok ? v : 0u;
  }
}
template <int CI>
__device__ __forceinline__ void commit_patch(__bf16* patch, const uint32_t (&reg)[NPATCH]) {
#pragma unroll
  for (int i = 0; i < NPATCH; ++i) {
    const int e = threadIdx.x + 256 * i;
    if (e < PH * PW) {
      if (CI == 2) reinterpret_cast<uint32_t*>(patch)[e] = reg[i];
      else reinterpret_cast<unsigned short*>(patch)[e] = (unsigned short)reg[i];
    }
  }
}

// standard pack -> LDS Wl[CO][KP] in stem K order (pad taps zero): one 4-byte load per (co, tap), a few in flight
template <int CI>
__device__ __forceinline__ void load_weights(const StemArgs& a, __bf16* wl, int KP) {
  for (int e = threadIdx.x; e < CO * KP / 2; e += 256) reinterpret_cast<uint32_t*>(wl)[e] = 0u;
  __syncthreads();
#pragma unroll 4
  for (int e = threadIdx.x; e < CO * 49; e += 256) {
    const int co = e / 49, tap = e - co * 49, ky = tap / 7, kx = tap - ky * 7;
    const __bf16* src = a.w + (long)co * a.ld_w + tap * a.Cpad;
    if (CI == 2) reinterpret_cast<uint32_t*>(wl)[(co * KP + ky * 16 + kx * 2) >> 1] = *GPTR(const uint32_t, src);
    else reinterpret_cast<unsigned short*>(wl)[co * KP + ky * 8 + kx] = *GPTR(const unsigned short, src);
  }
}

// A fragment of MFMA row m (tile pixel ty = m / 16, tx = m % 16), K-step ks, lane half h: 8 contiguous bf16 of a patch row
template <int CI>
__device__ __forceinline__ bf16x8 patch_frag(const __bf16* patch, int ty, int tx, int ks, int h) {
  const uint32_t* p32 = reinterpret_cast<const uint32_t*>(patch);
  int idx;                                                        // in 4-byte units
  if (CI == 2) idx = (2 * ty + ks) * PW + 2 * tx + 4 * h;         // ky = ks, kx = 4h .. 4h+3, both channels
  else idx = ((2 * ty + 2 * ks + h) * PW + 2 * tx) >> 1;          // ky = 2ks + h, kx = 0..7 (PW and 2tx are even)
  const uint4 v = make_uint4(p32[idx], p32[idx + 1], p32[idx + 2], p32[idx + 3]);
  return __builtin_bit_cast(bf16x8, v);
}

template <int CI>
__global__ void __launch_bounds__(256) stem_fwd_kernel(const StemArgs a) {
  constexpr int K = Stem<CI>::K, KSTEPS = Stem<CI>::KSTEPS, KP = K + 8, OP = CO + 8;
  __shared__ __attribute__((aligned(16))) __bf16 wl[CO * KP];
  __shared__ __attribute__((aligned(16))) __bf16 patch[PH * PW * CI + 8];
  __shared__ __attribute__((aligned(16))) __bf16 otile[TH * TW * OP];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  load_weights<CI>(a, wl, KP);
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
  const long ntiles = (long)a.B * tiles_y * tiles_x;
  const int m = wid * 32 + r, ty = m >> 4, tx = m & 15;           // this lane's MFMA row = tile pixel
  uint32_t preg[NPATCH];
  if (blockIdx.x < ntiles) fetch_patch<CI>(a, tile_of(blockIdx.x, tiles_x, tiles_y), preg);
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const TileId id = tile_of(t, tiles_x, tiles_y);
    const int b = id.b, oy0 = id.oy0, ox0 = id.ox0;
    __syncthreads();                                              // previous tile's patch / otile are consumed
    commit_patch<CI>(patch, preg);
    __syncthreads();
    if (t + gridDim.x < ntiles) fetch_patch<CI>(a, tile_of(t + gridDim.x, tiles_x, tiles_y), preg);   // next tile's loads in flight
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const bf16x8 af = patch_frag<CI>(patch, ty, tx, ks, h);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wl + (j * 32 + r) * KP + ks * 16 + 8 * h);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
      }
    }
    // accumulators -> LDS [128 px][CO] (bias, LeakyReLU) -> 16-byte row stores
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = j * 32 + r;
      const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        float v = acc[j][e] + bv;
        v = v > 0.f ? v : v * a.slope;
        otile[row * OP + co] = (__bf16)v;
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TH * TW * (CO / 8); e += 256) {
      const int px = e / (CO / 8), gch = e - px * (CO / 8);
      const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
      if (oy < a.Ho && ox < a.Wo)
        *GPTR(uint4, a.y + (((long)b * a.Ho + oy) * a.Wo + ox) * a.ld_y + gch * 8) = *reinterpret_cast<const uint4*>(otile + px * OP + gch * 8);
    }
  }
}

// backward-weights: out[k][co] = sum_px P[px][k] dy[px][co]; waves split the K rows (32 each), K of the MFMA = pixels
template <int CI>
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const StemArgs a) {
  constexpr int K = Stem<CI>::K, DP = CO + 8;
  __shared__ __attribute__((aligned(16))) __bf16 patch[PH * PW * CI + 8];
  __shared__ __attribute__((aligned(16))) __bf16 dl[TH * TW * DP];      // dy tile [128 px][CO]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int kidx = wid * 32 + r;                                  // this lane's A row = stem K index
  const bool kok = kidx < K;
  const int ky = CI == 2 ? kidx >> 4 : kidx >> 3, kx = CI == 2 ? (kidx & 15) >> 1 : kidx & 7, ci = CI == 2 ? kidx & 1 : 0;
  const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;       // transposing-read geometry
  const int rowoff = 8 * (gq >> 1) + q, coloff = 16 * (gq & 1) + 4 * pp;
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
  const long ntiles = (long)a.B * tiles_y * tiles_x;
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const unsigned short* p16 = reinterpret_cast<const unsigned short*>(patch);
  uint32_t preg[NPATCH];
  uint4 dreg[4];                                                  // dy tile: 128 px x 8 granules = 4 per thread
  auto fetch_dy = [&](const TileId id) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + 256 * i;
      const int px = e / (CO / 8), gch = e - px * (CO / 8);
      const int oy = id.oy0 + (px >> 4), ox = id.ox0 + (px & 15);
      const bool ok = oy < a.Ho && ox < a.Wo;
      const uint4 v = *GPTR(const uint4, a.y + (((long)id.b * a.Ho + (ok ? oy : 0)) * a.Wo + (ok ? ox : 0)) * a.ld_y + gch * 8);
      dreg[i] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  if (blockIdx.x < ntiles) { const TileId id = tile_of(blockIdx.x, tiles_x, tiles_y); fetch_patch<CI>(a, id, preg); fetch_dy(id); }
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();
    commit_patch<CI>(patch, preg);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + 256 * i;
      const int px = e / (CO / 8), gch = e - px * (CO / 8);
      *reinterpret_cast<uint4*>(dl + px * DP + gch * 8) = dreg[i];
    }
    __syncthreads();
    if (t + gridDim.x < ntiles) { const TileId id = tile_of(t + gridDim.x, tiles_x, tiles_y); fetch_patch<CI>(a, id, preg); fetch_dy(id); }
    if (wid * 32 < K) {                                           // wave-uniform: CI = 1 needs only two K-row tiles
#pragma unroll
      for (int s = 0; s < TH; ++s) {                              // MFMA K-step = the 16 pixels of tile row s
        unsigned short av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int tx = 8 * h + j;
          av[j] = kok ? p16[((2 * s + ky) * PW + 2 * tx + kx) * CI + ci] : (unsigned short)0;
        }
        const uint4 apk = make_uint4(av[0] | ((uint32_t)av[1] << 16), av[2] | ((uint32_t)av[3] << 16), av[4] | ((uint32_t)av[5] << 16),
                                     av[6] | ((uint32_t)av[7] << 16));
        const bf16x8 af = __builtin_bit_cast(bf16x8, apk);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const __bf16* base = dl + (s * 16 + rowoff) * DP + j * 32 + coloff;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 4 * DP));
          const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, v), acc[j], 0, 0, 0);
        }
      }
    }
  }
  // acc[j][e]: row = stem K index wid*32 + (e&3) + 8*(e>>2) + 4h, column = co j*32 + r  ->  standard slab layout
  float* slab = a.slab + (long)blockIdx.x * a.slab_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = j * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (k >= K) continue;
      const int kky = CI == 2 ? k >> 4 : k >> 3, kkx = CI == 2 ? (k & 15) >> 1 : k & 7, kci = CI == 2 ? k & 1 : 0;
      if (kky < 7 && kkx < 7) *GPTR(float, slab + (long)co * a.ld_w + (kky * 7 + kkx) * a.Cpad + kci) = acc[j][e];
    }
  }
}

inline bool stem_ok(const void* x, long ld_x, const void* w, long ld_w, int Cpad, const void* y, long ld_y, int B, int H, int W, int Ci,
                    int Co) {
  return x && w && y && B > 0 && H > 0 && W > 0 && (Ci == 1 || Ci == 2) && Co == CO && Cpad >= Ci && ld_w >= 49L * Cpad &&
         ld_x >= Ci && (ld_x * 2) % 4 == 0 && ((uintptr_t)x % 4) == 0 && ld_y % 8 == 0 && ((uintptr_t)y % 16) == 0 && ld_y >= CO;
}

}  // namespace

extern "C" {

int mireg_stem_conv_blocks(int B, int H, int W) {
  const long tiles = (long)B * (((H + 1) / 2 + TH - 1) / TH) * (((W + 1) / 2 + TW - 1) / TW);
  return (int)(tiles < 256 ? tiles : 256);
}

int mireg_stem_conv_fwd(const void* x, long ld_x, const void* w, long ld_w, int Cpad, const float* bias, float slope, void* y,
                        long ld_y, int B, int H, int W, int Ci, int Co, hipStream_t stream) {
  MIREG_CHECK_ARG(stem_ok(x, ld_x, w, ld_w, Cpad, y, ld_y, B, H, W, Ci, Co));
  StemArgs a{};
  a.x = (const __bf16*)x; a.ld_x = ld_x; a.w = (const __bf16*)w; a.ld_w = ld_w; a.Cpad = Cpad; a.bias = bias; a.slope = slope;
  a.y = (__bf16*)y; a.ld_y = ld_y; a.B = B; a.H = H; a.W = W; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2;   // (H + 6 - 7) / 2 + 1
  const long tiles = (long)B * ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW);
  const int grid = (int)(tiles < 512 ? tiles : 512);
  if (Ci == 2) hipLaunchKernelGGL(stem_fwd_kernel<2>, dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(stem_fwd_kernel<1>, dim3(grid), dim3(256), 0, stream, a);
  MIREG_LAUNCH_RET();
}

int mireg_stem_conv_wgrad(const void* x, long ld_x, const void* dy, long ld_dy, float* slab, long ld_w, int Cpad, int nblocks, int B,
                          int H, int W, int Ci, int Co, hipStream_t stream) {
  MIREG_CHECK_ARG(stem_ok(x, ld_x, slab, ld_w, Cpad, dy, ld_dy, B, H, W, Ci, Co) && nblocks == mireg_stem_conv_blocks(B, H, W));
  StemArgs a{};
  a.x = (const __bf16*)x; a.ld_x = ld_x; a.ld_w = ld_w; a.Cpad = Cpad; a.y = (__bf16*)const_cast<void*>(dy); a.ld_y = ld_dy;
  a.slab = slab; a.slab_stride = (long)Co * ld_w; a.B = B; a.H = H; a.W = W; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2;
  if (Ci == 2) hipLaunchKernelGGL(stem_wgrad_kernel<2>, dim3(nblocks), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(stem_wgrad_kernel<1>, dim3(nblocks), dim3(256), 0, stream, a);
  MIREG_LAUNCH_RET();
}

}  // extern "C"
