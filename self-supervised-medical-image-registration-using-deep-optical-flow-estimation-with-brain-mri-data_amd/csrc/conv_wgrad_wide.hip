// conv_wgrad_wide.hip -- backward-weights GEMM on a 256 x 256 output tile with 8 waves (round 3).
//
// dW[co][kidx] = sum_pix dy[pix][co] * x[gather(pix, tap(kidx))][c(kidx)]   (autograd of FlowNetS/util.py:17-46, PWC/models/PWCNet.py:24-31)
// The same contraction, descriptor and slab layout as conv_wgrad_dma_kernel (conv_gemm.hip); what changes is the tile.  Both
// operand tiles are [pixel][256 channels] (512-byte rows, channel-contiguous as they sit in HBM), so one K-step of 32 pixels moves
// 32 KiB into LDS for 2 x 256 x 256 x 32 FLOP: 128 FLOP per byte against 64 for the 128 x 128 tile, which is what the L2 -> LDS
// rate of a CU (about 30 B/clk) asks for.  Where it pays: the deep layers, whose pixel count (384 / 1536) is ONE short K loop per
// output tile (no split, the slab IS the gradient), and the 256/512-channel layers, where a 256-wide tile halves the operand
// re-reads per slab byte.
//   8 waves = 2 x 4, 128 x 64 per wave as 4 x 2 MFMA 32x32x16 tiles; LDS-DMA ring of 4 stages (buffer_load ... lds, one DMA =
//   2 pixel rows x 512 B), counted vmcnt, one raw s_barrier per K-step, lgkmcnt(0) before it (write-after-read on a ring stage);
//   fragments by the transposing LDS read ds_read_b64_tr_b16, 16-byte chunks XOR-swizzled with (pixel row & 3) << 2 on the DMA
//   source side and on the read side; epilogue through LDS in four 64-row passes, 16-byte slab stores.
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_void_t;

namespace {

__device__ __forceinline__ void stg_u4(void* p, uint4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  *reinterpret_cast<__attribute__((address_space(1))) uint4*>(reinterpret_cast<uintptr_t>(p)) = v;
#else
  *reinterpret_cast<uint4*>(p) = v;
#endif
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {                   // n is wave-uniform, a multiple of 4 here
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 4: wait_vmcnt<4>(); break;
    default: wait_vmcnt<8>(); break;
  }
}

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int ROWB = 512;                                                  // bytes per pixel row of a tile (256 bf16)
constexpr int TILE_BYTES = BK * ROWB;                                      // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int STAGES = 4;
constexpr int WTM = 128, WTN = 64, TM = 4, TN = 2;
constexpr int EPI_ROWS = 64;
constexpr unsigned kOOB = 0x80000000u;

__global__ void __launch_bounds__(512)
conv_wgrad_wide_kernel(const mireg_conv_desc p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * STAGE_BYTES];      // 128 KiB; the epilogue reuses 64 KiB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int r = lane & 31, h = lane >> 5;
  const int Ktot = p.taps_y * p.taps_x * p.x_C;                            // GEMM N
  const int tiles_n = (Ktot + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
  }
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int gHW = p.g_H * p.g_W;
  const int P = p.n_img * gHW;                                             // reduction length (pixels)
  const int nk_total = (P + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk_total;
  if (p.split_k > 1) {
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.z * per;
    kt_end = min(nk_total, kt_begin + per);
  }
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.y), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);

  // lane -> (row inside the 2-row DMA instruction, logical 16-B chunk); rows are swizzled by (tile row & 3) << 2
  const int lrow = lane >> 5, pch = lane & 31;
  const int lch = pch ^ (((2 * wid + lrow) & 3) << 2);                     // tile row = 2 * (wid + 8c) + lrow: & 3 does not depend on c
  const int a_col = m0 + lch * 8;                                          // A operand (dy): this lane's 8 output channels
  const bool a_cok = a_col < (p.N + 7) / 8 * 8;
  const int b_n = n0 + lch * 8;                                            // B operand (x): this lane's kidx chunk fixes (tap, channel)
  const bool b_cok = b_n < Ktot;
  const int b_tap = (b_cok ? b_n : 0) / p.x_C, b_ch = (b_cok ? b_n : 0) - b_tap * p.x_C;
  const int b_ty = b_tap / p.taps_x, b_tx = b_tap - b_ty * p.taps_x;
  const int b_dy = p.off_y + b_ty * p.step_y, b_dx = p.off_x + b_tx * p.step_x;

  // this lane's pixel for its two DMA instructions per operand, kept decomposed and advanced by BK per K-step: running 32-bit byte
  // offsets and source coordinates, adds on the carries only (as conv_wgrad_dma_kernel)
  const unsigned PIXB = (unsigned)p.x_ld * 2;
  const unsigned STEPX = (unsigned)p.mul_x * PIXB, ROWB_X = (unsigned)p.x_W * PIXB, STEPY = (unsigned)p.mul_y * ROWB_X;
  const unsigned IMGB = (unsigned)p.x_H * ROWB_X;
  const unsigned y_ldb = (unsigned)p.y_ld * 2;
  int pix[2], p_gy[2], p_gx[2], s_iy[2], s_ix[2];
  unsigned a_off[2], b_off[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    pix[c] = kt_begin * BK + (wid + 8 * c) * 2 + lrow;
    const int pp = min(pix[c], P - 1);
    const int img = pp / gHW;
    const int rem = pp - img * gHW;
    p_gy[c] = rem / p.g_W;
    p_gx[c] = rem - p_gy[c] * p.g_W + (pix[c] - pp);                      // a tail overshoot stays on the x axis (never used: pok)
    s_iy[c] = p_gy[c] * p.mul_y + b_dy;
    s_ix[c] = p_gx[c] * p.mul_x + b_dx;
    a_off[c] = (unsigned)pix[c] * y_ldb + (unsigned)a_col * 2;
    b_off[c] = (unsigned)(img * (int)IMGB) + (unsigned)(s_iy[c] * (int)ROWB_X) + (unsigned)(s_ix[c] * (int)PIXB) + (unsigned)b_ch * 2;
  }

  auto issue = [&](int stage) {
    unsigned char* At = smem + stage * STAGE_BYTES;
    unsigned char* Bt = At + TILE_BYTES;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool pok = pix[c] < P;
      const unsigned aoff = (pok && a_cok) ? a_off[c] : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void_t)(At + (wid + 8 * c) * 1024), 16, aoff, 0, 0, 0);
      const bool bok = pok && b_cok && (unsigned)s_iy[c] < (unsigned)p.x_H && (unsigned)s_ix[c] < (unsigned)p.x_W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t)(Bt + (wid + 8 * c) * 1024), 16, bok ? b_off[c] : kOOB, 0, 0, 0);
      pix[c] += BK;
      a_off[c] += BK * y_ldb;
      p_gx[c] += BK;
      s_ix[c] += BK * p.mul_x;
      b_off[c] += BK * STEPX;
      while (p_gx[c] >= p.g_W) {                                            // carry into rows / images
        p_gx[c] -= p.g_W;
        s_ix[c] -= p.g_W * p.mul_x;
        b_off[c] += STEPY - (unsigned)p.g_W * STEPX;
        s_iy[c] += p.mul_y;
        if (++p_gy[c] == p.g_H) {
          p_gy[c] = 0;
          s_iy[c] -= p.g_H * p.mul_y;
          b_off[c] += IMGB - (unsigned)p.g_H * STEPY;
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposing reads: 16-lane group g covers columns 16 (g & 1) .. of a 32-wide fragment and pixel rows 8 (g >> 1) + q (+4)
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp4 = i16 & 3;
  const int rowoff = 8 * (g >> 1) + q;                                     // tile row & 3 == q
  const int colch = 2 * (g & 1) + (pp4 >> 1), sub = (pp4 & 1) * 8;
  int a_lane[TM], b_lane[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_lane[i] = rowoff * ROWB + (((((wm * WTM + i * 32) >> 3) + colch) ^ (q << 2)) << 4) + sub;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_lane[j] = TILE_BYTES + rowoff * ROWB + (((((wn * WTN + j * 32) >> 3) + colch) ^ (q << 2)) << 4) + sub;

  auto tr_pair = [&](const unsigned char* base) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 4 * ROWB));
    const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&](int stage) {
    const unsigned char* St = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = tr_pair(St + ks * 16 * ROWB + a_lane[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = tr_pair(St + ks * 16 * ROWB + b_lane[j]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = max(kt_end - kt_begin, 0);
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st)
    if (st < nk) issue(st);
  int it = 0;
  const int steady = nk - (STAGES - 1);
  for (; it < steady; ++it) {
    wait_vmcnt<(STAGES - 2) * 4>();                                        // STAGES-2 younger K-steps x 4 DMAs stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue((it + STAGES - 1) % STAGES);
    compute(it % STAGES);
  }
  for (; it < nk; ++it) {
    wait_vmcnt_dyn(min(STAGES - 2, nk - 1 - it) * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    compute(it % STAGES);
  }

  // ---- epilogue: accumulators -> LDS fp32 [64][256] per pass -> 16-byte rows of the slab [z][Cout][Ktot] -------------------------
  wait_vmcnt<0>();
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
  const int Cout = p.N;
  const long sld = p.slab_ld > 0 ? p.slab_ld : Ktot;
  float* __restrict__ slab = p.slab + (long)blockIdx.z * Cout * sld;
  const bool vec = (Ktot % 4) == 0 && (sld % 4) == 0;
#pragma unroll
  for (int hp = 0; hp < BM / EPI_ROWS; ++hp) {                             // rows [64 hp, 64 hp + 64) of the tile (unrolled: static fragment indices)
    if (hp) __syncthreads();
    if (wm == (hp >> 1)) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            ct[(i2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * BN + wn * WTN + j * 32 + r] = acc[2 * (hp & 1) + i2][j][e];
    }
    __syncthreads();
    for (int c = tid; c < EPI_ROWS * (BN / 4); c += 512) {
      const int mr = c / (BN / 4), nl = (c - mr * (BN / 4)) * 4;
      const int m = m0 + hp * EPI_ROWS + mr, n = n0 + nl;
      if (m >= Cout || n >= Ktot) continue;
      const float4 v = *reinterpret_cast<const float4*>(ct + mr * BN + nl);
      float* d = slab + (long)m * sld + n;
      if (vec) stg_u4(d, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
      else { const float vv[4] = {v.x, v.y, v.z, v.w}; for (int qq = 0; qq < 4 && n + qq < Ktot; ++qq) d[qq] = vv[qq]; }
    }
  }
}

}  // namespace

// 1 when the 256 x 256 backward-weights tile can run desc: bf16, 2-D, buffer extents below 2 GiB, at least 129 output channels
// (a 256-row tile on <= 128 channels is half empty: those layers stay on the 128 x 128 kernels).
extern "C" int mireg_conv_wgrad_wide_eligible(const mireg_conv_desc* p) {
  if (!p || p->dtype != MIREG_DTYPE_BF16) return 0;
  if (p->g_D > 1 || p->x_D > 1 || p->taps_z > 1 || p->mul_z > 1 || p->off_z != 0) return 0;
  if (!(p->x_bytes > 0 && p->w_bytes > 0 && p->x_bytes < (1L << 31) && p->w_bytes < (1L << 31))) return 0;
  if (p->N <= 128) return 0;
  return 1;
}

extern "C" int mireg_conv_wgrad_wide_try(const mireg_conv_desc* p, hipStream_t stream) {
  if (!mireg_conv_wgrad_wide_eligible(p)) return -100;
  const int Ktot = p->taps_y * p->taps_x * p->x_C;
  const int z = p->split_k > 1 ? p->split_k : 1;
  dim3 grid((unsigned)(((p->N + BM - 1) / BM) * ((Ktot + BN - 1) / BN)), 1, z);
  hipLaunchKernelGGL(conv_wgrad_wide_kernel, grid, dim3(512), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
}
