// conv_gemm.hip -- implicit-GEMM convolution family on the gfx950 matrix cores (K1-K4).
//
// One gather-GEMM kernel serves every dense contraction of the predictors:
//   FWD   conv forward                (FlowNetS/util.py:17-46, PWC/models/PWCNet.py:24-31)
//   DGRAD conv backward-data == ConvTranspose2d forward (FlowNetS/util.py:49-55, PWCNet.py:33-34),
//         decomposed per output-pixel parity for stride 2 so no MAC is wasted on zero taps
//   WGRAD conv backward-weights (second kernel: reduction over pixels)
// Activations are NHWC with a pixel stride (ld) so producers write straight into channel slices of
// the concat buffers (no torch.cat copies, K5).  Operands are bf16 (v_mfma_f32_32x32x16_bf16) or
// fp32 (v_mfma_f32_32x32x2_f32, exact fp32: the parity path); accumulation is always fp32.
//
// Tiling: 256 threads = 4 waves, block tile BM x BN x (64 bytes of K), wave tile of 32x32 MFMA sub-tiles; operand tiles
// go L2 -> LDS by LDS-DMA (buffer_load ... lds) into a 3/4-stage ring of unpadded 64-byte rows whose 16-byte chunks are
// XOR-swizzled on the source side and on the ds_read_b128 side (bank-conflict free); counted vmcnt, one raw s_barrier
// per K-step.  Unit-stride gathers whose grid is 16/32/64 pixels wide take the halo-staged kernel of conv_halo.hip instead
// (each input pixel staged once per channel chunk, not once per tap); this file keeps the strided and small-grid cases.
#include <cstdlib>
#include "mireg_common.h"
#include "../../include/mireg.h"
#include <stdlib.h>

using namespace mireg;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

struct alignas(16) Chunk { uint32_t w[4]; };
// Tile loads must be GLOBAL (address space 1) loads: through a struct-passed pointer hipcc otherwise emits
// flat_load, which also counts on lgkmcnt and serialises against the LDS fragment reads.
template <typename T> __device__ __forceinline__ Chunk ldg_chunk(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4 v = *reinterpret_cast<const __attribute__((address_space(1))) uint4*>(reinterpret_cast<uintptr_t>(p));
  return Chunk{{v.x, v.y, v.z, v.w}};
#else
  return *reinterpret_cast<const Chunk*>(p);
#endif
}

__device__ __forceinline__ void stg_u4(void* p, uint4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  *reinterpret_cast<__attribute__((address_space(1))) uint4*>(reinterpret_cast<uintptr_t>(p)) = v;
#else
  *reinterpret_cast<uint4*>(p) = v;
#endif
}

template <typename T> struct Cfg;
template <> struct Cfg<float>  { static constexpr int CPC = 4, BK = 16; };   // elements per 16-B chunk, K per step
template <> struct Cfg<__bf16> { static constexpr int CPC = 8, BK = 32; };

constexpr int kRowB = 80;   // LDS row pitch of the [row][k] tiles: 64 B of K + 16 B pad

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// =====================================================================================================
// FWD / DGRAD: C[m][n] = sum_k A[m][k] * W[n][k], rows m = output pixels (gathered), cols n = channels.
// LDS-DMA ring.  Tiles go HBM/L2 -> LDS directly (buffer_load ... lds, 16 B per
// lane, 1 KiB per wave-instruction) into a 4-stage ring, so three K-steps of loads are always in flight behind
// the MFMAs; one raw s_barrier per K-step, counted s_waitcnt vmcnt(N) (never 0 in steady state).
//   * zero padding / tile tails: the lane's buffer offset is pushed out of range, the hardware writes zeros;
//   * LDS rows are 64 B unpadded (DMA writes lane-linear), bank conflicts are removed by XOR-swizzling the
//     16-B chunk index with (row >> 2) & 3 on the SOURCE side and again on the ds_read_b128 side.
// =====================================================================================================
typedef __attribute__((address_space(3))) void* lds_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {   // n is wave-uniform
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    default: wait_vmcnt<8>(); break;
  }
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256)
conv_gemm_dma_kernel(const mireg_conv_desc pd) {
  constexpr int CPC = Cfg<T>::CPC, BK = Cfg<T>::BK;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
  constexpr int STAGES = 4;
  constexpr int A_GROUPS = BM / 16, B_GROUPS = BN / 16;          // 16-row groups = one DMA wave-instruction each
  constexpr int A_PW = A_GROUPS / 4;                               // per wave
  constexpr int B_PW = (B_GROUPS + 3) / 4;
  constexpr int STAGE_BYTES = (BM + BN) * 64;
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && A_GROUPS % 4 == 0, "bad tile");
  constexpr int EPI_BYTES = BM * BN * 4 + BM * 8;
  constexpr int SMEM_BYTES = STAGES * STAGE_BYTES > EPI_BYTES ? STAGES * STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

  // stride-2 backward-data / deconvolution: blockIdx.y selects the output-pixel parity class (own taps, weights,
  // sub-grid and output offset); one launch covers all classes so the deep layers still fill the chip
  mireg_conv_desc p = pd;
  const int cls = blockIdx.y;
  if (pd.n_cls > 1) {
    const mireg_conv_cls k = pd.cls[cls];
    p.taps_y = k.taps_y; p.taps_x = k.taps_x; p.off_y = k.off_y; p.off_x = k.off_x; p.g_H = k.g_H; p.g_W = k.g_W;
    p.y_off_y = k.y_off_y; p.y_off_x = k.y_off_x; p.w = k.w; p.w_ld = k.w_ld; p.w_bytes = k.w_bytes;
    if (k.g_D > 0) { p.taps_z = k.taps_z; p.off_z = k.off_z; p.g_D = k.g_D; p.y_off_z = k.y_off_z; }      // Conv3d parity classes
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_n = (p.N + BN - 1) / BN;
  // optional depth axis (Conv3d): all *_D / *_z fields are 0 for 2-D launches and then collapse to extent 1
  const int xD = max(p.x_D, 1), gD = max(p.g_D, 1), tapsZ = max(p.taps_z, 1), yD = max(p.y_D, 1), ymz = max(p.y_mul_z, 1);
  const int gHW = p.g_H * p.g_W, gDHW = gD * gHW;
  const int M = p.n_img * gDHW;
  const int my_tiles = ((M + BM - 1) / BM) * tiles_n;
  // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), so give each XCD a contiguous
  // run of tiles (neighbouring m-tiles x all n-tiles): its 4 MiB L2 then holds that run's pixels and the weights.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
  }
  if (bid >= my_tiles) return;                                      // classes with fewer tiles than the grid
  // An XCD's run of tiles shares whichever operand the tile order keeps fixed.  Many pixels (M >= N): m-major, the run shares its
  // pixels' gathers and re-reads the (small) weights.  Few pixels, many channels (the 4x4 / 8x8 levels: M = 384..1536 against
  // up to 19 MB of weights): n-major, so that all row tiles of a weight slice sit on ONE XCD and the slice leaves HBM once
  // instead of once per row tile (conv6_1 forward: 60 MB at the memory side for 26 MB of operands before this).
  const int tiles_m = (M + BM - 1) / BM;
  int tile_m, tile_n;
  if (M < p.N) { tile_n = bid / tiles_m; tile_m = bid - tile_n * tiles_m; }
  else { tile_m = bid / tiles_n; tile_n = bid - tile_m * tiles_n; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int K = tapsZ * p.taps_y * p.taps_x * p.x_C;
  const int nk_total = (K + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk_total;
  if (p.split_k > 1) {
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.z * per;
    kt_end = min(nk_total, kt_begin + per);
  }
  float* const slab_base = p.slab ? p.slab + (long)cls * pd.slab_cls_stride : nullptr;

  // ---- DMA source state: lane L of a group covers row L>>2, physical chunk L&3 = logical chunk ^ swizzle ----
  constexpr unsigned kOOB = 0x80000000u;
  const int lrow = lane >> 2;
  const int kc = (lane & 3) ^ ((lane >> 4) & 3);                   // logical 16-B chunk of the K-step this lane fetches
  unsigned a_base[A_PW];                                            // byte offset of the image, or OOB
  int a_iz0[A_PW], a_iy0[A_PW], a_ix0[A_PW];
#pragma unroll
  for (int c = 0; c < A_PW; ++c) {
    const int m = m0 + (wid + 4 * c) * 16 + lrow;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int img = mm / gDHW, r3 = mm - img * gDHW;
    const int gz = r3 / gHW, rem = r3 - gz * gHW;
    const int gy = rem / p.g_W, gx = rem - gy * p.g_W;
    a_base[c] = ok ? (unsigned)((long)img * xD * p.x_H * p.x_W * p.x_ld * (long)sizeof(T)) : kOOB;
    a_iz0[c] = gz * p.mul_z + p.off_z;
    a_iy0[c] = gy * p.mul_y + p.off_y;
    a_ix0[c] = gx * p.mul_x + p.off_x;
  }
  unsigned b_base[B_PW];
#pragma unroll
  for (int c = 0; c < B_PW; ++c) {
    const int g = wid + 4 * c;
    const int n = n0 + g * 16 + lrow;
    b_base[c] = (g < B_GROUPS && n < p.N) ? (unsigned)((long)n * p.w_ld * (long)sizeof(T)) : kOOB;
  }
  const int cpt = p.x_C / CPC;
  int tz, ty, tx, cc;
  {
    const int q = kt_begin * 4 + kc;
    const int tap = q / cpt;
    cc = q - tap * cpt;
    const int tyx = p.taps_y * p.taps_x;
    tz = tap / tyx;
    const int t2 = tap - tz * tyx;
    ty = t2 / p.taps_x;
    tx = t2 - ty * p.taps_x;
  }
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  // running source offsets: a_cur = byte offset of (row, current tap, channel 0) or OOB; recomputed only when the
  // lane's chunk walks into the next tap, so the steady-state K-step costs a handful of VALU ops per DMA
  unsigned a_cur[A_PW];
  auto set_tap = [&]() {
    const bool kvalid = tz < tapsZ;
#pragma unroll
    for (int c = 0; c < A_PW; ++c) {
      const int iz = a_iz0[c] + tz * p.step_z, iy = a_iy0[c] + ty * p.step_y, ix = a_ix0[c] + tx * p.step_x;
      const bool ok = kvalid && a_base[c] != kOOB && (unsigned)iz < (unsigned)xD && (unsigned)iy < (unsigned)p.x_H &&
                      (unsigned)ix < (unsigned)p.x_W;
      a_cur[c] = ok ? a_base[c] + (unsigned)(((((long)iz * p.x_H + iy) * p.x_W + ix) * p.x_ld) * (long)sizeof(T)) : kOOB;
    }
  };
  set_tap();
  unsigned b_k = (unsigned)((kt_begin * BK + kc * CPC) * (int)sizeof(T));      // byte offset along K of this lane's chunk
  const unsigned b_kend = (unsigned)(K * (int)sizeof(T));

  auto issue = [&](int stage) {
    unsigned char* As = smem + stage * STAGE_BYTES;
    unsigned char* Bs = As + BM * 64;
    const unsigned ccb = (unsigned)(cc * 16);
#pragma unroll
    for (int c = 0; c < A_PW; ++c) {
      const unsigned off = a_cur[c] != kOOB ? a_cur[c] + ccb : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t)(As + (wid + 4 * c) * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < B_PW; ++c) {
      if (wid + 4 * c < B_GROUPS) {                                 // wave-uniform
        const unsigned off = (b_base[c] != kOOB && b_k < b_kend) ? b_base[c] + b_k : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t)(Bs + (wid + 4 * c) * 1024), 16, off, 0, 0, 0);
      }
    }
    b_k += BK * (int)sizeof(T);
    cc += 4;
    if (cc >= cpt) {
      do { cc -= cpt; if (++tx == p.taps_x) { tx = 0; if (++ty == p.taps_y) { ty = 0; ++tz; } } } while (cc >= cpt);
      set_tap();
    }
  };
  int loads_per_tile = A_PW;                                         // DMA instructions this wave issues per K-step
#pragma unroll
  for (int c = 0; c < B_PW; ++c) loads_per_tile += (wid + 4 * c < B_GROUPS) ? 1 : 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = kt_end - kt_begin;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s);

  // fragment read offsets (bytes inside a stage): row*64 + ((chunk ^ swz(row)) * 16), swz(row) = (row>>2)&3
  int a_off[TM], b_off[TN], a_swz[TM], b_swz[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) { const int row = wm * WTM + i * 32 + r; a_off[i] = row * 64; a_swz[i] = (row >> 2) & 3; }
#pragma unroll
  for (int j = 0; j < TN; ++j) { const int row = wn * WTN + j * 32 + r; b_off[j] = BM * 64 + row * 64; b_swz[j] = (row >> 2) & 3; }

  auto compute = [&](int stage) {
    const unsigned char* st = smem + stage * STAGE_BYTES;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + (((ks * 2 + h) ^ a_swz[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + (((ks * 2 + h) ^ b_swz[j]) << 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        f32x4 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(st + a_off[i] + (((g * 2 + h) ^ a_swz[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const f32x4*>(st + b_off[j] + (((g * 2 + h) ^ b_swz[j]) << 4));
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bfr[j][t], acc[i][j], 0, 0, 0);
      }
    }
  };

  // steady state: tile `it` has landed once at most 2 younger tiles (2 * loads_per_tile DMAs) are outstanding
  int it = 0;
  const int steady = nk - (STAGES - 1);
  if (loads_per_tile == A_PW + B_PW) {                               // every wave of 128/64-wide tiles: immediate count
    for (; it < steady; ++it) {
      wait_vmcnt<2 * (A_PW + B_PW)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own fragment reads done before another wave's DMA re-fills the stage
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue((it + STAGES - 1) % STAGES);
      compute(it % STAGES);
    }
  } else {
    for (; it < steady; ++it) {
      wait_vmcnt_dyn(2 * loads_per_tile);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own fragment reads done before another wave's DMA re-fills the stage
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue((it + STAGES - 1) % STAGES);
      compute(it % STAGES);
    }
  }
  for (; it < nk; ++it) {                                            // drain: nothing left to issue
    wait_vmcnt_dyn(min(STAGES - 2, nk - 1 - it) * loads_per_tile);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    compute(it % STAGES);
  }

  // ---- epilogue: accumulators -> LDS (fp32 [BM][BN], reusing the ring) -> 16-byte coalesced row stores ------
  __syncthreads();                                                   // every wave is done with the ring
  float* ct = reinterpret_cast<float*>(smem);
  long* rowoff = reinterpret_cast<long*>(smem + BM * BN * 4);        // BM entries (ring is >= BM*BN*4 + BM*8 bytes)
  const bool slab_out = p.split_k > 1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = wn * WTN + j * 32 + r, n = n0 + nl;
      const float bv = (!slab_out && p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ml = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        float v = acc[i][j][e];
        if (!slab_out) { v += bv; v = v > 0.f ? v : v * p.slope; }
        ct[ml * BN + nl] = v;
      }
    }
  if (tid < BM) {
    const int m = m0 + tid;
    long off = -1;
    if (m < M) {
      if (slab_out) off = ((long)blockIdx.z * M + m) * p.N;
      else {
        const int img = m / gDHW, r3 = m - img * gDHW;
        const int gz = r3 / gHW, rem = r3 - gz * gHW;
        const int gy = rem / p.g_W, gx = rem - gy * p.g_W;
        off = (((long)img * yD + gz * ymz + p.y_off_z) * p.y_H + gy * p.y_mul_y + p.y_off_y) * p.y_W + gx * p.y_mul_x + p.y_off_x;
      }
    }
    rowoff[tid] = off;
  }
  __syncthreads();
  if (slab_out) {
    constexpr int CPR = BN / 4;                                      // float4 chunks per row
    const bool vec = (p.N % 4) == 0;
    for (int c = tid; c < BM * CPR; c += 256) {
      const int ml = c / CPR, nl = (c - ml * CPR) * 4, n = n0 + nl;
      const long off = rowoff[ml];
      if (off < 0 || n >= p.N) continue;
      const float4 v = *reinterpret_cast<const float4*>(ct + ml * BN + nl);
      float* d = slab_base + off + n;
      if (vec) stg_u4(d, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
      else { const float vv[4] = {v.x, v.y, v.z, v.w}; for (int q = 0; q < 4 && n + q < p.N; ++q) d[q] = vv[q]; }
    }
    return;
  }
  T* __restrict__ yp = reinterpret_cast<T*>(p.y);
  constexpr int V = 16 / (int)sizeof(T);                              // output elements per 16-byte store
  constexpr int CPR = BN / V;
  const bool vec = yp && (p.y_ld % V) == 0 && (reinterpret_cast<uintptr_t>(yp) % 16) == 0;
  for (int c = tid; c < BM * CPR; c += 256) {
    const int ml = c / CPR, nl = (c - ml * CPR) * V, n = n0 + nl;
    const long pix = rowoff[ml];
    if (pix < 0 || n >= p.N) continue;
    float v[V];
#pragma unroll
    for (int q = 0; q < V; q += 4) {
      const float4 t4 = *reinterpret_cast<const float4*>(ct + ml * BN + nl + q);
      v[q] = t4.x; v[q + 1] = t4.y; v[q + 2] = t4.z; v[q + 3] = t4.w;
    }
    if (yp) {
      T* d = yp + pix * p.y_ld + n;
      if (vec && n + V <= p.N) {
        Chunk o;
        if (p.accumulate & 1) {
          const Chunk old = ldg_chunk(d);
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[2 * q] += __uint_as_float(old.w[q] << 16); v[2 * q + 1] += __uint_as_float(old.w[q] & 0xffff0000u); }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += __uint_as_float(old.w[q]);
          }
        }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            const bf2 t2 = {(__bf16)v[2 * q], (__bf16)v[2 * q + 1]};
            o.w[q] = __builtin_bit_cast(uint32_t, t2);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) o.w[q] = __float_as_uint(v[q]);
        }
        stg_u4(d, make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]));
      } else {
        for (int q = 0; q < V && n + q < p.N; ++q) {
          float vv = v[q];
          if (p.accumulate & 1) vv += to_f32(d[q]);
          d[q] = from_f32<T>(vv);
          v[q] = vv;
        }
      }
    }
    if (p.y32) for (int q = 0; q < V && n + q < p.N; ++q) p.y32[pix * p.y32_ld + n + q] = v[q];
  }
}

// split-K second pass: y = act(sum_z slab[z] + bias) with the same pixel mapping as the main epilogue
template <typename T>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const mireg_conv_desc pd) {
  mireg_conv_desc p = pd;
  if (pd.n_cls > 1) {
    const mireg_conv_cls k = pd.cls[blockIdx.y];
    p.g_H = k.g_H; p.g_W = k.g_W; p.y_off_y = k.y_off_y; p.y_off_x = k.y_off_x;
    if (k.g_D > 0) { p.g_D = k.g_D; p.y_off_z = k.y_off_z; }
  }
  const float* __restrict__ slab = p.slab + (long)blockIdx.y * pd.slab_cls_stride;
  const int gD = max(p.g_D, 1), yD = max(p.y_D, 1), ymz = max(p.y_mul_z, 1);
  const int gHW = p.g_H * p.g_W, gDHW = gD * gHW;
  const long M = (long)p.n_img * gDHW, total = M * p.N;
  auto out_pix = [&](long m) -> long {
    const int img = (int)(m / gDHW), r3 = (int)(m - (long)img * gDHW);
    const int gz = r3 / gHW, rem = r3 - gz * gHW;
    const int gy = rem / p.g_W, gx = rem - gy * p.g_W;
    return (((long)img * yD + gz * ymz + p.y_off_z) * p.y_H + gy * p.y_mul_y + p.y_off_y) * p.y_W + gx * p.y_mul_x + p.y_off_x;
  };
  T* __restrict__ yp = reinterpret_cast<T*>(p.y);
  const bool vec = (p.N % 4) == 0 && yp && (p.y_ld % 4) == 0 && !p.y32;
  if (vec) {                                                        // 4 channels per thread: 16-B slab reads
    const int n4 = p.N >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M * n4; i += (long)gridDim.x * blockDim.x) {
      const long m = i / n4;
      const int n = (int)(i - m * n4) * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      for (int z = 0; z < p.split_k; ++z) {
        const Chunk c = ldg_chunk(slab + (long)z * total + m * p.N + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += __uint_as_float(c.w[q]);
      }
      const long pix = out_pix(m);
      T* d = yp + pix * p.y_ld + n;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float vv = v[q];
        if (p.bias) vv += p.bias[n + q];
        vv = vv > 0.f ? vv : vv * p.slope;
        if (p.accumulate & 1) vv += to_f32(d[q]);
        d[q] = from_f32<T>(vv);
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / p.N;
    const int n = (int)(i - m * p.N);
    float v = 0.f;
    for (int z = 0; z < p.split_k; ++z) v += slab[(long)z * total + i];
    if (p.bias) v += p.bias[n];
    v = v > 0.f ? v : v * p.slope;
    const long pix = out_pix(m);
    if (yp) {
      T* d = yp + pix * p.y_ld + n;
      if (p.accumulate & 1) v += to_f32(*d);
      *d = from_f32<T>(v);
    }
    if (p.y32) p.y32[pix * p.y32_ld + n] = v;
  }
}

// =====================================================================================================
// WGRAD: dW[co][kidx] = sum_pix dy[pix][co] * x[gather(pix, tap(kidx))][c(kidx)]   (fp32 slabs out)
// LDS tiles are [pixel][channel] (channel-contiguous, as both operands sit in HBM); bf16 fragments are
// formed with the gfx950 transposing LDS read (ds_read_b64_tr_b16), fp32 fragments with plain b32 reads.
// =====================================================================================================
template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const mireg_conv_desc p) {
  constexpr int CPC = Cfg<T>::CPC, BK = Cfg<T>::BK;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + (sizeof(T) == 2 ? 32 : 0);   // elements; bf16 pitch 320 B == 64 mod 256 (tr-read friendly)
  constexpr int LDB = BN + (sizeof(T) == 2 ? 32 : 0);
  constexpr int A_CPR = BM / CPC, B_CPR = BN / CPC;     // chunks per pixel row
  constexpr int A_CH = BK * A_CPR / 256, B_CH = BK * B_CPR / 256;
  static_assert(A_CH >= 1 && B_CH >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) T smem[2 * BK * (LDA + LDB)];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int r = lane & 31, h = lane >> 5;
  const int Ktot = p.taps_y * p.taps_x * p.x_C;          // GEMM N
  const int tiles_n = (Ktot + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int gHW = p.g_H * p.g_W;
  const int P = p.n_img * gHW;                            // reduction length (pixels)
  const int nk_total = (P + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk_total;
  if (p.split_k > 1) {
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.z * per;
    kt_end = min(nk_total, kt_begin + per);
  }
  const T* __restrict__ dyp = reinterpret_cast<const T*>(p.y);
  const T* __restrict__ xp = reinterpret_cast<const T*>(p.x);

  // loader state: A chunk = (pixel row, co chunk); B chunk = (pixel row, kidx chunk -> fixed tap / channel)
  int a_prow[A_CH], a_col[A_CH];
  bool a_cok[A_CH];
#pragma unroll
  for (int c = 0; c < A_CH; ++c) {
    const int id = tid + 256 * c;
    a_prow[c] = id / A_CPR;
    a_col[c] = (id - a_prow[c] * A_CPR) * CPC;
    a_cok[c] = m0 + a_col[c] < (p.N + CPC - 1) / CPC * CPC;   // dy rows are readable up to the CPC pad
  }
  int b_prow[B_CH], b_col[B_CH], b_ty[B_CH], b_tx[B_CH], b_ch[B_CH];
  bool b_cok[B_CH];
#pragma unroll
  for (int c = 0; c < B_CH; ++c) {
    const int id = tid + 256 * c;
    b_prow[c] = id / B_CPR;
    b_col[c] = (id - b_prow[c] * B_CPR) * CPC;
    const int n = n0 + b_col[c];
    b_cok[c] = n < Ktot;
    const int tap = (b_cok[c] ? n : 0) / p.x_C;
    b_ch[c] = (b_cok[c] ? n : 0) - tap * p.x_C;
    b_ty[c] = tap / p.taps_x;
    b_tx[c] = tap - b_ty[c] * p.taps_x;
  }
  Chunk ra[A_CH], rb[B_CH];
  const Chunk zero = {{0u, 0u, 0u, 0u}};
  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int c = 0; c < A_CH; ++c) {
      const int pix = kt * BK + a_prow[c];
      const bool ok = a_cok[c] && pix < P;
      ra[c] = ok ? ldg_chunk(dyp + (long)pix * p.y_ld + m0 + a_col[c]) : zero;
    }
#pragma unroll
    for (int c = 0; c < B_CH; ++c) {
      const int pix = kt * BK + b_prow[c];
      bool ok = b_cok[c] && pix < P;
      const int pp = ok ? pix : 0;
      const int img = pp / gHW, rem = pp - img * gHW;
      const int gy = rem / p.g_W, gx = rem - gy * p.g_W;
      const int iy = gy * p.mul_y + p.off_y + b_ty[c] * p.step_y, ix = gx * p.mul_x + p.off_x + b_tx[c] * p.step_x;
      ok = ok && (unsigned)iy < (unsigned)p.x_H && (unsigned)ix < (unsigned)p.x_W;
      rb[c] = ok ? ldg_chunk(xp + (((long)img * p.x_H + iy) * p.x_W + ix) * p.x_ld + b_ch[c]) : zero;
    }
  };
  auto store_tiles = [&](int buf) {
    T* At = smem + buf * BK * (LDA + LDB);
    T* Bt = At + BK * LDA;
#pragma unroll
    for (int c = 0; c < A_CH; ++c) *reinterpret_cast<Chunk*>(At + a_prow[c] * LDA + a_col[c]) = ra[c];
#pragma unroll
    for (int c = 0; c < B_CH; ++c) *reinterpret_cast<Chunk*>(Bt + b_prow[c] * LDB + b_col[c]) = rb[c];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (kt_begin < kt_end) {
    load_tiles(kt_begin);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const bool more = kt + 1 < kt_end;
      if (more) load_tiles(kt + 1);
      const T* At = smem + cur * BK * (LDA + LDB);
      const T* Bt = At + BK * LDA;
      if constexpr (sizeof(T) == 2) {
        // transposing read: 16-lane group g covers columns 16*(g&1).., rows 8*(g>>1) + 4*t + q of a 16-pixel step
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
        const int rowoff = 8 * (g >> 1) + q, coloff = 16 * (g & 1) + 4 * pp;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 af[TM], bfr[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const T* base = At + (ks * 16 + rowoff) * LDA + wm * WTM + i * 32 + coloff;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 4 * LDA));
            const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            af[i] = __builtin_bit_cast(bf16x8, v);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const T* base = Bt + (ks * 16 + rowoff) * LDB + wn * WTN + j * 32 + coloff;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 4 * LDB));
            const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bfr[j] = __builtin_bit_cast(bf16x8, v);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int t = 0; t < BK / 2; ++t) {
          float af[TM], bfr[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = to_f32(At[(2 * t + h) * LDA + wm * WTM + i * 32 + r]);
#pragma unroll
          for (int j = 0; j < TN; ++j) bfr[j] = to_f32(Bt[(2 * t + h) * LDB + wn * WTN + j * 32 + r]);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
      }
      if (more) store_tiles(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  // epilogue: fp32 slab [z][Cout][Ktot]
  const int Cout = p.N;
  float* __restrict__ slab = p.slab + (long)blockIdx.z * Cout * Ktot;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (m >= Cout) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + r;
        if (n < Ktot) slab[(long)m * Ktot + n] = acc[i][j][e];
      }
    }
}

// =====================================================================================================
// WGRAD on the LDS-DMA ring: dW[co][kidx] = sum_pix dy[pix][co] * x[gather(pix, tap(kidx))][c(kidx)]
// Both operand tiles are [pixel][128 channels] (channel-contiguous in HBM), filled by buffer_load ... lds
// (bf16: one DMA = 4 pixel rows x 256 B; fp32: 2 rows x 512 B) into a 4-stage ring.  bf16 fragments come from
// the transposing read ds_read_b64_tr_b16; its 4 rows-per-lane-group would hit the same banks on 256-B rows, so
// 16-B chunks are XOR-swizzled with (pixel_row & 3) << 2 on the DMA source side and on the read side.
// =====================================================================================================
// STAGES = 3: with the two-pass epilogue the kernel needs 48 KiB of LDS, so three workgroups share a CU -- split-K launches
// have 700-800 workgroups, which then are all resident at once (conv2's wgrad: 90 -> 69 us); 4 stages gain nothing elsewhere.
// NARROW: layers with few output channels (PWC's 32 / 64-channel dense layers, conv1 of FlowNetS over volumes) fill a quarter / half of
// the 128 tile rows.  Same LDS tiles and DMA schedule, but the four waves then sit side by side over the 128 columns and own only the
// real rows -- NARROW 1: 64 rows x 32 columns per wave (half the MFMAs and fragment reads), NARROW 2: 32 x 32 (a quarter).
template <typename T, int STAGES = 3, int NARROW = 0>
__global__ void __launch_bounds__(256)
conv_wgrad_dma_kernel(const mireg_conv_desc p) {
  constexpr int CPC = Cfg<T>::CPC, BK = Cfg<T>::BK, BM = 128, BN = 128;
  constexpr int WTM = NARROW == 2 ? 32 : 64, WTN = NARROW ? 32 : 64, TM = NARROW == 2 ? 1 : 2, TN = NARROW ? 1 : 2;
  constexpr int ROWB = 128 * (int)sizeof(T);                     // bytes per pixel row of a tile
  constexpr int RPI = 1024 / ROWB;                               // pixel rows per DMA instruction (4 bf16 / 2 fp32)
  constexpr int CPR = ROWB / 16;                                 // 16-B chunks per row (16 / 32)
  constexpr int TILE_BYTES = BK * ROWB;                          // 8 KiB
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int EPI_BYTES = WTM * BN * 4;                        // the epilogue stages 64 rows per pass (two passes)
  constexpr int SMEM_BYTES = STAGES * STAGE_BYTES > EPI_BYTES ? STAGES * STAGE_BYTES : EPI_BYTES;
  constexpr int INSTR = TILE_BYTES / 1024;                       // 8 DMA instructions per tile, 2 per wave
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = NARROW ? 0 : wid >> 1, wn = NARROW ? wid : wid & 1;
  const int r = lane & 31, h = lane >> 5;
  const int Ktot = p.taps_y * p.taps_x * p.x_C;                  // GEMM N
  const int tiles_n = (Ktot + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
  }
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int gHW = p.g_H * p.g_W;
  // Conv3d: one launch per depth tap; an "image" is then one (volume, output slice) pair whose source slice is
  // gz*mul_z + off_z of an x_D deep volume (2-D launches: gD = xD = mulz = 1, off_z = 0)
  const int gD = max(p.g_D, 1), xD = max(p.x_D, 1), mulz = max(p.mul_z, 1);
  const int P = p.n_img * gD * gHW;                              // reduction length (pixels)
  const int nk_total = (P + BK - 1) / BK;
  int kt_begin = 0, kt_end = nk_total;
  if (p.split_k > 1) {
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.z * per;
    kt_end = min(nk_total, kt_begin + per);
  }
  constexpr unsigned kOOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.y), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);

  // lane -> (row inside the DMA instruction, logical 16-B chunk); bf16 rows are swizzled by (row & 3) << 2
  const int lrow = lane / CPR, pch = lane % CPR;
  const int lch = sizeof(T) == 2 ? (pch ^ ((lrow & 3) << 2)) : pch;
  // A operand (dy): column part of the offset is fixed per lane
  const int a_col = m0 + lch * CPC;
  const bool a_cok = a_col < (p.N + CPC - 1) / CPC * CPC;
  // B operand (x): the lane's kidx chunk fixes (tap, channel)
  const int b_n = n0 + lch * CPC;
  const bool b_cok = b_n < Ktot;
  const int b_tap = (b_cok ? b_n : 0) / p.x_C, b_ch = (b_cok ? b_n : 0) - b_tap * p.x_C;
  const int b_ty = b_tap / p.taps_x, b_tx = b_tap - b_ty * p.taps_x;
  const int b_dy = p.off_y + b_ty * p.step_y, b_dx = p.off_x + b_tx * p.step_x;

  // this lane's pixel for its two DMA instructions, kept decomposed (img, gy, gx) and advanced by BK per K-step.
  // Everything the DMA needs is a running 32-bit quantity (buffer offsets are 32-bit; x_bytes, w_bytes < 2^31):
  // byte offsets of the pixel in dy and of its tap-shifted source pixel in x, plus the source coordinates for the
  // zero-padding test, all updated with adds on the carries -- no multiplies in the steady state.
  const unsigned sz = (unsigned)sizeof(T);
  const unsigned PIXB = (unsigned)p.x_ld * sz;                                    // one source pixel
  const unsigned STEPX = (unsigned)p.mul_x * PIXB, ROWB_X = (unsigned)p.x_W * PIXB, STEPY = (unsigned)p.mul_y * ROWB_X;
  const unsigned IMGB = (unsigned)p.x_H * ROWB_X;
  const unsigned y_ldb = (unsigned)p.y_ld * sz;
  int pix[2], p_gy[2], p_gx[2], s_iy[2], s_ix[2], p_gz[2], s_iz[2];
  unsigned a_off[2], b_off[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    pix[c] = kt_begin * BK + (wid + 4 * c) * RPI + lrow;
    const int pp = min(pix[c], P - 1);
    const int img = pp / gHW;
    const int rem = pp - img * gHW;
    p_gy[c] = rem / p.g_W;
    p_gx[c] = rem - p_gy[c] * p.g_W + (pix[c] - pp);            // a tail overshoot stays on the x axis (never used: pok)
    s_iy[c] = p_gy[c] * p.mul_y + b_dy;
    s_ix[c] = p_gx[c] * p.mul_x + b_dx;
    const int vol = img / gD;
    p_gz[c] = img - vol * gD;
    s_iz[c] = p_gz[c] * mulz + p.off_z + (int)blockIdx.y;      // Conv3d: blockIdx.y = depth tap (one launch for all of them)
    a_off[c] = (unsigned)pix[c] * y_ldb + (unsigned)a_col * sz;
    b_off[c] = (unsigned)((vol * xD + s_iz[c]) * (int)IMGB) + (unsigned)(s_iy[c] * (int)ROWB_X) + (unsigned)(s_ix[c] * (int)PIXB) + (unsigned)b_ch * sz;
  }

  auto issue = [&](int stage) {
    unsigned char* At = smem + stage * STAGE_BYTES;
    unsigned char* Bt = At + TILE_BYTES;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool pok = pix[c] < P;
      const unsigned aoff = (pok && a_cok) ? a_off[c] : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void_t)(At + (wid + 4 * c) * 1024), 16, aoff, 0, 0, 0);
      const bool bok = pok && b_cok && (unsigned)s_iy[c] < (unsigned)p.x_H && (unsigned)s_ix[c] < (unsigned)p.x_W && (unsigned)s_iz[c] < (unsigned)xD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t)(Bt + (wid + 4 * c) * 1024), 16, bok ? b_off[c] : kOOB, 0, 0, 0);
      pix[c] += BK;
      a_off[c] += BK * y_ldb;
      p_gx[c] += BK;
      s_ix[c] += BK * p.mul_x;
      b_off[c] += BK * STEPX;
      while (p_gx[c] >= p.g_W) {                                    // carry into rows / images
        p_gx[c] -= p.g_W;
        s_ix[c] -= p.g_W * p.mul_x;
        b_off[c] += STEPY - (unsigned)p.g_W * STEPX;
        s_iy[c] += p.mul_y;
        if (++p_gy[c] == p.g_H) {
          p_gy[c] = 0;
          s_iy[c] -= p.g_H * p.mul_y;
          b_off[c] += (unsigned)mulz * IMGB - (unsigned)p.g_H * STEPY;
          s_iz[c] += mulz;
          if (++p_gz[c] == gD) {                                    // next volume
            p_gz[c] = 0;
            s_iz[c] -= gD * mulz;
            b_off[c] += (unsigned)(xD - gD * mulz) * IMGB;
          }
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

  auto compute = [&](int stage) {
    const unsigned char* At = smem + stage * STAGE_BYTES;
    const unsigned char* Bt = At + TILE_BYTES;
    if constexpr (sizeof(T) == 2) {
      const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
      const int rowoff = 8 * (g >> 1) + q;                       // row & 3 == q
      const int colch = 2 * (g & 1) + (pp >> 1), sub = (pp & 1) * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int ch = ((wm * WTM + i * 32) >> 3) + colch;
          const unsigned char* base = At + (ks * 16 + rowoff) * ROWB + ((ch ^ (q << 2)) << 4) + sub;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 4 * ROWB));
          const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int ch = ((wn * WTN + j * 32) >> 3) + colch;
          const unsigned char* base = Bt + (ks * 16 + rowoff) * ROWB + ((ch ^ (q << 2)) << 4) + sub;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + 4 * ROWB));
          const __attribute__((ext_vector_type(8))) short v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfr[j] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const float* Af = reinterpret_cast<const float*>(At);
      const float* Bf = reinterpret_cast<const float*>(Bt);
#pragma unroll
      for (int t = 0; t < BK / 2; ++t) {
        float af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = Af[(2 * t + h) * 128 + wm * WTM + i * 32 + r];
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = Bf[(2 * t + h) * 128 + wn * WTN + j * 32 + r];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  const int nk = kt_end - kt_begin;
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st)
    if (st < nk) issue(st);
  int it = 0;
  const int steady = nk - (STAGES - 1);
  for (; it < steady; ++it) {
    wait_vmcnt<(STAGES - 2) * 4>();                                // STAGES-2 younger tiles x 4 DMAs stay in flight
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

  // ---- epilogue: accumulators -> LDS fp32 [64][128] per pass -> float4 rows of the slab [z][Cout][Ktot] ----------
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
  const int Cout = p.N;
  const long sld = p.slab_ld > 0 ? p.slab_ld : Ktot;               // Conv3d: the depth taps share one [Cout][taps_z*Ktot] slab
  float* __restrict__ slab = p.slab + (long)blockIdx.z * Cout * sld + (long)blockIdx.y * Ktot;   // depth tap t fills columns [t*Ktot, (t+1)*Ktot)
  const bool vec = (Ktot % 4) == 0 && (sld % 4) == 0;
  for (int hp = 0; hp < (NARROW ? 1 : 2); ++hp) {                   // rows [64 hp, 64 hp + 64) of the tile (NARROW: the real rows only)
    if (hp) __syncthreads();
    if (wm == hp) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            ct[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * BN + wn * WTN + j * 32 + r] = acc[i][j][e];
    }
    __syncthreads();
    for (int c = tid; c < WTM * (BN / 4); c += 256) {
      const int mr = c / (BN / 4), nl = (c - mr * (BN / 4)) * 4;
      const int m = m0 + hp * WTM + mr, n = n0 + nl;
      if (m >= Cout || n >= Ktot) continue;
      const float4 v = *reinterpret_cast<const float4*>(ct + mr * BN + nl);
      float* d = slab + (long)m * sld + n;
      if (vec) stg_u4(d, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
      else { const float vv[4] = {v.x, v.y, v.z, v.w}; for (int q = 0; q < 4 && n + q < Ktot; ++q) d[q] = vv[q]; }
    }
  }
}

template <typename T>
void launch_reduce(const mireg_conv_desc& p, long M, int ncls, hipStream_t stream) {
  const long total = M * p.N;
  long g = (total / 4 + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)g, ncls), dim3(256), 0, stream, p);
}

long largest_class_rows(const mireg_conv_desc& p) {
  const int ncls = p.n_cls > 1 ? p.n_cls : 1;
  long M = 0;
  for (int c = 0; c < ncls; ++c) {
    const long m = (ncls > 1 ? (long)p.n_img * p.cls[c].g_H * p.cls[c].g_W * (p.cls[c].g_D > 0 ? p.cls[c].g_D : (p.g_D > 0 ? p.g_D : 1))
                             : (long)p.n_img * p.g_H * p.g_W * (p.g_D > 0 ? p.g_D : 1));
    M = m > M ? m : M;
  }
  return M;
}

template <typename T>
int launch_fwd(const mireg_conv_desc& p, hipStream_t stream) {
  const int ncls = p.n_cls > 1 ? p.n_cls : 1;
  long M = 0;                                                       // largest class decides the grid
  for (int c = 0; c < ncls; ++c) {
    const long m = (ncls > 1 ? (long)p.n_img * p.cls[c].g_H * p.cls[c].g_W * (p.cls[c].g_D > 0 ? p.cls[c].g_D : (p.g_D > 0 ? p.g_D : 1))
                             : (long)p.n_img * p.g_H * p.g_W * (p.g_D > 0 ? p.g_D : 1));
    M = m > M ? m : M;
  }
  const int z = p.split_k > 1 ? p.split_k : 1;
  const int bn = (p.tile_n == 64 || p.tile_n == 128) ? p.tile_n : (p.N > 64 ? 128 : (p.N > 32 ? 64 : 32));
  if (bn == 128) {
    dim3 grid((unsigned)(((M + 127) / 128) * ((p.N + 127) / 128)), ncls, z);
    hipLaunchKernelGGL((conv_gemm_dma_kernel<T, 128, 128, 2, 2>), grid, dim3(256), 0, stream, p);
  } else if (bn == 64) {
    dim3 grid((unsigned)(((M + 127) / 128) * ((p.N + 63) / 64)), ncls, z);
    hipLaunchKernelGGL((conv_gemm_dma_kernel<T, 128, 64, 2, 2>), grid, dim3(256), 0, stream, p);
  } else {
    dim3 grid((unsigned)((M + 127) / 128), ncls, z);
    hipLaunchKernelGGL((conv_gemm_dma_kernel<T, 128, 32, 4, 1>), grid, dim3(256), 0, stream, p);
  }
  if (z > 1) launch_reduce<T>(p, M, ncls, stream);
  return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
}

// MIREG_WGRAD_WIDE_ROWS=1 (environment, read once): A/B switch that keeps layers with <= 64 output channels on the 2 x 2 wave layout
static const bool kWgradWideRows = [] { const char* e = getenv("MIREG_WGRAD_WIDE_ROWS"); return e && e[0] == '1'; }();

template <typename T>
int launch_wgrad(const mireg_conv_desc& p, hipStream_t stream) {
  const int Cout = p.N;
  const int Ktot = p.taps_y * p.taps_x * p.x_C;
  const int z = p.split_k > 1 ? p.split_k : 1;
  // Conv3d: all depth taps in one launch (grid.y): the tap-variants of a pixel chunk run back to back and share its dy / x tiles in the
  // cache hierarchy instead of sweeping both operands from HBM once per tap (conv1 of FlowNetS-3D: 7 launches x 112 us -> one)
  dim3 grid((unsigned)(((Cout + 127) / 128) * ((Ktot + 127) / 128)), (unsigned)(p.taps_z > 1 ? p.taps_z : 1), z);
  if (p.x_bytes > 0 && p.w_bytes > 0 && p.x_bytes < (1L << 31) && p.w_bytes < (1L << 31))
    // 3 stages = 48 KiB LDS, three workgroups per CU; 4 stages = 64 KiB, two per CU (leaves more of the CU to a concurrent stream)
    if (Cout <= 32 && !kWgradWideRows) hipLaunchKernelGGL((conv_wgrad_dma_kernel<T, 3, 2>), grid, dim3(256), 0, stream, p);
    else if (Cout <= 64 && !kWgradWideRows) hipLaunchKernelGGL((conv_wgrad_dma_kernel<T, 3, 1>), grid, dim3(256), 0, stream, p);
    else if (p.stages == 4) hipLaunchKernelGGL((conv_wgrad_dma_kernel<T, 4>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_dma_kernel<T, 3>), grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<T, 128, 128>), grid, dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
}

bool desc_ok(const mireg_conv_desc* p, bool wgrad) {
  if (!p || !p->x || p->x_ld <= 0 || p->x_H <= 0 || p->x_W <= 0 || p->n_img <= 0 || p->g_H <= 0 || p->g_W <= 0) return false;
  if (p->taps_y <= 0 || p->taps_x <= 0 || p->N <= 0) return false;
  const int cpc = p->dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  if (p->dtype != MIREG_DTYPE_BF16 && p->dtype != MIREG_DTYPE_F32) return false;
  if (p->x_C <= 0 || p->x_C % cpc || p->x_ld % cpc || ((uintptr_t)p->x % 16)) return false;
  if ((long)p->n_img * p->g_H * p->g_W * (p->g_D > 0 ? p->g_D : 1) >= (1L << 31)) return false;
  if (wgrad && p->taps_z > 1 && !(p->x_bytes > 0 && p->w_bytes > 0 && p->x_bytes < (1L << 31) && p->w_bytes < (1L << 31) && p->algo != 3 &&
                                   p->slab_ld >= (long)p->taps_z * p->taps_y * p->taps_x * p->x_C)) return false;   // depth taps = grid.y of the LDS-DMA kernel only
  if (wgrad && (p->g_D > 1 || p->x_D > 1) && !(p->x_bytes > 0 && p->w_bytes > 0 && p->x_bytes < (1L << 31) && p->w_bytes < (1L << 31))) return false;
  if (wgrad) {
    if (!p->y || !p->slab || p->y_ld % cpc || ((uintptr_t)p->y % 16)) return false;
  } else {
    if (p->n_cls < 0 || p->n_cls > 8) return false;
    if (p->n_cls <= 1 && (!p->w || p->w_ld % cpc || ((uintptr_t)p->w % 16) || p->w_bytes <= 0 || p->w_bytes >= (1L << 31))) return false;
    if (p->x_bytes <= 0 || p->x_bytes >= (1L << 31)) return false;
    for (int c = 0; c < (p->n_cls > 1 ? p->n_cls : 0); ++c)
      if (!p->cls[c].w || p->cls[c].w_ld % cpc || ((uintptr_t)p->cls[c].w % 16) || p->cls[c].w_bytes <= 0 || p->cls[c].w_bytes >= (1L << 31) || p->cls[c].g_H <= 0 || p->cls[c].g_W <= 0 || p->cls[c].taps_y <= 0 || p->cls[c].taps_x <= 0) return false;
    if (!p->y && !p->y32) return false;
    if (p->split_k > 1 && !p->slab) return false;
  }
  return true;
}

}  // namespace

extern "C" int mireg_conv_halo_try(const mireg_conv_desc* p, hipStream_t stream);         // conv_halo.hip
extern "C" int mireg_conv_wgrad_halo_try(const mireg_conv_desc* p, hipStream_t stream);   // conv_wgrad_halo.hip
extern "C" int mireg_conv_wide_try(const mireg_conv_desc* p, hipStream_t stream);         // conv_wide.hip
extern "C" int mireg_conv_wgrad_wide_try(const mireg_conv_desc* p, hipStream_t stream);   // conv_wgrad_wide.hip

extern "C" {

int mireg_conv_gemm(const mireg_conv_desc* desc, hipStream_t stream) {
  if (!desc_ok(desc, false)) return MIREG_ERR_ARG;
  if (desc->algo == 3) {                                            // 3: the 256-pixel 8-wave tile (conv_wide.hip) or UNSUPPORTED
    const int rc = mireg_conv_wide_try(desc, stream);
    if (rc == -100) return MIREG_ERR_UNSUPPORTED;
    if (rc != MIREG_OK) return rc;
    if (desc->split_k > 1) {
      launch_reduce<__bf16>(*desc, largest_class_rows(*desc), desc->n_cls > 1 ? desc->n_cls : 1, stream);
      return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
    }
    return MIREG_OK;
  }
  if (desc->algo != 1) {                                            // 0: halo-staged kernel when it applies, 2: require it
    const int rc = mireg_conv_halo_try(desc, stream);
    if (rc != -100) return rc;
    if (desc->algo == 2) return MIREG_ERR_UNSUPPORTED;
  }
  return desc->dtype == MIREG_DTYPE_BF16 ? launch_fwd<__bf16>(*desc, stream) : launch_fwd<float>(*desc, stream);
}

int mireg_conv_wgrad(const mireg_conv_desc* desc, hipStream_t stream) {
  if (!desc_ok(desc, true)) return MIREG_ERR_ARG;
  if (desc->algo == 3) {                                            // 3: the 256 x 256 8-wave tile (conv_wgrad_wide.hip) or UNSUPPORTED
    const int rc = mireg_conv_wgrad_wide_try(desc, stream);
    return rc == -100 ? MIREG_ERR_UNSUPPORTED : rc;
  }
  if (desc->algo != 1) {                                            // 0: halo-staged kernel when it applies, 2: require it
    const int rc = mireg_conv_wgrad_halo_try(desc, stream);
    if (rc != -100) return rc;
    if (desc->algo == 2) return MIREG_ERR_UNSUPPORTED;
  }
  return desc->dtype == MIREG_DTYPE_BF16 ? launch_wgrad<__bf16>(*desc, stream) : launch_wgrad<float>(*desc, stream);
}

}  // extern "C"
