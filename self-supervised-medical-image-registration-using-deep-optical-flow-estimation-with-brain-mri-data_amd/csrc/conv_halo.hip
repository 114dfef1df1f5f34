// conv_halo.hip -- halo-staged implicit GEMM for unit-stride gathers (K1-K4, second generation).
//
// Serves the same contractions as conv_gemm.hip's ring kernel whenever the gather is unit-stride:
//   conv forward with stride 1            (FlowNetS/util.py:17-30 conv3_1/4_1/5_1/6_1, PWC/models/PWCNet.py:24-31)
//   conv backward-data, every stride      (per output-pixel parity class the gather over dy is unit-stride)
//   ConvTranspose2d forward               (FlowNetS/util.py:49-55, PWCNet.py:33-34: the backward-data form)
//
// Why: the ring kernel re-fetches every input pixel once per tap.  A 128x128 tile then moves 16 KB of L2->LDS traffic per
// MFLOP (64 FLOP/B) and a CU's LDS-DMA path delivers ~65 GB/s, i.e. <= 4.2 TFLOP/s per CU = 43 % of its matrix peak before any
// stall.  Here K is ordered (channel chunk, tap): the (rows + ty - 1) x (W + tx - 1) input halo of a
// pixel tile is staged ONCE per 64-byte channel chunk and all ty*tx taps read their A fragments from it at shifted LDS rows;
// only the weights still stream per tap.  256 pixels x 128 channels x 3x3: 200 FLOP/B instead of 64.
//
// Structure (one workgroup = 4 waves, 2 x 2 over a BM-pixel x BN-channel tile):
//   * pixel tile = BM/16 segments of 16 consecutive pixels of one image row (full rows of a 16/32/64 wide grid);
//   * A halo: double-buffered, (R+ty-1) rows x (W+tx-1) pixels x 64 B, filled by buffer_load ... lds one chunk ahead;
//   * B (weights of one tap x chunk): BN rows x 64 B in a 3/4-stage ring, one stage per K-step;
//   * one raw s_barrier per K-step, counted s_waitcnt vmcnt(N), N recomputed per step (halo DMAs are issued in bursts);
//   * LDS rows are 64 B, 16-B chunks XOR-swizzled with (row >> 2) & 3 on the DMA source side and on the read side;
//     the M rows of a 32-row MFMA block are assigned to pixels so that each hardware ds_read_b128 lane group
//     ({0-3,12-15,20-27} / {4-11,16-19,28-31}) reads 16 CONSECUTIVE halo rows: conflict-free at any tap shift.
#include "mireg_common.h"
#include "../../include/mireg.h"
#include <stdlib.h>

using namespace mireg;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

struct alignas(16) Chunk { uint32_t w[4]; };
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
template <> struct Cfg<float>  { static constexpr int CPC = 4, BK = 16; };
template <> struct Cfg<__bf16> { static constexpr int CPC = 8, BK = 32; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

typedef __attribute__((address_space(3))) void* lds_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {   // n is wave-uniform; waiting for fewer outstanding is always safe
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 11: wait_vmcnt<11>(); break;
    default: wait_vmcnt<12>(); break;
  }
}

// halo capacity (16-row DMA instructions per buffer): BM=128 -> 17 (2x66+..=264 rows), BM=256 -> 25 (6x66=396 rows)
template <int BM> struct HaloCap { static constexpr int INSTR = BM == 128 ? 17 : 25; };

template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256, 2)
conv_halo_kernel(const mireg_conv_desc pd) {
  constexpr int CPC = Cfg<T>::CPC, BK = Cfg<T>::BK;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int S = BM / 16;                                       // 16-pixel segments per tile
  constexpr int B_GROUPS = BN / 16, B_PW = B_GROUPS / 4;           // DMA instructions per stage / per wave
  constexpr int BSTAGES = BM == 128 ? 4 : 3;
  constexpr int D = BSTAGES - 1;                                   // K-steps of weights in flight
  constexpr int B_STAGE_BYTES = BN * 64;
  constexpr int A_INSTR = HaloCap<BM>::INSTR, A_BUF_BYTES = A_INSTR * 1024, A_PW = (A_INSTR + 3) / 4;
  constexpr int RING_BYTES = 2 * A_BUF_BYTES + BSTAGES * B_STAGE_BYTES;
  constexpr int EPI_BYTES = 128 * BN * 4 + 128 * 8;                // one 128-row pass of the epilogue
  constexpr int SMEM_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static_assert(TM >= 1 && TN >= 1 && B_PW >= 1 && A_PW <= 7, "bad tile");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

  mireg_conv_desc p = pd;
  const int cls = blockIdx.y;
  if (pd.n_cls > 1) {
    const mireg_conv_cls k = pd.cls[cls];
    p.taps_y = k.taps_y; p.taps_x = k.taps_x; p.off_y = k.off_y; p.off_x = k.off_x; p.g_H = k.g_H; p.g_W = k.g_W;
    p.y_off_y = k.y_off_y; p.y_off_x = k.y_off_x; p.w = k.w; p.w_ld = k.w_ld; p.w_bytes = k.w_bytes;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int r = lane & 31, h = lane >> 5;

  // ---- tile -> (image, first grid row, column tile) ----------------------------------------------------------------
  const int W = p.g_W, Hg = p.g_H;
  const int spr = W >> 4;                                          // segments per grid row
  const int R = S / spr;                                           // grid rows per tile
  const int tiles_img = Hg / R;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int my_tiles = p.n_img * tiles_img * tiles_n;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
  }
  if (bid >= my_tiles) return;
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int img = tile_m / tiles_img, y0 = (tile_m - img * tiles_img) * R;
  const int n0 = tile_n * BN;

  // ---- halo geometry: input rows iy_min .. iy_min+HR-1, pixels ix_min .. ix_min+HP-1 ------------------------------
  const int ty_n = p.taps_y, tx_n = p.taps_x, taps = ty_n * tx_n;
  const int HR = R + ty_n - 1, HP = W + tx_n - 1;
  const int iy_min = y0 + p.off_y + (p.step_y < 0 ? -(ty_n - 1) : 0);
  const int ix_min = p.off_x + (p.step_x < 0 ? -(tx_n - 1) : 0);
  const int halo_rows = HR * HP;
  const int NA = (halo_rows + 15) >> 4;                            // DMA instructions per halo buffer (<= A_INSTR, host-checked)
  const int cpt = p.x_C / CPC;                                     // 16-byte chunks per pixel
  const int nchunks = (cpt + 3) >> 2;                              // 64-byte channel chunks
  const int nsteps = nchunks * taps;

  constexpr unsigned kOOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // DMA lane roles: lane L of an instruction covers LDS row L>>2, physical chunk L&3 = logical chunk ^ ((row>>2)&3)
  const int lrow = lane >> 2;
  const int kc = (lane & 3) ^ ((lane >> 4) & 3);
  unsigned a_src[A_PW];                                            // byte offset of the halo pixel (channel 0) or OOB
  int nA = 0;
#pragma unroll
  for (int c = 0; c < A_PW; ++c) {
    const int q = wid + 4 * c;
    const int l = q * 16 + lrow;
    const int hy = l / HP, hx = l - hy * HP;
    const int iy = iy_min + hy, ix = ix_min + hx;
    const bool ok = q < NA && l < halo_rows && (unsigned)iy < (unsigned)p.x_H && (unsigned)ix < (unsigned)p.x_W;
    a_src[c] = ok ? (unsigned)((((long)img * p.x_H + iy) * p.x_W + ix) * p.x_ld * (long)sizeof(T)) + (unsigned)(kc * 16) : kOOB;
    nA += q < NA ? 1 : 0;
  }
  nA = __builtin_amdgcn_readfirstlane(nA);
  unsigned b_src[B_PW];
#pragma unroll
  for (int c = 0; c < B_PW; ++c) {
    const int n = n0 + (wid + 4 * c) * 16 + lrow;
    b_src[c] = n < p.N ? (unsigned)((long)n * p.w_ld * (long)sizeof(T)) + (unsigned)(kc * 16) : kOOB;
  }

  auto issueA = [&](int chunk) {
    unsigned char* Ab = smem + (chunk & 1) * A_BUF_BYTES;
    const bool cok = chunk * 4 + kc < cpt;
    const unsigned cb = (unsigned)(chunk * 64);
#pragma unroll
    for (int c = 0; c < A_PW; ++c) {
      if (wid + 4 * c < NA) {                                       // wave-uniform
        const unsigned off = (cok && a_src[c] != kOOB) ? a_src[c] + cb : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t)(Ab + (wid + 4 * c) * 1024), 16, off, 0, 0, 0);
      }
    }
  };
  // weights of K-step (chunk, tap): k = tap * x_C + chunk * BK
  int i_chunk = 0, i_tap = 0, i_stage = 0;                           // next weight stage to issue
  auto issueB = [&]() {
    unsigned char* Bs = smem + 2 * A_BUF_BYTES + i_stage * B_STAGE_BYTES;
    const bool cok = i_chunk * 4 + kc < cpt;
    const unsigned kb = (unsigned)((i_tap * p.x_C + i_chunk * BK) * (int)sizeof(T));
#pragma unroll
    for (int c = 0; c < B_PW; ++c) {
      const unsigned off = (cok && b_src[c] != kOOB) ? b_src[c] + kb : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t)(Bs + (wid + 4 * c) * 1024), 16, off, 0, 0, 0);
    }
    if (++i_tap == taps) { i_tap = 0; ++i_chunk; }
    if (++i_stage == BSTAGES) i_stage = 0;
  };

  // ---- fragment rows: lane r of a 32-row block -> (segment 2i+sel, pixel xi) so that hardware lane groups read 16
  // consecutive halo rows (header comment) ----------------------------------------------------------------------------
  const int quad = r >> 2;
  const int sel = (0x96 >> quad) & 1, xi = ((quad >> 1) << 2) | (r & 3);
  int a_row0[TM];                                                    // LDS row of this lane's pixel at tap shift (0, 0)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int seg = wm * (S / 2) + 2 * i + sel;
    const int j = seg / spr, xs = (seg - j * spr) * 16 + xi;
    a_row0[i] = j * HP + xs;
  }
  int b_off[TN], b_swz[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { const int row = wn * WTN + j * 32 + r; b_off[j] = row * 64; b_swz[j] = (row >> 2) & 3; }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int abuf, int bstage, int shift) {
    const unsigned char* As = smem + abuf * A_BUF_BYTES;
    const unsigned char* Bs = smem + 2 * A_BUF_BYTES + bstage * B_STAGE_BYTES;
    int a_off[TM], a_swz[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = a_row0[i] + shift; a_off[i] = row * 64; a_swz[i] = (row >> 2) & 3; }
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + a_off[i] + (((ks * 2 + h) ^ a_swz[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + b_off[j] + (((ks * 2 + h) ^ b_swz[j]) << 4));
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
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + a_off[i] + (((g * 2 + h) ^ a_swz[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const f32x4*>(Bs + b_off[j] + (((g * 2 + h) ^ b_swz[j]) << 4));
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

  // ---- pipeline ---------------------------------------------------------------------------------------------------
  issueA(0);
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nsteps) issueB();
  // DMA issue order per wave: A(0) B(0..D-1) | step s: [A(c+1) if t == 0] B(s+D).  B(s) has landed once at most
  // nB * min(D-1, steps left) weight DMAs plus (if the chunk's halo burst was issued after it, i.e. 1 <= t <= D-1) nA halo
  // DMAs are still outstanding; the halo of chunk c is older than B(c, 0) by construction.
  int c = 0, ky = 0, kx = 0, t = 0, bstage = 0;
  const int sy_sign = p.step_y < 0 ? -1 : 1, sx_sign = p.step_x < 0 ? -1 : 1;
  const int sy0 = p.step_y < 0 ? ty_n - 1 : 0, sx0 = p.step_x < 0 ? tx_n - 1 : 0;
  for (int s = 0; s < nsteps; ++s) {
    const int leftB = min(D - 1, nsteps - 1 - s);
    const bool more_chunks = c + 1 < nchunks;
    const int extraA = (t >= 1 && t <= D - 1 && more_chunks) ? nA : 0;
    wait_vmcnt_dyn(B_PW * leftB + extraA);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t == 0 && more_chunks) issueA(c + 1);
    if (s + D < nsteps) issueB();
    const int shift = (sy0 + sy_sign * ky) * HP + (sx0 + sx_sign * kx);
    compute(c & 1, bstage, shift);
    if (++bstage == BSTAGES) bstage = 0;
    ++t;
    if (++kx == tx_n) { kx = 0; if (++ky == ty_n) { ky = 0; t = 0; ++c; } }
  }

  // ---- epilogue: 128 virtual rows per pass -> LDS fp32 -> 16-byte coalesced row stores ------------------------------
  float* ct = reinterpret_cast<float*>(smem);
  long* rowoff = reinterpret_cast<long*>(smem + 128 * BN * 4);
  T* __restrict__ yp = reinterpret_cast<T*>(p.y);
  constexpr int V = 16 / (int)sizeof(T);
  constexpr int CPR = BN / V;
  const bool vec = (p.y_ld % V) == 0 && (reinterpret_cast<uintptr_t>(yp) % 16) == 0;
#pragma unroll
  for (int hp = 0; hp < BM / 128; ++hp) {
    __syncthreads();                                                 // ring (or the previous pass) no longer read
    constexpr int WPP = 128 / WTM;                                   // m-halves of the wave grid per pass (BM=128: 2, BM=256: 1)
    if (wm / WPP == hp || WPP == 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nl = wn * WTN + j * 32 + r, n = n0 + nl;
          const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ml = (WPP == 2 ? wm * WTM : 0) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            float v = acc[i][j][e] + bv;
            v = v > 0.f ? v : v * p.slope;
            ct[ml * BN + nl] = v;
          }
        }
    }
    if (tid < 128) {                                                 // virtual row -> output pixel
      const int mv = hp * 128 + tid;                                 // virtual row of the tile
      const int wmv = mv / WTM, rem = mv - wmv * WTM, i = rem >> 5, rho = rem & 31, q = rho >> 2;
      const int sl = (0x96 >> q) & 1, xq = ((q >> 1) << 2) | (rho & 3);
      const int seg = wmv * (S / 2) + 2 * i + sl;
      const int j = seg / spr, gx = (seg - j * spr) * 16 + xq, gy = y0 + j;
      rowoff[tid] = ((long)img * p.y_H + gy * p.y_mul_y + p.y_off_y) * p.y_W + gx * p.y_mul_x + p.y_off_x;
    }
    __syncthreads();
    for (int cix = tid; cix < 128 * CPR; cix += 256) {
      const int ml = cix / CPR, nl = (cix - ml * CPR) * V, n = n0 + nl;
      if (n >= p.N) continue;
      const long pix = rowoff[ml];
      float v[V];
#pragma unroll
      for (int q = 0; q < V; q += 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(ct + ml * BN + nl + q);
        v[q] = t4.x; v[q + 1] = t4.y; v[q + 2] = t4.z; v[q + 3] = t4.w;
      }
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
        }
      }
    }
  }
}

// geometry of one class: does the halo path apply, and with how many tiles at BM rows per tile?
bool class_ok(int gH, int gW, int ty, int tx, int bm, long* tiles_m, int n_img) {
  if (gW != 16 && gW != 32 && gW != 64) return false;
  const int S = bm / 16, spr = gW / 16;
  if (S % spr) return false;
  const int R = S / spr;
  if (R < 1 || gH % R) return false;
  if (ty * tx < 4 || ty > 5 || tx > 5) return false;                 // >= 4 K-steps per chunk: the halo burst lands in time
  const int cap = bm == 128 ? HaloCap<128>::INSTR : HaloCap<256>::INSTR;
  if (((R + ty - 1) * (gW + tx - 1) + 15) / 16 > cap) return false;
  *tiles_m = (long)n_img * (gH / R);
  return true;
}

}  // namespace

// 1 = eligible; tiles_out[0] / [1] = workgroups per class-launch at BM = 128 / 256 (0 when that tile does not apply)
extern "C" int mireg_conv_halo_eligible(const mireg_conv_desc* p, long* tiles_out) {
  tiles_out[0] = tiles_out[1] = 0;
  if (!p || p->split_k > 1 || p->y32 || !p->y) return 0;
  if (p->mul_y != 1 || p->mul_x != 1 || abs(p->step_y) != 1 || abs(p->step_x) != 1) return 0;
  if (p->x_D > 1 || p->g_D > 1 || p->taps_z > 1 || p->y_D > 1 || p->off_z != 0 || p->y_off_z != 0) return 0;
  if (p->N < 64) return 0;
  if (p->dtype != MIREG_DTYPE_BF16 && p->dtype != MIREG_DTYPE_F32) return 0;
  const int ncls = p->n_cls > 1 ? p->n_cls : 1;
  for (int b = 0; b < 2; ++b) {
    const int bm = b ? 256 : 128;
    long worst = 0;
    bool ok = true;
    for (int c = 0; c < ncls && ok; ++c) {
      long tm = 0;
      ok = ncls > 1 ? class_ok(p->cls[c].g_H, p->cls[c].g_W, p->cls[c].taps_y, p->cls[c].taps_x, bm, &tm, p->n_img)
                    : class_ok(p->g_H, p->g_W, p->taps_y, p->taps_x, bm, &tm, p->n_img);
      worst = tm > worst ? tm : worst;
    }
    if (ok) tiles_out[b] = worst;
  }
  return tiles_out[0] > 0 || tiles_out[1] > 0;
}

template <typename T>
static int launch_halo_t(const mireg_conv_desc& p, int bm, long tiles_m, hipStream_t stream) {
  const int ncls = p.n_cls > 1 ? p.n_cls : 1;
  const int bn = (p.tile_n == 64 || p.N <= 64) ? 64 : 128;
  const unsigned gx = (unsigned)(tiles_m * ((p.N + bn - 1) / bn));
  dim3 grid(gx, ncls, 1);
  if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_halo_kernel<T, 128, 128>), grid, dim3(256), 0, stream, p);
  else if (bm == 128) hipLaunchKernelGGL((conv_halo_kernel<T, 128, 64>), grid, dim3(256), 0, stream, p);
  else if (bn == 128) hipLaunchKernelGGL((conv_halo_kernel<T, 256, 128>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv_halo_kernel<T, 256, 64>), grid, dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
}

// Called by mireg_conv_gemm (conv_gemm.hip): >= 0 -> launched (MIREG_OK / error), -100 -> not applicable.
extern "C" int mireg_conv_halo_try(const mireg_conv_desc* p, hipStream_t stream) {
  long tiles[2];
  if (!mireg_conv_halo_eligible(p, tiles)) return -100;
  int bm;
  if (p->tile_m == 128 || p->tile_m == 256) {
    bm = p->tile_m;
    if (!tiles[bm == 256]) return -100;
  } else {
    // default: the larger tile (200 FLOP/B) when it still gives every CU a workgroup, else the smaller one
    const int bn = (p->tile_n == 64 || p->N <= 64) ? 64 : 128;
    const long tn = (p->N + bn - 1) / bn;
    const int ncls = p->n_cls > 1 ? p->n_cls : 1;
    bm = (tiles[1] && tiles[1] * tn * ncls >= 224) ? 256 : (tiles[0] ? 128 : 256);
  }
  const long tm = tiles[bm == 256];
  return p->dtype == MIREG_DTYPE_BF16 ? launch_halo_t<__bf16>(*p, bm, tm, stream) : launch_halo_t<float>(*p, bm, tm, stream);
}
