// conv_wide.hip -- the implicit-GEMM convolution on a 256-pixel, 8-wave macro-tile (round 3).
//
// Same contraction and descriptor as conv_gemm.hip (FWD: FlowNetS/util.py:17-46, PWC/models/PWCNet.py:24-31; DGRAD form ==
// ConvTranspose2d forward / Conv2d backward-data: FlowNetS/util.py:49-55), bf16 operands, fp32 accumulate.  What changes is the
// tile: what bounds these GEMMs on MI355X is the L2 -> LDS rate of a CU (about 30 B/clk) against 4096 MFMA FLOP/clk, so a
// workgroup needs BM*BN/(BM+BN) >= 128 FLOP per loaded byte to keep the matrix pipe fed; the 128x128 tile of conv_gemm.hip has 64.
//
//   tile      256 pixels x 256 channels (2 x 4 waves, 128 x 64 per wave)  or  256 x 128 (4 x 2 waves, 64 x 64 per wave)
//   K-step    64 bf16 = one full 128-byte line per row and DMA lane group (8 rows x 128 B per buffer_load ... lds)
//   waves     8 = two per SIMD, running half a phase apart ("ping-pong"): in every barrier interval one wave of a SIMD
//             issues 16 v_mfma_f32_16x16x32_bf16 while its partner reads the next fragments from LDS, issues the LDS-DMAs
//             of a later K-step and does the gather address arithmetic
//   LDS       256x256: 2 stages x 64 KiB, refilled piece by piece (a piece = 64 rows of both pixel halves, or 128 weight
//             rows) one phase after its last fragment read, two pieces always in flight (counted vmcnt(4));
//             256x128: 3 stages x 48 KiB, refilled per K-step, one K-step always in flight (vmcnt(6))
//   swizzle   16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7): conflict-free for ds_read_b128 fragment reads of
//             16 rows x 4 chunks; applied on the DMA SOURCE side (the LDS image of a DMA is lane-linear) and on the read side
//   MFMA      operands swapped (A = weight rows, B = pixel rows) so that a lane ends up with 4 consecutive channels of one
//             pixel: the accumulators go to the LDS staging tile as 16-byte writes
//
// Synchronisation rules the schedule below is built on (two wave groups G0 = waves 0-3, G1 = waves 4-7; G1 runs one barrier
// behind G0; a phase = load segment L, barrier, MFMA segment M, barrier):
//   RAW  a DMA issued by any wave is covered by that wave's counted vmcnt at the end of ITS L(q-1) or earlier before any
//        wave reads it in L(q);
//   WAR  every L segment ends with s_waitcnt lgkmcnt(0) before its barrier, so a buffer read in L(q) may be re-filled by any
//        wave from L(q+1) on.
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_void_t;

namespace {

struct alignas(16) Chunk { uint32_t w[4]; };
// global (address space 1) accesses: through a struct-passed pointer hipcc otherwise emits flat loads / stores
__device__ __forceinline__ Chunk ldg_chunk(const void* p) {
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

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void seg_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int BM = 256;
constexpr unsigned kOOB = 0x80000000u;

template <int BN>
__global__ void __launch_bounds__(512)
conv_wide_kernel(const mireg_conv_desc pd) {
  constexpr int NST = BN == 256 ? 2 : 3;                  // LDS stages
  constexpr int BOFF = BM * 128;                          // weights follow the pixel rows inside a stage
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NBC = BN / 64;                            // 64-row weight blocks = DMAs per wave and K-step
  constexpr int TI = BN == 256 ? 8 : 4;                   // 16-pixel fragments per wave
  constexpr int EPI_ROWS = BN == 256 ? 128 : 256;         // rows staged per epilogue pass
  constexpr int RING = NST * STAGE;
  constexpr int EPI = EPI_ROWS * BN * 4 + BM * 8;
  constexpr int SMEM = RING > EPI ? RING : EPI;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

  mireg_conv_desc p = pd;
  const int cls = blockIdx.y;
  if (pd.n_cls > 1) {
    const mireg_conv_cls k = pd.cls[cls];
    p.taps_y = k.taps_y; p.taps_x = k.taps_x; p.off_y = k.off_y; p.off_x = k.off_x; p.g_H = k.g_H; p.g_W = k.g_W;
    p.y_off_y = k.y_off_y; p.y_off_x = k.y_off_x; p.w = k.w; p.w_ld = k.w_ld; p.w_bytes = k.w_bytes;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;                                // ping-pong group: waves 0-3 / 4-7 = one wave of each SIMD
  // optional depth axis (Conv3d): all *_D / *_z fields are 0 for 2-D launches and then collapse to extent 1
  const int xD = max(p.x_D, 1), gD = max(p.g_D, 1), tapsZ = max(p.taps_z, 1), yD = max(p.y_D, 1), ymz = max(p.y_mul_z, 1);
  const int gHW = p.g_H * p.g_W, gDHW = gD * gHW;
  const int M = p.n_img * gDHW;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int bid = blockIdx.x;
  {   // XCD-aware tile order (blocks b, b+8, .. share an XCD): each XCD gets a contiguous run of tiles
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
  }
  if (bid >= tiles_m * tiles_n) return;                   // classes with fewer tiles than the grid (whole workgroup leaves)
  int tile_m, tile_n;
  if (M < p.N) { tile_n = bid / tiles_m; tile_m = bid - tile_n * tiles_m; }
  else { tile_m = bid / tiles_n; tile_n = bid - tile_m * tiles_n; }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int K = tapsZ * p.taps_y * p.taps_x * p.x_C;
  const int nk_total = (K + 63) / 64;
  int kt_begin = 0, kt_end = nk_total;
  if (p.split_k > 1) {
    const int per = (nk_total + p.split_k - 1) / p.split_k;
    kt_begin = blockIdx.z * per;
    kt_end = min(nk_total, kt_begin + per);
  }
  const int nk = max(kt_end - kt_begin, 0);
  float* const slab_base = p.slab ? p.slab + (long)cls * pd.slab_cls_stride : nullptr;

  // ---- LDS-DMA source state.  Lane L of a DMA covers row L>>3 of its 8-row block, physical chunk L&7 ----
  const int lrow = lane >> 3;
  const int kc = (lane & 7) ^ ((4 * wid + (lane >> 4)) & 7);       // logical 16-B chunk of the K-step this lane fetches
  const int ldb = (int)p.x_ld * 2;                                   // bytes per input pixel
  int a_pix[4], a_iz0[4], a_iy0[4], a_ix0[4];                       // byte offset of (img, iz0, iy0, ix0) (mod 2^32), tap-0 coordinates
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int m = m0 + 8 * wid + 64 * c + lrow;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int img = mm / gDHW, r3 = mm - img * gDHW;
    const int gz = r3 / gHW, rem = r3 - gz * gHW;
    const int gy = rem / p.g_W, gx = rem - gy * p.g_W;
    a_iz0[c] = gz * p.mul_z + p.off_z;
    a_iy0[c] = ok ? gy * p.mul_y + p.off_y : -(1 << 28);             // rows past M never pass the range test
    a_ix0[c] = gx * p.mul_x + p.off_x;
    a_pix[c] = (int)((unsigned)((long)img * xD * p.x_H * p.x_W * ldb) +
                     (unsigned)(((a_iz0[c] * p.x_H + (gy * p.mul_y + p.off_y)) * p.x_W + a_ix0[c]) * ldb));
  }
  unsigned b_base[NBC];
#pragma unroll
  for (int c = 0; c < NBC; ++c) {
    const int n = n0 + 8 * wid + 64 * c + lrow;
    b_base[c] = n < p.N ? (unsigned)((long)n * p.w_ld * 2) : kOOB;
  }
  const int cpt = p.x_C >> 3;                                       // 16-B chunks per tap (>= 8: at most one tap step per K-step)
  int tz, ty, tx, cc;
  {
    const int q = kt_begin * 8 + kc;
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
  // per-tap quantities of this lane: coordinate shifts, byte shift; a_cur[c] = source byte offset of row c at the current tap
  int dzv = tz * p.step_z, dyv = ty * p.step_y, dxv = tx * p.step_x, tap_off = ((dzv * p.x_H + dyv) * p.x_W + dxv) * ldb;
  unsigned a_cur[4];
  bool tap_moved = true;                                              // wave-uniform: some lane stepped to a new tap
  auto row_update = [&](int c) {
    const int iz = a_iz0[c] + dzv, iy = a_iy0[c] + dyv, ix = a_ix0[c] + dxv;
    const bool ok = tz < tapsZ && (unsigned)iz < (unsigned)xD && (unsigned)iy < (unsigned)p.x_H && (unsigned)ix < (unsigned)p.x_W;
    a_cur[c] = ok ? (unsigned)(a_pix[c] + tap_off) : kOOB;
  };
#pragma unroll
  for (int c = 0; c < 4; ++c) row_update(c);
  unsigned b_k = (unsigned)((kt_begin * 64 + kc * 8) * 2);         // byte offset along K of this lane's chunk
  const unsigned b_kend = (unsigned)(K * 2);
  int a_left = nk, b_left = nk;                                      // K-steps not yet issued (wave-uniform)

  auto issueA = [&](int stage, int c) {
    const unsigned off = (a_left > 0 && a_cur[c] != kOOB) ? a_cur[c] + (unsigned)(cc * 16) : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t)(smem + stage * STAGE + wid * 1024 + c * 8192), 16, off, 0, 0, 0);
  };
  auto advanceA = [&]() {                                            // step this lane's chunk to the next K-step (branch-free per lane)
    --a_left;
    cc += 8;
    const bool wrap = cc >= cpt;
    tap_moved = __any(wrap) != 0;
    if (tap_moved) {
      cc -= wrap ? cpt : 0;
      tx += wrap ? 1 : 0;
      const bool wx = tx >= p.taps_x;
      tx = wx ? 0 : tx;
      ty += wx ? 1 : 0;
      const bool wy = ty >= p.taps_y;
      ty = wy ? 0 : ty;
      tz += wy ? 1 : 0;
      dzv = tz * p.step_z; dyv = ty * p.step_y; dxv = tx * p.step_x;
      tap_off = ((dzv * p.x_H + dyv) * p.x_W + dxv) * ldb;
    }
  };
  auto rows_if_moved = [&](int c0, int c1) { if (tap_moved) { row_update(c0); row_update(c1); } };
  auto issueB = [&](int stage, int c) {
    const unsigned off = (b_left > 0 && b_base[c] != kOOB && b_k < b_kend) ? b_base[c] + b_k : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t)(smem + stage * STAGE + BOFF + wid * 1024 + c * 8192), 16, off, 0, 0, 0);
  };
  auto advanceB = [&]() { --b_left; b_k += 128; };

  // ---- fragment read offsets: row (lane & 15) of a 16-row fragment, chunk (4 ks + (lane >> 4)) ^ ((lane >> 1) & 7) ----
  const int rd0 = (lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4);      // ks = 1: rd0 ^ 64
  const int xrow0 = BN == 256 ? (wid >> 2) * 128 : (wid >> 1) * 64;                   // first pixel row of this wave
  const int wrow0 = BN == 256 ? (wid & 3) * 64 : (wid & 1) * 64;                      // first weight row of this wave
  const int x_rd = xrow0 * 128 + rd0, w_rd = BOFF + wrow0 * 128 + rd0;

  f32x4 acc[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[4], wf[4];

  auto read_x = [&](int stage, int i0, int ks) {          // 4 pixel fragments i0 .. i0+3 of K half ks
    const unsigned char* b = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(b + ((x_rd + (i0 + i) * 2048) ^ (ks * 64)));
  };
  auto read_w = [&](int stage, int ks) {
    const unsigned char* b = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(b + ((w_rd + j * 2048) ^ (ks * 64)));
  };
  auto mfma16 = [&](int i0) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i0 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  if constexpr (BN == 256) {
    // pieces of a K-step: PB0 = weight blocks {0,1}, PB1 = {2,3}, PA0 = pixel blocks {0,2} (first 64 rows of each group's
    // half), PA1 = {1,3}.  Issue order, one piece per phase: phase 0: PB1(t+1), 1: PA0(t+1), 2: PA1(t+1), 3: PB0(t+2).
    issueB(0, 0); issueB(0, 1); issueB(0, 2); issueB(0, 3); advanceB();
    issueA(0, 0); issueA(0, 2); issueA(0, 1); issueA(0, 3);
    advanceA(); rows_if_moved(0, 2); rows_if_moved(1, 3);
    issueB(1, 0); issueB(1, 1);
    wait_vm<4>();                                          // PA1(0) and PB0(1) may still be in flight
    seg_barrier();
    if (grp) seg_barrier();                                // G1 runs one barrier behind G0
    for (int t = 0; t < nk; ++t) {
      const int s = t & 1;
      // phase 0: K half 0, pixel fragments 0-3
      read_w(s, 0); read_x(s, 0, 0);
      issueB(s ^ 1, 2); issueB(s ^ 1, 3); advanceB();
      if (t) rows_if_moved(0, 2);                          // (the prologue has set K-step 1's rows)
      wait_lgkm0(); wait_vm<4>(); seg_barrier();
      mfma16(0);
      seg_barrier();
      // phase 1: K half 0, pixel fragments 4-7
      read_x(s, 4, 0);
      issueA(s ^ 1, 0); issueA(s ^ 1, 2);
      if (t) rows_if_moved(1, 3);
      wait_lgkm0(); wait_vm<4>(); seg_barrier();
      mfma16(4);
      seg_barrier();
      // phase 2: K half 1, pixel fragments 0-3
      read_w(s, 1); read_x(s, 0, 1);
      issueA(s ^ 1, 1); issueA(s ^ 1, 3);
      wait_lgkm0(); wait_vm<4>(); seg_barrier();
      mfma16(0);
      seg_barrier();
      // phase 3: K half 1, pixel fragments 4-7; the weight blocks {0,1} of this stage were last read in phase 2
      read_x(s, 4, 1);
      issueB(s, 0); issueB(s, 1);
      advanceA();                                          // tap state of K-step t+2; its rows follow in phases 0 and 1
      wait_lgkm0(); wait_vm<4>(); seg_barrier();
      mfma16(4);
      seg_barrier();
    }
    if (!grp) seg_barrier();
  } else {
    // three whole K-steps in LDS: K-step t+2 is issued while t is computed, t+1 has landed before t ends
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      issueA(s, 0); issueA(s, 1); issueA(s, 2); issueA(s, 3);
      advanceA(); rows_if_moved(0, 1); rows_if_moved(2, 3);
      issueB(s, 0); issueB(s, 1); advanceB();
    }
    wait_vm<6>();
    seg_barrier();
    if (grp) seg_barrier();
    int s = 0, s2 = 2;
    for (int t = 0; t < nk; ++t) {
      // phase 0: K half 0
      read_w(s, 0); read_x(s, 0, 0);
      issueA(s2, 0); issueA(s2, 1); issueB(s2, 0);
      if (t) rows_if_moved(2, 3);
      wait_lgkm0(); seg_barrier();
      mfma16(0);
      seg_barrier();
      // phase 1: K half 1
      read_w(s, 1); read_x(s, 0, 1);
      issueA(s2, 2); issueA(s2, 3); issueB(s2, 1); advanceB();
      advanceA(); rows_if_moved(0, 1);
      wait_lgkm0(); wait_vm<6>(); seg_barrier();
      mfma16(0);
      seg_barrier();
      s = s == 2 ? 0 : s + 1;
      s2 = s2 == 2 ? 0 : s2 + 1;
    }
    if (!grp) seg_barrier();
  }

  // ---- epilogue: accumulators -> LDS fp32 [rows][BN] (16-B chunk q of row r at q ^ (r & 15)) -> 16-byte row stores --------
  wait_vm<0>();                                            // the (out-of-range, zero-filling) DMAs of the last phases too
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
  long* rowoff = reinterpret_cast<long*>(smem + EPI_ROWS * BN * 4);
  const bool slab_out = p.split_k > 1;
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
  const int lq = lane >> 4, lr = lane & 15;
  __bf16* __restrict__ yp = reinterpret_cast<__bf16*>(p.y);
  constexpr int NPASS = BM / EPI_ROWS;
  for (int hp = 0; hp < NPASS; ++hp) {
    if (hp) __syncthreads();
    if (NPASS == 1 || grp == hp) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nl = wrow0 + j * 16 + 4 * lq, n = n0 + nl;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (!slab_out && p.bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = n + e < p.N ? p.bias[n + e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int ml = (NPASS == 1 ? xrow0 : 0) + i * 16 + lr;       // row inside the staged block
          f32x4 v = acc[i][j];
          if (!slab_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float t = v[e] + bv[e]; v[e] = t > 0.f ? t : t * p.slope; }
          }
          *reinterpret_cast<f32x4*>(ct + ml * BN + (((nl >> 2) ^ (ml & 15)) << 2)) = v;
        }
      }
    }
    __syncthreads();
    const int mbase = hp * EPI_ROWS;
    if (slab_out) {
      constexpr int CPR = BN / 4;
      const bool vec = (p.N % 4) == 0;
      for (int c = tid; c < EPI_ROWS * CPR; c += 512) {
        const int ml = c / CPR, q4 = c - ml * CPR, n = n0 + q4 * 4;
        const long off = rowoff[mbase + ml];
        if (off < 0 || n >= p.N) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ct + ml * BN + ((q4 ^ (ml & 15)) << 2));
        float* d = slab_base + off + n;
        if (vec) stg_u4(d, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
        else { for (int q = 0; q < 4 && n + q < p.N; ++q) d[q] = v[q]; }
      }
      continue;
    }
    constexpr int CPR = BN / 8;                                        // 16-byte bf16 stores per row
    const bool vec = yp && (p.y_ld % 8) == 0 && (reinterpret_cast<uintptr_t>(yp) % 16) == 0;
    for (int c = tid; c < EPI_ROWS * CPR; c += 512) {
      const int ml = c / CPR, q8 = c - ml * CPR, n = n0 + q8 * 8;
      const long pix = rowoff[mbase + ml];
      if (pix < 0 || n >= p.N) continue;
      float v[8];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(ct + ml * BN + (((2 * q8 + h2) ^ (ml & 15)) << 2));
        v[4 * h2] = t4[0]; v[4 * h2 + 1] = t4[1]; v[4 * h2 + 2] = t4[2]; v[4 * h2 + 3] = t4[3];
      }
      if (yp) {
        __bf16* d = yp + pix * p.y_ld + n;
        if (vec && n + 8 <= p.N) {
          if (p.accumulate & 1) {
            const Chunk old = ldg_chunk(d);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[2 * q] += __uint_as_float(old.w[q] << 16); v[2 * q + 1] += __uint_as_float(old.w[q] & 0xffff0000u); }
          }
          uint32_t o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            const bf2 t2 = {(__bf16)v[2 * q], (__bf16)v[2 * q + 1]};
            o[q] = __builtin_bit_cast(uint32_t, t2);
          }
          stg_u4(d, make_uint4(o[0], o[1], o[2], o[3]));
        } else {
          for (int q = 0; q < 8 && n + q < p.N; ++q) {
            float vv = v[q];
            if (p.accumulate & 1) vv += (float)d[q];
            d[q] = (__bf16)vv;
            v[q] = vv;
          }
        }
      }
      if (p.y32) for (int q = 0; q < 8 && n + q < p.N; ++q) p.y32[pix * p.y32_ld + n + q] = v[q];
    }
  }
}

}  // namespace

// 1 when the wide kernel can run desc (bf16, 2-D or single-class 3-D, at least 64 channels per tap so that a lane's chunk steps at most one tap
// per K-step); tiles_out = workgroups per class at the tile width it would use (tile_n 128 / 256, 0 = by N).
extern "C" int mireg_conv_wide_eligible(const mireg_conv_desc* p, long* tiles_out) {
  if (tiles_out) *tiles_out = 0;
  if (!p || p->dtype != MIREG_DTYPE_BF16) return 0;
  const bool depth = p->x_D > 1 || p->g_D > 1 || p->taps_z > 1 || p->y_D > 1;
  if ((depth && p->n_cls > 1) || p->n_cls > 4) return 0;             // Conv3d parity classes in one launch: ring kernel only
  if (p->x_C < 64 || p->N < 16) return 0;
  const int ncls = p->n_cls > 1 ? p->n_cls : 1;
  long M = 0;
  for (int c = 0; c < ncls; ++c) {
    const long m = ncls > 1 ? (long)p->n_img * p->cls[c].g_H * p->cls[c].g_W : (long)p->n_img * p->g_H * p->g_W * (p->g_D > 0 ? p->g_D : 1);
    M = m > M ? m : M;
    const long xc = p->x_C;
    const long k = (ncls > 1 ? (long)p->cls[c].taps_y * p->cls[c].taps_x : (long)p->taps_y * p->taps_x * (p->taps_z > 0 ? p->taps_z : 1)) * xc;
    if (k < 64) return 0;
  }
  const int bn = p->tile_n == 256 ? 256 : (p->tile_n == 128 ? 128 : (p->N > 128 ? 256 : 128));
  if (tiles_out) *tiles_out = ((M + BM - 1) / BM) * ((p->N + bn - 1) / bn);
  return 1;
}

// -100: not applicable (the caller falls back); otherwise MIREG_OK / MIREG_ERR_LAUNCH.  split_k > 1 writes slabs only: the
// caller runs the split-K reduce pass of conv_gemm.hip.
extern "C" int mireg_conv_wide_try(const mireg_conv_desc* p, hipStream_t stream) {
  long tiles = 0;
  if (!mireg_conv_wide_eligible(p, &tiles)) return -100;
  const int ncls = p->n_cls > 1 ? p->n_cls : 1;
  const int z = p->split_k > 1 ? p->split_k : 1;
  const int bn = p->tile_n == 256 ? 256 : (p->tile_n == 128 ? 128 : (p->N > 128 ? 256 : 128));
  dim3 grid((unsigned)tiles, ncls, z);
  if (bn == 256) hipLaunchKernelGGL((conv_wide_kernel<256>), grid, dim3(512), 0, stream, *p);
  else hipLaunchKernelGGL((conv_wide_kernel<128>), grid, dim3(512), 0, stream, *p);
  return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
}
