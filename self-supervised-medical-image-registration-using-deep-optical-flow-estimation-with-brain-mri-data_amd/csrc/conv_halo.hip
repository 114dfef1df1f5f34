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
//   * A halo: 3 buffers of (R+ty-1) rows x (W+tx-1) pixels x 64 B, filled by buffer_load ... lds two chunks ahead, one
//     instruction per wave and K-step;
//   * B (weights of one tap x chunk): BN rows x 64 B in a 4-5 stage ring, one stage per K-step;
//   * taps unrolled at compile time (3x3, 3x2, 2x3, 2x2), one raw s_barrier per K-step, s_waitcnt vmcnt(immediate);
//   * LDS rows are 64 B, 16-B chunks XOR-swizzled with (row >> 2) & 3 on the DMA source side and on the read side;
//     the M rows of a 32-row MFMA block are assigned to pixels so that each hardware ds_read_b128 lane group
//     ({0-3,12-15,20-27} / {4-11,16-19,28-31}) reads 16 CONSECUTIVE halo rows: conflict-free at any tap shift.
#include "mireg_common.h"
#include "../../include/mireg.h"
#include <stdlib.h>
#include <type_traits>

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
#define MIREG_VM_CASE(k) case k: wait_vmcnt<k>(); break;
  switch (n) {
    MIREG_VM_CASE(0) MIREG_VM_CASE(1) MIREG_VM_CASE(2) MIREG_VM_CASE(3) MIREG_VM_CASE(4) MIREG_VM_CASE(5) MIREG_VM_CASE(6)
    MIREG_VM_CASE(7) MIREG_VM_CASE(8) MIREG_VM_CASE(9) MIREG_VM_CASE(10) MIREG_VM_CASE(11) MIREG_VM_CASE(12) MIREG_VM_CASE(13)
    MIREG_VM_CASE(14) MIREG_VM_CASE(15) MIREG_VM_CASE(16) MIREG_VM_CASE(17) MIREG_VM_CASE(18) MIREG_VM_CASE(19) MIREG_VM_CASE(20)
    MIREG_VM_CASE(21) MIREG_VM_CASE(22) MIREG_VM_CASE(23) MIREG_VM_CASE(24)
    default: wait_vmcnt<25>(); break;
  }
#undef MIREG_VM_CASE
}

// halo capacity (16-row DMA instructions per buffer): BM=128 -> 17 (4 x 66 = 264 rows), BM=256 -> 25 (6 x 66 = 396 rows)
template <int BM> struct HaloCap { static constexpr int INSTR = BM == 128 ? 17 : 25; };

// LDS plan of one launch (host and kernel agree through these): three halo buffers (two chunks ahead), the weight ring,
// one 1-KiB dump slot for the padding DMAs; the epilogue reuses the space
struct HaloPlan { int na, ka, d, bytes; bool ok; };
template <int BM, int BN> __host__ __device__ inline HaloPlan halo_plan(int R, int W, int ty, int tx) {
  HaloPlan h;
  const int taps = ty * tx;
  h.na = ((R + ty - 1) * (W + tx - 1) + 15) >> 4;                  // DMA instructions (1 KiB) per halo buffer
  h.ka = (h.na + 3) >> 2;                                           // K-steps over which a halo is issued (one instruction per wave and step)
  h.d = taps >= 6 ? 4 : 3;                                          // weight stages in flight (ring of d + 1)
  const int epi = 128 * BN * 4 + 128 * 8;
  const int ring = 3 * h.na * 1024 + (h.d + 1) * BN * 64 + 1024;
  h.bytes = ring > epi ? ring : epi;
  // tap shapes with an unrolled loop; the halo of chunk c+2 (issued from step (c, 0) on) must be older than every weight stage
  // waited for from step (c+2, 0) on, and must not run into the next chunk's issue window
  const bool shape = (ty == 3 && tx == 3) || (ty == 3 && tx == 2) || (ty == 2 && tx == 3) || (ty == 2 && tx == 2);
  h.ok = shape && h.ka <= taps && 2 * taps - h.ka >= h.d && h.bytes <= 160 * 1024;
  return h;
}

extern __shared__ __attribute__((aligned(1024))) unsigned char halo_smem[];

// one output tile, start to finish (the body of both kernels below)
template <typename T, int BM, int BN, int TY, int TX>
__device__ __forceinline__ void halo_tile(const mireg_conv_desc& pd) {
  constexpr int CPC = Cfg<T>::CPC, BK = Cfg<T>::BK;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
  constexpr int S = BM / 16;                                       // 16-pixel segments per tile
  constexpr int B_GROUPS = BN / 16;                                // weight DMA instructions per K-step
  constexpr int B_PW = B_GROUPS / 4;                               // per wave
  constexpr int B_STAGE_BYTES = BN * 64;
  static_assert(TM >= 1 && TN >= 1, "bad tile");
  unsigned char* const smem = halo_smem;

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
  constexpr int ty_n = TY, tx_n = TX;                              // every class of this launch has this tap shape (host-grouped)
  const int HR = R + ty_n - 1, HP = W + tx_n - 1;
  const int iy_min = y0 + p.off_y + (p.step_y < 0 ? -(ty_n - 1) : 0);
  const int ix_min = p.off_x + (p.step_x < 0 ? -(tx_n - 1) : 0);
  const HaloPlan plan = halo_plan<BM, BN>(R, W, ty_n, tx_n);       // the launch's dynamic LDS was sized with the same call
  const int NA = plan.na, KA = plan.ka;
  const int A_BUF_BYTES = NA * 1024, B_BASE = 3 * A_BUF_BYTES;
  const int cpt = p.x_C / CPC;                                     // 16-byte chunks per pixel
  const int nchunks = (cpt + 3) >> 2;                              // 64-byte channel chunks

  constexpr unsigned kOOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // DMA schedule.  Two measured facts shape it (profiles/README.md, round 2): (1) a CU has ONE scalar unit shared by its
  // eight resident waves, so every scalar instruction of the K loop costs all of them a cycle -- a loop with run-time tap
  // bookkeeping and a computed vmcnt ran 4.6x the scalar instructions of the ring kernel and 1.4x its time at 0.6x its bytes;
  // (2) the LDS-DMA path only overlaps with the MFMAs when every wave issues the same few instructions every step.
  // So: taps are unrolled at compile time, every K-step every wave issues exactly BN/64 weight DMAs (stage s+D) and ONE
  // halo-slot DMA -- instruction 4k+wave of the halo two chunks ahead during the first KA steps of a chunk, otherwise a
  // padding DMA (all lanes out of range: no memory access, zeros into a dump slot) -- which makes every vmcnt an immediate.
  // lane L of a DMA instruction covers LDS row L>>2, physical chunk L&3 = logical chunk ^ ((row>>2)&3)
  const int lrow = lane >> 2;
  const int kc = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned pixB = (unsigned)(p.x_ld * (long)sizeof(T));
  const unsigned img_off = (unsigned)((long)img * p.x_H * p.x_W * p.x_ld * (long)sizeof(T)) + (unsigned)(kc * 16);
  const float inv_hp = 1.0f / (float)HP;
  unsigned char* const dump = smem + plan.bytes - 1024;            // padding DMAs land here (never read)
  // Out-of-range handling is arithmetic (an offset with bit 31 set is beyond every buffer: the DMA writes zeros): per-lane
  // booleans would live in SGPR pairs and turn into exec-mask branches around the DMA instruction.
  const int lim = (cpt - kc + 3) >> 2;                             // first 64-byte chunk in which this lane's 16 bytes lie beyond x_C
  // halo-slot DMA of this wave for step k of the issue window of `chunk` (buffer `buf`); valid = uniform "real piece"
  auto issueA = [&](int chunk, int buf, int k, bool valid) {
    const int q = 4 * k + wid;
    const bool real = valid && q < NA;                              // wave-uniform
    const int l = q * 16 + lrow;
    const int hy = (int)(((float)l + 0.5f) * inv_hp), hx = l - hy * HP;        // exact: l < 2^10, HP <= 68
    const int iy = iy_min + hy, ix = ix_min + hx;
    const unsigned bad = (unsigned)(iy | (p.x_H - 1 - iy) | ix | (p.x_W - 1 - ix) | (HR - 1 - hy) | (lim - 1 - chunk)) & kOOB;
    const unsigned off = (img_off + (unsigned)(iy * p.x_W + ix) * pixB + (unsigned)(chunk * 64)) | bad | (real ? 0u : kOOB);
    unsigned char* dst = real ? smem + buf * A_BUF_BYTES + q * 1024 : dump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t)dst, 16, off, 0, 0, 0);
  };
  // weight rows: groups g = wid + 4c (rows beyond N carry bit 31 from the start)
  unsigned b_src[B_PW];
#pragma unroll
  for (int c = 0; c < B_PW; ++c) {
    const int n = n0 + (wid + 4 * c) * 16 + lrow;
    b_src[c] = (unsigned)((long)min(n, p.N - 1) * p.w_ld * (long)sizeof(T)) + (unsigned)(kc * 16);
    b_src[c] |= (unsigned)(p.N - 1 - n) & kOOB;
  }
  const int xc_bytes = p.x_C * (int)sizeof(T);
  // weights of K-step (chunk, tap) into ring stage `stage`: k = tap * x_C + chunk * BK
  auto issueB = [&](int chunk, int tap_bytes, int stage) {           // tap_bytes = tap * x_C * sizeof(T); chunks past the end: zeros
    unsigned char* Bs = smem + B_BASE + stage * B_STAGE_BYTES;
    const unsigned kb = (unsigned)(tap_bytes + chunk * 64);
    const unsigned bad = (unsigned)(lim - 1 - chunk) & kOOB;
#pragma unroll
    for (int c = 0; c < B_PW; ++c) {          // braces matter: hipcc 7.2 drops the kernel's host stub for a braceless builtin-call body here
      const unsigned off = (b_src[c] + kb) | bad;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t)(Bs + (wid + 4 * c) * 1024), 16, off, 0, 0, 0);
    }
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
    const unsigned char* Bs = smem + B_BASE + bstage * B_STAGE_BYTES;
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

  // ---- pipeline: everything per step is a compile-time constant or a running scalar cursor ---------------------------
  {
    constexpr int TAPS = TY * TX;
    constexpr int D = TAPS >= 6 ? 4 : 3, BST = D + 1;               // == HaloPlan::d
    // DMAs of a wave younger than the weight stage it waits for: the halo slot of step s-D, then [weights, slot] of steps
    // s-D+1 .. s-1 -- the same at every step, because weight stages past the end are still issued (out of range: zeros)
    constexpr int YOUNG = 1 + (D - 1) * (1 + B_PW);
    // prologue: halos of chunks 0 and 1 (older than any wait), then stages 0..D-1 in the steady pattern [weights, slot]
    for (int k = 0; k < KA; ++k) issueA(0, 0, k, true);
    for (int k = 0; k < KA; ++k) issueA(1, 1, k, nchunks > 1);
    int i_kb = 0, i_tap = 0, istage = 0;                             // issue cursor of the weight ring: byte offset along K, tap, stage
    int i_chunk = 0;
    auto issue_next_B = [&]() {
      issueB(i_chunk, i_kb, istage);
      i_kb += xc_bytes;
      if (++i_tap == TAPS) { i_tap = 0; i_kb = 0; ++i_chunk; }
      if (++istage == BST) istage = 0;
    };
#pragma unroll
    for (int s = 0; s < D; ++s) { issue_next_B(); issueA(0, 0, 0, false); }
    int bstage = 0, abuf = 0, ibuf = 2;                             // read cursors; ibuf = buffer of the halo two chunks ahead
    const int sx_sign = p.step_x < 0 ? -1 : 1;
    const int shift0 = (p.step_y < 0 ? TY - 1 : 0) * HP + (p.step_x < 0 ? TX - 1 : 0);
    const int d_ky = (p.step_y < 0 ? -HP : HP) - sx_sign * (TX - 1);  // row shift from the last tap of a row to the first of the next
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 2 < nchunks;                            // a halo two chunks ahead exists
      int shift = shift0;
      // one K-step; t is a literal (a generic lambda here would keep hipcc from emitting the kernel's host handle)
#define MIREG_HALO_STEP(t)                                                             \
      if constexpr ((t) < TAPS) {                                                      \
        wait_vmcnt<YOUNG>();                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  /* own fragment reads done before the ring is re-filled */ \
        __builtin_amdgcn_s_barrier();                                                  \
        asm volatile("" ::: "memory");                                                 \
        issue_next_B();                                                                \
        issueA(c + 2, ibuf, (t), more && (t) < KA);                                    \
        compute(abuf, bstage, shift);                                                  \
        shift += ((t) % TX == TX - 1) ? d_ky : sx_sign;                                \
        if (++bstage == BST) bstage = 0;                                               \
      }
      MIREG_HALO_STEP(0) MIREG_HALO_STEP(1) MIREG_HALO_STEP(2) MIREG_HALO_STEP(3) MIREG_HALO_STEP(4)
      MIREG_HALO_STEP(5) MIREG_HALO_STEP(6) MIREG_HALO_STEP(7) MIREG_HALO_STEP(8)
#undef MIREG_HALO_STEP
      if (++abuf == 3) abuf = 0;
      if (++ibuf == 3) ibuf = 0;
    }
    wait_vmcnt<0>();                                                 // padding / past-the-end DMAs still target the ring
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

// every class of the launch has the tap shape TY x TX
template <typename T, int BM, int BN, int TY, int TX>
__global__ void __launch_bounds__(256, BM == 128 ? 2 : 1)          // 256-pixel tiles: 128 accumulators + pipelined fragments per lane
conv_halo_kernel(const mireg_conv_desc pd) {
  halo_tile<T, BM, BN, TY, TX>(pd);
}

// geometry of one class: does the halo path apply, and with how many tiles at BM rows per tile?
bool class_ok(int gH, int gW, int ty, int tx, int bm, int bn, long* tiles_m, int n_img) {
  if (gW != 16 && gW != 32 && gW != 64) return false;
  const int S = bm / 16, spr = gW / 16;
  if (S % spr) return false;
  const int R = S / spr;
  if (R < 1 || gH % R) return false;
  const HaloPlan h = bm == 128 ? (bn == 128 ? halo_plan<128, 128>(R, gW, ty, tx) : halo_plan<128, 64>(R, gW, ty, tx))
                               : (bn == 128 ? halo_plan<256, 128>(R, gW, ty, tx) : halo_plan<256, 64>(R, gW, ty, tx));
  if (!h.ok) return false;
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
  if (p->n_cls > 4) return 0;
  for (int c = 0; c < (p->n_cls > 1 ? p->n_cls : 0); ++c) if (p->cls[c].g_D > 0) return 0;
  if (p->N < 32) return 0;                                  // a 64-column tile half full still beats the ring kernel's im2col traffic (PWC's 32-channel layers)
  if (p->dtype != MIREG_DTYPE_BF16 && p->dtype != MIREG_DTYPE_F32) return 0;
  const int ncls = p->n_cls > 1 ? p->n_cls : 1;
  // classes of different tap shapes (backward-data of 5x5 / stride 2) stay on the ring kernel: a one-launch dispatch over
  // separately compiled tile functions was measured at 64-90 us stand-alone but 235 us next to the backward-weights stream
  // (scratch for the calls, 248 VGPRs), and one launch per shape leaves 192-tile grids (profiles/README.md round 2)
  for (int c = 1; c < ncls; ++c)
    if (p->cls[c].taps_y != p->cls[0].taps_y || p->cls[c].taps_x != p->cls[0].taps_x) return 0;
  const int bn = (p->tile_n == 64 || p->N <= 64) ? 64 : 128;
  for (int b = 0; b < 2; ++b) {
    const int bm = b ? 256 : 128;
    long worst = 0;
    bool ok = true;
    for (int c = 0; c < ncls && ok; ++c) {
      long tm = 0;
      ok = ncls > 1 ? class_ok(p->cls[c].g_H, p->cls[c].g_W, p->cls[c].taps_y, p->cls[c].taps_x, bm, bn, &tm, p->n_img)
                    : class_ok(p->g_H, p->g_W, p->taps_y, p->taps_x, bm, bn, &tm, p->n_img);
      worst = tm > worst ? tm : worst;
    }
    if (ok) tiles_out[b] = worst;
  }
  return tiles_out[0] > 0 || tiles_out[1] > 0;
}

template <typename T, int BM, int BN, int TY, int TX>
static int launch_halo_shape(const mireg_conv_desc& p, long tiles_m, int bytes, hipStream_t stream) {
  static bool attr_set = false;                                      // per instantiation; > 64 KiB of dynamic LDS needs the opt-in
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<T, BM, BN, TY, TX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return MIREG_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid((unsigned)(tiles_m * ((p.N + BN - 1) / BN)), p.n_cls > 1 ? p.n_cls : 1, 1);
  hipLaunchKernelGGL((conv_halo_kernel<T, BM, BN, TY, TX>), grid, dim3(256), bytes, stream, p);
  return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
}

template <typename T, int BM, int BN>
static int launch_halo_k(const mireg_conv_desc& p, long tiles_m, hipStream_t stream) {
  const int ncls = p.n_cls > 1 ? p.n_cls : 1;
  int bytes = 0;                                                     // dynamic LDS: the largest plan over the classes
  bool uniform = true;
  const int ty0 = ncls > 1 ? p.cls[0].taps_y : p.taps_y, tx0 = ncls > 1 ? p.cls[0].taps_x : p.taps_x;
  for (int c = 0; c < ncls; ++c) {
    const int gW = ncls > 1 ? p.cls[c].g_W : p.g_W, ty = ncls > 1 ? p.cls[c].taps_y : p.taps_y, tx = ncls > 1 ? p.cls[c].taps_x : p.taps_x;
    const HaloPlan h = halo_plan<BM, BN>((BM / 16) / (gW / 16), gW, ty, tx);
    bytes = h.bytes > bytes ? h.bytes : bytes;
    uniform = uniform && ty == ty0 && tx == tx0;
  }
  if (uniform) {
    if (ty0 == 3 && tx0 == 3) return launch_halo_shape<T, BM, BN, 3, 3>(p, tiles_m, bytes, stream);
    if (ty0 == 3 && tx0 == 2) return launch_halo_shape<T, BM, BN, 3, 2>(p, tiles_m, bytes, stream);
    if (ty0 == 2 && tx0 == 3) return launch_halo_shape<T, BM, BN, 2, 3>(p, tiles_m, bytes, stream);
    if (ty0 == 2 && tx0 == 2) return launch_halo_shape<T, BM, BN, 2, 2>(p, tiles_m, bytes, stream);
    return MIREG_ERR_UNSUPPORTED;
  }
  return MIREG_ERR_UNSUPPORTED;                                      // mixed tap shapes: not eligible (mireg_conv_halo_eligible)
}

template <typename T>
static int launch_halo_t(const mireg_conv_desc& p, int bm, long tiles_m, hipStream_t stream) {
  const int bn = (p.tile_n == 64 || p.N <= 64) ? 64 : 128;
  if (bm == 128 && bn == 128) return launch_halo_k<T, 128, 128>(p, tiles_m, stream);
  if (bm == 128) return launch_halo_k<T, 128, 64>(p, tiles_m, stream);
  if (bn == 128) return launch_halo_k<T, 256, 128>(p, tiles_m, stream);
  return launch_halo_k<T, 256, 64>(p, tiles_m, stream);
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
