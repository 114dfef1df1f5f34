// conv_wgrad_halo.hip -- halo-staged backward-weights GEMM (K3, second generation), bf16 operands / fp32 accumulate.
//
//   dW[co][(ky,kx,ci)] = sum_{img,y,x} dy[img,y,x][co] * x[img, y*s + ky - p, x*s + kx - p][ci]
// (autograd of Conv2d / ConvTranspose2d: FlowNetS/util.py:17-30,49-55; PWC/models/PWCNet.py:24-34).
//
// The ring kernel (conv_gemm.hip, conv_wgrad_dma_kernel) gathers the x operand once per TAP: a 128 x 128 output tile moves
// 16 KB of L2->LDS traffic per MFLOP, which the CU's LDS-DMA path (~27 B/clk) turns into <= 43 % of the matrix peak, and
// every operand byte is re-read by 18-36 output tiles.  Here a workgroup owns 128 output channels x (all taps of a tap
// group) x 32 input channels: per 32-pixel K-step it stages the dy tile (32 pixels x 128 channels, the MFMA A operand) and,
// once per 128-pixel chunk, the x halo of that chunk (the B operand); every tap's B fragment is the same halo read at a
// shifted LDS row (ds_read_b64_tr_b16 takes per-lane addresses).  3x3: 4.4 KB per MFLOP, 18 MFMAs per wave and barrier.
//
// A stride-s convolution is decomposed into its s*s tap-parity classes: the taps ky = py + s*a of class (py, px) see the
// input sub-image x[ry::s, rx::s] at unit stride, so every class is a unit-stride problem with (ceil-ish k/s)^2 taps
// (5x5/s2: 3x3, 3x2, 2x3, 2x2; 4x4/s2: four 2x2).  One launch per class group of equal tap shape.
//
// Loop discipline (measured on conv_halo.hip, profiles/README.md round 2): taps and steps unrolled at compile time, every
// wave issues exactly 2 dy DMAs + 1 halo-slot DMA per K-step (padding slots are all-out-of-range DMAs into a dump slot),
// s_waitcnt vmcnt(immediate), one raw s_barrier per K-step, out-of-range handling by offset arithmetic (bit 31).
#include "mireg_common.h"
#include "../../include/mireg.h"
#include <stdlib.h>

using namespace mireg;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

typedef __attribute__((address_space(3))) void* lds_void_t;
typedef s16x4 __attribute__((address_space(3)))* lds_tr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one unit-stride class of one layer (built on the host from mireg_conv_desc)
struct WgArgs {
  const void* x; const void* dy; float* slab;
  long x_bytes, dy_bytes;
  int W, H, R;                 // dy grid per image; rows per 128-pixel chunk (R * W == 128 * SPC / 4)
  int total_chunks, cpz, nz;   // chunks over all images; chunks per pixel split; splits (slabs)
  int Cout, Cip;               // dy channels (GEMM M), x channels padded to 8 (GEMM N per tap)
  long y_ldb, x_pixb;          // bytes per dy pixel / per x pixel
  int x_H, x_W;                // the full input image
  int s, ry, rx, oy, ox;       // sub-image: pixel (u, v) = input (u*s + ry, v*s + rx); tap (a, b) reads sub-pixel (y+a+oy, x+b+ox)
  int ky0, kx0, kstep, KW;     // tap (a, b) is weight tap (ky0 + a*kstep, kx0 + b*kstep) of a KW-wide kernel
  long slab_ld, slab_z;        // floats per Cout row of a slab; floats per slab
  int nci;                     // 32-channel chunks of x (ceil(Cip / 32))
};

constexpr int kStage = 8192;                                        // dy ring stage: 32 pixels x 128 channels bf16
constexpr int kD = 3;                                               // dy stages in flight; ring of kD + 1

struct WgPlan { int na, ka, bytes; bool ok; };
__host__ __device__ inline WgPlan wg_plan(int R, int W, int ty, int tx, int spc) {
  WgPlan p;
  p.na = ((R + ty - 1) * (W + tx - 1) + 15) >> 4;
  p.ka = (p.na + 3) >> 2;
  p.bytes = 3 * p.na * 1024 + (kD + 1) * kStage + 1024;
  p.ok = p.ka <= spc && 2 * spc - p.ka >= kD && p.bytes <= 160 * 1024;
  return p;
}

extern __shared__ __attribute__((aligned(1024))) unsigned char wg_smem[];

// all classes of one layer (blockIdx.y picks one): they run side by side so that a stride-2 layer fills the chip with few splits
struct WgLayer { WgArgs c[4]; };

template <int TY, int TX, int SPC>
__device__ __forceinline__ void wgrad_halo_tile(const WgArgs& a) {
  constexpr int TC = TY * TX;
  constexpr unsigned kOOB = 0x80000000u;
  unsigned char* const smem = wg_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, h = lane >> 5;

  // ---- job: (pixel split z, output tile) -- blocks of one z are contiguous so that one XCD reads a pixel range once ----
  const int ntiles = ((a.Cout + 127) >> 7) * a.nci;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
  }
  const int z = bid / ntiles, tile = bid - z * ntiles;
  const int tco = tile / a.nci, tci = tile - tco * a.nci;
  const int co0 = tco * 128, ci0 = tci * 32;
  const int g_begin = z * a.cpz, g_end = min(a.total_chunks, g_begin + a.cpz);
  const int nch = max(0, g_end - g_begin);
  const int nsteps = nch * SPC;

  const int W = a.W, R = a.R;
  const int HR = R + TY - 1, HP = W + TX - 1;
  const WgPlan plan = wg_plan(R, W, TY, TX, SPC);
  const int NA = plan.na, KA = plan.ka;
  const int A_BUF = NA * 1024, RING = 3 * A_BUF;
  unsigned char* const dump = smem + plan.bytes - 1024;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);

  // ---- dy tile DMA: instruction i covers pixel rows 4i .. 4i+3 of the 32-pixel step (256 B each); wave w issues i = w, w+4.
  // 16-B chunks are XOR-swizzled with (pixel & 3) << 2 on the source side (ds_read_b64_tr_b16 reads 4 rows per lane group)
  unsigned dy_lane[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int px = (wid + 4 * c) * 4 + (lane >> 4);
    const int lch = (lane & 15) ^ ((px & 3) << 2);
    const int co = co0 + lch * 8;
    dy_lane[c] = (unsigned)((long)(g_begin * 32 * SPC + px) * a.y_ldb) + (unsigned)(co * 2);
    dy_lane[c] |= (unsigned)(((a.Cout + 7) & ~7) - 1 - co) & kOOB;           // channels beyond the (padded) row: zeros
  }
  const unsigned dy_step = (unsigned)(32 * a.y_ldb);
  unsigned dy_off = 0;                                               // running byte offset of the next step to issue
  int i_step = 0, i_stage = 0;
  auto issue_dy = [&]() {
    unsigned char* St = smem + RING + i_stage * kStage;
    const unsigned bad = (unsigned)(nsteps - 1 - i_step) & kOOB;     // steps past the end: zeros (never read)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const unsigned off = (dy_lane[c] + dy_off) | bad;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void_t)(St + (wid + 4 * c) * 1024), 16, off, 0, 0, 0);
    }
    dy_off += dy_step;
    ++i_step;
    if (++i_stage == kD + 1) i_stage = 0;
  };

  // ---- x halo DMA: one slot per wave and step; instruction q = 4k + wave covers halo rows 16q .. 16q+15 (64 B each) ----
  const int lrow = lane >> 2, kc = lane & 3;
  const float inv_hp = 1.0f / (float)HP;
  const unsigned ci_lane = (unsigned)((ci0 + kc * 8) * 2) | ((unsigned)(a.Cip - 1 - (ci0 + kc * 8)) & kOOB);
  const int cpi = a.H / R;                                           // chunks per image
  int h_img = g_begin / cpi, h_y0 = (g_begin - h_img * cpi) * R;     // chunk whose halo is issued next
  const unsigned imgB = (unsigned)((long)a.x_H * a.x_W * a.x_pixb);
  auto issue_halo = [&](int buf, int k, bool valid) {
    const int q = 4 * k + wid;
    const bool real = valid && q < NA;
    const int l = q * 16 + lrow;
    const int hy = (int)(((float)l + 0.5f) * inv_hp), hx = l - hy * HP;
    const int iy = (h_y0 + hy + a.oy) * a.s + a.ry, ix = (hx + a.ox) * a.s + a.rx;
    const unsigned bad = (unsigned)(iy | (a.x_H - 1 - iy) | ix | (a.x_W - 1 - ix) | (HR - 1 - hy)) & kOOB;
    const unsigned off = ((unsigned)h_img * imgB + (unsigned)(iy * a.x_W + ix) * (unsigned)a.x_pixb + ci_lane) | bad | (real ? 0u : kOOB);
    unsigned char* dst = real ? smem + buf * A_BUF + q * 1024 : dump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t)dst, 16, off, 0, 0, 0);
  };
  auto next_halo_chunk = [&]() { h_y0 += R; if (h_y0 >= a.H) { h_y0 = 0; ++h_img; } };

  // ---- fragment addressing ----------------------------------------------------------------------------------------------
  // A (dy^T, 32 channels x 16 pixels): lane group g covers pixel rows 8(g>>1) + q4 (+4) and 16-B chunk 4w + 2(g&1) + (pp>>1)
  int a_lane[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int row = ks * 16 + 8 * (g >> 1) + q4;
    const int ch = 4 * wid + 2 * (g & 1) + (pp >> 1);
    a_lane[ks] = row * 256 + ((ch ^ (q4 << 2)) << 4) + (pp & 1) * 8;
  }
  // B (x halo, 16 pixels x 32 channels, rows of 64 B, no swizzle: 4 consecutive rows x 64 B fill all banks exactly once)
  const int b_lane = (8 * (g >> 1) + q4) * 64 + (2 * (g & 1) + (pp >> 1)) * 16 + (pp & 1) * 8;
  // first halo row of the 16-pixel slab (step j, half ks) at tap (0, 0): pixel 32j + 16ks of the chunk -> (row, column)
  const int wsh = W == 16 ? 4 : (W == 32 ? 5 : 6);
  int slab_row[SPC][2];
#pragma unroll
  for (int j = 0; j < SPC; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int idx = 32 * j + 16 * ks;
      slab_row[j][ks] = ((idx >> wsh) * HP + (idx & (W - 1))) * 64;
    }
  const int hp64 = HP * 64;

  f32x16 acc[TC];
#pragma unroll
  for (int t = 0; t < TC; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  auto tr_pair = [&](const unsigned char* base) -> bf16x8 {          // rows r .. r+3 and r+4 .. r+7 of a 16-row slab, transposed
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + 1024));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto tr_pair_b = [&](const unsigned char* base) -> bf16x8 {        // the same on 64-byte rows
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + 256));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  // ---- pipeline -----------------------------------------------------------------------------------------------------------
  constexpr int YOUNG = 1 + (kD - 1) * 3;                            // halo slot of step s-D, then [2 dy, slot] of steps s-D+1 .. s-1
  for (int k = 0; k < KA; ++k) issue_halo(0, k, nch > 0);
  next_halo_chunk();
  for (int k = 0; k < KA; ++k) issue_halo(1, k, nch > 1);
  next_halo_chunk();
#pragma unroll
  for (int s = 0; s < kD; ++s) { issue_dy(); issue_halo(0, 0, false); }
  int stage = 0, abuf = 0, ibuf = 2;
  for (int c = 0; c < nch; ++c) {
    const bool more = c + 2 < nch;
    const unsigned char* Hb = smem + abuf * A_BUF + b_lane;
#define MIREG_WG_STEP(j)                                                                           \
    if constexpr ((j) < SPC) {                                                                     \
      wait_vmcnt<YOUNG>();                                                                         \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  /* own fragment reads done before the ring is re-filled */ \
      __builtin_amdgcn_s_barrier();                                                                \
      asm volatile("" ::: "memory");                                                               \
      issue_dy();                                                                                  \
      issue_halo(ibuf, (j), more && (j) < KA);                                                     \
      const unsigned char* St = smem + RING + stage * kStage;                                      \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
        const bf16x8 af = tr_pair(St + a_lane[ks]);                                                \
        _Pragma("unroll") for (int ty = 0; ty < TY; ++ty) {                                        \
          const unsigned char* rowp = Hb + slab_row[(j)][ks] + ty * hp64;                          \
          _Pragma("unroll") for (int tx = 0; tx < TX; ++tx) {                                      \
            const bf16x8 bf = tr_pair_b(rowp + tx * 64);                                           \
            acc[ty * TX + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[ty * TX + tx], 0, 0, 0); \
          }                                                                                        \
        }                                                                                          \
      }                                                                                            \
      if (++stage == kD + 1) stage = 0;                                                            \
    }
    MIREG_WG_STEP(0) MIREG_WG_STEP(1) MIREG_WG_STEP(2) MIREG_WG_STEP(3)
    MIREG_WG_STEP(4) MIREG_WG_STEP(5) MIREG_WG_STEP(6) MIREG_WG_STEP(7)
#undef MIREG_WG_STEP
    if (more) next_halo_chunk();
    if (++abuf == 3) abuf = 0;
    if (++ibuf == 3) ibuf = 0;
  }
  wait_vmcnt<0>();

  // ---- epilogue: accumulator rows are 32 consecutive floats of one slab row (128 B per half wave) -----------------------
  float* __restrict__ slab = a.slab + (long)z * a.slab_z;
  const int ci = ci0 + (lane & 31);
  const bool ci_ok = ci < a.Cip;
#pragma unroll
  for (int ty = 0; ty < TY; ++ty)
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) {
      const long col = (long)((a.ky0 + ty * a.kstep) * a.KW + a.kx0 + tx * a.kstep) * a.Cip + ci;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + 32 * wid + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (ci_ok && co < a.Cout) slab[(long)co * a.slab_ld + col] = acc[ty * TX + tx][e];
      }
    }
}

// every class of the layer has the tap shape TY x TX (3x3 stride 1; 4x4 stride 2: four 2x2 classes)
template <int TY, int TX, int SPC>
__global__ void __launch_bounds__(256, 2)
conv_wgrad_halo_kernel(const WgLayer L) {
  wgrad_halo_tile<TY, TX, SPC>(L.c[blockIdx.y]);
}

// 5x5 stride 2: classes 3x3, 3x2, 2x3, 2x2 in one launch, each shape a separately compiled tile function (conv_halo.hip has the
// measurement behind this: inlining all unrolled loops into one kernel spills 60-160 SGPRs)
typedef const __attribute__((address_space(4))) WgLayer* kernarg_layer_t;
template <int TY, int TX, int SPC>
__device__ __noinline__ void wgrad_halo_tile_call(kernarg_layer_t kp, int cls) {
  const WgLayer& L = *(const WgLayer*)kp;                            // read in place: scalar loads from the kernarg segment
  wgrad_halo_tile<TY, TX, SPC>(L.c[cls]);
}
template <int SPC>
__global__ void __launch_bounds__(256, 2)
conv_wgrad_halo_5x5s2_kernel(const WgLayer L) {
  kernarg_layer_t kp = (kernarg_layer_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int cls = blockIdx.y;                                        // class order (py, px) = (0,0) (0,1) (1,0) (1,1)
  if (cls == 0) wgrad_halo_tile_call<3, 3, SPC>(kp, 0);
  else if (cls == 1) wgrad_halo_tile_call<3, 2, SPC>(kp, 1);
  else if (cls == 2) wgrad_halo_tile_call<2, 3, SPC>(kp, 2);
  else wgrad_halo_tile_call<2, 2, SPC>(kp, 3);
}

template <typename K>
int launch_kernel(K kernel, const WgLayer& L, int ncls, int bytes, bool* attr_set, hipStream_t stream) {
  if (!*attr_set) {                                                  // > 64 KiB of dynamic LDS needs the opt-in, once per kernel
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return MIREG_ERR_LAUNCH;
    *attr_set = true;
  }
  const WgArgs& a = L.c[0];
  const int ntiles = ((a.Cout + 127) / 128) * a.nci;
  hipLaunchKernelGGL(kernel, dim3((unsigned)(ntiles * a.nz), ncls), dim3(256), bytes, stream, L);
  return hipGetLastError() == hipSuccess ? MIREG_OK : MIREG_ERR_LAUNCH;
}

template <int TY, int TX, int SPC>
int launch_uniform(const WgLayer& L, int ncls, int bytes, hipStream_t stream) {
  static bool attr_set = false;
  return launch_kernel(&conv_wgrad_halo_kernel<TY, TX, SPC>, L, ncls, bytes, &attr_set, stream);
}
template <int SPC>
int launch_5x5s2(const WgLayer& L, int bytes, hipStream_t stream) {
  static bool attr_set = false;
  return launch_kernel(&conv_wgrad_halo_5x5s2_kernel<SPC>, L, 4, bytes, &attr_set, stream);
}

// chunk geometry for a W-wide dy grid: 128-pixel chunks (4 steps) for W = 16 / 32, 256-pixel chunks (8 steps) for W = 64
bool grid_plan(int H, int W, int* R, int* spc) {
  if (W == 16) { *R = 8; *spc = 4; }
  else if (W == 32) { *R = 4; *spc = 4; }
  else if (W == 64) { *R = 4; *spc = 8; }
  else return false;
  return H % *R == 0;
}

}  // namespace

// Class decomposition of a layer: fills up to 4 classes; returns the count (0 = not applicable)
static int wg_classes(const mireg_conv_desc* p, WgArgs* out, int* tys, int* txs, int* spc_out) {
  if (!p || p->dtype != MIREG_DTYPE_BF16 || p->taps_z > 1 || p->g_D > 1 || p->x_D > 1) return 0;
  if (p->step_y != 1 || p->step_x != 1 || p->mul_y != p->mul_x || (p->mul_y != 1 && p->mul_y != 2)) return 0;
  if (p->x_bytes <= 0 || p->w_bytes <= 0 || p->x_bytes >= (1L << 31) || p->w_bytes >= (1L << 31)) return 0;
  if (p->taps_y != p->taps_x || p->off_y != p->off_x) return 0;
  int R, spc;
  if (!grid_plan(p->g_H, p->g_W, &R, &spc)) return 0;
  const int s = p->mul_y, k = p->taps_y, pad = -p->off_y;
  int n = 0;
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px) {
      const int ty = (k - py + s - 1) / s, tx = (k - px + s - 1) / s;       // taps ky = py + s*a < k
      if (ty < 1 || tx < 1) continue;
      if (!((ty == 3 || ty == 2) && (tx == 3 || tx == 2))) return 0;
      if (s == 1 && !(ty == 3 && tx == 3)) return 0;
      if (!wg_plan(R, p->g_W, ty, tx, spc).ok) return 0;
      WgArgs a;
      a.x = p->x; a.dy = p->y; a.slab = p->slab; a.x_bytes = p->x_bytes; a.dy_bytes = p->w_bytes;
      a.W = p->g_W; a.H = p->g_H; a.R = R;
      a.total_chunks = p->n_img * (p->g_H / R);
      a.nz = p->split_k > 1 ? p->split_k : 1;
      a.cpz = (a.total_chunks + a.nz - 1) / a.nz;
      a.Cout = p->N; a.Cip = p->x_C;
      a.y_ldb = p->y_ld * 2; a.x_pixb = p->x_ld * 2;
      a.x_H = p->x_H; a.x_W = p->x_W;
      // input row of output y, tap a: y*s + (py + s*a) - pad = s*(y + a + oy) + ry
      const int dy0 = py - pad, dx0 = px - pad;
      a.s = s;
      a.ry = ((dy0 % s) + s) % s; a.rx = ((dx0 % s) + s) % s;
      a.oy = (dy0 - a.ry) / s; a.ox = (dx0 - a.rx) / s;
      a.ky0 = py; a.kx0 = px; a.kstep = s; a.KW = k;
      a.slab_ld = p->slab_ld > 0 ? p->slab_ld : (long)k * k * p->x_C;
      a.slab_z = (long)p->N * a.slab_ld;
      a.nci = (p->x_C + 31) / 32;
      out[n] = a; tys[n] = ty; txs[n] = tx; ++n;
    }
  *spc_out = spc;
  return n;
}

extern "C" int mireg_conv_wgrad_halo_eligible(const mireg_conv_desc* p) {
  WgArgs cls[4]; int ty[4], tx[4], spc;
  return wg_classes(p, cls, ty, tx, &spc) > 0 ? 1 : 0;
}

// Called by mireg_conv_wgrad (conv_gemm.hip): >= 0 launched (MIREG_OK / error), -100 not applicable.
extern "C" int mireg_conv_wgrad_halo_try(const mireg_conv_desc* p, hipStream_t stream) {
  WgLayer L; int ty[4], tx[4], spc;
  const int n = wg_classes(p, L.c, ty, tx, &spc);
  if (n <= 0) return -100;
  int bytes = 0, max_taps = 0;
  for (int c = 0; c < n; ++c) {
    const WgPlan pl = wg_plan(L.c[c].R, L.c[c].W, ty[c], tx[c], spc);
    bytes = pl.bytes > bytes ? pl.bytes : bytes;
    max_taps = ty[c] * tx[c] > max_taps ? ty[c] * tx[c] : max_taps;
  }
  // every class gets the grid of split_k pixel splits, but a class with fewer taps spreads its pixels over proportionally
  // fewer of them (equal work per workgroup); its remaining workgroups only write their zero slabs
  for (int c = 0; c < n; ++c) {
    int nzc = (L.c[c].nz * ty[c] * tx[c] + max_taps - 1) / max_taps;
    nzc = nzc < 1 ? 1 : nzc;
    L.c[c].cpz = (L.c[c].total_chunks + nzc - 1) / nzc;
  }
  for (int c = n; c < 4; ++c) L.c[c] = L.c[0];
  const bool uniform = n == 1 || (ty[0] == ty[1] && ty[0] == ty[2] && ty[0] == ty[3] && tx[0] == tx[1] && tx[0] == tx[2] && tx[0] == tx[3]);
  if (uniform) {
    const int code = ty[0] * 4 + tx[0];
    if (spc == 4) {
      if (code == 15) return launch_uniform<3, 3, 4>(L, n, bytes, stream);
      if (code == 10) return launch_uniform<2, 2, 4>(L, n, bytes, stream);
    } else {
      if (code == 15) return launch_uniform<3, 3, 8>(L, n, bytes, stream);
      if (code == 10) return launch_uniform<2, 2, 8>(L, n, bytes, stream);
    }
    return MIREG_ERR_UNSUPPORTED;
  }
  if (n == 4 && ty[0] == 3 && tx[0] == 3 && ty[1] == 3 && tx[1] == 2 && ty[2] == 2 && tx[2] == 3 && ty[3] == 2 && tx[3] == 2)
    return spc == 4 ? launch_5x5s2<4>(L, bytes, stream) : launch_5x5s2<8>(L, bytes, stream);
  return MIREG_ERR_UNSUPPORTED;
}
