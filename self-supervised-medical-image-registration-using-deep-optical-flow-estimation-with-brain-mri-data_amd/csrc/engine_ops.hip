// engine_ops.hip -- the HBM-bound glue around the MFMA contractions (all NHWC [rows][ld] views):
//   weight packing / gradient unpacking between torch layout and the GEMM layouts (LDS-tiled transposes),
//   train-mode BatchNorm2d + LeakyReLU(0.1) forward / backward (FlowNetS/util.py:17-30),
//   LeakyReLU backward, fp32 <-> compute-dtype casts with accumulate, NCHW -> NHWC input staging,
//   fused Adam exactly as train.py:129 configures it (eps = 1e-4).
// Every kernel streams 16-byte vectors through GLOBAL (address-space 1) accesses; reductions are two-stage
// (per-block partial rows, then a tiny finalize) so they are deterministic and free of atomic contention.
#include <cstdlib>
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

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

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<__bf16> { static constexpr int N = 8; };

// 16-byte vector load / store of VecOf<T>::N elements <-> float registers
__device__ __forceinline__ void ldv(const float* p, float (&v)[4]) {
  const float4 q = *GPTR(const float4, p);
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void stv(float* p, const float (&v)[4]) { *GPTR(float4, p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void ldv(const __bf16* p, float (&v)[8]) {
  const uint4 q = *GPTR(const uint4, p);
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  const bf2 t = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ void stv(__bf16* p, const float (&v)[8]) {
  *GPTR(uint4, p) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// ==============================================================================================
// weight layout plumbing.  Per conv weight W[co][ci][tap] (torch layout, tap = ky*kw+kx):
//   FWD pack   F[co][tap*Cip + ci]                       (wave unit = one co x 64 ci, wave-private LDS)
//   DGRAD pack D_c[ci][tt*Cop + co] = F[co][tap(tt)*Cip + ci]   (block = 64 co x 64 ci transpose of one tap)
//   unpack     grad[co][ci][tap] = sum_z slab[z][co][tap*Cip + ci]   (wave unit, reversed FWD pack)
// Work items are flattened over all jobs: job j owns units [unit0, unit0 + nunits).
// ==============================================================================================
constexpr int kPackMaxTaps = 25;       // wave-private LDS path; larger kernels (7x7 stems) use the scalar path

__device__ __forceinline__ int find_job(const mireg_pack_job* jobs, int njobs, int unit) {
  int lo = 0;
  for (int i = 1; i < njobs; ++i) if (GPTR(const mireg_pack_job, jobs + i)->unit0 <= unit) lo = i;
  return lo;
}

__device__ __forceinline__ int find_job_d(const mireg_pack_job* __restrict__ jobs, int njobs, int unit) {
  int lo = 0, hi = njobs;                                // last job with dunit0 <= unit (unit0 sequences are non-decreasing)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (jobs[mid].dunit0 <= unit) lo = mid; else hi = mid;
  }
  return lo;
}

// MODE 0: FWD pack (src = torch weight fp32, dst = F in T).  MODE 1: unpack (src = slab fp32, dst = torch grad fp32)
template <typename T, int MODE>
__global__ void __launch_bounds__(256) pack_fwd_kernel(const mireg_pack_job* __restrict__ jobs, int njobs, int total_units) {
  __shared__ float lds[4][64 * kPackMaxTaps];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* tile = lds[wid];
  for (int unit = blockIdx.x * 4 + wid; unit < total_units; unit += gridDim.x * 4) {
    const mireg_pack_job j = *GPTR(const mireg_pack_job, jobs + find_job(jobs, njobs, unit));
    const int taps = j.kh * j.kw, tp = taps | 1;
    const int C = MODE == 0 ? j.Cpad : j.Ci;            // ci extent walked by this mode
    const int chunks = (C + 63) / 64;
    const int u = unit - j.unit0;
    const int co = u / chunks, ci0 = (u - co * chunks) * 64;
    const int nci = max(0, min(64, j.Ci - ci0));        // real channels in this chunk
    const int run = nci * taps;
    const unsigned magic = (unsigned)((0x100000000ull + taps - 1) / taps);
    if (taps > kPackMaxTaps) {                           // 7x7 stems and every Conv3d (27..343 taps): same tile, tap chunks
      for (int t0 = 0; t0 < taps; t0 += kPackMaxTaps) {
        const int nt = min(kPackMaxTaps, taps - t0), ntp = nt | 1;
        const unsigned mg = (unsigned)((0x100000000ull + nt - 1) / nt);
        __builtin_amdgcn_wave_barrier();
        if (MODE == 0) {
          const float* src = j.src + ((long)co * j.Ci + ci0) * taps + t0;
          for (int r = lane; r < nci * nt; r += 64) {     // runs of nt contiguous floats per input channel
            const int ci = (int)__umulhi((unsigned)r, mg);
            tile[ci * ntp + (r - ci * nt)] = ldf(src + (long)ci * taps + (r - ci * nt));
          }
          __builtin_amdgcn_wave_barrier();
          T* dst = reinterpret_cast<T*>(j.dst) + (long)co * j.ld + (long)t0 * j.Cpad + ci0 + lane;
          if (ci0 + lane < j.Cpad) {
#pragma unroll 4
            for (int t = 0; t < nt; ++t) stf(dst + (long)t * j.Cpad, lane < nci ? tile[lane * ntp + t] : 0.f);
          }
        } else {
          if (lane < nci) {
            const float* sp = j.src + (long)co * j.ld + (long)t0 * j.Cpad + ci0 + lane;
            const long slab_sz = (long)j.Co * j.ld;
#pragma unroll 4
            for (int t = 0; t < nt; ++t) {
              float v = 0.f;
#pragma unroll 4
              for (int z = 0; z < j.nsplit; ++z) v += ldf(sp + (long)t * j.Cpad + z * slab_sz);
              tile[lane * ntp + t] = v;
            }
          }
          __builtin_amdgcn_wave_barrier();
          float* d = reinterpret_cast<float*>(j.dst) + ((long)co * j.Ci + ci0) * taps + t0;
          for (int r = lane; r < nci * nt; r += 64) {
            const int ci = (int)__umulhi((unsigned)r, mg), t = r - ci * nt;
            const float v = tile[ci * ntp + t];
            float* q = d + (long)ci * taps + t;
            stf(q, j.accumulate ? ldf(q) + v : v);
          }
        }
      }
      continue;
    }
    if (MODE == 0) {
      const float* src = j.src + ((long)co * j.Ci + ci0) * taps;
#pragma unroll 4
      for (int r = lane; r < run; r += 64) {
        const int ci = (int)__umulhi((unsigned)r, magic);
        tile[ci * tp + (r - ci * taps)] = ldf(src + r);
      }
      T* dst = reinterpret_cast<T*>(j.dst) + (long)co * j.ld + ci0 + lane;
      if (ci0 + lane < j.Cpad) {
#pragma unroll 4
        for (int tap = 0; tap < taps; ++tap) stf(dst + (long)tap * j.Cpad, lane < nci ? tile[lane * tp + tap] : 0.f);
      }
    } else {
      if (lane < nci) {
        const float* sp = j.src + (long)co * j.ld + ci0 + lane;
        const long slab_sz = (long)j.Co * j.ld;
#pragma unroll 4
        for (int tap = 0; tap < taps; ++tap) {
          float v = 0.f;
#pragma unroll 4
          for (int z = 0; z < j.nsplit; ++z) v += ldf(sp + (long)tap * j.Cpad + z * slab_sz);
          tile[lane * tp + tap] = v;
        }
      }
      float* d = reinterpret_cast<float*>(j.dst) + ((long)co * j.Ci + ci0) * taps;
#pragma unroll 4
      for (int r = lane; r < run; r += 64) {
        const int ci = (int)__umulhi((unsigned)r, magic);
        const float v = tile[ci * tp + (r - ci * taps)];
        stf(d + r, j.accumulate ? ldf(d + r) + v : v);
      }
    }
  }
}

// Conv3d backward-data packs, all output-voxel parity classes of a layer in one pass over its weights:
// tap t_a of axis a reaches the voxels of parity p_a with (p_a + pad_a) % s_a == t_a % s_a, as that class's tap j_a = t_a / s_a;
// D_class[ci][(jz, jy, jx)][Cop] = W[co][ci][tz][ty][tx].  Unit = (ci, 64 output channels): lanes walk co, so every store is a
// contiguous run; each lane reads its own taps-long run of the torch-layout weight (served from L2 after the first tap).
template <typename T>
__global__ void __launch_bounds__(256) pack_dgrad3d_kernel(const mireg_pack3d_job* __restrict__ jobs, int njobs, int total_units) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int unit = blockIdx.x * 4 + wid; unit < total_units; unit += gridDim.x * 4) {
    int ji = 0;
    for (int i = 1; i < njobs; ++i) if (jobs[i].unit0 <= unit) ji = i;
    const mireg_pack3d_job* __restrict__ jp = jobs + ji;        // fields are read through the table pointer (no private copy)
    const int Co = jp->Co, Ci = jp->Ci, Cop = jp->Cop, kd = jp->kd, kh = jp->kh, kw = jp->kw;
    const int sz = jp->sz, sy = jp->sy, sx = jp->sx, pz = jp->pz, py = jp->py, px = jp->px;
    const int chunks = (Cop + 63) / 64, u = unit - jp->unit0;
    const int ci = u / chunks, co = (u - ci * chunks) * 64 + lane;
    if (co >= Cop) continue;
    const int taps = kd * kh * kw;
    const float* src = jp->src + ((long)co * Ci + ci) * taps;
    const bool real = co < Co;
    for (int tz = 0; tz < kd; ++tz) {
      const int rz = tz % sz, cz = ((rz - pz) % sz + sz) % sz, ntz = (kd - rz + sz - 1) / sz;
      for (int ty = 0; ty < kh; ++ty) {
        const int ry = ty % sy, cy = ((ry - py) % sy + sy) % sy, nty = (kh - ry + sy - 1) / sy;
        // the kw (<= 8) loads of a tap row are issued together, then stored (one load in flight per wave is latency-bound)
        float v[8];
#pragma unroll
        for (int tx = 0; tx < 8; ++tx) v[tx] = (real && tx < kw) ? ldf(src + (tz * kh + ty) * kw + tx) : 0.f;
#pragma unroll
        for (int tx = 0; tx < 8; ++tx) {
          if (tx >= kw) break;
          const int rx = tx % sx, cx = ((rx - px) % sx + sx) % sx, ntx = (kw - rx + sx - 1) / sx;
          const int cls = (cz * sy + cy) * sx + cx;
          const long jidx = ((long)(tz / sz) * nty + ty / sy) * ntx + tx / sx;
          T* dst = reinterpret_cast<T*>(jp->dst[cls]) + ((long)ci * (ntz * nty * ntx) + jidx) * Cop + co;
          stf(dst, v[tx]);
        }
      }
    }
  }
}

// The same Conv3d backward-data packs taken from the layer's (fresh) FWD pack F[co][tap*Cip + ci] instead of the fp32 master weights:
// a 64 co x 64 ci tile of one tap is a 16-byte-row transpose through LDS (both sides whole 128-byte runs), half the bytes read and
// none of the per-lane strided gathers of pack_dgrad3d_kernel (0.72 -> 0.2 ms for the 133 M parameters of FlowNetS-3D).
// job.src = the FWD pack (element type T); unit = (tap, co tile, ci tile).
template <typename T>
__global__ void __launch_bounds__(256) pack_dgrad3d_fwd_kernel(const mireg_pack3d_job* __restrict__ jobs, int njobs) {
  __shared__ __attribute__((aligned(16))) T tile[64][64 + VecOf<T>::N];
  int ji = 0;
  for (int i = 1; i < njobs; ++i) if (jobs[i].unit0 <= (int)blockIdx.x) ji = i;
  const mireg_pack3d_job* __restrict__ jp = jobs + ji;
  const int Co = jp->Co, Ci = jp->Ci, Cop = jp->Cop, kd = jp->kd, kh = jp->kh, kw = jp->kw;
  const int sz = jp->sz, sy = jp->sy, sx = jp->sx, pz = jp->pz, py = jp->py, px = jp->px;
  const int Cip = (Ci + 7) & ~7, taps = kd * kh * kw;
  const long ld = (long)taps * Cip;
  const int co_t = (Cop + 63) / 64, ci_t = (Ci + 63) / 64;
  const int u = blockIdx.x - jp->unit0;
  const int tap = u / (co_t * ci_t), rem = u - tap * (co_t * ci_t);
  const int co0 = (rem / ci_t) * 64, ci0 = (rem % ci_t) * 64;
  const int tz = tap / (kh * kw), ty = (tap / kw) % kh, tx = tap % kw;
  const int rz = tz % sz, cz = ((rz - pz) % sz + sz) % sz, ntz = (kd - rz + sz - 1) / sz;
  const int ry = ty % sy, cy = ((ry - py) % sy + sy) % sy, nty = (kh - ry + sy - 1) / sy;
  const int rx = tx % sx, cx = ((rx - px) % sx + sx) % sx, ntx = (kw - rx + sx - 1) / sx;
  const int cls = (cz * sy + cy) * sx + cx;
  const long jidx = ((long)(tz / sz) * nty + ty / sy) * ntx + tx / sx, ncls = (long)ntz * nty * ntx;
  constexpr int V = VecOf<T>::N, CH = 64 / V;
  const T* F = reinterpret_cast<const T*>(jp->src);
  for (int idx = threadIdx.x; idx < 64 * CH; idx += 256) {   // rows = co, granules along ci
    const int r = idx / CH, ch = idx - r * CH;
    const int co = co0 + r, ci = ci0 + ch * V;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (co < Co && ci < Cip) v = *GPTR(const uint4, F + (long)co * ld + (long)tap * Cip + ci);
    *reinterpret_cast<uint4*>(&tile[r][ch * V]) = v;
  }
  __syncthreads();
  T* D = reinterpret_cast<T*>(jp->dst[cls]);
  for (int idx = threadIdx.x; idx < 64 * CH; idx += 256) {   // rows = ci, granules along co (zero beyond Co: the tile rows were zero-filled)
    const int r = idx / CH, ch = idx - r * CH;
    const int ci = ci0 + r, co = co0 + ch * V;
    if (ci < Ci && co < Cop) {
      T v[V];
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = tile[ch * V + e][r];
      *GPTR(uint4, D + ((long)ci * ncls + jidx) * Cop + co) = *reinterpret_cast<const uint4*>(v);
    }
  }
}

// DGRAD packs from the FWD pack: D_c[ci][tt*Cop + co] = F[co][tap*Cip + ci]; block = (64 co x 64 ci) of one (class, tt)
template <typename T>
__global__ void __launch_bounds__(256) pack_dgrad_kernel(const mireg_pack_job* __restrict__ jobs, int njobs) {
  __shared__ __attribute__((aligned(16))) T tile[64][64 + VecOf<T>::N];
  // every field is read through the (block-uniform) table pointer: scalar loads, no private copy of the struct
  const mireg_pack_job* __restrict__ jp = jobs + find_job_d(jobs, njobs, blockIdx.x);
  struct { int Co, Ci, Cpad, Cop, kw, stride; long ld; const void* dst; } j = {jp->Co, jp->Ci, jp->Cpad, jp->Cop, jp->kw, jp->stride, jp->ld, jp->dst};
  const int co_t = (j.Cop + 63) / 64, ci_t = (j.Ci + 63) / 64, nclass = jp->nclass;
  int u = blockIdx.x - jp->dunit0, c = 0;
  for (; c < nclass - 1; ++c) {                         // which class / tap / tile
    const int n = jp->cls[c].nty * jp->cls[c].ntx * co_t * ci_t;
    if (u < n) break;
    u -= n;
  }
  const mireg_pack_class* __restrict__ kp = jp->cls + c;
  struct { void* dst; long ld; int ky0, kx0, ntx; } k = {kp->dst, kp->ld, kp->ky0, kp->kx0, kp->ntx};
  const int tt = u / (co_t * ci_t), rem = u - tt * (co_t * ci_t);
  const int co0 = (rem / ci_t) * 64, ci0 = (rem % ci_t) * 64;
  const int ty = tt / k.ntx, tx = tt - ty * k.ntx;
  const int tap = (k.ky0 + j.stride * ty) * j.kw + k.kx0 + j.stride * tx;
  constexpr int V = VecOf<T>::N, CH = 64 / V;              // 16-byte granules per 64-wide tile row
  const T* F = reinterpret_cast<const T*>(j.dst);
  for (int idx = threadIdx.x; idx < 64 * CH; idx += 256) {   // rows = co, granules along ci: 16-byte loads
    const int r = idx / CH, ch = idx - r * CH;
    const int co = co0 + r, ci = ci0 + ch * V;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (co < j.Co && ci < j.Cpad) v = *GPTR(const uint4, F + (long)co * j.ld + (long)tap * j.Cpad + ci);
    *reinterpret_cast<uint4*>(&tile[r][ch * V]) = v;
  }
  __syncthreads();
  T* D = reinterpret_cast<T*>(k.dst);
  for (int idx = threadIdx.x; idx < 64 * CH; idx += 256) {   // rows = ci, granules along co: gather V, 16-byte stores
    const int r = idx / CH, ch = idx - r * CH;               // 8 (4) lanes cover one 128-byte (64-byte) run of a D row
    const int ci = ci0 + r, co = co0 + ch * V;
    if (ci < j.Ci && co < j.Cop) {
      T v[V];
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = tile[ch * V + e][r];
      *GPTR(uint4, D + (long)ci * k.ld + (long)tt * j.Cop + co) = *reinterpret_cast<const uint4*>(v);
    }
  }
}

// ==============================================================================================
// BatchNorm (train mode = batch statistics over all rows); vector path: C % VEC == 0, 16-B aligned views
// ==============================================================================================
// stage 1: partial[blk][2*C]:  FWD {sum y, sum y^2}   BWD {sum dz, sum dz*xhat}, dz = da * lrelu'(y*scale+shift)
template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
bn_partial_kernel(const T* __restrict__ y, long ld_y, const T* __restrict__ da, long ld_da, const float* __restrict__ ss,
                  float* __restrict__ partial, long M, int C, float slope) {
  constexpr int V = VecOf<T>::N;
  __shared__ float red[256][2 * V + 1];
  const int cpr = C / V;                                // 16-B chunks per row (<= 256)
  const int rpp = 256 / cpr;                            // rows per pass
  const int tx = threadIdx.x % cpr, ty = threadIdx.x / cpr;
  float a0[V], a1[V], sc[V], sh[V], mu[V], rs[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    a0[v] = a1[v] = 0.f;
    if (BWD) { const int c = tx * V + v; sc[v] = ss[c]; sh[v] = ss[C + c]; mu[v] = ss[2 * C + c]; rs[v] = ss[3 * C + c]; }
  }
  if (ty < rpp) {
    for (long m = (long)blockIdx.x * rpp + ty; m < M; m += (long)gridDim.x * rpp) {
      float yv[V];
      ldv(y + m * ld_y + tx * V, yv);
      if (!BWD) {
#pragma unroll
        for (int v = 0; v < V; ++v) { a0[v] += yv[v]; a1[v] += yv[v] * yv[v]; }
      } else {
        float g[V];
        ldv(da + m * ld_da + tx * V, g);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const float z = yv[v] * sc[v] + sh[v];
          const float dz = g[v] * (z > 0.f ? 1.f : slope);
          a0[v] += dz; a1[v] += dz * ((yv[v] - mu[v]) * rs[v]);
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) { red[threadIdx.x][v] = a0[v]; red[threadIdx.x][V + v] = a1[v]; }
  __syncthreads();
  if (ty == 0) {
    for (int r = 1; r < rpp; ++r)
#pragma unroll
      for (int v = 0; v < V; ++v) { a0[v] += red[r * cpr + tx][v]; a1[v] += red[r * cpr + tx][V + v]; }
    float* dst = partial + (long)blockIdx.x * 2 * C;
#pragma unroll
    for (int v = 0; v < V; ++v) { dst[tx * V + v] = a0[v]; dst[C + tx * V + v] = a1[v]; }
  }
}

// scalar fallback (any C): one atomic-free pass per block over a strided row set
template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
bn_partial_scalar_kernel(const T* __restrict__ y, long ld_y, const T* __restrict__ da, long ld_da, const float* __restrict__ ss,
                         float* __restrict__ partial, long M, int C, float slope) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a0 = 0.f, a1 = 0.f;
    for (long m = blockIdx.x; m < M; m += gridDim.x) {
      const float yv = ldf(y + m * ld_y + c);
      if (!BWD) { a0 += yv; a1 += yv * yv; }
      else {
        const float z = yv * ss[c] + ss[C + c];
        const float dz = ldf(da + m * ld_da + c) * (z > 0.f ? 1.f : slope);
        a0 += dz; a1 += dz * ((yv - ss[2 * C + c]) * ss[3 * C + c]);
      }
    }
    partial[(long)blockIdx.x * 2 * C + c] = a0;
    partial[(long)blockIdx.x * 2 * C + C + c] = a1;
  }
}

// stage 2: block = 8 channels x 32 partial-row lanes (the partial rows are few but the loads are latency-bound, so
// they are spread over many lanes); sums in float64, fixed order
__device__ __forceinline__ void sum_partials(const float* __restrict__ partial, int nblk, int C, int c, int rl,
                                             double (*sh)[8][2], double& s0, double& s1) {
  s0 = 0; s1 = 0;
  if (c < C) {
#pragma unroll 4
    for (int b = rl; b < nblk; b += 32) { s0 += ldf(partial + (long)b * 2 * C + c); s1 += ldf(partial + (long)b * 2 * C + C + c); }
  }
  sh[rl][threadIdx.x & 7][0] = s0; sh[rl][threadIdx.x & 7][1] = s1;
  __syncthreads();
  if (rl == 0)
    for (int r = 1; r < 32; ++r) { s0 += sh[r][threadIdx.x & 7][0]; s1 += sh[r][threadIdx.x & 7][1]; }
}

// forward: -> ss = [scale | shift | mean | rstd], running stats (momentum, unbiased variance)
__global__ void __launch_bounds__(256)
bn_finalize_kernel(const float* __restrict__ partial, int nblk, long M, int C, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
                   int training, float* __restrict__ ss) {
  __shared__ double sh[32][8][2];
  const int c = blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
  double s0 = 0, s1 = 0;
  if (training) sum_partials(partial, nblk, C, c, rl, sh, s0, s1);
  if (rl != 0 || c >= C) return;
  float mean, var;
  if (training) {
    const double mu = s0 / (double)M;
    double vr = s1 / (double)M - mu * mu;
    if (vr < 0) vr = 0;
    mean = (float)mu; var = (float)vr;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      const float unbiased = M > 1 ? (float)(vr * (double)M / (double)(M - 1)) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  } else { mean = running_mean[c]; var = running_var[c]; }
  const float rstd = 1.f / sqrtf(var + eps);
  const float sc = gamma[c] * rstd;
  ss[c] = sc; ss[C + c] = beta[c] - mean * sc; ss[2 * C + c] = mean; ss[3 * C + c] = rstd;
}

// backward: red[c] = sum dz / M, red[C+c] = sum dz*xhat / M; dgamma, dbeta
__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, long M, int C, float* __restrict__ red,
                       float* __restrict__ dgamma, float* __restrict__ dbeta, int acc_param_grads) {
  __shared__ double sh[32][8][2];
  const int c = blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
  double s0, s1;
  sum_partials(partial, nblk, C, c, rl, sh, s0, s1);
  if (rl != 0 || c >= C) return;
  red[c] = (float)(s0 / (double)M); red[C + c] = (float)(s1 / (double)M);
  if (dgamma) {
    dgamma[c] = (acc_param_grads ? dgamma[c] : 0.f) + (float)s1;
    dbeta[c] = (acc_param_grads ? dbeta[c] : 0.f) + (float)s0;
  }
}

// Single-launch BatchNorm for the deep layers (few rows, many channels), where three launches are pure launch
// latency: a block owns 2 channel granules and ALL rows; pass 1 = column sums (fixed order: per-lane serial,
// wave xor-tree, 4 waves in order), then the owner lanes finalize, pass 2 re-reads the rows (L2-resident).
//   FWD: out = lrelu(bn(y)), ss / running stats as bn_finalize_kernel        BWD: dy, red, dgamma, dbeta
template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
bn_fused_kernel(const T* __restrict__ y, long ld_y, const T* __restrict__ da, long ld_da, T* __restrict__ out, long ld_o, long M, int C,
                const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean, float* running_var,
                float momentum, float eps, float slope, float* __restrict__ ss, float* __restrict__ red, float* __restrict__ dgamma,
                float* __restrict__ dbeta, int acc_param_grads) {
  constexpr int V = VecOf<T>::N;
  __shared__ double wsum[4][2][2 * V];
  __shared__ float coef[2][4 * V];                      // per granule: FWD {scale, shift}, BWD {scale, shift, mean, rstd} + red in coef2
  __shared__ float coef2[2][2 * V];
  const int gsel = threadIdx.x & 1, rl = threadIdx.x >> 1, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gr = blockIdx.x * 2 + gsel;                 // channel granule of this thread
  const bool gok = gr * V < C;
  const int c0 = gr * V;
  float sc[V], sh[V], mu[V], rs[V];
  if (BWD && gok) {
#pragma unroll
    for (int v = 0; v < V; ++v) { sc[v] = ss[c0 + v]; sh[v] = ss[C + c0 + v]; mu[v] = ss[2 * C + c0 + v]; rs[v] = ss[3 * C + c0 + v]; }
  }
  float a0[V], a1[V];
#pragma unroll
  for (int v = 0; v < V; ++v) a0[v] = a1[v] = 0.f;
  if (gok) {
#pragma unroll 4
    for (long m = rl; m < M; m += 128) {
      float yv[V];
      ldv(y + m * ld_y + c0, yv);
      if (!BWD) {
#pragma unroll
        for (int v = 0; v < V; ++v) { a0[v] += yv[v]; a1[v] += yv[v] * yv[v]; }
      } else {
        float g[V];
        ldv(da + m * ld_da + c0, g);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const float z = yv[v] * sc[v] + sh[v];
          const float dz = g[v] * (z > 0.f ? 1.f : slope);
          a0[v] += dz; a1[v] += dz * ((yv[v] - mu[v]) * rs[v]);
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int off = 2; off < 64; off <<= 1) { a0[v] += __shfl_xor(a0[v], off, 64); a1[v] += __shfl_xor(a1[v], off, 64); }
  if (lane < 2) {
#pragma unroll
    for (int v = 0; v < V; ++v) { wsum[wid][lane][v] = (double)a0[v]; wsum[wid][lane][V + v] = (double)a1[v]; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * V) {                            // one lane per (granule, channel)
    const int gs = threadIdx.x / V, v = threadIdx.x % V, c = (blockIdx.x * 2 + gs) * V + v;
    if (c < C) {
      double s0 = 0, s1 = 0;
      for (int w = 0; w < 4; ++w) { s0 += wsum[w][gs][v]; s1 += wsum[w][gs][V + v]; }
      if (!BWD) {
        const double m_ = s0 / (double)M;
        double vr = s1 / (double)M - m_ * m_;
        if (vr < 0) vr = 0;
        const float mean = (float)m_, var = (float)vr;
        if (running_mean) {
          running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
          const float unbiased = M > 1 ? (float)(vr * (double)M / (double)(M - 1)) : var;
          running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
        const float rstd = 1.f / sqrtf(var + eps);
        const float s_ = gamma[c] * rstd, t_ = beta[c] - mean * s_;
        ss[c] = s_; ss[C + c] = t_; ss[2 * C + c] = mean; ss[3 * C + c] = rstd;
        coef[gs][v] = s_; coef[gs][V + v] = t_;
      } else {
        const float r0 = (float)(s0 / (double)M), r1 = (float)(s1 / (double)M);
        red[c] = r0; red[C + c] = r1;
        if (dgamma) {
          dgamma[c] = (acc_param_grads ? dgamma[c] : 0.f) + (float)s1;
          dbeta[c] = (acc_param_grads ? dbeta[c] : 0.f) + (float)s0;
        }
        coef2[gs][v] = r0; coef2[gs][V + v] = r1;
      }
    }
  }
  __syncthreads();
  if (!gok) return;
  float r0[V], r1[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    if (!BWD) { sc[v] = coef[gsel][v]; sh[v] = coef[gsel][V + v]; }
    else { r0[v] = coef2[gsel][v]; r1[v] = coef2[gsel][V + v]; }
  }
#pragma unroll 4
  for (long m = rl; m < M; m += 128) {
    float yv[V], o[V];
    ldv(y + m * ld_y + c0, yv);
    if (!BWD) {
#pragma unroll
      for (int v = 0; v < V; ++v) { const float z = yv[v] * sc[v] + sh[v]; o[v] = z > 0.f ? z : z * slope; }
    } else {
      float g[V];
      ldv(da + m * ld_da + c0, g);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float z = yv[v] * sc[v] + sh[v];
        const float dz = g[v] * (z > 0.f ? 1.f : slope);
        const float xh = (yv[v] - mu[v]) * rs[v];
        o[v] = sc[v] * (dz - r0[v] - xh * r1[v]);
      }
    }
    stv(out + m * ld_o + c0, o);
  }
}

// out = lrelu(y*scale + shift)            (MODE 0)
// dy  = scale*(dz - red0 - xhat*red1)     (MODE 1; dz as above)
template <typename T, int MODE, bool VECT>
__global__ void __launch_bounds__(256)
bn_elementwise_kernel(const T* __restrict__ y, long ld_y, const T* __restrict__ da, long ld_da, T* __restrict__ out, long ld_o,
                      const float* __restrict__ ss, const float* __restrict__ red, long M, int C, float slope) {
  constexpr int V = VECT ? VecOf<T>::N : 1;
  const int cpr = C / V;
  const long total = M * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long rc;
    const long m = fast_divmod(i, cpr, rc);
    const int c0 = (int)rc * V;
    float yv[V], g[V], o[V];
    if constexpr (VECT) { ldv(y + m * ld_y + c0, yv); if (MODE == 1) ldv(da + m * ld_da + c0, g); }
    else { yv[0] = ldf(y + m * ld_y + c0); if (MODE == 1) g[0] = ldf(da + m * ld_da + c0); }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c = c0 + v;
      const float z = yv[v] * ss[c] + ss[C + c];
      if (MODE == 0) o[v] = z > 0.f ? z : z * slope;
      else {
        const float dz = g[v] * (z > 0.f ? 1.f : slope);
        const float xh = (yv[v] - ss[2 * C + c]) * ss[3 * C + c];
        o[v] = ss[c] * (dz - red[c] - xh * red[C + c]);
      }
    }
    if constexpr (VECT) stv(out + m * ld_o + c0, o); else stf(out + m * ld_o + c0, o[0]);
  }
}

// in-place LeakyReLU backward through a stored activation: g *= (a > 0 ? 1 : slope)
template <typename T, bool VECT>
__global__ void __launch_bounds__(256)
lrelu_bwd_kernel(T* __restrict__ g, long ld_g, const T* __restrict__ a, long ld_a, long M, int C, float slope) {
  constexpr int V = VECT ? VecOf<T>::N : 1;
  const int cpr = C / V;
  const long total = M * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long rc;
    const long m = fast_divmod(i, cpr, rc);
    const int c0 = (int)rc * V;
    float gv[V], av[V];
    if constexpr (VECT) { ldv(g + m * ld_g + c0, gv); ldv(a + m * ld_a + c0, av); }
    else { gv[0] = ldf(g + m * ld_g + c0); av[0] = ldf(a + m * ld_a + c0); }
#pragma unroll
    for (int v = 0; v < V; ++v) gv[v] = av[v] > 0.f ? gv[v] : gv[v] * slope;
    if constexpr (VECT) stv(g + m * ld_g + c0, gv); else stf(g + m * ld_g + c0, gv[0]);
  }
}

// dst(T)[m][c] = beta*dst + alpha*src(f32)[m][c]     and the reverse direction
template <typename T>
__global__ void __launch_bounds__(256)
cast_from_f32_kernel(T* __restrict__ dst, long ld_d, const float* __restrict__ src, long ld_s, long M, int C, float alpha, float beta) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long rc;
    const long m = fast_divmod(i, C, rc);
    const int c = (int)rc;
    const float v = alpha * ldf(src + m * ld_s + c) + (beta != 0.f ? beta * ldf(dst + m * ld_d + c) : 0.f);
    stf(dst + m * ld_d + c, v);
  }
}
template <typename T>
__global__ void __launch_bounds__(256)
cast_to_f32_kernel(float* __restrict__ dst, long ld_d, const T* __restrict__ src, long ld_s, long M, int C, float alpha, float beta) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long rc;
    const long m = fast_divmod(i, C, rc);
    const int c = (int)rc;
    stf(dst + m * ld_d + c, alpha * ldf(src + m * ld_s + c) + (beta != 0.f ? beta * ldf(dst + m * ld_d + c) : 0.f));
  }
}

// NCHW fp32 (B, C, H, W) channels [c0, c0+nc) -> NHWC dst[(b,y,x)][ld] channels [0, nc)
template <typename T>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int Ctot, int c0, int nc, long HW, long ld) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, pix = i - b * HW;
    for (int c = 0; c < nc; ++c) stf(dst + i * ld + c, ldf(src + (b * Ctot + c0 + c) * HW + pix));
  }
}

// per-column sums over rows (bias gradients): out[c] (+)= sum_m g[m][c]
template <typename T>
__global__ void __launch_bounds__(256)
colsum_kernel(const T* __restrict__ g, long ld, long M, int C, float* __restrict__ out, float* __restrict__ partial, int accumulate) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float acc = 0.f;
  for (long m = threadIdx.x + (long)blockIdx.y * blockDim.x; m < M; m += (long)blockDim.x * gridDim.y) acc += ldf(g + m * ld + c);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  // no float atomics: row segments go to partial[segment][c] and are summed in order by colsum_finalize_kernel (one segment: direct)
  if (threadIdx.x == 0) {
    if (partial) partial[(long)blockIdx.y * C + c] = red[0];
    else out[c] = accumulate ? out[c] + red[0] : red[0];
  }
}

// bias gradient = column sums of dy, deterministic: block = 2 channel granules x one row segment (128 row lanes,
// xor-tree + 4 waves in order); one segment writes `out` directly, several go through partial[seg][C] + a tiny finalize
template <typename T>
__global__ void __launch_bounds__(256)
colsum_vec_kernel(const T* __restrict__ g, long ld, long M, int C, float* __restrict__ dst, int accumulate, long rows_per_seg) {
  constexpr int V = VecOf<T>::N;
  __shared__ float wsum[4][2][V];
  const int gsel = threadIdx.x & 1, rl = threadIdx.x >> 1, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = (blockIdx.x * 2 + gsel) * V;
  const long m0 = (long)blockIdx.y * rows_per_seg, m1 = m0 + rows_per_seg < M ? m0 + rows_per_seg : M;
  float a[V];
#pragma unroll
  for (int v = 0; v < V; ++v) a[v] = 0.f;
  if (c0 < C) {
#pragma unroll 4
    for (long m = m0 + rl; m < m1; m += 128) {
      float yv[V];
      ldv(g + m * ld + c0, yv);
#pragma unroll
      for (int v = 0; v < V; ++v) a[v] += yv[v];
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int off = 2; off < 64; off <<= 1) a[v] += __shfl_xor(a[v], off, 64);
  if (lane < 2) {
#pragma unroll
    for (int v = 0; v < V; ++v) wsum[wid][lane][v] = a[v];
  }
  __syncthreads();
  if (threadIdx.x < 2 * V) {
    const int gs = threadIdx.x / V, v = threadIdx.x % V, c = (blockIdx.x * 2 + gs) * V + v;
    if (c < C) {
      const float s = ((wsum[0][gs][v] + wsum[1][gs][v]) + wsum[2][gs][v]) + wsum[3][gs][v];
      float* d = dst + (long)blockIdx.y * C + c;                       // gridDim.y == 1: dst = out, else partial[seg][C]
      *d = (gridDim.y == 1 && accumulate) ? *d + s : s;
    }
  }
}

__global__ void __launch_bounds__(256)
colsum_finalize_kernel(const float* __restrict__ partial, int nseg, int C, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < nseg; ++k) s += ldf(partial + (long)k * C + c);      // independent loads, summed in order
  out[c] = accumulate ? out[c] + s : s;
}

// one launch that clears a table of buffers (gradient accumulators of a backward pass): block = 16 KiB of one buffer
__global__ void __launch_bounds__(256) zero_many_kernel(const mireg_zero_job* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (jobs[mid].unit0 <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const mireg_zero_job j = jobs[lo];
  const long base = (long)(blockIdx.x - j.unit0) * 16384;
  unsigned char* p = reinterpret_cast<unsigned char*>(j.p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long o = base + ((long)i * 256 + threadIdx.x) * 16;
    if (o + 16 <= j.bytes) *GPTR(uint4, p + o) = make_uint4(0u, 0u, 0u, 0u);
    else for (long b = o; b < j.bytes && b < o + 16; ++b) p[b] = 0;
  }
}

// fused multi-tensor Adam (torch.optim.Adam semantics, no weight decay / amsgrad); step lives on device so
// the launch is hipGraph-replayable.  float4 streams: 28 B of traffic per parameter.
__global__ void adam_tick_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }

__global__ void __launch_bounds__(256)
adam_kernel(const mireg_adam_job* __restrict__ jobs, const int* __restrict__ step, float lr, float b1, float b2, float eps,
            float grad_scale) {
  const mireg_adam_job j = *GPTR(const mireg_adam_job, jobs + blockIdx.y);
  const float t = (float)*step;
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  const float step_size = lr / bc1;
  const long n4 = (((uintptr_t)j.p | (uintptr_t)j.g | (uintptr_t)j.m | (uintptr_t)j.v) & 15) ? 0 : (j.n >> 2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float p[4], g[4], m[4], v[4];
    ldv(j.p + 4 * i, p); ldv(j.g + 4 * i, g); ldv(j.m + 4 * i, m); ldv(j.v + 4 * i, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = g[e] * grad_scale;
      m[e] = b1 * m[e] + (1.f - b1) * gg;
      v[e] = b2 * v[e] + (1.f - b2) * gg * gg;
      p[e] -= step_size * (m[e] / (sqrtf(v[e]) / bc2s + eps));
    }
    stv(j.p + 4 * i, p); stv(j.m + 4 * i, m); stv(j.v + 4 * i, v);
  }
  for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < j.n; i += (long)gridDim.x * blockDim.x) {
    const float g = ldf(j.g + i) * grad_scale;
    const float m = b1 * ldf(j.m + i) + (1.f - b1) * g;
    const float v = b2 * ldf(j.v + i) + (1.f - b2) * g * g;
    stf(j.m + i, m); stf(j.v + i, v);
    stf(j.p + i, ldf(j.p + i) - step_size * (m / (sqrtf(v) / bc2s + eps)));
  }
}


// ==============================================================================================
// packed-domain optimizer: split-K wgrad slabs -> packed gradient -> Adam on the torch-layout master
// weights -> refreshed FWD pack, without ever materialising the torch-layout gradient.
// ==============================================================================================
__device__ __forceinline__ int find_wopt(const mireg_wopt_job* __restrict__ jobs, int njobs, int unit, bool reduce) {
  int lo = 0, hi = njobs;                                // last job whose first unit is <= unit (block-uniform: scalar loads)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((reduce ? jobs[mid].runit0 : jobs[mid].unit0) <= unit) lo = mid; else hi = mid;
  }
  return lo;
}

// g[e] = sum_z slab[z][e]; block = 256 consecutive floats (64 float4 lanes) x 4 z-groups, fixed summation order
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const mireg_wopt_job* __restrict__ jobs, int njobs) {
  __shared__ float4 part[3][64];
  const mireg_wopt_job j = jobs[find_wopt(jobs, njobs, blockIdx.x, true)];
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const long E = (long)j.Co * j.ld, e = ((long)(blockIdx.x - j.runit0) * 64 + lane) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < E) {
    const float* sp = j.slab + e;
#pragma unroll 8
    for (int z = zg; z < j.nsplit; z += 4) {
      const float4 q = *GPTR(const float4, sp + (long)z * j.slab_stride);
      acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
    }
  }
  if (zg) part[zg - 1][lane] = acc;
  __syncthreads();
  if (zg == 0 && e < E) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float4 q = part[k][lane]; acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
    *GPTR(float4, j.g + e) = acc;
  }
}

constexpr int kOptMaxTaps = 125;     // 5 x 5 x 5 (Conv3d); the LDS tile is sized by the launch's largest tap count
// block = one (co, 64-ci chunk): packed gradient -> LDS (transposed) -> coalesced torch-order Adam -> FWD pack row
template <typename T>
__global__ void __launch_bounds__(256)
adam_pack_kernel(const mireg_wopt_job* __restrict__ jobs, int njobs, const int* __restrict__ step, float lr, float b1, float b2,
                 float eps, float grad_scale) {
  extern __shared__ float opt_tile[];                                  // 64 * (max_taps | 1) floats
  const mireg_wopt_job j = jobs[find_wopt(jobs, njobs, blockIdx.x, false)];
  const int taps = j.taps, tp = taps | 1, tid = threadIdx.x;
  const int chunks = (j.Cpad + 63) / 64;
  const int u = blockIdx.x - j.unit0;
  const int co = u / chunks, ci0 = (u - co * chunks) * 64;
  const int nci = max(0, min(64, j.Ci - ci0));
  const int run = nci * taps;
  const float t = (float)*step;
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  const float step_size = lr / bc1;
  const float* g = j.g + (long)co * j.ld + ci0;
  for (int e = tid; e < taps * 64; e += 256) {
    const int tap = e >> 6, ci = e & 63;
    if (ci < nci) opt_tile[ci * tp + tap] = ldf(g + (long)tap * j.Cpad + ci);
  }
  __syncthreads();
  const unsigned magic = (unsigned)((0x100000000ull + taps - 1) / taps);
  const long base = ((long)co * j.Ci + ci0) * taps;
  for (int r = tid; r < run; r += 256) {
    const int ci = (int)__umulhi((unsigned)r, magic), idx = ci * tp + (r - ci * taps);
    const float gg = opt_tile[idx] * grad_scale;
    const float m = b1 * ldf(j.m + base + r) + (1.f - b1) * gg;
    const float v = b2 * ldf(j.v + base + r) + (1.f - b2) * gg * gg;
    const float pn = ldf(j.p + base + r) - step_size * (m / (sqrtf(v) / bc2s + eps));
    stf(j.m + base + r, m); stf(j.v + base + r, v); stf(j.p + base + r, pn);
    opt_tile[idx] = pn;
  }
  __syncthreads();
  T* F = reinterpret_cast<T*>(j.F) + (long)co * j.ld + ci0;
  for (int e = tid; e < taps * 64; e += 256) {
    const int tap = e >> 6, ci = e & 63;
    if (ci0 + ci < j.Cpad) stf(F + (long)tap * j.Cpad + ci, ci < nci ? opt_tile[ci * tp + tap] : 0.f);
  }
}

constexpr long kBnFusedMaxRows = 2048;   // rows up to which BatchNorm runs as one launch (deep layers)
// backward reads two tensors per pass: at 1,536 rows the one-launch form (C/16 = 32 blocks) takes 29-33 us as run, three launches ~18
static const long kBnFusedMaxRowsBwd = [] { const char* e = getenv("MIREG_BN_FUSED_BWD_ROWS"); return e ? atol(e) : 512L; }();

inline int grid1(long work, int cap = 4096) {
  long g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
inline bool vec_ok(int dtype, int C, std::initializer_list<long> lds, std::initializer_list<const void*> ptrs) {
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  if (C % V || C / V > 256) return false;
  for (long l : lds) if (l % V) return false;
  for (const void* p : ptrs) if ((uintptr_t)p % 16) return false;
  return true;
}
inline int bn_blocks(long M, int C, int dtype, bool vec) {
  if (!vec) { long g = M < 256 ? M : 256; return (int)(g < 1 ? 1 : g); }
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  const int rpp = 256 / (C / V);
  long g = (M + (long)rpp * 8 - 1) / ((long)rpp * 8);
  return (int)(g < 1 ? 1 : (g > MIREG_BN_MAX_BLOCKS ? MIREG_BN_MAX_BLOCKS : g));
}

template <typename T, bool BWD>
void launch_partial(const void* y, long ld_y, const void* da, long ld_da, const float* ss, float* partial, long M, int C,
                    float slope, int nblk, bool vec, hipStream_t stream) {
  if (vec) hipLaunchKernelGGL((bn_partial_kernel<T, BWD>), dim3(nblk), dim3(256), 0, stream, (const T*)y, ld_y, (const T*)da, ld_da, ss, partial, M, C, slope);
  else hipLaunchKernelGGL((bn_partial_scalar_kernel<T, BWD>), dim3(nblk), dim3(256), 0, stream, (const T*)y, ld_y, (const T*)da, ld_da, ss, partial, M, C, slope);
}
template <typename T, int MODE>
void launch_elem(const void* y, long ld_y, const void* da, long ld_da, void* out, long ld_o, const float* ss, const float* red,
                 long M, int C, float slope, bool vec, hipStream_t stream) {
  const int V = vec ? VecOf<T>::N : 1;
  const dim3 g(grid1(M * (C / V)));
  if (vec) hipLaunchKernelGGL((bn_elementwise_kernel<T, MODE, true>), g, dim3(256), 0, stream, (const T*)y, ld_y, (const T*)da, ld_da, (T*)out, ld_o, ss, red, M, C, slope);
  else hipLaunchKernelGGL((bn_elementwise_kernel<T, MODE, false>), g, dim3(256), 0, stream, (const T*)y, ld_y, (const T*)da, ld_da, (T*)out, ld_o, ss, red, M, C, slope);
}

}  // namespace

extern "C" {

int mireg_pack_weights(const mireg_pack_job* jobs_dev, int njobs, int total_units, int total_dgrad_units, int dtype,
                       hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 256 && total_units >= 0 && total_dgrad_units >= 0 &&
                  total_units + total_dgrad_units > 0 && (dtype == MIREG_DTYPE_F32 || dtype == MIREG_DTYPE_BF16));
  const int g = total_units / 4 + 1 < 2048 ? total_units / 4 + 1 : 2048;     // total_units == 0: DGRAD packs only (FWD packs are fresh)
  if (dtype == MIREG_DTYPE_BF16) {
    if (total_units) hipLaunchKernelGGL((pack_fwd_kernel<__bf16, 0>), dim3(g), dim3(256), 0, stream, jobs_dev, njobs, total_units);
    if (total_dgrad_units) hipLaunchKernelGGL((pack_dgrad_kernel<__bf16>), dim3(total_dgrad_units), dim3(256), 0, stream, jobs_dev, njobs);
  } else {
    if (total_units) hipLaunchKernelGGL((pack_fwd_kernel<float, 0>), dim3(g), dim3(256), 0, stream, jobs_dev, njobs, total_units);
    if (total_dgrad_units) hipLaunchKernelGGL((pack_dgrad_kernel<float>), dim3(total_dgrad_units), dim3(256), 0, stream, jobs_dev, njobs);
  }
  MIREG_LAUNCH_RET();
}

int mireg_pack_dgrad3d(const mireg_pack3d_job* jobs_dev, int njobs, int total_units, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 256 && total_units > 0);       // kernel extents up to 8 per axis
  MIREG_CHECK_ARG(dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
  const int g = total_units / 4 + 1 < 8192 ? total_units / 4 + 1 : 8192;
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((pack_dgrad3d_kernel<__bf16>), dim3(g), dim3(256), 0, stream, jobs_dev, njobs, total_units);
  else hipLaunchKernelGGL((pack_dgrad3d_kernel<float>), dim3(g), dim3(256), 0, stream, jobs_dev, njobs, total_units);
  MIREG_LAUNCH_RET();
}

int mireg_pack_dgrad3d_fwd(const mireg_pack3d_job* jobs_dev, int njobs, int total_units, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 256 && total_units > 0);
  MIREG_CHECK_ARG(dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((pack_dgrad3d_fwd_kernel<__bf16>), dim3(total_units), dim3(256), 0, stream, jobs_dev, njobs);
  else hipLaunchKernelGGL((pack_dgrad3d_fwd_kernel<float>), dim3(total_units), dim3(256), 0, stream, jobs_dev, njobs);
  MIREG_LAUNCH_RET();
}

int mireg_unpack_wgrad(const mireg_pack_job* jobs_dev, int njobs, int total_units, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 256 && total_units > 0);
  const int g = total_units / 4 + 1 < 2048 ? total_units / 4 + 1 : 2048;
  hipLaunchKernelGGL((pack_fwd_kernel<float, 1>), dim3(g), dim3(256), 0, stream, jobs_dev, njobs, total_units);
  MIREG_LAUNCH_RET();
}

int mireg_wgrad_reduce(const mireg_wopt_job* jobs_dev, int njobs, int total_runits, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 256 && total_runits > 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(total_runits), dim3(256), 0, stream, jobs_dev, njobs);
  MIREG_LAUNCH_RET();
}

int mireg_adam_pack(const mireg_wopt_job* jobs_dev, int njobs, int total_units, int max_taps, int* step_dev, int tick, float lr,
                    float beta1, float beta2, float eps, float grad_scale, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 256 && total_units > 0 && step_dev && max_taps > 0 && max_taps <= kOptMaxTaps &&
                  (dtype == MIREG_DTYPE_F32 || dtype == MIREG_DTYPE_BF16));
  if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, step_dev);
  const size_t lds = (size_t)64 * (max_taps | 1) * sizeof(float);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((adam_pack_kernel<__bf16>), dim3(total_units), dim3(256), lds, stream, jobs_dev, njobs, step_dev, lr, beta1, beta2, eps, grad_scale);
  else
    hipLaunchKernelGGL((adam_pack_kernel<float>), dim3(total_units), dim3(256), lds, stream, jobs_dev, njobs, step_dev, lr, beta1, beta2, eps, grad_scale);
  MIREG_LAUNCH_RET();
}

int mireg_bn_forward(const void* y, long ld_y, void* out, long ld_o, long M, int C, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int training, float slope,
                     float* partial, float* ss, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(y && out && gamma && beta && ss && partial && M > 0 && C > 0 && C <= 4096);
  MIREG_CHECK_ARG(training || (running_mean && running_var));
  const bool vec = vec_ok(dtype, C, {ld_y, ld_o}, {y, out});
  const int nblk = bn_blocks(M, C, dtype, vec);
  if (training && vec && M <= kBnFusedMaxRows) {
    const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4, blocks = (C / V + 1) / 2;
    if (dtype == MIREG_DTYPE_BF16)
      hipLaunchKernelGGL((bn_fused_kernel<__bf16, false>), dim3(blocks), dim3(256), 0, stream, (const __bf16*)y, ld_y, (const __bf16*)nullptr, 0L,
                         (__bf16*)out, ld_o, M, C, gamma, beta, running_mean, running_var, momentum, eps, slope, ss, (float*)nullptr,
                         (float*)nullptr, (float*)nullptr, 0);
    else
      hipLaunchKernelGGL((bn_fused_kernel<float, false>), dim3(blocks), dim3(256), 0, stream, (const float*)y, ld_y, (const float*)nullptr, 0L,
                         (float*)out, ld_o, M, C, gamma, beta, running_mean, running_var, momentum, eps, slope, ss, (float*)nullptr,
                         (float*)nullptr, (float*)nullptr, 0);
    MIREG_LAUNCH_RET();
  }
  if (training) {
    if (dtype == MIREG_DTYPE_BF16) launch_partial<__bf16, false>(y, ld_y, nullptr, 0, nullptr, partial, M, C, slope, nblk, vec, stream);
    else launch_partial<float, false>(y, ld_y, nullptr, 0, nullptr, partial, M, C, slope, nblk, vec, stream);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, partial, nblk, M, C, gamma, beta, running_mean,
                     running_var, momentum, eps, training, ss);
  if (dtype == MIREG_DTYPE_BF16) launch_elem<__bf16, 0>(y, ld_y, nullptr, 0, out, ld_o, ss, nullptr, M, C, slope, vec, stream);
  else launch_elem<float, 0>(y, ld_y, nullptr, 0, out, ld_o, ss, nullptr, M, C, slope, vec, stream);
  MIREG_LAUNCH_RET();
}

int mireg_bn_backward(const void* y, long ld_y, const void* da, long ld_da, void* dy, long ld_dy, const float* ss,
                      float* partial, float* red, float* dgamma, float* dbeta, int acc_param_grads, long M, int C,
                      float slope, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(y && da && dy && ss && partial && red && M > 0 && C > 0 && C <= 4096);
  const bool vec = vec_ok(dtype, C, {ld_y, ld_da, ld_dy}, {y, da, dy});
  const int nblk = bn_blocks(M, C, dtype, vec);
  if (vec && M <= kBnFusedMaxRowsBwd) {
    const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4, blocks = (C / V + 1) / 2;
    float* ssm = const_cast<float*>(ss);
    if (dtype == MIREG_DTYPE_BF16)
      hipLaunchKernelGGL((bn_fused_kernel<__bf16, true>), dim3(blocks), dim3(256), 0, stream, (const __bf16*)y, ld_y, (const __bf16*)da, ld_da,
                         (__bf16*)dy, ld_dy, M, C, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0.f,
                         slope, ssm, red, dgamma, dbeta, acc_param_grads);
    else
      hipLaunchKernelGGL((bn_fused_kernel<float, true>), dim3(blocks), dim3(256), 0, stream, (const float*)y, ld_y, (const float*)da, ld_da,
                         (float*)dy, ld_dy, M, C, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f, 0.f,
                         slope, ssm, red, dgamma, dbeta, acc_param_grads);
    MIREG_LAUNCH_RET();
  }
  if (dtype == MIREG_DTYPE_BF16) launch_partial<__bf16, true>(y, ld_y, da, ld_da, ss, partial, M, C, slope, nblk, vec, stream);
  else launch_partial<float, true>(y, ld_y, da, ld_da, ss, partial, M, C, slope, nblk, vec, stream);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, stream, partial, nblk, M, C, red, dgamma, dbeta,
                     acc_param_grads);
  if (dtype == MIREG_DTYPE_BF16) launch_elem<__bf16, 1>(y, ld_y, da, ld_da, dy, ld_dy, ss, red, M, C, slope, vec, stream);
  else launch_elem<float, 1>(y, ld_y, da, ld_da, dy, ld_dy, ss, red, M, C, slope, vec, stream);
  MIREG_LAUNCH_RET();
}

int mireg_lrelu_bwd(void* g, long ld_g, const void* a, long ld_a, long M, int C, float slope, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(g && a && M > 0 && C > 0);
  const bool vec = vec_ok(dtype, C, {ld_g, ld_a}, {g, a});
  const int V = vec ? (dtype == MIREG_DTYPE_BF16 ? 8 : 4) : 1;
  const dim3 grid(grid1(M * (C / V)));
  if (dtype == MIREG_DTYPE_BF16) {
    if (vec) hipLaunchKernelGGL((lrelu_bwd_kernel<__bf16, true>), grid, dim3(256), 0, stream, (__bf16*)g, ld_g, (const __bf16*)a, ld_a, M, C, slope);
    else hipLaunchKernelGGL((lrelu_bwd_kernel<__bf16, false>), grid, dim3(256), 0, stream, (__bf16*)g, ld_g, (const __bf16*)a, ld_a, M, C, slope);
  } else {
    if (vec) hipLaunchKernelGGL((lrelu_bwd_kernel<float, true>), grid, dim3(256), 0, stream, (float*)g, ld_g, (const float*)a, ld_a, M, C, slope);
    else hipLaunchKernelGGL((lrelu_bwd_kernel<float, false>), grid, dim3(256), 0, stream, (float*)g, ld_g, (const float*)a, ld_a, M, C, slope);
  }
  MIREG_LAUNCH_RET();
}

int mireg_cast_from_f32(void* dst, long ld_d, const float* src, long ld_s, long M, int C, float alpha, float beta, int dtype,
                        hipStream_t stream) {
  MIREG_CHECK_ARG(dst && src && M > 0 && C > 0);
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((cast_from_f32_kernel<__bf16>), dim3(grid1(M * C)), dim3(256), 0, stream, (__bf16*)dst, ld_d, src, ld_s, M, C, alpha, beta);
  else hipLaunchKernelGGL((cast_from_f32_kernel<float>), dim3(grid1(M * C)), dim3(256), 0, stream, (float*)dst, ld_d, src, ld_s, M, C, alpha, beta);
  MIREG_LAUNCH_RET();
}

int mireg_cast_to_f32(float* dst, long ld_d, const void* src, long ld_s, long M, int C, float alpha, float beta, int dtype,
                      hipStream_t stream) {
  MIREG_CHECK_ARG(dst && src && M > 0 && C > 0);
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((cast_to_f32_kernel<__bf16>), dim3(grid1(M * C)), dim3(256), 0, stream, dst, ld_d, (const __bf16*)src, ld_s, M, C, alpha, beta);
  else hipLaunchKernelGGL((cast_to_f32_kernel<float>), dim3(grid1(M * C)), dim3(256), 0, stream, dst, ld_d, (const float*)src, ld_s, M, C, alpha, beta);
  MIREG_LAUNCH_RET();
}

int mireg_nchw_to_nhwc(const float* src, void* dst, int B, int Ctot, int c0, int nc, long HW, long ld, int dtype,
                       hipStream_t stream) {
  MIREG_CHECK_ARG(src && dst && B > 0 && nc > 0 && c0 >= 0 && c0 + nc <= Ctot && HW > 0 && ld >= nc);
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16>), dim3(grid1((long)B * HW)), dim3(256), 0, stream, src, (__bf16*)dst, B, Ctot, c0, nc, HW, ld);
  else hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(grid1((long)B * HW)), dim3(256), 0, stream, src, (float*)dst, B, Ctot, c0, nc, HW, ld);
  MIREG_LAUNCH_RET();
}

int mireg_colsum(const void* g, long ld, long M, int C, float* out, int accumulate, float* workspace, int dtype,
                 hipStream_t stream) {
  MIREG_CHECK_ARG(g && out && M > 0 && C > 0);
  if (vec_ok(dtype, C, {ld}, {g})) {
    const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4, gp = (C / V + 1) / 2;
    int nseg = 1;
    if (M > 8192 && workspace) {
      nseg = (256 + gp - 1) / gp;
      if (nseg > MIREG_COLSUM_MAX_SEGMENTS) nseg = MIREG_COLSUM_MAX_SEGMENTS;
      if ((long)nseg * 2048 > M) nseg = (int)(M / 2048);
      if (nseg < 1) nseg = 1;
    }
    const long rps = (M + nseg - 1) / nseg;
    float* dst = nseg == 1 ? out : workspace;
    if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((colsum_vec_kernel<__bf16>), dim3(gp, nseg), dim3(256), 0, stream, (const __bf16*)g, ld, M, C, dst, accumulate, rps);
    else hipLaunchKernelGGL((colsum_vec_kernel<float>), dim3(gp, nseg), dim3(256), 0, stream, (const float*)g, ld, M, C, dst, accumulate, rps);
    if (nseg > 1) hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, workspace, nseg, C, out, accumulate);
    MIREG_LAUNCH_RET();
  }
  // channel counts that are not a granule multiple (the 2-channel heads and upsamplers): one block column per channel, row segments
  // through the workspace and a fixed-order finalize (deterministic; without a workspace a single segment per channel)
  long gy = workspace ? (M + 256 * 16 - 1) / (256 * 16) : 1;
  if (gy > MIREG_COLSUM_MAX_SEGMENTS) gy = MIREG_COLSUM_MAX_SEGMENTS;
  if (gy < 1) gy = 1;
  float* partial = gy > 1 ? workspace : nullptr;
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((colsum_kernel<__bf16>), dim3(C, (unsigned)gy), dim3(256), 0, stream, (const __bf16*)g, ld, M, C, out, partial, accumulate);
  else hipLaunchKernelGGL((colsum_kernel<float>), dim3(C, (unsigned)gy), dim3(256), 0, stream, (const float*)g, ld, M, C, out, partial, accumulate);
  if (gy > 1) hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const float*)workspace, (int)gy, C, out, accumulate);
  MIREG_LAUNCH_RET();
}

int mireg_zero_many(const mireg_zero_job* jobs_dev, int njobs, int total_units, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && total_units > 0);
  hipLaunchKernelGGL(zero_many_kernel, dim3(total_units), dim3(256), 0, stream, jobs_dev, njobs);
  MIREG_LAUNCH_RET();
}

int mireg_adam_step(const mireg_adam_job* jobs_dev, int njobs, int* step_dev, int tick, float lr, float beta1, float beta2,
                    float eps, float grad_scale, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && step_dev);
  if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, step_dev);
  hipLaunchKernelGGL(adam_kernel, dim3(njobs == 1 ? 4096 : (njobs <= 8 ? 256 : 8), njobs), dim3(256), 0, stream, jobs_dev, step_dev, lr, beta1, beta2, eps, grad_scale);
  MIREG_LAUNCH_RET();
}

}  // extern "C"
