// thin_conv.hip -- 3x3 / stride-1 convolutions with TWO output channels (the predict_flow heads of every
// predictor: FlowNetS/util.py:33-34, flownet2/networks/submodules.py:32-33, PWC/models/PWCNet.py:31-32).
// A two-column GEMM wastes 15/16 of an MFMA tile and is bound by the operand gather, so these layers run on
// the vector ALUs instead, as HBM/L2 streaming kernels over NHWC rows:
//   fwd   y[p][co]       = b[co] + sum_{tap,ci} x[p + tap - 1][ci] F[co][tap][ci]      (lane group per pixel, shuffle-reduced)
//   dgrad dx[q][ci]     (+)= sum_{tap,co} dy[q - tap + 1][co] F[co][tap][ci]            (one 16-byte channel granule per thread)
//   wgrad slab[z][co][tap][ci] = sum_{q in tile z} dy[q - tap + 1][co] x[q][ci]         (register tile per thread, LDS fold)
// All three read the SAME FWD pack F[co][tap*Cpad + ci] the GEMM path uses and wgrad emits the same slab layout,
// so packing, the split-K reduce and the optimizer are shared.
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

namespace {

#if defined(__HIP_DEVICE_COMPILE__)
#define GPTR(T, p) (reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p)))
#else
#define GPTR(T, p) (reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p)))
#endif

constexpr int CO = 2, TAPS = 9;

template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<__bf16> { static constexpr int N = 8; };

__device__ __forceinline__ void unpack16(const uint4 q, float (&v)[4], float) {
  v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w);
}
__device__ __forceinline__ void unpack16(const uint4 q, float (&v)[8], __bf16) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  const bf2 t = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ uint4 pack16(const float (&v)[4], float) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
__device__ __forceinline__ uint4 pack16(const float (&v)[8], __bf16) {
  return make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}
__device__ __forceinline__ float ld1(const float* p) { return *GPTR(const float, p); }
__device__ __forceinline__ float ld1(const __bf16* p) { return (float)*GPTR(const __bf16, p); }
__device__ __forceinline__ void st1(float* p, float v) { *GPTR(float, p) = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *GPTR(__bf16, p) = (__bf16)v; }
// the two gradient / flow components of one pixel (adjacent in memory)
__device__ __forceinline__ void ld2(const float* p, float& a, float& b) { const float2 q = *GPTR(const float2, p); a = q.x; b = q.y; }
__device__ __forceinline__ void ld2(const __bf16* p, float& a, float& b) {
  const uint32_t q = *GPTR(const uint32_t, p);
  a = __uint_as_float(q << 16); b = __uint_as_float(q & 0xffff0000u);
}

struct ThinArgs {
  const void* x; long ld_x;       // wide tensor [pix][ld_x], Cpad channels walked (16-byte granules)
  const void* w; long ld_w;       // FWD pack [CO][TAPS*Cpad]
  const void* t; long ld_t;       // thin tensor [pix][ld_t] (fwd: output y, dgrad/wgrad: incoming dy)
  float* t32; long ld_t32;        // fwd: optional fp32 copy of y
  const float* bias;
  void* dx; long ld_dx;           // dgrad output
  float* slab;                    // wgrad partial slabs [tiles][CO][TAPS*Cpad]
  int B, H, W, Cpad, accumulate, ppl;
  int tpp;                        // dgrad: threads sharing one pixel's dy loads
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
typedef __attribute__((ext_vector_type(2))) float f2_t;

// 16-byte granule dot products: bf16 pairs go through v_dot2c_f32_bf16 (fp32 accumulate), fp32 through fma
__device__ __forceinline__ float dot16(const uint4 x, const uint4 w, float acc, __bf16) {
#if defined(__HIP_DEVICE_COMPILE__)
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, x.x), __builtin_bit_cast(bf2_t, w.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, x.y), __builtin_bit_cast(bf2_t, w.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, x.z), __builtin_bit_cast(bf2_t, w.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, x.w), __builtin_bit_cast(bf2_t, w.w), acc, false);
#endif
  return acc;
}
__device__ __forceinline__ float dot16(const uint4 x, const uint4 w, float acc, float) {
  acc = fmaf(__uint_as_float(x.x), __uint_as_float(w.x), acc);
  acc = fmaf(__uint_as_float(x.y), __uint_as_float(w.y), acc);
  acc = fmaf(__uint_as_float(x.z), __uint_as_float(w.z), acc);
  return fmaf(__uint_as_float(x.w), __uint_as_float(w.w), acc);
}

// ---- forward: LPP lanes share one pixel.  LPP = 8 (fine levels; weights staged in LDS once per block) or
// LPP = 64 (coarse levels: few pixels, many channels; weights straight from L2) --------------------------------
template <typename T, int LPP>
__global__ void __launch_bounds__(256) thin_fwd_kernel(const ThinArgs a) {
  constexpr int V = Vec<T>::N;
  constexpr bool STAGE = LPP < 64;
  extern __shared__ uint4 wl[];
  const int nchunk = a.Cpad / V;
  const T* wg = reinterpret_cast<const T*>(a.w);
  if (STAGE) {
    for (int i = threadIdx.x; i < CO * TAPS * nchunk; i += 256) {
      const int co = i / (TAPS * nchunk), rem = i - co * TAPS * nchunk;
      wl[i] = *GPTR(const uint4, wg + (long)co * a.ld_w + (long)rem * V);
    }
    __syncthreads();
  }
  const int sub = (threadIdx.x & 63) % LPP;
  const long NP = (long)a.B * a.H * a.W;
  const int ppb = 256 / LPP;
  const T* x = reinterpret_cast<const T*>(a.x);
  for (long p0 = (long)blockIdx.x * ppb; p0 < NP; p0 += (long)gridDim.x * ppb) {
    const long p = p0 + threadIdx.x / LPP;
    const bool live = p < NP;
    const long pp = live ? p : 0;
    const int b = (int)(pp / ((long)a.H * a.W)), r = (int)(pp - (long)b * a.H * a.W), y = r / a.W, xx = r - y * a.W;
    // out-of-image taps read the centre pixel and are masked at the end: no divergent branch around the loads,
    // and all nine taps of a channel granule are in flight together
    const T* row[TAPS];
    bool inb[TAPS];
    float t0[TAPS], t1[TAPS];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int iy = y + tap / 3 - 1, ix = xx + tap % 3 - 1;
      inb[tap] = live && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      row[tap] = x + (((long)b * a.H + (inb[tap] ? iy : y)) * a.W + (inb[tap] ? ix : xx)) * a.ld_x;
      t0[tap] = t1[tap] = 0.f;
    }
    for (int c = sub; c < nchunk; c += LPP) {
      uint4 xv[TAPS];
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) xv[tap] = *GPTR(const uint4, row[tap] + c * V);
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        uint4 u, v;
        if (STAGE) { u = wl[tap * nchunk + c]; v = wl[(TAPS + tap) * nchunk + c]; }
        else {
          u = *GPTR(const uint4, wg + ((long)tap * nchunk + c) * V);
          v = *GPTR(const uint4, wg + a.ld_w + ((long)tap * nchunk + c) * V);
        }
        t0[tap] = dot16(xv[tap], u, t0[tap], T());
        t1[tap] = dot16(xv[tap], v, t1[tap], T());
      }
    }
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) { acc0 += inb[tap] ? t0[tap] : 0.f; acc1 += inb[tap] ? t1[tap] : 0.f; }
#pragma unroll
    for (int off = LPP / 2; off > 0; off >>= 1) { acc0 += __shfl_xor(acc0, off, 64); acc1 += __shfl_xor(acc1, off, 64); }
    if (live && sub == 0) {
      if (a.bias) { acc0 += a.bias[0]; acc1 += a.bias[1]; }
      if (a.t) { T* o = reinterpret_cast<T*>(const_cast<void*>(a.t)) + p * a.ld_t; st1(o, acc0); st1(o + 1, acc1); }
      if (a.t32) { float* o = a.t32 + p * a.ld_t32; st1(o, acc0); st1(o + 1, acc1); }
    }
  }
}

// ---- backward-data: one 16-byte channel granule of one pixel per thread; the weights sit in LDS interleaved as
// (co0, co1) pairs per channel, so one bf16 dot2 with the raw (dy0, dy1) pair handles a channel and a tap ----------
__device__ __forceinline__ uint32_t ld_pair(const __bf16* p, bool ok) { const uint32_t g = *GPTR(const uint32_t, p); return ok ? g : 0u; }
__device__ __forceinline__ float2 ld_pair(const float* p, bool ok) { const float2 g = *GPTR(const float2, p); return ok ? g : make_float2(0.f, 0.f); }
__device__ __forceinline__ void dgrad_tap(const uint32_t g, const uint4* wp, float (&acc)[8]) {
  const uint4 w0 = wp[0], w1 = wp[1];
  const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int e = 0; e < 8; ++e)
    acc[e] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, g), __builtin_bit_cast(bf2_t, w[e]), acc[e], false);
#endif
}
__device__ __forceinline__ void dgrad_tap(const float2 g, const uint4* wp, float (&acc)[4]) {
  const uint4 w0 = wp[0], w1 = wp[1];
  const float w[8] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w),
                      __uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z), __uint_as_float(w1.w)};
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = fmaf(g.x, w[2 * e], fmaf(g.y, w[2 * e + 1], acc[e]));
}

template <typename T>
__global__ void __launch_bounds__(256) thin_dgrad_kernel(const ThinArgs a) {
  constexpr int V = Vec<T>::N;
  extern __shared__ uint4 wl[];                       // [tap][chunk][2]: V channels x (co0, co1)
  const int nchunk = a.Cpad / V;
  {
    const T* wg = reinterpret_cast<const T*>(a.w);
    for (int i = threadIdx.x; i < TAPS * nchunk; i += 256) {            // 16-byte loads, pairs interleaved in registers
      const uint4 u = *GPTR(const uint4, wg + (long)i * V), v = *GPTR(const uint4, wg + a.ld_w + (long)i * V);
      if constexpr (sizeof(T) == 2) {
        wl[2 * i] = make_uint4((u.x & 0xffffu) | (v.x << 16), (u.x >> 16) | (v.x & 0xffff0000u),
                               (u.y & 0xffffu) | (v.y << 16), (u.y >> 16) | (v.y & 0xffff0000u));
        wl[2 * i + 1] = make_uint4((u.z & 0xffffu) | (v.z << 16), (u.z >> 16) | (v.z & 0xffff0000u),
                                   (u.w & 0xffffu) | (v.w << 16), (u.w >> 16) | (v.w & 0xffff0000u));
      } else {
        wl[2 * i] = make_uint4(u.x, v.x, u.y, v.y);
        wl[2 * i + 1] = make_uint4(u.z, v.z, u.w, v.w);
      }
    }
    __syncthreads();
  }
  // a.tpp threads share a pixel: each loads the pixel's nine (dy0, dy1) pairs once and walks the channel granules c0, c0 + tpp, ...
  // with them (the loads are the latency chain of this kernel; one granule per thread re-loaded them for every granule)
  const int tpp = a.tpp;
  const long NP = (long)a.B * a.H * a.W, total = NP * tpp;
  const T* dy = reinterpret_cast<const T*>(a.t);
  T* dx = reinterpret_cast<T*>(a.dx);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / tpp;
    const int c0 = (int)(i - p * tpp);
    if (c0 >= nchunk) continue;
    const int b = (int)(p / ((long)a.H * a.W)), r = (int)(p - (long)b * a.H * a.W), y = r / a.W, xx = r - y * a.W;
    decltype(ld_pair(dy, true)) g[TAPS];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {                              // nine (dy0, dy1) pairs in flight, branch-free
      const int qy = y - (tap / 3 - 1), qx = xx - (tap % 3 - 1);       // the output pixel whose window covers (y, xx) at this tap
      const bool ok = (unsigned)qy < (unsigned)a.H && (unsigned)qx < (unsigned)a.W;
      g[tap] = ld_pair(dy + (((long)b * a.H + (ok ? qy : y)) * a.W + (ok ? qx : xx)) * a.ld_t, ok);
    }
    for (int c = c0; c < nchunk; c += tpp) {
      float acc[V];
      T* out = dx + p * a.ld_dx + c * V;
      if (a.accumulate) unpack16(*GPTR(const uint4, out), acc, T());
      else {
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
      }
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) dgrad_tap(g[tap], wl + ((long)tap * nchunk + c) * 2, acc);
      *GPTR(uint4, out) = pack16(acc, T());
    }
  }
}

// ---- backward-weights: block = 32 channel granules x 8 pixel lanes; pixel lanes fold through LDS tap by tap --------
template <typename T>
__global__ void __launch_bounds__(256) thin_wgrad_kernel(const ThinArgs a) {
  constexpr int V = Vec<T>::N;
  __shared__ float fold[8][32][CO * V + 1];
  const int nchunk = a.Cpad / V;
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.y * 32 + cl;
  const bool cok = c < nchunk;
  const long NP = (long)a.B * a.H * a.W;
  const long q0 = ((long)blockIdx.x * 8 + pl) * a.ppl;
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* dy = reinterpret_cast<const T*>(a.t);
  f2_t acc[TAPS][CO][V / 2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
      for (int e = 0; e < V / 2; ++e) acc[t][o][e] = f2_t{0.f, 0.f};
  if (cok) {
    // register double buffer: pixel i+1's granule and its nine (dy0, dy1) pairs are loading while pixel i is accumulated
    uint4 xq;
    float g0[TAPS], g1[TAPS];
    auto fetch = [&](long q, uint4& xo, float (&h0)[TAPS], float (&h1)[TAPS]) {
      const bool qok = q < NP;
      const long qq = qok ? q : NP - 1;
      const int b = (int)(qq / ((long)a.H * a.W)), r = (int)(qq - (long)b * a.H * a.W), y = r / a.W, xx = r - y * a.W;
      xo = *GPTR(const uint4, x + qq * a.ld_x + c * V);
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int py = y - (tap / 3 - 1), px = xx - (tap % 3 - 1);     // output pixel that read x[q] through this tap
        const bool inb = qok && (unsigned)py < (unsigned)a.H && (unsigned)px < (unsigned)a.W;
        ld2(dy + (((long)b * a.H + (inb ? py : y)) * a.W + (inb ? px : xx)) * a.ld_t, h0[tap], h1[tap]);
        h0[tap] = inb ? h0[tap] : 0.f;
        h1[tap] = inb ? h1[tap] : 0.f;
      }
    };
    fetch(q0, xq, g0, g1);
    for (int i = 0; i < a.ppl; ++i) {
      uint4 xn;
      float n0[TAPS], n1[TAPS];
      fetch(q0 + i + 1 < q0 + a.ppl ? q0 + i + 1 : NP, xn, n0, n1);    // past the lane's run: masked to zero
      float xv[V];
      unpack16(xq, xv, T());
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const f2_t a0 = {g0[tap], g0[tap]}, a1 = {g1[tap], g1[tap]};
#pragma unroll
        for (int e = 0; e < V / 2; ++e) {
          const f2_t xp = {xv[2 * e], xv[2 * e + 1]};
          acc[tap][0][e] = __builtin_elementwise_fma(a0, xp, acc[tap][0][e]);
          acc[tap][1][e] = __builtin_elementwise_fma(a1, xp, acc[tap][1][e]);
        }
      }
      xq = xn;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) { g0[tap] = n0[tap]; g1[tap] = n1[tap]; }
    }
  }
  float* slab = a.slab + (long)blockIdx.x * CO * TAPS * a.Cpad;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    __syncthreads();
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
      for (int e = 0; e < V / 2; ++e) { fold[pl][cl][o * V + 2 * e] = acc[tap][o][e].x; fold[pl][cl][o * V + 2 * e + 1] = acc[tap][o][e].y; }
    __syncthreads();
    // 32 granules x CO x V values of this tap, summed over the 8 pixel lanes in fixed order
    for (int k = threadIdx.x; k < 32 * CO * V; k += 256) {
      const int g = k / (CO * V), rem = k - g * (CO * V), o = rem / V, e = rem - o * V;
      const int cc = blockIdx.y * 32 + g;
      if (cc < nchunk) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 8; ++l) s += fold[l][g][rem];
        *GPTR(float, slab + (long)o * TAPS * a.Cpad + (long)tap * a.Cpad + cc * V + e) = s;
      }
    }
  }
}

inline bool thin_ok(const void* x, long ld_x, const void* w, long ld_w, int B, int H, int W, int Cpad, int dtype) {
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  const long es = dtype == MIREG_DTYPE_BF16 ? 2 : 4;
  return x && w && B > 0 && H > 0 && W > 0 && Cpad > 0 && Cpad % V == 0 && (ld_x * es) % 16 == 0 && (ld_w * es) % 16 == 0 &&
         ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ld_w >= (long)TAPS * Cpad &&
         (dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32) && (long)CO * TAPS * Cpad * es <= 160 * 1024 - 4096;
}

}  // namespace

// ---- GEMM formulation of the heads at the fine levels (>= 16k pixels): z[p][co*9 + tap] = sum_c x[p][c] * w[co][tap][c] is a 1x1
// convolution to 18 columns (the FWD pack [2][9*Cpad] IS [18][Cpad] row-major) that reads x once instead of nine times, and
// y[p][co] = bias + sum_tap z[p + tap][co*9 + tap] a 9-tap shift-sum over 72 B per pixel; backward-weights gathers
// dz[q][co*9 + tap] = dy[q - tap][co] and runs the ring backward-weights GEMM on (dz, x).  The GEMMs are mireg_conv_gemm /
// mireg_conv_wgrad; these two kernels are the pixel-wise ends.
template <typename T>
__global__ void __launch_bounds__(256)
thin_shift_sum_kernel(const float* __restrict__ z, long ld_z, const float* __restrict__ bias, T* __restrict__ y, long ld_y,
                      float* __restrict__ y32, long ld_y32, int B, int H, int W) {
  const long n = (long)B * H * W * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i & 1);
    const long p = i >> 1;
    long rx, ry;
    const long qy = fast_divmod(p, W, rx);
    (void)fast_divmod(qy, H, ry);
    const int xx = (int)rx, yy = (int)ry;
    float acc = bias ? bias[co] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = yy + ky - 1, ix = xx + kx - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
          acc += z[(p + (long)(ky - 1) * W + (kx - 1)) * ld_z + co * 9 + ky * 3 + kx];
      }
    if (y) y[p * ld_y + co] = (T)acc;
    if (y32) y32[p * ld_y32 + co] = acc;
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
thin_gather18_kernel(const T* __restrict__ dy, long ld_dy, T* __restrict__ dz, long ld_dz, int B, int H, int W) {
  const long n = (long)B * H * W * 18;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 18);
    const long q = i / 18;
    const int co = c / 9, tap = c - co * 9, ky = tap / 3, kx = tap - ky * 3;
    const int xx = (int)(q % W), yy = (int)((q / W) % H);
    const int sy = yy - (ky - 1), sx = xx - (kx - 1);                  // y[s] received z[q] through tap (ky, kx) when q = s + tap
    T v = (T)0.f;
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) v = dy[(q - (long)(ky - 1) * W - (kx - 1)) * ld_dy + co];
    dz[q * ld_dz + c] = v;
  }
}

// ---- the 2 -> 2 channel flow / feature upsamplers, ConvTranspose2d(2, 2, 4, 2, 1) (FlowNetS/FlowNetS.py:37-40,
// flownet2/networks/FlowNetC.py:53-56, PWC/models/PWCNet.py:80,95,...): a few MFLOP each, 12 GEMM launches per FlowNetS step
// when run through the 128-wide tiles.  Here: one thread per pixel on the fp32 master weight Wc[co][ci][ky][kx] (the Conv2d
// weight of the adjoint, stride-2 convolution fine(ci) -> coarse(co); 32 floats, read through the scalar cache).
template <typename T>
__global__ void __launch_bounds__(256)
tiny_deconv_fwd_kernel(const T* __restrict__ xc, long ld_c, const float* __restrict__ w, const float* __restrict__ bias,
                       T* __restrict__ yf, long ld_f, float* __restrict__ y32, long ld_32, int B, int Hc, int Wc_) {
  const int Hf = 2 * Hc, Wf = 2 * Wc_;
  const long n = (long)B * Hf * Wf;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
    long rx, ry;
    const long qy = fast_divmod(p, Wf, rx), b = fast_divmod(qy, Hf, ry);
    const int ix = (int)rx, iy = (int)ry;
    float a0 = bias ? bias[0] : 0.f, a1 = bias ? bias[1] : 0.f;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
      const int ky = ((iy + 1) & 1) + 2 * ty, oy = (iy + 1 - ky) >> 1;          // fine row iy receives taps ky = (iy+1) mod 2, +2
      if ((unsigned)oy >= (unsigned)Hc) continue;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int kx = ((ix + 1) & 1) + 2 * tx, ox = (ix + 1 - kx) >> 1;
        if ((unsigned)ox >= (unsigned)Wc_) continue;
        const T* src = xc + ((b * Hc + oy) * Wc_ + ox) * ld_c;
        const float c0 = (float)src[0], c1 = (float)src[1];
        a0 += c0 * w[(0 * 2 + 0) * 16 + ky * 4 + kx] + c1 * w[(1 * 2 + 0) * 16 + ky * 4 + kx];
        a1 += c0 * w[(0 * 2 + 1) * 16 + ky * 4 + kx] + c1 * w[(1 * 2 + 1) * 16 + ky * 4 + kx];
      }
    }
    if (yf) { T* d = yf + p * ld_f; d[0] = (T)a0; d[1] = (T)a1; }
    if (y32) { float* d = y32 + p * ld_32; d[0] = a0; d[1] = a1; }              // fp32 copy of the flow for the loss tail
  }
}

// backward-data of the upsampler = the stride-2 convolution itself: coarse[o][co] (+)= sum fine[2o + k - 1][ci] * Wc[co][ci][k]
template <typename T>
__global__ void __launch_bounds__(256)
tiny_conv_fwd_kernel(const T* __restrict__ xf, long ld_f, const float* __restrict__ w, T* __restrict__ yc, long ld_c, int accumulate,
                     const float* __restrict__ add_nchw, int B, int Hc, int Wc_) {
  const int Hf = 2 * Hc, Wf = 2 * Wc_;
  const long n = (long)B * Hc * Wc_;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
    long rx, ry;
    const long qy = fast_divmod(p, Wc_, rx), b = fast_divmod(qy, Hc, ry);
    const int ox = (int)rx, oy = (int)ry;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = 2 * oy + ky - 1;
      if ((unsigned)iy >= (unsigned)Hf) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = 2 * ox + kx - 1;
        if ((unsigned)ix >= (unsigned)Wf) continue;
        const T* src = xf + ((b * Hf + iy) * Wf + ix) * ld_f;
        const float f0 = (float)src[0], f1 = (float)src[1];
        a0 += f0 * w[(0 * 2 + 0) * 16 + ky * 4 + kx] + f1 * w[(0 * 2 + 1) * 16 + ky * 4 + kx];
        a1 += f0 * w[(1 * 2 + 0) * 16 + ky * 4 + kx] + f1 * w[(1 * 2 + 1) * 16 + ky * 4 + kx];
      }
    }
    T* d = yc + p * ld_c;
    if (add_nchw) {                                                    // a planar fp32 (B, 2, Hc, Wc) term, e.g. the loss gradient
      const long hw = (long)Hc * Wc_, o = b * 2 * hw + (long)oy * Wc_ + ox;
      a0 += add_nchw[o]; a1 += add_nchw[o + hw];
    }
    if (accumulate) { a0 += (float)d[0]; a1 += (float)d[1]; }
    d[0] = (T)a0; d[1] = (T)a1;
  }
}

// backward-weights: slab[blk][co][(ky*4+kx)*Cpad + ci] = sum over the block's coarse pixels of dy[o][co] * fine[2o + k - 1][ci]
// (fixed pixel ranges per block, fixed-order tree sum: deterministic); pad slots are never written (zeroed slabs)
template <typename T>
__global__ void __launch_bounds__(256)
tiny_wgrad_kernel(const T* __restrict__ xf, long ld_f, const T* __restrict__ dyc, long ld_c, float* __restrict__ slab, int Cpad,
                  int B, int Hc, int Wc_) {
  __shared__ float red[4][64];
  const int Hf = 2 * Hc, Wf = 2 * Wc_;
  const long n = (long)B * Hc * Wc_;
  const long per = (n + gridDim.x - 1) / gridDim.x, p0 = (long)blockIdx.x * per, p1 = min(n, p0 + per);
  float acc[64];                                                      // [co][ci][ky][kx]
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 0.f;
  for (long p = p0 + threadIdx.x; p < p1; p += 256) {
    long rx, ry;
    const long qy = fast_divmod(p, Wc_, rx), b = fast_divmod(qy, Hc, ry);
    const int ox = (int)rx, oy = (int)ry;
    const float g0 = (float)dyc[p * ld_c], g1 = (float)dyc[p * ld_c + 1];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = 2 * oy + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = 2 * ox + kx - 1;
        float f0 = 0.f, f1 = 0.f;
        if ((unsigned)iy < (unsigned)Hf && (unsigned)ix < (unsigned)Wf) {
          const T* src = xf + ((b * Hf + iy) * Wf + ix) * ld_f;
          f0 = (float)src[0]; f1 = (float)src[1];
        }
        acc[(0 * 2 + 0) * 16 + ky * 4 + kx] += g0 * f0; acc[(0 * 2 + 1) * 16 + ky * 4 + kx] += g0 * f1;
        acc[(1 * 2 + 0) * 16 + ky * 4 + kx] += g1 * f0; acc[(1 * 2 + 1) * 16 + ky * 4 + kx] += g1 * f1;
      }
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const float v = wave_sum(acc[i]);
    if (lane == 0) red[wid][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int i = threadIdx.x, co = i >> 5, ci = (i >> 4) & 1, tap = i & 15;
    slab[(long)blockIdx.x * 2 * 16 * Cpad + (long)co * 16 * Cpad + tap * Cpad + ci] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

extern "C" {

int mireg_thin_conv_fwd(const void* x, long ld_x, const void* w, long ld_w, const float* bias, void* y, long ld_y,
                        float* y32, long ld_y32, int B, int H, int W, int Cpad, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(thin_ok(x, ld_x, w, ld_w, B, H, W, Cpad, dtype) && (y || y32) && (!y || ld_y >= 2) && (!y32 || ld_y32 >= 2));
  ThinArgs a{};
  a.x = x; a.ld_x = ld_x; a.w = w; a.ld_w = ld_w; a.t = y; a.ld_t = ld_y; a.t32 = y32; a.ld_t32 = ld_y32; a.bias = bias;
  a.B = B; a.H = H; a.W = W; a.Cpad = Cpad;
  const long NP = (long)B * H * W;
  const size_t lds = (size_t)CO * TAPS * Cpad * (dtype == MIREG_DTYPE_BF16 ? 2 : 4);
  const bool wide = NP < 16384;                       // few pixels, many channels: a whole wave per pixel
  const long ppb = wide ? 4 : 32;
  const int grid = (int)((NP + ppb - 1) / ppb < 2048 ? (NP + ppb - 1) / ppb : 2048);
#define MIREG_THIN_FWD(T, LPP) { \
    const size_t l = LPP < 64 ? lds : 0; \
    if (l > 48 * 1024) (void)hipFuncSetAttribute((const void*)thin_fwd_kernel<T, LPP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); \
    hipLaunchKernelGGL((thin_fwd_kernel<T, LPP>), dim3(grid), dim3(256), l, stream, a); }
  if (dtype == MIREG_DTYPE_BF16) { if (wide) MIREG_THIN_FWD(__bf16, 64) else MIREG_THIN_FWD(__bf16, 8) }
  else { if (wide) MIREG_THIN_FWD(float, 64) else MIREG_THIN_FWD(float, 8) }
#undef MIREG_THIN_FWD
  MIREG_LAUNCH_RET();
}

int mireg_thin_conv_dgrad(const void* dy, long ld_dy, const void* w, long ld_w, void* dx, long ld_dx, int accumulate,
                          int B, int H, int W, int Cpad, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(thin_ok(dx, ld_dx, w, ld_w, B, H, W, Cpad, dtype) && dy && ld_dy >= 2 &&
                  ((uintptr_t)dy % (dtype == MIREG_DTYPE_BF16 ? 4 : 8)) == 0 && ld_dy % 2 == 0);
  ThinArgs a{};
  a.w = w; a.ld_w = ld_w; a.t = dy; a.ld_t = ld_dy; a.dx = dx; a.ld_dx = ld_dx; a.accumulate = accumulate;
  a.B = B; a.H = H; a.W = W; a.Cpad = Cpad;
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  // threads per pixel: 8 share the dy loads on the big levels; small levels (few pixels, many channels) spread the granules
  // over more threads instead, down to one granule per thread
  int tpp = 8;
  while ((long)B * H * W * tpp < 65536 && tpp < Cpad / V) tpp *= 2;
  if (tpp > Cpad / V) tpp = Cpad / V;
  a.tpp = tpp;
  const long total = (long)B * H * W * tpp;
  const size_t lds = (size_t)CO * TAPS * Cpad * (dtype == MIREG_DTYPE_BF16 ? 2 : 4);
  // latency-bound (nine dependent dy loads per item): as many resident blocks per CU as the weight tile in LDS allows
  const long per_cu = lds + 1024 > 160 * 1024 / 8 ? (long)(160 * 1024 / (lds + 1024)) : 8;
  const long cap = 256 * (per_cu < 1 ? 1 : per_cu);
  const int grid = (int)((total + 255) / 256 < cap ? (total + 255) / 256 : cap);
  if (dtype == MIREG_DTYPE_BF16) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)thin_dgrad_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((thin_dgrad_kernel<__bf16>), dim3(grid), dim3(256), lds, stream, a);
  } else {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)thin_dgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((thin_dgrad_kernel<float>), dim3(grid), dim3(256), lds, stream, a);
  }
  MIREG_LAUNCH_RET();
}

int mireg_thin_conv_wgrad_tiles(int B, int H, int W, int Cpad, int dtype, int* pixels_per_lane) {
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  const long NP = (long)B * H * W;
  const int groups = (Cpad / V + 31) / 32;
  long ppl = NP / (8L * (256 / groups > 0 ? 256 / groups : 1));
  ppl = ppl < 1 ? 1 : (ppl > 64 ? 64 : ppl);
  if (pixels_per_lane) *pixels_per_lane = (int)ppl;
  return (int)((NP + 8 * ppl - 1) / (8 * ppl));
}

int mireg_thin_conv_wgrad(const void* x, long ld_x, const void* dy, long ld_dy, float* slab, int ntiles, int B, int H, int W,
                          int Cpad, int dtype, hipStream_t stream) {
  int ppl = 0;
  MIREG_CHECK_ARG(x && dy && slab && B > 0 && H > 0 && W > 0 && Cpad > 0 && ld_dy >= 2 && ld_dy % 2 == 0 &&
                  (dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32) && Cpad % (dtype == MIREG_DTYPE_BF16 ? 8 : 4) == 0 &&
                  ((uintptr_t)x % 16) == 0 && (ld_x * (dtype == MIREG_DTYPE_BF16 ? 2 : 4)) % 16 == 0 &&
                  ((uintptr_t)dy % (dtype == MIREG_DTYPE_BF16 ? 4 : 8)) == 0 &&
                  mireg_thin_conv_wgrad_tiles(B, H, W, Cpad, dtype, &ppl) == ntiles);
  ThinArgs a{};
  a.x = x; a.ld_x = ld_x; a.t = dy; a.ld_t = ld_dy; a.slab = slab; a.B = B; a.H = H; a.W = W; a.Cpad = Cpad; a.ppl = ppl;
  const int V = dtype == MIREG_DTYPE_BF16 ? 8 : 4;
  const dim3 grid(ntiles, (Cpad / V + 31) / 32);
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((thin_wgrad_kernel<__bf16>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((thin_wgrad_kernel<float>), grid, dim3(256), 0, stream, a);
  MIREG_LAUNCH_RET();
}


int mireg_thin_shift_sum(const float* z, long ld_z, const float* bias, void* y, long ld_y, float* y32, long ld_y32, int B, int H,
                         int W, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(z && ld_z >= 18 && (y || y32) && B > 0 && H > 0 && W > 0 && (dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32));
  const long n = (long)B * H * W * 2;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((thin_shift_sum_kernel<__bf16>), dim3(grid), dim3(256), 0, stream, z, ld_z, bias, (__bf16*)y, ld_y, y32, ld_y32, B, H, W);
  else
    hipLaunchKernelGGL((thin_shift_sum_kernel<float>), dim3(grid), dim3(256), 0, stream, z, ld_z, bias, (float*)y, ld_y, y32, ld_y32, B, H, W);
  MIREG_LAUNCH_RET();
}

int mireg_thin_gather18(const void* dy, long ld_dy, void* dz, long ld_dz, int B, int H, int W, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(dy && dz && ld_dy >= 2 && ld_dz >= 18 && B > 0 && H > 0 && W > 0 && (dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32));
  const long n = (long)B * H * W * 18;
  const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((thin_gather18_kernel<__bf16>), dim3(grid), dim3(256), 0, stream, (const __bf16*)dy, ld_dy, (__bf16*)dz, ld_dz, B, H, W);
  else
    hipLaunchKernelGGL((thin_gather18_kernel<float>), dim3(grid), dim3(256), 0, stream, (const float*)dy, ld_dy, (float*)dz, ld_dz, B, H, W);
  MIREG_LAUNCH_RET();
}


int mireg_tiny_deconv_blocks(int B, int Hc, int Wc) {
  const long n = (long)B * Hc * Wc;
  const long g = (n + 255) / 256;                      // one coarse pixel per thread: the kernel is a chain of dependent gathers,
  return (int)(g < 1 ? 1 : (g > 192 ? 192 : g));       // so it wants blocks, not work per block (slab: 1 KiB per block)
}

int mireg_tiny_deconv_fwd(const void* x_coarse, long ld_c, const float* w, const float* bias, void* y_fine, long ld_f, float* y32,
                          long ld_y32, int B, int Hc, int Wc, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(x_coarse && w && (y_fine || y32) && ld_c >= 2 && (!y_fine || ld_f >= 2) && (!y32 || ld_y32 >= 2) && B > 0 && Hc > 0 && Wc > 0);
  MIREG_CHECK_ARG(dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
  const long n = (long)B * Hc * Wc * 4;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((tiny_deconv_fwd_kernel<__bf16>), dim3(grid), dim3(256), 0, stream, (const __bf16*)x_coarse, ld_c, w, bias, (__bf16*)y_fine, ld_f, y32, ld_y32, B, Hc, Wc);
  else
    hipLaunchKernelGGL((tiny_deconv_fwd_kernel<float>), dim3(grid), dim3(256), 0, stream, (const float*)x_coarse, ld_c, w, bias, (float*)y_fine, ld_f, y32, ld_y32, B, Hc, Wc);
  MIREG_LAUNCH_RET();
}

int mireg_tiny_deconv_bwd_data(const void* g_fine, long ld_f, const float* w, void* dx_coarse, long ld_c, int accumulate,
                               const float* add_nchw, int B, int Hc, int Wc, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(g_fine && w && dx_coarse && ld_c >= 2 && ld_f >= 2 && B > 0 && Hc > 0 && Wc > 0);
  MIREG_CHECK_ARG(dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32);
  const long n = (long)B * Hc * Wc;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((tiny_conv_fwd_kernel<__bf16>), dim3(grid), dim3(256), 0, stream, (const __bf16*)g_fine, ld_f, w, (__bf16*)dx_coarse, ld_c, accumulate, add_nchw, B, Hc, Wc);
  else
    hipLaunchKernelGGL((tiny_conv_fwd_kernel<float>), dim3(grid), dim3(256), 0, stream, (const float*)g_fine, ld_f, w, (float*)dx_coarse, ld_c, accumulate, add_nchw, B, Hc, Wc);
  MIREG_LAUNCH_RET();
}

int mireg_tiny_deconv_bwd_weights(const void* g_fine, long ld_f, const void* x_coarse, long ld_c, float* slab, int nblocks, int Cpad,
                                  int B, int Hc, int Wc, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(g_fine && x_coarse && slab && ld_c >= 2 && ld_f >= 2 && B > 0 && Hc > 0 && Wc > 0 && Cpad >= 2);
  MIREG_CHECK_ARG(nblocks == mireg_tiny_deconv_blocks(B, Hc, Wc) && (dtype == MIREG_DTYPE_BF16 || dtype == MIREG_DTYPE_F32));
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((tiny_wgrad_kernel<__bf16>), dim3(nblocks), dim3(256), 0, stream, (const __bf16*)g_fine, ld_f, (const __bf16*)x_coarse, ld_c, slab, Cpad, B, Hc, Wc);
  else
    hipLaunchKernelGGL((tiny_wgrad_kernel<float>), dim3(nblocks), dim3(256), 0, stream, (const float*)g_fine, ld_f, (const float*)x_coarse, ld_c, slab, Cpad, B, Hc, Wc);
  MIREG_LAUNCH_RET();
}

}  // extern "C"
