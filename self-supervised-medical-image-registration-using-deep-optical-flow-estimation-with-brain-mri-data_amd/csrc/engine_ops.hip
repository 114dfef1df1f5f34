// engine_ops.hip -- the HBM-bound glue around the MFMA contractions (all NHWC [rows][ld] views):
//   weight packing / gradient unpacking between torch layout and the GEMM layouts,
//   train-mode BatchNorm2d + LeakyReLU(0.1) forward / backward (FlowNetS/util.py:17-30),
//   LeakyReLU backward, fp32 <-> compute-dtype casts with accumulate, NCHW -> NHWC input staging,
//   fused multi-tensor Adam exactly as train.py:129 configures it (eps = 1e-4).
// Table-driven launches (one grid.y slot per job) keep the per-step launch count flat.
#include "mireg_common.h"
#include "../../include/mireg.h"

using namespace mireg;

namespace {

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(__bf16* p, float v) { *p = (__bf16)v; }

// ----------------------------------------------------------------------------------------------
// weight pack: dst[row][(ty*ntx+tx)*Cpad + c] = W[co][ci][ky0+sy*ty][kx0+sx*tx]   (0 in the pad)
//   kind 0 (FWD)  : row = co, c = ci         kind 1 (DGRAD): row = ci, c = co
template <typename T>
__global__ void __launch_bounds__(256) pack_weights_kernel(const mireg_pack_job* __restrict__ jobs) {
  const mireg_pack_job j = jobs[blockIdx.y];
  const long total = (long)j.rows * j.ld;
  T* __restrict__ dst = reinterpret_cast<T*>(j.dst);
  const int ktaps = j.nty * j.ntx * j.Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / j.ld), k = (int)(i - (long)row * j.ld);
    float v = 0.f;
    if (k < ktaps) {
      const int tap = k / j.Cpad, c = k - tap * j.Cpad;
      const int ty = tap / j.ntx, tx = tap - ty * j.ntx;
      const int ky = j.ky0 + j.sy * ty, kx = j.kx0 + j.sx * tx;
      const int co = j.kind == 0 ? row : c, ci = j.kind == 0 ? c : row;
      if (co < j.Co && ci < j.Ci && ky < j.kh && kx < j.kw) v = j.src[(((long)co * j.Ci + ci) * j.kh + ky) * j.kw + kx];
    }
    stf(dst + i, v);
  }
}

// gradient unpack: grad[co][ci][ky][kx] = sum_z slab[z][co][(ky*kw+kx)*Cpad + ci]   (torch layout, fp32)
__global__ void __launch_bounds__(256) unpack_wgrad_kernel(const mireg_pack_job* __restrict__ jobs) {
  const mireg_pack_job j = jobs[blockIdx.y];   // src = slab, dst = grad, rows = Co, ld = taps*Cpad, sy = nsplit
  const long total = (long)j.Co * j.Ci * j.kh * j.kw;
  const long slab_sz = (long)j.Co * j.ld;
  float* __restrict__ dst = reinterpret_cast<float*>(j.dst);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int kx = (int)(i % j.kw);
    const int ky = (int)((i / j.kw) % j.kh);
    const int ci = (int)((i / ((long)j.kw * j.kh)) % j.Ci);
    const int co = (int)(i / ((long)j.kw * j.kh * j.Ci));
    const long s = (long)co * j.ld + (long)(ky * j.kw + kx) * j.Cpad + ci;
    float v = 0.f;
    for (int z = 0; z < j.sy; ++z) v += j.src[z * slab_sz + s];
    dst[i] = j.kind ? dst[i] + v : v;
  }
}

// ----------------------------------------------------------------------------------------------
// BatchNorm (train mode = batch statistics over all rows)
template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
bn_reduce_kernel(const T* __restrict__ y, long ld_y, const T* __restrict__ da, long ld_da, const float* __restrict__ ss,
                 double* __restrict__ sums, long M, int C, float slope) {
  // FWD: sums[c] += y, sums[C+c] += y^2.   BWD: dz = da*lrelu'(z); sums[c] += dz, sums[C+c] += dz*xhat
  constexpr int VEC = 4;
  __shared__ float red[2][256][VEC];
  const int cpr = (C + VEC - 1) / VEC;                 // channel groups per row
  const int tx = threadIdx.x % min(cpr, 256), tyy = threadIdx.x / min(cpr, 256);
  const int rpp = 256 / min(cpr, 256);                 // rows per pass
  const float* scale = ss, *shift = ss + C, *mean = ss + 2 * C, *rstd = ss + 3 * C;
  for (int cbase = 0; cbase < cpr; cbase += 256) {   // uniform trip count: the loop body holds barriers
    const int cg = cbase + tx;
    float a0[VEC] = {0, 0, 0, 0}, a1[VEC] = {0, 0, 0, 0};
    float sc[VEC], sh[VEC], mu[VEC], rs[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int c = cg * VEC + v;
      const bool ok = BWD && c < C;
      sc[v] = ok ? scale[c] : 0.f; sh[v] = ok ? shift[c] : 0.f; mu[v] = ok ? mean[c] : 0.f; rs[v] = ok ? rstd[c] : 0.f;
    }
    if (tyy < rpp && cg < cpr) {
      for (long m = (long)blockIdx.x * rpp + tyy; m < M; m += (long)gridDim.x * rpp) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int c = cg * VEC + v;
          if (c >= C) continue;
          const float yv = ldf(y + m * ld_y + c);
          if (!BWD) { a0[v] += yv; a1[v] += yv * yv; }
          else {
            const float z = yv * sc[v] + sh[v];
            const float dz = ldf(da + m * ld_da + c) * (z > 0.f ? 1.f : slope);
            a0[v] += dz; a1[v] += dz * ((yv - mu[v]) * rs[v]);
          }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) { red[0][threadIdx.x][v] = a0[v]; red[1][threadIdx.x][v] = a1[v]; }
    __syncthreads();
    if (tyy == 0 && cpr <= 256) {
      const int w = min(cpr, 256);
      for (int rr = 1; rr < rpp; ++rr)
#pragma unroll
        for (int v = 0; v < VEC; ++v) { a0[v] += red[0][rr * w + tx][v]; a1[v] += red[1][rr * w + tx][v]; }
    }
    if (tyy == 0 && cg < cpr) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int c = cg * VEC + v;
        if (c < C) { atomicAdd(&sums[c], (double)a0[v]); atomicAdd(&sums[C + c], (double)a1[v]); }
      }
    }
    __syncthreads();
  }
}

// ss = [scale | shift | mean | rstd] (4*C floats)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, long M, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   float momentum, float eps, int training, float* __restrict__ ss) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    const double mu = sums[c] / (double)M;
    double vr = sums[C + c] / (double)M - mu * mu;
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

// out = lrelu(y*scale + shift)
template <typename T>
__global__ void __launch_bounds__(256)
bn_apply_kernel(const T* __restrict__ y, long ld_y, T* __restrict__ out, long ld_o, const float* __restrict__ ss,
                long M, int C, float slope) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C; const int c = (int)(i - m * C);
    const float z = ldf(y + m * ld_y + c) * ss[c] + ss[C + c];
    stf(out + m * ld_o + c, z > 0.f ? z : z * slope);
  }
}

// dy = scale * (dz - mean(dz) - xhat * mean(dz*xhat));  also dgamma = sum dz*xhat, dbeta = sum dz (block 0)
template <typename T>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const T* __restrict__ y, long ld_y, const T* __restrict__ da, long ld_da, T* __restrict__ dy, long ld_dy,
                    const float* __restrict__ ss, const double* __restrict__ sums, float* __restrict__ dgamma,
                    float* __restrict__ dbeta, int acc_param_grads, long M, int C, float slope) {
  const long total = M * C;
  const float invM = 1.f / (float)M;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C; const int c = (int)(i - m * C);
    const float yv = ldf(y + m * ld_y + c);
    const float z = yv * ss[c] + ss[C + c];
    const float dz = ldf(da + m * ld_da + c) * (z > 0.f ? 1.f : slope);
    const float xh = (yv - ss[2 * C + c]) * ss[3 * C + c];
    stf(dy + m * ld_dy + c, ss[c] * (dz - (float)sums[c] * invM - xh * (float)sums[C + c] * invM));
  }
  if (blockIdx.x == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      dgamma[c] = (acc_param_grads ? dgamma[c] : 0.f) + (float)sums[C + c];
      dbeta[c] = (acc_param_grads ? dbeta[c] : 0.f) + (float)sums[c];
    }
  }
}

// in-place LeakyReLU backward through a stored activation: g *= (a > 0 ? 1 : slope)
template <typename T>
__global__ void __launch_bounds__(256)
lrelu_bwd_kernel(T* __restrict__ g, long ld_g, const T* __restrict__ a, long ld_a, long M, int C, float slope) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C; const int c = (int)(i - m * C);
    if (!(ldf(a + m * ld_a + c) > 0.f)) stf(g + m * ld_g + c, ldf(g + m * ld_g + c) * slope);
  }
}

// dst(T)[m][c] = (beta ? dst : 0) + alpha * src(f32)[m][c]     and the reverse direction
template <typename T>
__global__ void __launch_bounds__(256)
cast_from_f32_kernel(T* __restrict__ dst, long ld_d, const float* __restrict__ src, long ld_s, long M, int C, float alpha, float beta) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C; const int c = (int)(i - m * C);
    const float v = alpha * src[m * ld_s + c] + (beta != 0.f ? beta * ldf(dst + m * ld_d + c) : 0.f);
    stf(dst + m * ld_d + c, v);
  }
}
template <typename T>
__global__ void __launch_bounds__(256)
cast_to_f32_kernel(float* __restrict__ dst, long ld_d, const T* __restrict__ src, long ld_s, long M, int C, float alpha, float beta) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C; const int c = (int)(i - m * C);
    dst[m * ld_d + c] = alpha * ldf(src + m * ld_s + c) + (beta != 0.f ? beta * dst[m * ld_d + c] : 0.f);
  }
}

// NCHW fp32 (B, C, H, W) channels [c0, c0+nc) -> NHWC dst[(b,y,x)][ld] channels [0, nc)
template <typename T>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int Ctot, int c0, int nc, long HW, long ld) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, pix = i - b * HW;
    for (int c = 0; c < nc; ++c) stf(dst + i * ld + c, src[(b * Ctot + c0 + c) * HW + pix]);
  }
}

// per-column sums over rows (bias gradients): out[c] (+)= sum_m g[m][c]
template <typename T>
__global__ void __launch_bounds__(256)
colsum_kernel(const T* __restrict__ g, long ld, long M, int C, float* __restrict__ out) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float acc = 0.f;
  for (long m = threadIdx.x + (long)blockIdx.y * blockDim.x; m < M; m += (long)blockDim.x * gridDim.y) acc += ldf(g + m * ld + c);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(&out[c], red[0]);
}

// fused multi-tensor Adam (torch.optim.Adam semantics, no weight decay / amsgrad); step lives on device so
// the launch is hipGraph-replayable.
__global__ void adam_tick_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }

__global__ void __launch_bounds__(256)
adam_kernel(const mireg_adam_job* __restrict__ jobs, const int* __restrict__ step, float lr, float b1, float b2, float eps,
            float grad_scale) {
  const mireg_adam_job j = jobs[blockIdx.y];
  const float t = (float)*step;
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < j.n; i += (long)gridDim.x * blockDim.x) {
    const float g = j.g[i] * grad_scale;
    const float m = b1 * j.m[i] + (1.f - b1) * g;
    const float v = b2 * j.v[i] + (1.f - b2) * g * g;
    j.m[i] = m; j.v[i] = v;
    j.p[i] -= step_size * (m / (sqrtf(v) / bc2s + eps));
  }
}

inline int grid1(long work, int cap = 2048) {
  long g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
#define DISPATCH_T(dtype, KERNEL, GRID, ...)                                                                      \
  do {                                                                                                              \
    if ((dtype) == MIREG_DTYPE_BF16) hipLaunchKernelGGL((KERNEL<__bf16>), GRID, dim3(256), 0, stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<float>), GRID, dim3(256), 0, stream, __VA_ARGS__);                              \
  } while (0)

}  // namespace

extern "C" {

int mireg_pack_weights(const mireg_pack_job* jobs_dev, int njobs, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && (dtype == MIREG_DTYPE_F32 || dtype == MIREG_DTYPE_BF16));
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((pack_weights_kernel<__bf16>), dim3(64, njobs), dim3(256), 0, stream, jobs_dev);
  else hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(64, njobs), dim3(256), 0, stream, jobs_dev);
  MIREG_LAUNCH_RET();
}

int mireg_unpack_wgrad(const mireg_pack_job* jobs_dev, int njobs, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0);
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(64, njobs), dim3(256), 0, stream, jobs_dev);
  MIREG_LAUNCH_RET();
}

int mireg_bn_stats(const void* y, long ld_y, long M, int C, double* sums, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(y && sums && M > 0 && C > 0 && C <= 4096);
  if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, stream) != hipSuccess) return MIREG_ERR_LAUNCH;
  const int cpr = (C + 3) / 4, rpp = 256 / (cpr < 256 ? cpr : 256);
  long g = (M + rpp * 8 - 1) / (rpp * 8);
  if (g > 1024) g = 1024;
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((bn_reduce_kernel<__bf16, false>), dim3((unsigned)g), dim3(256), 0, stream, (const __bf16*)y, ld_y, (const __bf16*)nullptr, 0L, (const float*)nullptr, sums, M, C, 0.f);
  else
    hipLaunchKernelGGL((bn_reduce_kernel<float, false>), dim3((unsigned)g), dim3(256), 0, stream, (const float*)y, ld_y, (const float*)nullptr, 0L, (const float*)nullptr, sums, M, C, 0.f);
  MIREG_LAUNCH_RET();
}

int mireg_bn_finalize(const double* sums, long M, int C, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int training, float* ss, hipStream_t stream) {
  MIREG_CHECK_ARG(gamma && beta && ss && M > 0 && C > 0 && (training ? sums != nullptr : (running_mean && running_var)));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums, M, C, gamma, beta, running_mean,
                     running_var, momentum, eps, training, ss);
  MIREG_LAUNCH_RET();
}

int mireg_bn_apply(const void* y, long ld_y, void* out, long ld_o, const float* ss, long M, int C, float slope, int dtype,
                   hipStream_t stream) {
  MIREG_CHECK_ARG(y && out && ss && M > 0 && C > 0);
  if (dtype == MIREG_DTYPE_BF16)
    hipLaunchKernelGGL((bn_apply_kernel<__bf16>), dim3(grid1(M * C)), dim3(256), 0, stream, (const __bf16*)y, ld_y, (__bf16*)out, ld_o, ss, M, C, slope);
  else
    hipLaunchKernelGGL((bn_apply_kernel<float>), dim3(grid1(M * C)), dim3(256), 0, stream, (const float*)y, ld_y, (float*)out, ld_o, ss, M, C, slope);
  MIREG_LAUNCH_RET();
}

int mireg_bn_bwd(const void* y, long ld_y, const void* da, long ld_da, void* dy, long ld_dy, const float* ss, double* sums,
                 float* dgamma, float* dbeta, int acc_param_grads, long M, int C, float slope, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(y && da && dy && ss && sums && M > 0 && C > 0 && C <= 4096);
  if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, stream) != hipSuccess) return MIREG_ERR_LAUNCH;
  const int cpr = (C + 3) / 4, rpp = 256 / (cpr < 256 ? cpr : 256);
  long g = (M + rpp * 8 - 1) / (rpp * 8);
  if (g > 1024) g = 1024;
  if (dtype == MIREG_DTYPE_BF16) {
    hipLaunchKernelGGL((bn_reduce_kernel<__bf16, true>), dim3((unsigned)g), dim3(256), 0, stream, (const __bf16*)y, ld_y, (const __bf16*)da, ld_da, ss, sums, M, C, slope);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<__bf16>), dim3(grid1(M * C)), dim3(256), 0, stream, (const __bf16*)y, ld_y, (const __bf16*)da, ld_da, (__bf16*)dy, ld_dy, ss, sums, dgamma, dbeta, acc_param_grads, M, C, slope);
  } else {
    hipLaunchKernelGGL((bn_reduce_kernel<float, true>), dim3((unsigned)g), dim3(256), 0, stream, (const float*)y, ld_y, (const float*)da, ld_da, ss, sums, M, C, slope);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(grid1(M * C)), dim3(256), 0, stream, (const float*)y, ld_y, (const float*)da, ld_da, (float*)dy, ld_dy, ss, sums, dgamma, dbeta, acc_param_grads, M, C, slope);
  }
  MIREG_LAUNCH_RET();
}

int mireg_lrelu_bwd(void* g, long ld_g, const void* a, long ld_a, long M, int C, float slope, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(g && a && M > 0 && C > 0);
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((lrelu_bwd_kernel<__bf16>), dim3(grid1(M * C)), dim3(256), 0, stream, (__bf16*)g, ld_g, (const __bf16*)a, ld_a, M, C, slope);
  else hipLaunchKernelGGL((lrelu_bwd_kernel<float>), dim3(grid1(M * C)), dim3(256), 0, stream, (float*)g, ld_g, (const float*)a, ld_a, M, C, slope);
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

int mireg_colsum(const void* g, long ld, long M, int C, float* out, int accumulate, int dtype, hipStream_t stream) {
  MIREG_CHECK_ARG(g && out && M > 0 && C > 0);
  if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * C, stream) != hipSuccess) return MIREG_ERR_LAUNCH;
  long gy = (M + 256 * 16 - 1) / (256 * 16);
  if (gy > 64) gy = 64;
  if (dtype == MIREG_DTYPE_BF16) hipLaunchKernelGGL((colsum_kernel<__bf16>), dim3(C, (unsigned)gy), dim3(256), 0, stream, (const __bf16*)g, ld, M, C, out);
  else hipLaunchKernelGGL((colsum_kernel<float>), dim3(C, (unsigned)gy), dim3(256), 0, stream, (const float*)g, ld, M, C, out);
  MIREG_LAUNCH_RET();
}

int mireg_adam_step(const mireg_adam_job* jobs_dev, int njobs, int* step_dev, float lr, float beta1, float beta2, float eps,
                    float grad_scale, hipStream_t stream) {
  MIREG_CHECK_ARG(jobs_dev && njobs > 0 && step_dev);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, stream, step_dev);
  hipLaunchKernelGGL(adam_kernel, dim3(128, njobs), dim3(256), 0, stream, jobs_dev, step_dev, lr, beta1, beta2, eps, grad_scale);
  MIREG_LAUNCH_RET();
}

}  // extern "C"
