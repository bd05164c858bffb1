// HBM ceilings for the tiler's access mix (tooling only): pure 16-B streaming stores, a float4 copy, and a 1 : 4 read : write mix (4-byte
// loads, 16-byte stores: the byte ratio of gather NHWC f32: 196 608 B read + 786 432 B written per 256 x 256 tile), plain and
// non-temporal.  hipcc --offload-arch=gfx950 tools/hbm_mix_peak.hip -o tools/bin/hbm_mix_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <bool NT> __global__ __launch_bounds__(256) void fill_k(float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)i);
    if (NT) __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(out) + i); else out[i] = v;
  }
}
__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
template <bool NT> __global__ __launch_bounds__(256) void mix_k(const unsigned* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned v = NT ? __builtin_nontemporal_load(in + i) : in[i];
    const float4 f = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24));
    if (NT) __builtin_nontemporal_store(f4v{f.x, f.y, f.z, f.w}, reinterpret_cast<f4v*>(out) + i); else out[i] = f;
  }
}
int main() {
  const size_t n = (size_t)1 << 26;   // 64 Mi float4 = 1 GiB written; 256 MiB read in the mix
  float4 *a, *b; unsigned* c;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 4));
  CK(hipMemset(a, 1, n * 16)); CK(hipMemset(c, 7, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us  %7.0f GB/s\n", name, ms * 100.0, bytes / (ms * 1e-4) / 1e9);
  };
  for (int g : {2048, 8192, 32768}) {
    printf("grid %d\n", g);
    timeit("fill  16-B stores (1 GiB)", n * 16.0, [&] { fill_k<false><<<g, 256>>>(b, n); });
    timeit("fill  16-B stores, non-temporal", n * 16.0, [&] { fill_k<true><<<g, 256>>>(b, n); });
    timeit("copy  float4 (1 GiB -> 1 GiB)", n * 32.0, [&] { copy_k<<<g, 256>>>(a, b, n); });
    timeit("mix   4-B load + 16-B store (0.25 + 1 GiB)", n * 20.0, [&] { mix_k<false><<<g, 256>>>(c, b, n); });
    timeit("mix   non-temporal load + store", n * 20.0, [&] { mix_k<true><<<g, 256>>>(c, b, n); });
  }
  return 0;
}
